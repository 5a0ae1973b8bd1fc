// conv_gemm.h — Conv1d / ConvTranspose1d as an implicit GEMM: gemm_kernel (gemm.h) with TAP-SHIFTED activation rows.
//
//   out[l, n] = epilogue( sum_j sum_c  Y[l + shift0 + j * dstep, c] * W[n, j * cpad + c] )        (rows outside [0, L) read as zero)
//
// Y is ONE operand copy [L, cpad] per batch element (fp32 rows, fp16 rows or packed fp16 hi/lo rows), not the k-fold tap-gathered
// matrix of launch_im2col_taps: the k-loop walks the taps, and for tap j the activation tile is simply loaded from rows shifted by
// shift0 + j * dstep.  A tap's segment of an operand row is a whole number of 128-byte k-tiles (cpad % 32 == 0 in the 4-byte-per-
// element layouts, cpad % 64 == 0 for plain fp16), so a k-tile never straddles two taps and the only change against gemm_kernel is
// the activation row offset, recomputed per k-tile (a handful of integer VALU ops per 16-byte chunk) — same LDS image, same swizzle,
// same fragment reads, same epilogue.  HBM traffic of a conv drops from (1 + 2k) to 2 activation-tensor passes (the shifted re-reads
// hit L2: consecutive taps re-read the same rows).
//
// Used by the BigVGAN path (bigvgan_api.cpp, option conv_impl = 1); the tap-gathered path (conv_impl = 0) is the cross-check.
#pragma once
#include "gemm.h"

struct ConvTaps {
  int ntaps, shift0, dstep;
  int tiles_per_tap;  // 128-byte k-tiles per tap segment = cpad * bytes per element / 128
};

template <typename T, int NSPLIT, int TM, int TN, typename Epi, int WGM = 2, int WGN = 2>
__global__ __launch_bounds__(64 * WGM * WGN) void conv_gemm_kernel(GemmCore g, ConvTaps tp, Epi epi) {
  constexpr int NT = 64 * WGM * WGN;
  constexpr int BM = 32 * WGM * TM, BN = 32 * WGN * TN;
  constexpr int NPL = (NSPLIT == 3) ? 2 : 1;
  constexpr int CPR = GEMM_KTB / 16;
  constexpr int KSTEPS = NPL == 2 ? 2 : 4;
  constexpr int CA = BM * CPR / NT, CW = BN * CPR / NT;
  constexpr int TILE_A = BM * GEMM_KTB, TILE_W = BN * GEMM_KTB;
  constexpr int STAGE = TILE_A + TILE_W;
  static_assert(CA >= 1 && CW >= 1 && CA * NT == BM * CPR && CW * NT == BN * CPR, "tile does not split evenly over the threads");
  F5_DYN_LDS(char, smem);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave % WGM, wn = wave / WGM;
  const int z = blockIdx.z;
  int m0, n0;
  {  // XCD-contiguous tile order, channel tiles fastest (gemm.h)
    const int nt = (g.N + BN - 1) / BN, nwg = gridDim.x;
    const int bid = blockIdx.x, q8 = nwg >> 3, r8 = nwg & 7, xcd = bid & 7, slot = bid >> 3;
    const int L = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + slot;
    const int mt = L / nt;
    m0 = mt * BM;
    n0 = (L - mt * nt) * BN;
  }
  // g.K is the GEMM's K = ntaps * cpad; g.lda is the row stride of the SINGLE-copy activation operand (>= cpad * NPL elements)
  const int wbytes_row = g.K * (int)sizeof(T) * NPL;                       // bytes of one weight row
  const int abytes_row = tp.tiles_per_tap * GEMM_KTB;                      // bytes of one activation row (= one tap segment)
  const int64_t lda_b = g.lda * (int64_t)sizeof(T);
  const uint32_t a_bytes = (uint32_t)((int64_t)(g.a_rows - 1) * lda_b + abytes_row);
  const uint32_t w_bytes = (uint32_t)((int64_t)(g.w_rows - 1) * g.ldw * (int64_t)sizeof(T) + wbytes_row);
  const BufRsrc Ar = make_rsrc(reinterpret_cast<const T*>(g.A) + (int64_t)z * g.strideA, a_bytes);
  const BufRsrc Wr = make_rsrc(reinterpret_cast<const T*>(g.W) + (int64_t)z * g.strideW, w_bytes);
  int a_row[CA], a_c[CA];
  uint32_t w_off[CW];
  int w_c[CW];
#pragma unroll
  for (int i = 0; i < CA; ++i) {
    const int c = tid + i * NT, row = c / CPR;
    a_row[i] = m0 + row;
    a_c[i] = ((c % CPR) ^ ((row >> 1) & 7)) * 16;
  }
#pragma unroll
  for (int i = 0; i < CW; ++i) {
    const int c = tid + i * NT, row = c / CPR, lc = (c % CPR) ^ ((row >> 1) & 7);
    w_c[i] = lc * 16;
    w_off[i] = (n0 + row) < g.w_rows ? (uint32_t)((int64_t)(n0 + row) * g.ldw * (int64_t)sizeof(T) + lc * 16) : OOB_ROW;
  }

  uint4 ra0[CA], rw0[CW], ra1[CA], rw1[CW];
  // load_global is called for k-tiles 0, 1, 2, ... in order, exactly once each: the tap / tile-within-tap counters run along
  int ld_tap = 0, ld_within = 0;
  auto load_global = [&](int kt, uint4 (&ra)[CA], uint4 (&rw)[CW]) {
    const int kb = kt * GEMM_KTB;
    const int shift = tp.shift0 + ld_tap * tp.dstep;
    const int wb = ld_within * GEMM_KTB;
    const bool live = ld_tap < tp.ntaps;
#pragma unroll
    for (int i = 0; i < CA; ++i) {
      const int r = a_row[i] + shift;
      const bool ok = live && r >= 0 && r < g.a_rows;
      ra[i] = buffer_load_b128(Ar, ok ? (uint32_t)((int64_t)r * lda_b) + (uint32_t)(wb + a_c[i]) : OOB_OFF);
    }
#pragma unroll
    for (int i = 0; i < CW; ++i) rw[i] = buffer_load_b128(Wr, (kb + w_c[i]) < wbytes_row ? w_off[i] + (uint32_t)kb : OOB_OFF);
    if (++ld_within == tp.tiles_per_tap) { ld_within = 0; ++ld_tap; }
  };
  auto store_lds = [&](int stage, const uint4 (&ra)[CA], const uint4 (&rw)[CW]) {
    char* base = smem + stage * STAGE;
#pragma unroll
    for (int i = 0; i < CA; ++i) *reinterpret_cast<uint4*>(base + (tid + i * NT) * 16) = ra[i];
#pragma unroll
    for (int i = 0; i < CW; ++i) *reinterpret_cast<uint4*>(base + TILE_A + (tid + i * NT) * 16) = rw[i];
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int j = 0; j < TM; ++j)
#pragma unroll
    for (int i = 0; i < TN; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[j][i][r] = 0.f;

  const int frow = (lane & 31) * GEMM_KTB, fswz = ((lane & 31) >> 1) & 7, fhi = lane >> 5;
  int foff[NPL][KSTEPS];
#pragma unroll
  for (int p = 0; p < NPL; ++p)
#pragma unroll
    for (int ks = 0; ks < KSTEPS; ++ks) foff[p][ks] = frow + (((p * 4 + 2 * ks + fhi) ^ fswz) << 4);

  auto compute = [&](int stage) {
    const char* sA = smem + stage * STAGE + (wm * 32 * TM) * GEMM_KTB;
    const char* sW = smem + stage * STAGE + TILE_A + (wn * 32 * TN) * GEMM_KTB;
#pragma unroll
    for (int ks = 0; ks < KSTEPS; ++ks) {
      Frag fa[NPL][TM], fw[NPL][TN];
#pragma unroll
      for (int p = 0; p < NPL; ++p) {
#pragma unroll
        for (int j = 0; j < TM; ++j) fa[p][j].u = *reinterpret_cast<const uint4*>(sA + j * 32 * GEMM_KTB + foff[p][ks]);
#pragma unroll
        for (int i = 0; i < TN; ++i) fw[p][i].u = *reinterpret_cast<const uint4*>(sW + i * 32 * GEMM_KTB + foff[p][ks]);
      }
#pragma unroll
      for (int j = 0; j < TM; ++j)
#pragma unroll
        for (int i = 0; i < TN; ++i) {
          Mma32<T>::mma(acc[j][i], fw[0][i], fa[0][j]);
          if constexpr (NPL == 2) {
            Mma32<T>::mma(acc[j][i], fw[0][i], fa[1][j]);  // W_hi . A_lo
            Mma32<T>::mma(acc[j][i], fw[1][i], fa[0][j]);  // W_lo . A_hi
          }
        }
    }
  };

  const int nkt = tp.ntaps * tp.tiles_per_tap;
  load_global(0, ra0, rw0);
  load_global(1, ra1, rw1);
  store_lds(0, ra0, rw0);
  __syncthreads();

  int kt = 0;
  for (; kt + 1 < nkt; kt += 2) {
    load_global(kt + 2, ra0, rw0);
    __builtin_amdgcn_sched_barrier(0);
    compute(0);
    store_lds(1, ra1, rw1);
    __syncthreads();
    load_global(kt + 3, ra1, rw1);
    __builtin_amdgcn_sched_barrier(0);
    compute(1);
    store_lds(0, ra0, rw0);
    __syncthreads();
  }
  if (kt < nkt) compute(0);

#pragma unroll
  for (int j = 0; j < TM; ++j) {
    const int m = m0 + wm * 32 * TM + j * 32 + (lane & 31);
    if (m >= g.M) continue;
#pragma unroll
    for (int i = 0; i < TN; ++i) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int n = n0 + wn * 32 * TN + i * 32 + 8 * q + 4 * (lane >> 5);
        if (n < g.N) epi(m, n, make_float4(acc[j][i][4 * q], acc[j][i][4 * q + 1], acc[j][i][4 * q + 2], acc[j][i][4 * q + 3]), z);
      }
    }
  }
}
