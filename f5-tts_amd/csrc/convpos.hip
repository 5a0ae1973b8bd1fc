// convpos.hip — ConvPositionEmbedding's grouped Conv1d(dim, dim, k=31, groups=16) as an implicit GEMM
// on MFMA (reference model/modules.py:175-201).  One block = (sequence, group, 128 frames):
//   out[m, co] = mish( mask( bias[co] + sum_{tap, ci} x[m + tap - K/2, ci] * W[co, ci, tap] ) ) (+ residual)
// The haloed input tile (128 + K - 1 frames x cpg channels) is staged ONCE in LDS (converted to the
// operand type on the way); the 31 shifted GEMMs of depth cpg read it at a row offset of `tap`, while
// the per-tap weight tile [cpg(co)][cpg(ci)] streams through a double-buffered LDS slot.
// The weight axis is the MFMA "A" operand (accumulator rows) exactly as in gemm.h.
#include "kernels.h"

namespace {

template <typename T, int NPL, int CPG>
struct ConvCfg {
  static constexpr int BMR = 128;                                // frames per block
  static constexpr int COT = (CPG + 31) / 32;                    // 32-row co tiles
  static constexpr int ROWB = CPG * (int)sizeof(T) + 16;         // LDS row bytes (padded)
  static constexpr int KSTEPS = CPG * (int)sizeof(T) / 32;       // 32-byte k-steps per tap
  static constexpr int WROWS = COT * 32;
  // weight tile rows: when a row is exactly one 128-byte line (fp16, 64 channels per group: the Base models) they are stored UNPADDED
  // with the 16-byte chunk index XOR-swizzled by (row & 7) — conflict-free fragment reads like the padded layout, and 4 KB less LDS
  // per workgroup, which is what lets TWO workgroups share a CU (80.4 KB -> 76.4 KB; 160 KB per CU)
  static constexpr bool WSWZ = CPG * (int)sizeof(T) == 128;
  static constexpr int WROWB = WSWZ ? 128 : ROWB;
  static constexpr int WPLANE = WROWS * WROWB;
  static constexpr int WSTAGE = NPL * WPLANE;
};

template <typename T, int NPL, int CPG>
__global__ __launch_bounds__(256) void convpos_kernel(const float* __restrict__ x, const T* __restrict__ w, const T* __restrict__ w_lo,
                                                      const float* __restrict__ bias, const uint8_t* __restrict__ rowvalid,
                                                      const float* __restrict__ residual, int n, int D, int K, float* out, int out_n, int out_off,
                                                      int mtiles, int S) {
  using C = ConvCfg<T, NPL, CPG>;
  F5_DYN_LDS(char, smem);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // XCD-aware placement (round 6): workgroup b runs on XCD b % 8, and each XCD has its own L2.  With the plain (frame tile, group, sequence) grid
  // every XCD saw every group, so each of the eight L2s fetched the whole 31-tap weight set of all 16 groups (PMC: 90 MB per B = 1 launch for
  // 11.5 MB of activations + 8 MB of weights).  Here an XCD takes a CONTIGUOUS run of the order (group, sequence, frame tile): a group's
  // weights are fetched by one L2 (two at a run boundary).  Bijective for any grid size, as in the GEMM and attention kernels.
  int m0, g, s;
  {
    const int nwg = gridDim.x, bid = blockIdx.x, q8 = nwg >> 3, r8 = nwg & 7, xcd = bid & 7, slot = bid >> 3;
    const int L = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + slot;
    const int per_g = S * mtiles;
    g = L / per_g;
    const int rem = L - g * per_g;
    s = rem / mtiles;
    m0 = (rem - s * mtiles) * C::BMR;
  }
  const int halo = K / 2;
  const int xrows = C::BMR + K - 1;
  const int xplane = xrows * C::ROWB;
  char* sX = smem;                       // [NPL][xrows][ROWB]
  char* sW = smem + NPL * xplane;        // [2 stages][NPL][WROWS][ROWB]

  // ---- stage the haloed input tile (fp32 -> T [hi, lo]) -------------------------------------
  constexpr int C4 = CPG / 4;  // float4 chunks per row
  for (int c = tid; c < xrows * C4; c += 256) {
    const int r = c / C4, c4 = c - r * C4;
    const int pos = m0 + r - halo;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (pos >= 0 && pos < n) {
      const int64_t grow = (int64_t)s * n + pos;
      if (!rowvalid || rowvalid[grow]) v = *reinterpret_cast<const float4*>(x + grow * D + g * CPG + c4 * 4);
    }
    if constexpr (sizeof(T) == 4) {
      *reinterpret_cast<float4*>(sX + r * C::ROWB + c4 * 16) = v;
    } else {
      f16x4 hi, lo;
      const float xv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
      for (int e = 0; e < 4; ++e) { f16 h, l; split_f16(xv[e], h, l); hi[e] = h; lo[e] = l; }
      *reinterpret_cast<f16x4*>(sX + r * C::ROWB + c4 * 8) = hi;
      if constexpr (NPL == 2) *reinterpret_cast<f16x4*>(sX + xplane + r * C::ROWB + c4 * 8) = lo;
    }
  }
  // zero the co-padding rows of both weight stages once (CPG < 32 only)
  if constexpr (C::WROWS > CPG) {
    for (int c = tid; c < 2 * C::WSTAGE / 16; c += 256) *reinterpret_cast<uint4*>(sW + c * 16) = make_uint4(0, 0, 0, 0);
    __syncthreads();
  }

  // ---- weight tile streaming -------------------------------------------------------------------
  constexpr int WCH = CPG * CPG * (int)sizeof(T) / 16;    // 16-byte chunks per tap tile
  constexpr int CPR = CPG * (int)sizeof(T) / 16;          // chunks per row
  constexpr int WPT = (WCH + 255) / 256;
  // plain vector-typed temporaries: an array of HIP's uint4 CLASS here is not scalarised by the compiler and ends up in scratch (every
  // tap's weight chunks went global -> scratch -> LDS; found by tests/test_isa_hazards.py), an array of ext_vector_type values stays in VGPRs
  typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
  u32x4 rw[NPL][WPT];
  const int64_t wtile = (int64_t)CPG * CPG;
  auto load_w = [&](int tap) {
    const T* src = w + ((int64_t)g * K + tap) * wtile;
    const T* src_lo = NPL == 2 ? w_lo + ((int64_t)g * K + tap) * wtile : nullptr;
#pragma unroll
    for (int i = 0; i < WPT; ++i) {
      const int c = tid + i * 256;
      if (c < WCH) {
        rw[0][i] = reinterpret_cast<const u32x4*>(src)[c];
        if constexpr (NPL == 2) rw[1][i] = reinterpret_cast<const u32x4*>(src_lo)[c];
      }
    }
  };
  auto store_w = [&](int stage) {
#pragma unroll
    for (int i = 0; i < WPT; ++i) {
      const int c = tid + i * 256;
      if (c < WCH) {
        const int r = c / CPR, cc = c - r * CPR;
#pragma unroll
        for (int p = 0; p < NPL; ++p)
          *reinterpret_cast<u32x4*>(sW + stage * C::WSTAGE + p * C::WPLANE + r * C::WROWB + ((C::WSWZ ? (cc ^ (r & 7)) : cc) << 4)) = rw[p][i];
      }
    }
  };

  f32x16 acc[C::COT];
#pragma unroll
  for (int i = 0; i < C::COT; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

  load_w(0);
  store_w(0);
  __syncthreads();

  const int koff = (lane >> 5) * 16;
  for (int tap = 0; tap < K; ++tap) {
    if (tap + 1 < K) load_w(tap + 1);
    const char* wbase = sW + (tap & 1) * C::WSTAGE + (lane & 31) * C::WROWB;  // + chunk offset (swizzled per row) below
    const int wrow7 = lane & 7;                                               // (row & 7) of rows lane&31 + 32 i
    const char* xbase = sX + (wave * 32 + (lane & 31) + tap) * C::ROWB + koff;
#pragma unroll
    for (int ks = 0; ks < C::KSTEPS; ++ks) {
      Frag fx[NPL], fw[NPL][C::COT];
#pragma unroll
      for (int p = 0; p < NPL; ++p) {
        fx[p].u = *reinterpret_cast<const uint4*>(xbase + p * xplane + ks * 32);
#pragma unroll
        for (int i = 0; i < C::COT; ++i)
          fw[p][i].u = *reinterpret_cast<const uint4*>(wbase + p * C::WPLANE + i * 32 * C::WROWB +
                                                       (C::WSWZ ? (((2 * ks + (lane >> 5)) ^ wrow7) << 4) : koff + ks * 32));
      }
#pragma unroll
      for (int i = 0; i < C::COT; ++i) {
        Mma32<T>::mma(acc[i], fw[0][i], fx[0]);
        if constexpr (NPL == 2) {
          Mma32<T>::mma(acc[i], fw[0][i], fx[1]);
          Mma32<T>::mma(acc[i], fw[1][i], fx[0]);
        }
      }
    }
    if (tap + 1 < K) store_w((tap + 1) & 1);
    __syncthreads();
  }

  // ---- epilogue --------------------------------------------------------------------------------
  const int m = m0 + wave * 32 + (lane & 31);
  if (m >= n) return;
  const int64_t grow = (int64_t)s * n + m;
  const bool dead = rowvalid && !rowvalid[grow];
  const int64_t orow = (int64_t)s * out_n + m + out_off;  // output rows may live in a longer sequence (UNetT: time token first)
#pragma unroll
  for (int i = 0; i < C::COT; ++i) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int co = i * 32 + 8 * q + 4 * (lane >> 5);
      if (co >= CPG) continue;
      const int ch = g * CPG + co;
      const float4 b = *reinterpret_cast<const float4*>(bias + ch);
      float v[4] = {acc[i][4 * q] + b.x, acc[i][4 * q + 1] + b.y, acc[i][4 * q + 2] + b.z, acc[i][4 * q + 3] + b.w};
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = dead ? 0.f : act_mish(v[e]);
      if (residual) {
        const float4 r = *reinterpret_cast<const float4*>(residual + grow * D + ch);
        v[0] += r.x; v[1] += r.y; v[2] += r.z; v[3] += r.w;
      }
      *reinterpret_cast<float4*>(out + orow * D + ch) = make_float4(v[0], v[1], v[2], v[3]);
    }
  }
}

// ---- MX form (round 6; 64 channels per group: the Base models in `fp16m`) -------------------------------------------------------------------
// The fp16x3 form above issues THREE fp16 MFMAs per product (hi.hi + hi.lo + lo.hi): 24 per tap and wave, 744 per wave over the 31 taps —
// half of a B = 1 launch's 48 us is the matrix pipe.  Here the two correction products are ONE MX-fp6 MFMA per 32 input channels, as in the
// block GEMMs (common.h, "fp16 + MX-fp6 corrections"): 8 fp16 + 4 fp6 MFMAs per tap and wave.  Both operands are MX LINES — a row's 64 input
// channels = two 128-byte lines [32 hi | P_0 | P_1]:
//   * weights: packed once at finalize by pack_mx_rows_kernel<WEIGHT> from the per-tap tiles [G][K][co][ci] (api.cpp conv_wmx): 16 KB per
//     tap, the bytes of the hi + lo planes; streamed through the double-buffered LDS slot with the 16-byte chunk index XOR-swizzled by
//     (row & 15) (rows are 256 bytes: sixteen rows then cover every bank once);
//   * activations: the haloed tile is packed on the way into LDS — one thread per (row, 32-channel block, half-wave set of 16 channels):
//     four float4 loads, mx_pack16, three stores; rows 256 + 16 bytes apart.  The packed row serves all 31 taps (a tap is a row offset).
template <int CPG>
__global__ __launch_bounds__(256) void convpos_mx_kernel(const float* __restrict__ x, const f16* __restrict__ wmx, const float* __restrict__ bias,
                                                         const uint8_t* __restrict__ rowvalid, const float* __restrict__ residual, int n, int D, int K,
                                                         float* out, int out_n, int out_off, int mtiles, int S) {
  static_assert(CPG == 64, "two 32-channel MX lines per row");
  constexpr int BMR = 128, XROWB = 256 + 16, WROWB = 256, WSTAGE = CPG * WROWB, COT = 2;
  F5_DYN_LDS(char, smem);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, hi = lane >> 5;
  int m0, g, s;
  {  // XCD-contiguous (group, sequence, frame tile) order, as convpos_kernel
    const int nwg = gridDim.x, bid = blockIdx.x, q8 = nwg >> 3, r8 = nwg & 7, xcd = bid & 7, slot = bid >> 3;
    const int L = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + slot;
    const int per_g = S * mtiles;
    g = L / per_g;
    const int rem = L - g * per_g;
    s = rem / mtiles;
    m0 = (rem - s * mtiles) * BMR;
  }
  const int halo = K / 2, xrows = BMR + K - 1;
  char* sX = smem;                   // [xrows][XROWB]
  char* sW = smem + xrows * XROWB;   // [2 stages][CPG rows][256 B], chunks swizzled

  // ---- the haloed input tile as MX lines -----------------------------------------------------------------------------------------------
  for (int u = tid; u < xrows * 4; u += 256) {
    const int r = u >> 2, blk = (u >> 1) & 1, h = u & 1;
    const int pos = m0 + r - halo;
    float v[16];
#pragma unroll
    for (int t = 0; t < 16; ++t) v[t] = 0.f;
    if (pos >= 0 && pos < n) {
      const int64_t grow = (int64_t)s * n + pos;
      if (!rowvalid || rowvalid[grow]) {
        const float* src = x + grow * D + g * CPG + 32 * blk + 4 * h;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float4 t4 = *reinterpret_cast<const float4*>(src + 8 * q);
          v[4 * q] = t4.x; v[4 * q + 1] = t4.y; v[4 * q + 2] = t4.z; v[4 * q + 3] = t4.w;
        }
      }
    }
    uint32_t hv[8], pw[8];
    mx_pack16<false>(v, hv, pw);
    char* line = sX + r * XROWB + blk * 128;
#pragma unroll
    for (int q = 0; q < 4; ++q) *reinterpret_cast<uint2*>(line + (8 * q + 4 * h) * 2) = make_uint2(hv[2 * q], hv[2 * q + 1]);
    *reinterpret_cast<uint4*>(line + 64 + 32 * h) = make_uint4(pw[0], pw[1], pw[2], pw[3]);
    *reinterpret_cast<uint4*>(line + 80 + 32 * h) = make_uint4(pw[4], pw[5], pw[6], pw[7]);
  }

  // ---- weight tile streaming: 16 KB per tap = 4 chunks of 16 bytes per thread ------------------------------------------------------------
  typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
  u32x4 rw[4];
  auto load_w = [&](int tap) {
    const u32x4* src = reinterpret_cast<const u32x4*>(reinterpret_cast<const char*>(wmx) + ((int64_t)g * K + tap) * WSTAGE);
#pragma unroll
    for (int i = 0; i < 4; ++i) rw[i] = src[tid + i * 256];
  };
  auto store_w = [&](int stage) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int c = tid + i * 256, r = c >> 4, cc = c & 15;
      *reinterpret_cast<u32x4*>(sW + stage * WSTAGE + r * WROWB + ((cc ^ (r & 15)) << 4)) = rw[i];
    }
  };

  f32x16 acc[COT];
#pragma unroll
  for (int i = 0; i < COT; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

  load_w(0);
  store_w(0);
  __syncthreads();
  const int wr = lane & 31, wsw = wr & 15;  // weight row of this lane inside a 32-row co tile (rows wr + 32 i: the same swizzle key)
  for (int tap = 0; tap < K; ++tap) {
    if (tap + 1 < K) load_w(tap + 1);
    const char* wbase = sW + (tap & 1) * WSTAGE + wr * WROWB;
    const char* xbase = sX + (wave * 32 + wr + tap) * XROWB;
#pragma unroll
    for (int blk = 0; blk < 2; ++blk) {
      // a line's chunks: hi halves of k-step ks' at 2 ks' + hi, P_hi at 4 + 2 hi, 5 + 2 hi
      Frag fx[4];
#pragma unroll
      for (int c = 0; c < 4; ++c) fx[c].u = *reinterpret_cast<const uint4*>(xbase + blk * 128 + ((c < 2 ? 2 * c + hi : 4 + 2 * hi + (c - 2)) << 4));
#pragma unroll
      for (int i = 0; i < COT; ++i) {
        Frag fw[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const int chunk = blk * 8 + (c < 2 ? 2 * c + hi : 4 + 2 * hi + (c - 2));
          fw[c].u = *reinterpret_cast<const uint4*>(wbase + i * 32 * WROWB + ((chunk ^ wsw) << 4));
        }
        Mma32<f16>::mma(acc[i], fw[0], fx[0]);
        Mma32<f16>::mma(acc[i], fw[1], fx[1]);
        mx_mma(acc[i], fw[2].u, fw[3].u, fx[2].u, fx[3].u);
      }
    }
    if (tap + 1 < K) store_w((tap + 1) & 1);
    __syncthreads();
  }

  // ---- epilogue (as convpos_kernel) --------------------------------------------------------------------------------------------------------
  const int m = m0 + wave * 32 + (lane & 31);
  if (m >= n) return;
  const int64_t grow = (int64_t)s * n + m;
  const bool dead = rowvalid && !rowvalid[grow];
  const int64_t orow = (int64_t)s * out_n + m + out_off;
#pragma unroll
  for (int i = 0; i < COT; ++i) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int ch = g * CPG + i * 32 + 8 * q + 4 * hi;
      const float4 b = *reinterpret_cast<const float4*>(bias + ch);
      float v[4] = {acc[i][4 * q] + b.x, acc[i][4 * q + 1] + b.y, acc[i][4 * q + 2] + b.z, acc[i][4 * q + 3] + b.w};
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = dead ? 0.f : act_mish(v[e]);
      if (residual) {
        const float4 r = *reinterpret_cast<const float4*>(residual + grow * D + ch);
        v[0] += r.x; v[1] += r.y; v[2] += r.z; v[3] += r.w;
      }
      *reinterpret_cast<float4*>(out + orow * D + ch) = make_float4(v[0], v[1], v[2], v[3]);
    }
  }
}

template <typename T, int NPL, int CPG>
hipError_t launch_cfg(const float* x, const T* w, const T* w_lo, const float* bias, const uint8_t* rowvalid, const float* residual,
                      int S, int n, int D, int groups, int K, float* out, hipStream_t s, int out_n, int out_off) {
  using C = ConvCfg<T, NPL, CPG>;
  const int lds = NPL * (C::BMR + K - 1) * C::ROWB + 2 * C::WSTAGE;
  auto kern = convpos_kernel<T, NPL, CPG>;
  if (lds > 160 * 1024) return hipErrorInvalidValue;
  const int mtiles = (n + C::BMR - 1) / C::BMR;
  hipLaunchKernelGGL(kern, dim3(mtiles * groups * S), dim3(256), lds, s, x, w, w_lo, bias, rowvalid, residual, n, D, K, out, out_n, out_off, mtiles, S);
  return hipGetLastError();
}

template <int CPG>
hipError_t launch_cpg(int op, const float* x, const float* w32, const f16* whi, const f16* wlo, const float* bias,
                      const uint8_t* rowvalid, const float* residual, int S, int n, int D, int groups, int K, float* out, hipStream_t s, int out_n,
                      int out_off) {
  switch (op) {
    case OP_F32: return launch_cfg<float, 1, CPG>(x, w32, nullptr, bias, rowvalid, residual, S, n, D, groups, K, out, s, out_n, out_off);
    case OP_F16: return launch_cfg<f16, 1, CPG>(x, whi, nullptr, bias, rowvalid, residual, S, n, D, groups, K, out, s, out_n, out_off);
    case OP_F16X3: return launch_cfg<f16, 2, CPG>(x, whi, wlo, bias, rowvalid, residual, S, n, D, groups, K, out, s, out_n, out_off);
    default: return hipErrorInvalidValue;
  }
}

}  // namespace

hipError_t launch_convpos(int op, const float* x, const float* w32, const f16* whi, const f16* wlo, const float* bias,
                          const uint8_t* rowvalid, const float* residual, int S, int n, int D, int groups, int K, float* out,
                          hipStream_t s, int out_n, int out_off, const f16* wmx) {
  if (out_n <= 0) out_n = n;
  const int cpg = D / groups;
  if (cpg * groups != D || (K & 1) == 0) return hipErrorInvalidValue;
  if (wmx) {  // MX lines (fp16m calls; api.cpp builds them for 64 channels per group only)
    if (cpg != 64 || op != OP_F16X3) return hipErrorInvalidValue;
    const int mtiles = (n + 127) / 128, lds = (128 + K - 1) * (256 + 16) + 2 * 64 * 256;
    hipLaunchKernelGGL(convpos_mx_kernel<64>, dim3(mtiles * groups * S), dim3(256), lds, s, x, wmx, bias, rowvalid, residual, n, D, K, out, out_n, out_off, mtiles, S);
    return hipGetLastError();
  }
  switch (cpg) {
    case 16: return launch_cpg<16>(op, x, w32, whi, wlo, bias, rowvalid, residual, S, n, D, groups, K, out, s, out_n, out_off);
    case 32: return launch_cpg<32>(op, x, w32, whi, wlo, bias, rowvalid, residual, S, n, D, groups, K, out, s, out_n, out_off);
    case 48: return launch_cpg<48>(op, x, w32, whi, wlo, bias, rowvalid, residual, S, n, D, groups, K, out, s, out_n, out_off);  // dim 768 (the Small models)
    case 64: return launch_cpg<64>(op, x, w32, whi, wlo, bias, rowvalid, residual, S, n, D, groups, K, out, s, out_n, out_off);
    default: return hipErrorInvalidValue;
  }
}

namespace {
template <int CPG>
hipError_t set_attrs_cpg() {
  hipError_t e;
  const int lim = 160 * 1024;
  if ((e = hipFuncSetAttribute(reinterpret_cast<const void*>(convpos_kernel<float, 1, CPG>), hipFuncAttributeMaxDynamicSharedMemorySize, lim)) != hipSuccess) return e;
  if ((e = hipFuncSetAttribute(reinterpret_cast<const void*>(convpos_kernel<f16, 1, CPG>), hipFuncAttributeMaxDynamicSharedMemorySize, lim)) != hipSuccess) return e;
  if ((e = hipFuncSetAttribute(reinterpret_cast<const void*>(convpos_kernel<f16, 2, CPG>), hipFuncAttributeMaxDynamicSharedMemorySize, lim)) != hipSuccess) return e;
  return hipSuccess;
}
}  // namespace
hipError_t init_convpos_kernels() {
  hipError_t e;
  if ((e = hipFuncSetAttribute(reinterpret_cast<const void*>(convpos_mx_kernel<64>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)) != hipSuccess) return e;
  if ((e = set_attrs_cpg<16>()) != hipSuccess) return e;
  if ((e = set_attrs_cpg<32>()) != hipSuccess) return e;
  if ((e = set_attrs_cpg<48>()) != hipSuccess) return e;
  return set_attrs_cpg<64>();
}
