// gemm.h — LDS-tiled MFMA GEMM for gfx950:  out[M,N] = epilogue( A[M,K] . W[N,K]^T )
//
// Both operands are K-contiguous (torch nn.Linear weight layout [out, in]).  A k-tile is ONE 128-byte line per row
// (64 halves, 32 floats, or — fp16x3 — 32 k-elements as [32 hi | 32 lo] halves: the "packed" operand layout, written
// by every producer of an fp16x3 operand), so global reads are whole cache lines: half-line (64 B) segments reach only
// ~60 % of the L2->CU rate on MI355X (tools/probes/l2stride.hip).  Tiles are staged global -> registers -> LDS,
// double-buffered in LDS, with the loads of tile t+2 in flight during the MFMAs of tile t.  The LDS image is
// unpadded [rows][128 B] with the 16-byte chunk index XOR-swizzled by (row >> 1) & 7: the 32-row ds_read_b128 fragment
// reads and the 8-lane ds_write_b128 groups are then both bank-conflict free, and the image stays lane-linear
// (what a direct-to-LDS load needs).  WGM x WGN waves; each wave owns TM x TN tiles of 32x32
// (v_mfma_f32_32x32x16_f16 or 4x v_mfma_f32_32x32x2_f32).  The weight axis is the MFMA "A" operand (accumulator
// rows) so a lane owns 4 consecutive output channels of one activation row -> 16-byte epilogue accesses.
//
// NSPLIT == 3: fp16 hi/lo split operands, acc += A_hi.W_hi + A_lo.W_hi + A_hi.W_lo (~fp32 accuracy).
#pragma once
#include "common.h"

struct GemmCore {
  const void* A;      // [M, K] activations; fp16x3: packed hi/lo rows ([K/32][32 hi | 32 lo], lda >= 2K)
  const void* A_lo;   // unused (kept for call-site symmetry)
  const void* W;      // [N, K] weights; fp16x3: packed like A
  const void* W_lo;
  int64_t lda, ldw;   // row strides in elements (of the stored type: halves / floats)
  int64_t strideA, strideW;      // blockIdx.z batch strides in elements
  int M, N, K;
  int a_rows;         // rows of A that exist (<= M): rows beyond are read as zero
  int w_rows;         // rows of W that exist (<= N)
  int group_m;        // tile rasterisation: row-tiles per group (0/1 = channel tiles fastest over the whole grid), see gemm_kernel
  // Per-output-channel factors that undo the conditioning of W (null = none): the half-precision copies of a weight matrix hold
  // W[n, :] * 2^e[n], e[n] chosen so that the row's largest entry sits near 2^13 (engine finalize, csrc/api.cpp `carve`) — a row of 1e-5
  // entries would otherwise be fp16 SUBNORMALS (11 bits, no `lo` part at all).  Every kernel multiplies its accumulators by
  // w_alpha[n] = 2^-e[n] before the epilogue sees them: exact (powers of two), so results equal those of the unconditioned weights up to
  // the rounding the conditioning removes.
  const float* w_alpha;
};

// acc (4 consecutive output channels n .. n+3) with the weight conditioning undone
__device__ __forceinline__ float4 unscale4(const GemmCore& g, int n, float4 v) {
  if (g.w_alpha) {
    const float4 a = *reinterpret_cast<const float4*>(g.w_alpha + n);
    v.x *= a.x; v.y *= a.y; v.z *= a.z; v.w *= a.w;
  }
  return v;
}

// Generic store epilogue:
//   v = alpha * acc + bias[n];  v = act(v);  v *= colscale[n];
//   if (rowmask && !rowmask[m] && mask_mode == 1) v = 0;
//   if (out2) out2[m,n] = v + res2[m,n];
//   if (res) v += res[m,n];
//   if (rowmask && !rowmask[m] && mask_mode == 2) v = 0;
//   out32[m,n] = v;  out16(_lo)[m,n] = split(v)
struct EpiStore {
  float alpha;
  int act;
  const float* bias;
  const float* colscale;
  const float* res;
  int64_t ldres;
  const float* res2;
  float* out2;
  const uint8_t* rowmask;
  int mask_mode;
  int64_t smask;      // batch stride of rowmask: row m of batch z reads rowmask[(z / zdiv) * smask + m] (0 = shared by all batches)
  float* out32;
  f16* out16;
  f16* out16_lo;
  int64_t ldo;
  int pk16;           // out16/out16_lo are a packed fp16x3 operand (out16_lo == out16 + 32, row stride ldo16)
  int64_t ldo16;      // row stride of out16 (0 = ldo)
  int64_t so1_16, so2_16;  // batch offsets of out16 (used when ldo16 != 0 and zdiv != 0; so2_16 is a k index, packed like n)
  // batch (blockIdx.z) addressing of the outputs/residual: off = (z / zdiv) * so1 + (z % zdiv) * so2
  int zdiv;
  int64_t so1, so2;

  __device__ __forceinline__ void operator()(int m, int n, float4 v, int z) const {
    float x[4] = {v.x, v.y, v.z, v.w};
    if (bias) {
      const float4 b = *reinterpret_cast<const float4*>(bias + n);
      x[0] = alpha * x[0] + b.x; x[1] = alpha * x[1] + b.y; x[2] = alpha * x[2] + b.z; x[3] = alpha * x[3] + b.w;
    } else {
      x[0] *= alpha; x[1] *= alpha; x[2] *= alpha; x[3] *= alpha;
    }
    if (act != ACT_NONE) {
#pragma unroll
      for (int e = 0; e < 4; ++e) x[e] = apply_act(act, x[e]);
    }
    if (colscale) {
      const float4 c = *reinterpret_cast<const float4*>(colscale + n);
      x[0] *= c.x; x[1] *= c.y; x[2] *= c.z; x[3] *= c.w;
    }
    const bool dead = rowmask && !rowmask[(zdiv ? (int64_t)(z / zdiv) * smask : 0) + m];
    if (dead && mask_mode == 1) { x[0] = x[1] = x[2] = x[3] = 0.f; }
    const int64_t zoff = zdiv ? (int64_t)(z / zdiv) * so1 + (int64_t)(z % zdiv) * so2 : 0;
    const int64_t o = zoff + (int64_t)m * ldo + n;
    if (out2) {
      const float4 r2 = *reinterpret_cast<const float4*>(res2 + o);
      *reinterpret_cast<float4*>(out2 + o) = make_float4(x[0] + r2.x, x[1] + r2.y, x[2] + r2.z, x[3] + r2.w);
    }
    if (res) {
      const float4 r = *reinterpret_cast<const float4*>(res + zoff + (int64_t)m * ldres + n);
      x[0] += r.x; x[1] += r.y; x[2] += r.z; x[3] += r.w;
    }
    if (dead && mask_mode == 2) { x[0] = x[1] = x[2] = x[3] = 0.f; }
    if (out32) *reinterpret_cast<float4*>(out32 + o) = make_float4(x[0], x[1], x[2], x[3]);
    if (out16) {
      f16x4 hi, lo;
#pragma unroll
      for (int e = 0; e < 4; ++e) { f16 h, l; split_f16(x[e], h, l); hi[e] = h; lo[e] = l; }
      const int64_t o16 = ldo16 ? (zdiv ? (int64_t)(z / zdiv) * so1_16 : 0) + (int64_t)m * ldo16 + pk_off((zdiv ? (int)((z % zdiv) * so2_16) : 0) + n, pk16)
                                : o;
      *reinterpret_cast<f16x4*>(out16 + o16) = hi;
      if (out16_lo) *reinterpret_cast<f16x4*>(out16_lo + o16) = lo;
    }
  }
};

// QKV epilogue: bias, rotary embedding on (2i,2i+1) pairs of q and k (x_transformers
// apply_rotary_pos_emb, reference call sites model/modules.py:503-509), q * scale, scatter to
// per-head layouts.  n in [0, 3*inner): n / inner selects q|k|v.
//   half mode  (q16 != null): q16/k16 [B'*H, nseq, dh] f16, vt16 [B'*H, dh, ldvt] f16 (V transposed for the
//                              flash kernel's V^T tiles); optional *_lo planes (fp16 hi/lo split)
//   float mode (q32 != null): q32/k32 [B'*H, nseq, dh] f32, vt32 [B'*H, dh, ldvt] f32 (V transposed)
template <bool B, typename T, typename F> struct f5_conditional { typedef T type; };
template <typename T, typename F> struct f5_conditional<false, T, F> { typedef F type; };

template <bool FAST>  // FAST: the division-free, 32-bit index path (a kernel of its own: see EpiQKVFast below)
struct EpiQKVT {
  const float* bias;    // [3*inner]
  const float* rope_cs; // [nseq, dh/2, 2] (cos, sin) fp32
  int nseq, heads, dh;
  int pe_heads;         // -1 = all
  float qscale;
  f16 *q16, *k16, *vt16;
  f16 *q16_lo, *k16_lo, *vt16_lo;
  float *q32, *k32, *vt32;
  int64_t ldvt;
  int slab_n, pos_off;  // MMDiT joint attention: the per-(batch', head) slabs hold slab_n tokens and this GEMM's sequences of nseq rows
                        // land at token pos_off + pos (rope still uses pos); slab_n == 0 means slab_n = nseq, pos_off = 0
  int qk_raw;           // qk_norm variant: q and k leave as fp32 rows holding Wx + b only (q32/k32); qk_norm_rope_kernel finishes them

  // Fast index path (EpiQKVFast, chosen by launch_gemm_qkv when epi_qkv_prepare() accepts: dh a power of two and every element index
  // of the slabs within 31 bits — all shipped configurations): the general path spends ~400 quarter-rate integer multiplies and 24
  // integer divisions per 128x64 tile on (n / inner, c / dh, m / nseq) and 64-bit offset products — more VALU time than the tile's
  // MFMAs.  Here: `which` by two comparisons, head / channel by shift and mask, sequence / position by an invariant multiplier, 32-bit
  // offsets.  Same indices, same arithmetic on the values (tests/hipemu/qkv_index_check.cpp compares every store of both paths byte for
  // byte at the full sizes).  A compile-time choice: both paths inlined into one epilogue stopped the big tiles' epilogue loops from
  // unrolling (accumulators in scratch; caught by tests/test_isa_hazards.py).
  // Packed rows (engine option "packed_rows"): row m of the GEMM is token (rowinfo[m] & 0xffff) of sequence (rowinfo[m] >> 16) instead of
  // token m % nseq of sequence m / nseq — the block GEMMs then run over the VALID rows of a ragged batch only (null = padded layout).
  const uint32_t* rowinfo;
  int nslab;            // packed rows: number of sequences the q / k / V^T slabs hold (bounds of the pipelined kernel's buffer descriptors)
  int mx_qk;            // the lo planes of q and k receive MX-fp6 P words (gemm_pp.h PpEpiQKV::mx_qk): pipelined kernels only — a launch that
                        // cannot take one fails (ask gemm_qkv_takes_pp first)
  int fast;             // set by epi_qkv_prepare (host bookkeeping; the kernels do not read it)
  int inner_;           // heads * dh
  int dh_shift;         // log2(dh)
  uint32_t nseq_magic;  // m / nseq == mulhi(m, nseq_magic) >> nseq_shift for 0 <= m < 2^31
  int nseq_shift;

  __device__ __forceinline__ void store(int which, int hh, int d, int bp, int pos, float (&x)[4]) const {
    using I = typename f5_conditional<FAST, int, int64_t>::type;  // offset arithmetic: 32-bit on the fast path
    auto mul_dh = [&](I t) -> I { return FAST ? (I)(t << dh_shift) : (I)(t * dh); };
    const int sn = slab_n ? slab_n : nseq, tokp = pos_off + pos;  // slab length, token index inside the slab
    if (qk_raw && which < 2) {
      float* dst = (which == 0 ? q32 : k32) + (mul_dh(((I)bp * heads + hh) * sn + tokp) + d);
      *reinterpret_cast<float4*>(dst) = make_float4(x[0], x[1], x[2], x[3]);
      return;
    }
    if (which < 2 && (pe_heads < 0 || hh < pe_heads)) {
      const float4 cs = *reinterpret_cast<const float4*>(rope_cs + ((FAST ? (I)((I)pos << (dh_shift - 1)) : (I)pos * (dh / 2)) + d / 2) * 2);
      const float a0 = x[0] * cs.x - x[1] * cs.y, a1 = x[1] * cs.x + x[0] * cs.y;
      const float a2 = x[2] * cs.z - x[3] * cs.w, a3 = x[3] * cs.z + x[2] * cs.w;
      x[0] = a0; x[1] = a1; x[2] = a2; x[3] = a3;
    }
    if (which == 0) { x[0] *= qscale; x[1] *= qscale; x[2] *= qscale; x[3] *= qscale; }
    const I bh = (I)bp * heads + hh;
    const I ldv = (I)ldvt;
    if (q16) {
      f16x4 hv, lv;
#pragma unroll
      for (int e = 0; e < 4; ++e) { f16 h, l; split_f16(x[e], h, l); hv[e] = h; lv[e] = l; }
      if (which < 2) {
        const I off = mul_dh(bh * sn + tokp) + d;
        *reinterpret_cast<f16x4*>((which == 0 ? q16 : k16) + off) = hv;
        if (q16_lo) *reinterpret_cast<f16x4*>((which == 0 ? q16_lo : k16_lo) + off) = lv;
      } else {
        const I off = (mul_dh(bh) + d) * ldv + tokp;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          vt16[off + e * ldv] = hv[e];
          if (vt16_lo) vt16_lo[off + e * ldv] = lv[e];
        }
      }
    } else {
      if (which < 2) {
        float* dst = (which == 0 ? q32 : k32) + (mul_dh(bh * sn + tokp) + d);
        *reinterpret_cast<float4*>(dst) = make_float4(x[0], x[1], x[2], x[3]);
      } else {
        float* dst = vt32 + ((mul_dh(bh) + d) * ldv + tokp);
        dst[0] = x[0]; dst[ldv] = x[1]; dst[2 * ldv] = x[2]; dst[3 * ldv] = x[3];
      }
    }
  }

  __device__ __forceinline__ void operator()(int m, int n, float4 v, int /*z*/) const {
    const float4 b = *reinterpret_cast<const float4*>(bias + n);
    float x[4] = {v.x + b.x, v.y + b.y, v.z + b.z, v.w + b.w};
    if constexpr (FAST) {
      const int which = (n >= inner_ ? 1 : 0) + (n >= 2 * inner_ ? 1 : 0);
      const int c = n - (which == 0 ? 0 : which == 1 ? inner_ : 2 * inner_);
      const int hh = c >> dh_shift, d = c & (dh - 1);
      int bp, pos;
      if (rowinfo) { const uint32_t ri = rowinfo[m]; bp = (int)(ri >> 16); pos = (int)(ri & 0xffffu); }
      else { bp = (int)((uint32_t)(((uint64_t)(uint32_t)m * nseq_magic) >> 32) >> nseq_shift); pos = m - bp * nseq; }
      store(which, hh, d, bp, pos, x);
    } else {
      const int inner = heads * dh;
      const int which = n / inner;
      const int c = n - which * inner;
      const int hh = c / dh, d = c - hh * dh;
      int bp = m / nseq, pos = m - bp * nseq;
      if (rowinfo) { const uint32_t ri = rowinfo[m]; bp = (int)(ri >> 16); pos = (int)(ri & 0xffffu); }
      store(which, hh, d, bp, pos, x);
    }
  }
};
typedef EpiQKVT<false> EpiQKV;      // what the host code fills in
typedef EpiQKVT<true> EpiQKVFast;   // same fields, same layout; launch_gemm_qkv converts when epi_qkv_prepare() accepts

// Host side of the fast index path: fill the derived fields for a GEMM of M rows, or leave fast = 0 when a precondition fails.
inline void epi_qkv_prepare(EpiQKV& e, int M) {
  e.fast = 0;
  const int dh = e.dh, nseq = e.nseq;
  if (dh < 2 || (dh & (dh - 1)) || nseq < 2 || M <= 0 || e.heads <= 0) return;
  const int64_t bp_max = e.rowinfo ? e.nslab : (M + nseq - 1) / nseq, sn = e.slab_n ? e.slab_n : nseq;
  const int64_t lim = (int64_t)1 << 31;
  // largest offsets any store can form: the q / k slabs, the V^T slab (one row past the last channel for the e * ldvt steps), the rope table
  if (bp_max * e.heads * sn * dh >= lim || (bp_max * e.heads * dh + 4) * e.ldvt + sn + e.pos_off >= lim || (int64_t)nseq * dh >= lim) return;
  if ((int64_t)3 * e.heads * dh >= lim / 2) return;
  int s = 0;
  while (((int64_t)2 << s) <= nseq) ++s;  // s = floor(log2(nseq))
  if (((int64_t)1 << s) == nseq) {        // power of two: mulhi(m, 2^31) = m >> 1
    e.nseq_magic = 0x80000000u;
    e.nseq_shift = s - 1;
  } else {                                // ceil(2^(32+s) / nseq) lies in (2^31, 2^32); exact for every m < 2^31 (error term < nseq < 2^(s+1))
    e.nseq_magic = (uint32_t)((((uint64_t)1 << (32 + s)) + (uint64_t)nseq - 1) / (uint64_t)nseq);
    e.nseq_shift = s;
  }
  e.inner_ = e.heads * dh;
  e.dh_shift = 0;
  while ((1 << e.dh_shift) < dh) ++e.dh_shift;
  e.fast = 1;
}

constexpr int GEMM_KTB = 128;  // bytes of one operand row per k-tile = one cache line

template <typename T, int NSPLIT, int TM, int TN, int WGM = 2, int WGN = 2>
constexpr int gemm_lds_bytes() {
  return 2 * (32 * WGM * TM + 32 * WGN * TN) * GEMM_KTB;
}

// ABL (microbenchmark ablations only): bit0 = no global loads in the k-loop, bit1 = no LDS stores, bit2 = no MFMAs, bit3 = no epilogue.
template <typename T, int NSPLIT, int TM, int TN, typename Epi, int WGM = 2, int WGN = 2, int ABL = 0>
__global__ __launch_bounds__(64 * WGM * WGN) void gemm_kernel(GemmCore g, Epi epi) {
  constexpr int NT = 64 * WGM * WGN;
  constexpr int BM = 32 * WGM * TM, BN = 32 * WGN * TN;
  constexpr int NPL = (NSPLIT == 3) ? 2 : 1;
  constexpr int CPR = GEMM_KTB / 16;                        // 8 chunks of 16 bytes per tile row
  constexpr int KSTEPS = NPL == 2 ? 2 : 4;                  // MFMA k-steps per tile (fp16x3: hi chunks 0-3, lo chunks 4-7)
  constexpr int CA = BM * CPR / NT, CW = BN * CPR / NT;     // chunks per thread
  constexpr int TILE_A = BM * GEMM_KTB, TILE_W = BN * GEMM_KTB;
  constexpr int STAGE = TILE_A + TILE_W;
  static_assert(CA >= 1 && CW >= 1 && CA * NT == BM * CPR && CW * NT == BN * CPR, "tile does not split evenly over the threads");
  F5_DYN_LDS(char, smem);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave % WGM, wn = wave / WGM;
  // Tile order (speed only): blockIdx.x -> XCD = id % 8 (observed dispatch) gets a contiguous run of tiles, channel tiles
  // fastest, so the workgroups resident on one XCD share a few activation row-panels and sweep the weight panel through
  // that XCD's L2; each activation panel is fetched from HBM/MALL by one XCD only.  Bijective for any tile count.
  const int z = blockIdx.z;
  int m0, n0;
  {
    const int nt = (g.N + BN - 1) / BN, nwg = gridDim.x;
    const int bid = blockIdx.x, q8 = nwg >> 3, r8 = nwg & 7, xcd = bid & 7, slot = bid >> 3;
    const int L = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + slot;
    int mt, ntile;
    if (g.group_m > 1) {  // groups of group_m row-tiles, row-tile fastest inside a group: the ~64 workgroups resident on an XCD then
                          // cover group_m row panels x a few channel panels, so the row panels stay in that XCD's L2 for the whole sweep
      const int mtt = (g.M + BM - 1) / BM, per = g.group_m * nt;
      const int grp = L / per, first = grp * g.group_m, gsz = min(g.group_m, mtt - first), within = L - grp * per;
      ntile = within / gsz;
      mt = first + (within - ntile * gsz);
    } else {
      mt = L / nt;
      ntile = L - mt * nt;
    }
    m0 = mt * BM;
    n0 = ntile * BN;
  }

  // Operands are read through buffer descriptors: a 16-byte chunk outside the matrix (row >= rows, k >= K) gets its
  // offset forced out of range and the hardware returns zeros — the loads are unconditional and branch-free, so nothing
  // waits on them until the matching LDS store (a predicated `ok ? load : 0` makes hipcc branch around every load and
  // drain vmcnt(0) before re-initialising the destination registers).
  const int kbytes = g.K * (int)sizeof(T) * NPL;  // bytes of one operand row
  const uint32_t a_bytes = (uint32_t)((int64_t)(g.a_rows - 1) * g.lda * (int64_t)sizeof(T) + kbytes);
  const uint32_t w_bytes = (uint32_t)((int64_t)(g.w_rows - 1) * g.ldw * (int64_t)sizeof(T) + kbytes);
  const BufRsrc Ar = make_rsrc(reinterpret_cast<const T*>(g.A) + (int64_t)z * g.strideA, a_bytes);
  const BufRsrc Wr = make_rsrc(reinterpret_cast<const T*>(g.W) + (int64_t)z * g.strideW, w_bytes);
  // Thread t owns LDS chunk slots t, t + NT, ... (linear image); slot (row, pc) holds the row's logical chunk pc ^ swz(row).
  uint32_t a_off[CA], w_off[CW];
  int a_c[CA], w_c[CW];
#pragma unroll
  for (int i = 0; i < CA; ++i) {
    const int c = tid + i * NT, row = c / CPR, lc = (c % CPR) ^ ((row >> 1) & 7);
    a_c[i] = lc * 16;
    a_off[i] = (m0 + row) < g.a_rows ? (uint32_t)((int64_t)(m0 + row) * g.lda * (int64_t)sizeof(T) + lc * 16) : OOB_ROW;
  }
#pragma unroll
  for (int i = 0; i < CW; ++i) {
    const int c = tid + i * NT, row = c / CPR, lc = (c % CPR) ^ ((row >> 1) & 7);
    w_c[i] = lc * 16;
    w_off[i] = (n0 + row) < g.w_rows ? (uint32_t)((int64_t)(n0 + row) * g.ldw * (int64_t)sizeof(T) + lc * 16) : OOB_ROW;
  }

  // Two register sets: the global loads of tile t+2 are issued at the top of iteration t and written to LDS at the end of
  // iteration t+1, so every load has two full compute phases of flight time.  Static set indexing needs the k-loop
  // unrolled by two.  A tile index past the end reads zeros (k >= K) and is never stored.
  uint4 ra0[CA], rw0[CW], ra1[CA], rw1[CW];

  auto load_global = [&](int kt, uint4 (&ra)[CA], uint4 (&rw)[CW]) {
    const int kb = kt * GEMM_KTB;
#pragma unroll
    for (int i = 0; i < CA; ++i) ra[i] = buffer_load_b128(Ar, (kb + a_c[i]) < kbytes ? a_off[i] + (uint32_t)kb : OOB_OFF);
#pragma unroll
    for (int i = 0; i < CW; ++i) rw[i] = buffer_load_b128(Wr, (kb + w_c[i]) < kbytes ? w_off[i] + (uint32_t)kb : OOB_OFF);
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

  // fragment addressing: lane (i = lane & 31, hi = lane >> 5) reads logical chunk 2 ks + hi (+4 for the lo plane) of row i
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
          if constexpr (ABL & 4) {  // keep the fragment reads alive without the MFMAs
#pragma unroll
            for (int p = 0; p < NPL; ++p)
              asm volatile("" ::"v"(fw[p][i].u.x), "v"(fw[p][i].u.y), "v"(fw[p][i].u.z), "v"(fw[p][i].u.w), "v"(fa[p][j].u.x), "v"(fa[p][j].u.y),
                           "v"(fa[p][j].u.z), "v"(fa[p][j].u.w));
          } else {
            Mma32<T>::mma(acc[j][i], fw[0][i], fa[0][j]);
            if constexpr (NPL == 2) {
              Mma32<T>::mma(acc[j][i], fw[0][i], fa[1][j]);  // W_hi . A_lo
              Mma32<T>::mma(acc[j][i], fw[1][i], fa[0][j]);  // W_lo . A_hi
            }
          }
        }
    }
  };

  const int nkt = (kbytes + GEMM_KTB - 1) / GEMM_KTB;
  load_global(0, ra0, rw0);
  load_global(1, ra1, rw1);
  store_lds(0, ra0, rw0);
  __syncthreads();

  int kt = 0;
  for (; kt + 1 < nkt; kt += 2) {
    // even phase: tile kt in stage 0; set 1 holds tile kt+1 (in flight); set 0 is free
    if constexpr (!(ABL & 1)) load_global(kt + 2, ra0, rw0);
    __builtin_amdgcn_sched_barrier(0);  // keep the loads at the top of the phase (hipcc otherwise sinks them below the LDS stores)
    compute(0);
    if constexpr (!(ABL & 2)) store_lds(1, ra1, rw1);
    __syncthreads();
    // odd phase: tile kt+1 in stage 1; set 0 holds tile kt+2; set 1 is free
    if constexpr (!(ABL & 1)) load_global(kt + 3, ra1, rw1);
    __builtin_amdgcn_sched_barrier(0);
    compute(1);
    if constexpr (!(ABL & 2)) store_lds(0, ra0, rw0);
    __syncthreads();
  }
  if (kt < nkt) compute(0);  // odd tile count: the last tile sits in stage 0

  // epilogue: lane owns row m = .. + (lane & 31), channels n = .. + 8q + 4*(lane>>5) + 0..3
#pragma unroll
  for (int j = 0; j < TM; ++j) {
    const int m = m0 + wm * 32 * TM + j * 32 + (lane & 31);
    if (m >= g.M) continue;
#pragma unroll
    for (int i = 0; i < TN; ++i) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int n = n0 + wn * 32 * TN + i * 32 + 8 * q + 4 * (lane >> 5);
        if constexpr (ABL & 8) {  // no epilogue: keep the accumulators alive
          asm volatile("" ::"v"(acc[j][i][4 * q]), "v"(acc[j][i][4 * q + 1]), "v"(acc[j][i][4 * q + 2]), "v"(acc[j][i][4 * q + 3]));
        } else {
          if (n < g.N) epi(m, n, unscale4(g, n, make_float4(acc[j][i][4 * q], acc[j][i][4 * q + 1], acc[j][i][4 * q + 2], acc[j][i][4 * q + 3])), z);
        }
      }
    }
  }
}

#ifndef F5_HIPEMU  // inline asm: not part of the host-shim build
// ---------------------------------------------------------------------------------------------------------------------
// Direct-to-LDS variant: the k-tiles are fetched by LDS-DMA (buffer_load_dwordx4 ... lds: the wave writes 64 x 16 B to a
// lane-linear 1 KiB LDS run, no VGPR round trip, no ds_write) into a ring of NS stages.  The swizzle lives on the per-lane
// SOURCE offset (the LDS image is the same as gemm_kernel's).  Tile t+2 is issued at the top of phase t into the stage that
// phase t-1 finished reading (one barrier earlier), tile t+1 is awaited with a COUNTED vmcnt before the single barrier of the
// phase, so two tiles stay in flight across it.  Raw s_barrier (no __syncthreads: its fence would drain vmcnt to 0).
// ---------------------------------------------------------------------------------------------------------------------
template <typename T, int NSPLIT, int TM, int TN, int WGM = 2, int WGN = 2, int NS = 3>
constexpr int gemm_glds_lds_bytes() {
  return NS * (32 * WGM * TM + 32 * WGN * TN) * GEMM_KTB;
}

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
__device__ __forceinline__ void wg_barrier() { asm volatile("s_barrier" ::: "memory"); }

template <typename T, int NSPLIT, int TM, int TN, typename Epi, int WGM = 2, int WGN = 2, int NS = 3, int PRIO = 0>
__global__ __launch_bounds__(64 * WGM * WGN) void gemm_glds_kernel(GemmCore g, Epi epi) {
  constexpr int NT = 64 * WGM * WGN;
  constexpr int BM = 32 * WGM * TM, BN = 32 * WGN * TN;
  constexpr int NPL = (NSPLIT == 3) ? 2 : 1;
  constexpr int CPR = GEMM_KTB / 16;
  constexpr int KSTEPS = NPL == 2 ? 2 : 4;
  constexpr int CA = BM * CPR / NT, CW = BN * CPR / NT;  // DMA pieces per thread per tile
  constexpr int LPT = CA + CW;
  constexpr int TILE_A = BM * GEMM_KTB, TILE_W = BN * GEMM_KTB;
  constexpr int STAGE = TILE_A + TILE_W;
  static_assert(CA >= 1 && CW >= 1 && CA * NT == BM * CPR && CW * NT == BN * CPR, "tile does not split evenly over the threads");
  static_assert(NS == 3 || NS == 2, "ring depth 3 (two tiles in flight) or 2 (one tile in flight, for tiles whose stage is 64 KB)");
  F5_DYN_LDS(char, smem);

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave % WGM, wn = wave / WGM;
  const int z = blockIdx.z;
  int m0, n0;
  {
    const int nt = (g.N + BN - 1) / BN, nwg = gridDim.x;
    const int bid = blockIdx.x, q8 = nwg >> 3, r8 = nwg & 7, xcd = bid & 7, slot = bid >> 3;
    const int L = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + slot;
    int mt, ntile;
    if (g.group_m > 1) {  // groups of group_m row-tiles, row-tile fastest inside a group: the ~64 workgroups resident on an XCD then
                          // cover group_m row panels x a few channel panels, so the row panels stay in that XCD's L2 for the whole sweep
      const int mtt = (g.M + BM - 1) / BM, per = g.group_m * nt;
      const int grp = L / per, first = grp * g.group_m, gsz = min(g.group_m, mtt - first), within = L - grp * per;
      ntile = within / gsz;
      mt = first + (within - ntile * gsz);
    } else {
      mt = L / nt;
      ntile = L - mt * nt;
    }
    m0 = mt * BM;
    n0 = ntile * BN;
  }
  const int kbytes = g.K * (int)sizeof(T) * NPL;
  const uint32_t a_bytes = (uint32_t)((int64_t)(g.a_rows - 1) * g.lda * (int64_t)sizeof(T) + kbytes);
  const uint32_t w_bytes = (uint32_t)((int64_t)(g.w_rows - 1) * g.ldw * (int64_t)sizeof(T) + kbytes);
  const BufRsrc Ar = make_rsrc(reinterpret_cast<const T*>(g.A) + (int64_t)z * g.strideA, a_bytes);
  const BufRsrc Wr = make_rsrc(reinterpret_cast<const T*>(g.W) + (int64_t)z * g.strideW, w_bytes);
  uint32_t a_off[CA], w_off[CW];
  int a_c[CA], w_c[CW];
#pragma unroll
  for (int i = 0; i < CA; ++i) {
    const int c = tid + i * NT, row = c / CPR, lc = (c % CPR) ^ ((row >> 1) & 7);
    a_c[i] = lc * 16;
    a_off[i] = (m0 + row) < g.a_rows ? (uint32_t)((int64_t)(m0 + row) * g.lda * (int64_t)sizeof(T) + lc * 16) : OOB_ROW;
  }
#pragma unroll
  for (int i = 0; i < CW; ++i) {
    const int c = tid + i * NT, row = c / CPR, lc = (c % CPR) ^ ((row >> 1) & 7);
    w_c[i] = lc * 16;
    w_off[i] = (n0 + row) < g.w_rows ? (uint32_t)((int64_t)(n0 + row) * g.ldw * (int64_t)sizeof(T) + lc * 16) : OOB_ROW;
  }
  // LDS-DMA of one k-tile: piece i of this wave lands at stage + (wave * 64 + i * NT) * 16 + lane * 16
  auto issue = [&](int kt, int stage) {
    const int kb = kt * GEMM_KTB;
    char* base = smem + stage * STAGE + wave * 1024;
#pragma unroll
    for (int i = 0; i < CA; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(Ar, (__attribute__((address_space(3))) void*)(base + i * NT * 16), 16,
                                               (int)((kb + a_c[i]) < kbytes ? a_off[i] + (uint32_t)kb : OOB_OFF), 0, 0, 0);
#pragma unroll
    for (int i = 0; i < CW; ++i)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(Wr, (__attribute__((address_space(3))) void*)(base + TILE_A + i * NT * 16), 16,
                                               (int)((kb + w_c[i]) < kbytes ? w_off[i] + (uint32_t)kb : OOB_OFF), 0, 0, 0);
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

  // Fragment reads are inline asm: hipcc drains vmcnt(0) before any ds_read it can see while an LDS-DMA is in flight (it cannot
  // prove the read does not alias the DMA target), which would serialise the whole ring.  The asm reads are ordered by hand:
  // fragments of k-step ks+1 are issued before the MFMAs of k-step ks, `s_waitcnt lgkmcnt(0)` + sched_barrier precede their use.
  const uint32_t lds0 = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
  typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
  auto lds_read = [&](uint32_t addr) -> uint4 {
    u32x4 v;
    asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(addr));
    return make_uint4(v[0], v[1], v[2], v[3]);
  };
  auto read_frags = [&](uint32_t sA, uint32_t sW, int ks, Frag (&fa)[NPL][TM], Frag (&fw)[NPL][TN]) {
#pragma unroll
    for (int p = 0; p < NPL; ++p) {
#pragma unroll
      for (int j = 0; j < TM; ++j) fa[p][j].u = lds_read(sA + j * 32 * GEMM_KTB + foff[p][ks]);
#pragma unroll
      for (int i = 0; i < TN; ++i) fw[p][i].u = lds_read(sW + i * 32 * GEMM_KTB + foff[p][ks]);
    }
  };
  auto mma_step = [&](const Frag (&fa)[NPL][TM], const Frag (&fw)[NPL][TN]) {
    if constexpr (PRIO & 1) __builtin_amdgcn_s_setprio(1);  // favour the wave that is feeding the matrix pipe over its SIMD partner's loads
#pragma unroll
    for (int j = 0; j < TM; ++j)
#pragma unroll
      for (int i = 0; i < TN; ++i) {
        Mma32<T>::mma(acc[j][i], fw[0][i], fa[0][j]);
        if constexpr (NPL == 2) {
          Mma32<T>::mma(acc[j][i], fw[0][i], fa[1][j]);
          if constexpr (!(PRIO & 2)) Mma32<T>::mma(acc[j][i], fw[1][i], fa[0][j]);  // PRIO bit 1 (microbenchmark only): 2 of the 3 products
        }
      }
    if constexpr (PRIO & 1) __builtin_amdgcn_s_setprio(0);
  };
  auto compute = [&](int stage) {
    const uint32_t sA = lds0 + stage * STAGE + (wm * 32 * TM) * GEMM_KTB;
    const uint32_t sW = lds0 + stage * STAGE + TILE_A + (wn * 32 * TN) * GEMM_KTB;
    Frag fa0[NPL][TM], fw0[NPL][TN], fa1[NPL][TM], fw1[NPL][TN];
    read_frags(sA, sW, 0, fa0, fw0);
#pragma unroll
    for (int ks = 0; ks < KSTEPS; ks += 2) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      read_frags(sA, sW, ks + 1, fa1, fw1);
      mma_step(fa0, fw0);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      if (ks + 2 < KSTEPS) read_frags(sA, sW, ks + 2, fa0, fw0);
      mma_step(fa1, fw1);
    }
  };

  const int nkt = (kbytes + GEMM_KTB - 1) / GEMM_KTB;
  if constexpr (NS == 3) {
    issue(0, 0);
    issue(1, 1);
    wait_vmcnt<LPT>();  // tile 0 landed (this wave's pieces); tile 1 in flight
    wg_barrier();
    int st = 0;  // stage of tile kt
    for (int kt = 0; kt < nkt; ++kt) {
      const int st2 = st == 0 ? 2 : st - 1;  // (st + 2) % 3: the stage phase kt-1 finished reading before the last barrier
      issue(kt + 2, st2);                    // past the end: zeros, never read
      compute(st);
      wait_vmcnt<LPT>();                     // tile kt+1 landed; tile kt+2 stays in flight across the barrier
      wg_barrier();
      st = st == 2 ? 0 : st + 1;
    }
  } else {  // two stages: the phase is long enough (>= 48 MFMAs per wave) to cover one tile's flight
    issue(0, 0);
    wait_vmcnt<0>();
    wg_barrier();
    for (int kt = 0; kt < nkt; ++kt) {
      issue(kt + 1, (kt + 1) & 1);  // the stage phase kt-1 finished reading before the last barrier
      compute(kt & 1);
      wait_vmcnt<0>();              // tile kt+1 landed
      wg_barrier();
    }
  }
  wait_vmcnt<0>();  // drain the two dummy tiles before the workgroup's LDS is released

#pragma unroll
  for (int j = 0; j < TM; ++j) {
    const int m = m0 + wm * 32 * TM + j * 32 + (lane & 31);
    if (m >= g.M) continue;
#pragma unroll
    for (int i = 0; i < TN; ++i) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int n = n0 + wn * 32 * TN + i * 32 + 8 * q + 4 * (lane >> 5);
        if (n < g.N) epi(m, n, unscale4(g, n, make_float4(acc[j][i][4 * q], acc[j][i][4 * q + 1], acc[j][i][4 * q + 2], acc[j][i][4 * q + 3])), z);
      }
    }
  }
}
#endif  // F5_HIPEMU
