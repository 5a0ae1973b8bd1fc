// gemm_pp.h — the pipelined MFMA GEMM of the DiT / UNetT block projections:  epilogue( A[M,K] . W[N,K]^T ),  fp16 or fp16x3 operands.
//
// What round 1's counters said about gemm.h's kernels at these shapes (profiles/r02a_*): the 4-wave 128x64 workgroup reads 1 KB of LDS
// fragments per MFMA and stages through VGPRs with ds_write_b128 (79 B/clk/CU) — more LDS cycles than MFMA cycles per k-tile; its waves
// sit in s_waitcnt / s_barrier 34-45 % of their time and a fifth of every launch is an epilogue that pays one L2 round trip and 2-8
// narrow stores per 4 outputs.  This kernel is built the other way round:
//   * k-tiles (one 128-byte line per operand row, the layout of gemm.h) arrive by LDS-DMA (`buffer_load ... lds`) into a ring of NS
//     stages; the k offset rides in the instruction's scalar offset, so the loop has no per-lane address arithmetic;
//   * ONE s_barrier per k-tile, placed before the tile's LAST MFMA group: by then every wave has issued and retired all its fragment
//     reads of the tile, so the stage is refilled right behind the barrier (tile t+NS) and the next tile's first fragments are read
//     under the last MFMA group — neither the barrier, nor the DMA wait (counted vmcnt: NS-2 tiles stay in flight across it), nor the
//     first ds_reads of a tile are exposed;
//   * fragments are double-buffered per "slot" (JG activation tiles x all weight tiles of one 16-wide k-step), so LDS latency hides
//     behind the previous slot's MFMAs with 48-64 fragment VGPRs instead of 96;
//   * wave tiles up to 128x64 / 96x96 (0.5-0.67 KB of fragments per MFMA), workgroup tiles chosen so that a launch is a whole number
//     of rounds of the 256 CUs (192-row tiles at B = 1: 15 x 16 = 240 workgroups);
//   * epilogues work on a whole wave tile: bias / gate / rope operands are fetched up front, rows outside M are dropped by the buffer
//     descriptor (no branches), outputs leave as 16-byte stores (v_permlane32_swap pairs the two half-waves' 4-channel groups).
// Fragment and accumulator layouts are those of gemm.h (weights on the MFMA "A" side: a lane owns 4 consecutive channels of one row).
#pragma once
#include <type_traits>
#include <utility>

#include "gemm.h"

// ---- primitives: amdgcn instructions on the GPU, plain memory operations under the host shim (tests/hipemu) ------------------------
namespace pp {
#ifdef F5_HIPEMU
template <int N>
inline void wait_vmcnt() {}  // the shim's LDS-DMA is synchronous (what it CAN show: a refill racing the reads of the stage it overwrites)
inline void wg_barrier() { __syncthreads(); }
inline uint32_t lds_base(char* smem) { return (uint32_t)(smem - hipemu::dyn_lds()); }
template <int IMM>
inline uint4 lds_read_b128(uint32_t addr) { uint4 v; memcpy(&v, hipemu::dyn_lds() + addr + IMM, 16); return v; }
inline void lds_wait() {}
inline void dma_b128(BufRsrc r, char* lds_dst, uint32_t voff, uint32_t soff) {  // each lane: 16 bytes -> lds_dst + lane * 16
  const auto v = hipemu::raw_buffer_load_b128(r, (int)voff, (int)soff, 0);
  memcpy(lds_dst + 16 * hipemu::blk->cur->lane, &v, 16);
}
inline int uniform(int v) { return v; }
inline void pin() {}
inline void swap32(uint32_t& a, uint32_t& b) { hipemu::permlane32_swap(a, b); }
inline uint32_t xchg1(uint32_t v) { return hipemu::shfl_xor_u32(v, 1); }
inline void store_b128(BufRsrc r, uint32_t off, uint4 v) { hipemu::raw_buffer_store(r, off, &v, 16); }
inline void store_b32(BufRsrc r, uint32_t off, uint32_t v) { hipemu::raw_buffer_store(r, off, &v, 4); }
inline void store_b16(BufRsrc r, uint32_t off, uint16_t v) { hipemu::raw_buffer_store(r, off, &v, 2); }
#else
template <int N>
__device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
__device__ __forceinline__ void wg_barrier() { asm volatile("s_barrier" ::: "memory"); }
__device__ __forceinline__ uint32_t lds_base(char* smem) { return (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)smem; }
// inline asm: hipcc drains vmcnt(0) before any ds_read it can see while an LDS-DMA is in flight (it cannot prove they do not alias)
template <int IMM>
__device__ __forceinline__ uint4 lds_read_b128(uint32_t addr) {
  typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
  u32x4 v;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(IMM));
  return make_uint4(v[0], v[1], v[2], v[3]);
}
__device__ __forceinline__ void lds_wait() {  // every fragment read issued so far has landed; nothing may be scheduled across
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);
}
__device__ __forceinline__ void dma_b128(BufRsrc r, char* lds_dst, uint32_t voff, uint32_t soff) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)lds_dst, 16, (int)voff, (int)soff, 0, 0);
}
__device__ __forceinline__ int uniform(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ void pin() { __builtin_amdgcn_sched_barrier(0); }  // nothing is scheduled across this point
// half exchange: lanes 32-63 of a swap with lanes 0-31 of b (v_permlane32_swap)
__device__ __forceinline__ void swap32(uint32_t& a, uint32_t& b) {
  const auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
  a = r[0];
  b = r[1];
}
// neighbour lane (lane ^ 1) as ONE DPP move, quad_perm [1, 0, 3, 2] (__shfl_xor(v, 1) compiled to a ds_bpermute round trip: 72 of them in the q|k|v kernel)
__device__ __forceinline__ uint32_t xchg1(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xf, 0xf, true); }
__device__ __forceinline__ void store_b128(BufRsrc r, uint32_t off, uint4 v) {
  typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
  u32x4 d = {v.x, v.y, v.z, v.w};
  __builtin_amdgcn_raw_buffer_store_b128(d, r, (int)off, 0, 0);
}
__device__ __forceinline__ void store_b32(BufRsrc r, uint32_t off, uint32_t v) { __builtin_amdgcn_raw_buffer_store_b32(v, r, (int)off, 0, 0); }
__device__ __forceinline__ void store_b16(BufRsrc r, uint32_t off, uint16_t v) { __builtin_amdgcn_raw_buffer_store_b16(v, r, (int)off, 0, 0); }
#endif

__device__ __forceinline__ uint32_t pack2(f16 a, f16 b) {
  union { f16 h[2]; uint32_t u; } x;
  x.h[0] = a;
  x.h[1] = b;
  return x.u;
}
// 4 floats -> 4 halves (hi) and the 4 halves of the remainders (lo), each as two packed dwords.  The pairs are built as VECTORS of two halves:
// gfx950 then converts a pair with one v_cvt_pk_f16_f32 (the union form above compiled to a conversion per half plus a shift and an or)
__device__ __forceinline__ void split4(const float (&x)[4], uint32_t (&hi)[2], uint32_t (&lo)[2]) {
#pragma unroll
  for (int e = 0; e < 4; e += 2) {
    float xa = x[e], xb = x[e + 1];
    pin_f32(xa, xb);  // (common.h: stored halves and remainders off the same fp32 values)
    const f16x2 h = {(f16)xa, (f16)xb};
    const f16x2 l = {(f16)(xa - (float)h[0]), (f16)(xb - (float)h[1])};
    hi[e >> 1] = __builtin_bit_cast(uint32_t, h);
    lo[e >> 1] = __builtin_bit_cast(uint32_t, l);
  }
}
// The four register quads q = 0..3 of a 32x32 accumulator tile hold, per lane, channels 8q + 4h + 0..3 (h = lane >> 5) of one row.
// After swapping quads (2p, 2p+1) between the half-waves, lanes 0-31 hold channels 16p + 0..7 and lanes 32-63 channels 16p + 8..15
// as {a[0], a[1], b[0], b[1]}: one 16-byte store per pair instead of two 8-byte ones.
__device__ __forceinline__ uint4 widen(uint32_t (&a)[2], uint32_t (&b)[2]) {
  swap32(a[0], b[0]);
  swap32(a[1], b[1]);
  return make_uint4(a[0], a[1], b[0], b[1]);
}
}  // namespace pp

// ---- wave-tile epilogues ------------------------------------------------------------------------------------------------------------
// tile<TM, TN>(acc, m_w, n_w, lane): the wave's accumulators cover rows m_w + 32 j + (lane & 31), channels n_w + 32 i + 8 q + 4 (lane >> 5)
// + 0..3 (acc[j][i][4 q + e]).  Rows >= M and channels >= N fall outside the buffer descriptors / are skipped per 32-channel tile.

// FeedForward first linear (modules.py:353-364): tanh-GELU(acc + bias) -> the operand rows of the second linear, packed hi/lo (PK) or plain fp16
// FMT: 0 plain fp16 rows, 1 packed hi | lo rows (fp16x3), 2 MX lines (fp16m: hi | P_0 | P_1, common.h)
template <int FMT, int ACT, int NOSTORE = 0>  // NOSTORE: microbenchmark ablations — 1: the arithmetic without the stores, 2: the same bytes
                                             // stored lane-linearly (1 KB runs per instruction; WRONG layout: what does the row-strided pattern cost?)
struct PpEpiAct16 {
  static constexpr bool PK = FMT != 0;
  const float* bias;
  f16* out;          // [M, ld] halves: PK: [N/32][32 hi | 32 lo], ld = 2N;  plain: [N], ld = N
  int64_t ld;
  int M, N;
  // the weight conditioning's per-channel 2^-e[n] (GemmCore::w_alpha, or null) rides in the bias FMA: acc * 2^-e is exact, so fma(acc, 2^-e, bias)
  // rounds exactly as pp_unscale's product followed by the bias add did — one instruction instead of two per output
  static constexpr bool FUSES_UNSCALE = true;
  template <int TM, int TN>
  __device__ __forceinline__ void tile(f32x16 (&acc)[TM][TN], int m_w, int n_w, int lane, const float* alpha) const {
    const BufRsrc R = make_rsrc(out, (uint32_t)((int64_t)M * ld * 2));
    const int h = lane >> 5, r = lane & 31;
#pragma unroll
    for (int i = 0; i < TN; ++i) {
      const int nb = n_w + 32 * i;
      if (nb >= N) continue;  // wave-uniform
      float4 b[4], al[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        b[q] = *reinterpret_cast<const float4*>(bias + nb + 8 * q + 4 * h);
        al[q] = alpha ? *reinterpret_cast<const float4*>(alpha + nb + 8 * q + 4 * h) : make_float4(1.f, 1.f, 1.f, 1.f);  // wave-uniform choice
      }
      auto pre = [&](int j, int q, int e) {  // acc * 2^-e[n] + bias[n]
        const float a = e == 0 ? al[q].x : e == 1 ? al[q].y : e == 2 ? al[q].z : al[q].w, bb = e == 0 ? b[q].x : e == 1 ? b[q].y : e == 2 ? b[q].z : b[q].w;
        return __builtin_fmaf(acc[j][i][4 * q + e], a, bb);
      };
      const uint32_t col = PK ? (uint32_t)(nb >> 5) * 128u : (uint32_t)nb * 2u;  // byte offset of the tile's 32 channels in a row
#pragma unroll
      for (int j = 0; j < TM; ++j) {
        const uint32_t row = (uint32_t)(m_w + 32 * j + r) * (uint32_t)(ld * 2);
        if constexpr (FMT == 2) {  // the lane's 16 channels 8 q + 4 h + e are exactly the k-set of P_h
          float x[16];
#pragma unroll
          for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int e = 0; e < 4; ++e) x[4 * q + e] = apply_act(ACT, pre(j, q, e));
          uint32_t hv[8], pw[8];
          mx_pack16<false>(x, hv, pw);
#pragma unroll
          for (int p = 0; p < 2; ++p) {
            uint32_t a2[2] = {hv[4 * p], hv[4 * p + 1]}, b2[2] = {hv[4 * p + 2], hv[4 * p + 3]};
            pp::store_b128(R, row + col + (uint32_t)(16 * p + 8 * h) * 2u, pp::widen(a2, b2));
          }
          pp::store_b128(R, row + col + 64u + 32u * (uint32_t)h, make_uint4(pw[0], pw[1], pw[2], pw[3]));
          pp::store_b128(R, row + col + 80u + 32u * (uint32_t)h, make_uint4(pw[4], pw[5], pw[6], pw[7]));
          continue;
        }
        uint32_t hi[4][2], lo[4][2];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float x[4] = {apply_act(ACT, pre(j, q, 0)), apply_act(ACT, pre(j, q, 1)), apply_act(ACT, pre(j, q, 2)), apply_act(ACT, pre(j, q, 3))};
          pp::split4(x, hi[q], lo[q]);
        }
#pragma unroll
        for (int p = 0; p < 2; ++p) {
          const uint32_t o = row + col + (uint32_t)(16 * p + 8 * h) * 2u;
          if constexpr (NOSTORE == 2) {
            const uint32_t d = (uint32_t)(((m_w >> 5) + j) * (N >> 5) + (nb >> 5)) * 4096u + (uint32_t)p * 2048u + (uint32_t)lane * 16u;
            pp::store_b128(R, d, pp::widen(hi[2 * p], hi[2 * p + 1]));
            if constexpr (FMT == 1) pp::store_b128(R, d + 1024u, pp::widen(lo[2 * p], lo[2 * p + 1]));
            continue;
          }
          if constexpr (NOSTORE == 1) {
            const uint4 a = pp::widen(hi[2 * p], hi[2 * p + 1]), b2 = pp::widen(lo[2 * p], lo[2 * p + 1]);
            asm volatile("" ::"v"(a.x), "v"(a.y), "v"(a.z), "v"(a.w), "v"(b2.x), "v"(b2.y), "v"(b2.z), "v"(b2.w), "v"(o));
            continue;
          }
          pp::store_b128(R, o, pp::widen(hi[2 * p], hi[2 * p + 1]));
          if constexpr (FMT == 1) pp::store_b128(R, o + 64u, pp::widen(lo[2 * p], lo[2 * p + 1]));
        }
      }
    }
  }
};

// attention out-projection / FeedForward second linear (modules.py:548-556,751,755): x[m, n] += gate[n] * (acc + bias[n]); rows whose
// mask byte is 0 add nothing (mask_mode 1 of EpiStore).  GATE = false: no gate vector (UNetT residual adds, unett.py:300-301).
template <bool GATE>
struct PpEpiGateRes {
  const float* bias;
  const float* gate;
  const uint8_t* rowmask;  // or null
  float* x;
  int64_t ldx;
  int M, N;
  template <int TM, int TN>
  __device__ __forceinline__ void tile(f32x16 (&acc)[TM][TN], int m_w, int n_w, int lane) const {
    const BufRsrc R = make_rsrc(x, (uint32_t)((int64_t)M * ldx * 4));
    const int h = lane >> 5, r = lane & 31;
    bool live[TM];
#pragma unroll
    for (int j = 0; j < TM; ++j) {
      const int m = m_w + 32 * j + r;
      live[j] = !rowmask || (m < M && rowmask[m] != 0);
    }
#pragma unroll
    for (int i = 0; i < TN; ++i) {
      const int nb = n_w + 32 * i;
      if (nb >= N) continue;
      float4 b[4], c[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        b[q] = *reinterpret_cast<const float4*>(bias + nb + 8 * q + 4 * h);
        if constexpr (GATE) c[q] = *reinterpret_cast<const float4*>(gate + nb + 8 * q + 4 * h);
      }
#pragma unroll
      for (int j = 0; j < TM; ++j) {
        const uint32_t row = (uint32_t)(m_w + 32 * j + r) * (uint32_t)(ldx * 4) + (uint32_t)(nb + 4 * h) * 4u;
        uint4 rv[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) rv[q] = buffer_load_b128(R, row + 32u * q);  // the four residual quads in flight together
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          float y[4] = {acc[j][i][4 * q] + b[q].x, acc[j][i][4 * q + 1] + b[q].y, acc[j][i][4 * q + 2] + b[q].z, acc[j][i][4 * q + 3] + b[q].w};
          if constexpr (GATE) { y[0] *= c[q].x; y[1] *= c[q].y; y[2] *= c[q].z; y[3] *= c[q].w; }
          if (!live[j]) { y[0] = y[1] = y[2] = y[3] = 0.f; }
          union { uint4 u; float f[4]; } t;
          t.u = rv[q];
          t.f[0] += y[0]; t.f[1] += y[1]; t.f[2] += y[2]; t.f[3] += y[3];
          pp::store_b128(R, row + 32u * q, t.u);
        }
      }
    }
  }
};

// fused to_q | to_k | to_v (modules.py:481-509): bias, rotary embedding on (2i, 2i+1) pairs of q and k (x_transformers convention, call
// sites modules.py:503-509), q * 1/sqrt(dh), scatter into the flash kernel's layouts: q, k [B'*H, sn, 64] fp16 (+ lo planes), V^T
// [B'*H, 64, ldvt] fp16 (+ lo).  dh = 64 and inner % 64 == 0: a 32-channel tile lies in one head of one of q / k / v, so `which`, the
// head and the channel base are wave-uniform.  Same indices and arithmetic as EpiQKVT<true> (gemm.h), 16-byte stores for q / k, and the
// V^T columns of two neighbouring tokens leave as one 4-byte store when nseq is even (the lane pair (2t, 2t+1) holds tokens of one sequence).
struct PpEpiQKV {
  const float* bias;
  const float* rope_cs;  // [nseq, 32, 2]
  f16 *q16, *k16, *vt16, *q16_lo, *k16_lo, *vt16_lo;  // lo planes may be null
  int nseq, heads, pe_heads, slab_n, pos_off, ldvt;
  float qscale;
  uint32_t nseq_magic;
  int nseq_shift, inner;
  int M, N;
  uint32_t qk_bytes, vt_bytes;  // sizes of the q / k planes and of the V^T planes in bytes (< 2^31)
  const uint32_t* rowinfo;      // packed rows: (sequence << 16) | token of GEMM row m (EpiQKVT::rowinfo); null = m / nseq, m % nseq
  // 1: the second plane of q and of k receives, instead of the fp16 remainders, the MX-fp6 P words of the row's two 32-channel blocks
  // ([block][half-wave h], 32 bytes each; common.h mx_pack16: q as the activation, k as the weight) — what the flash kernel's NSPLIT = 2
  // form reads (attention_kernel.h).  The lane's 16 channels of a tile ARE the k-set of P_h: no lane exchange.
  int mx_qk;

  static constexpr bool FUSES_UNSCALE = true;  // the conditioning's 2^-e[n] inside the bias FMA (exact: see PpEpiAct16)
  template <int TM, int TN>
  __device__ __forceinline__ void tile(f32x16 (&acc)[TM][TN], int m_w, int n_w, int lane, const float* alpha) const {
    const int h = lane >> 5, r = lane & 31;
    const int sn = slab_n ? slab_n : nseq;
    int bp[TM], pos[TM];
#pragma unroll
    for (int j = 0; j < TM; ++j) {
      const int m = m_w + 32 * j + r;
      if (rowinfo) {  // (rows >= M: any valid (sequence, token) will do, their stores are dropped)
        const uint32_t ri = rowinfo[m < M ? m : M - 1];
        bp[j] = (int)(ri >> 16);
        pos[j] = (int)(ri & 0xffffu);
      } else {
        bp[j] = (int)((uint32_t)(((uint64_t)(uint32_t)m * nseq_magic) >> 32) >> nseq_shift);
        pos[j] = m - bp[j] * nseq;
      }
    }
    // V^T pairs: lanes (2t, 2t + 1) must hold tokens (p, p + 1) of ONE sequence with p even — true for the padded layout with an even nseq,
    // not for packed rows (a sequence may start at an odd row)
    const bool pair_ok = !(nseq & 1) && !(pos_off & 1) && !rowinfo;
#pragma unroll
    for (int i = 0; i < TN; ++i) {
      const int nb = n_w + 32 * i;
      if (nb >= N) continue;
      const int which = (nb >= inner ? 1 : 0) + (nb >= 2 * inner ? 1 : 0);
      const int c0 = nb - which * inner, hh = c0 >> 6, d0 = c0 & 63;  // head, first channel of the tile inside the head (0 or 32)
      float4 b[4], al[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        b[q] = *reinterpret_cast<const float4*>(bias + nb + 8 * q + 4 * h);
        al[q] = alpha ? *reinterpret_cast<const float4*>(alpha + nb + 8 * q + 4 * h) : make_float4(1.f, 1.f, 1.f, 1.f);  // wave-uniform choice
      }
      if (which < 2) {
        const bool rope = pe_heads < 0 || hh < pe_heads;
        f16* P = which == 0 ? q16 : k16;
        f16* Pl = which == 0 ? q16_lo : k16_lo;
        const BufRsrc R = make_rsrc(P, qk_bytes);
        const BufRsrc Rl = make_rsrc(Pl ? Pl : P, Pl ? qk_bytes : 0u);
        const float sc = which == 0 ? qscale : 1.0f;
#pragma unroll
        for (int j = 0; j < TM; ++j) {
          const bool ok = m_w + 32 * j + r < M;
          // (cos, sin) of this lane's two channel pairs per quad; always fetched (pos < nseq for every m, so the address is valid even for
          // rows >= M), replaced by the identity rotation for heads outside pe_attn_head: one code path, no value defined on one branch only
          float4 cs[4];
#pragma unroll
          for (int q = 0; q < 4; ++q) cs[q] = *reinterpret_cast<const float4*>(rope_cs + (pos[j] * 32 + ((d0 + 8 * q + 4 * h) >> 1)) * 2);
          if (!rope) {
#pragma unroll
            for (int q = 0; q < 4; ++q) cs[q] = make_float4(1.f, 0.f, 1.f, 0.f);
          }
          float xs[16];
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const float x[4] = {__builtin_fmaf(acc[j][i][4 * q], al[q].x, b[q].x), __builtin_fmaf(acc[j][i][4 * q + 1], al[q].y, b[q].y),
                                __builtin_fmaf(acc[j][i][4 * q + 2], al[q].z, b[q].z), __builtin_fmaf(acc[j][i][4 * q + 3], al[q].w, b[q].w)};
            xs[4 * q] = x[0] * cs[q].x - x[1] * cs[q].y; xs[4 * q + 1] = x[1] * cs[q].x + x[0] * cs[q].y;
            xs[4 * q + 2] = x[2] * cs[q].z - x[3] * cs[q].w; xs[4 * q + 3] = x[3] * cs[q].z + x[2] * cs[q].w;
            if (which == 0) { xs[4 * q] *= sc; xs[4 * q + 1] *= sc; xs[4 * q + 2] *= sc; xs[4 * q + 3] *= sc; }
          }
          const uint32_t rowb = ok ? (uint32_t)((((bp[j] * heads + hh) * sn + pos_off + pos[j]) << 6) + d0 + 8 * h) * 2u : OOB_ROW;
          if (mx_qk) {  // (uniform over the launch)
            uint32_t hv[8], pw[8];
            if (which == 0) mx_pack16<false>(xs, hv, pw); else mx_pack16<true>(xs, hv, pw);
            uint32_t h0[2] = {hv[0], hv[1]}, h1[2] = {hv[2], hv[3]}, h2[2] = {hv[4], hv[5]}, h3[2] = {hv[6], hv[7]};
            pp::store_b128(R, rowb, pp::widen(h0, h1));  // the first plane as ever: fp16 values in channel order
            pp::store_b128(R, rowb + 32u, pp::widen(h2, h3));
            const uint32_t rowp = ok ? rowb + 16u * h : OOB_ROW;  // byte 64 blk + 32 h of the row: (d0 + 8 h) * 2 + 16 h
            pp::store_b128(Rl, rowp, make_uint4(pw[0], pw[1], pw[2], pw[3]));
            pp::store_b128(Rl, rowp + 16u, make_uint4(pw[4], pw[5], pw[6], pw[7]));
          } else {
            uint32_t hi[4][2], lo[4][2];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              const float x[4] = {xs[4 * q], xs[4 * q + 1], xs[4 * q + 2], xs[4 * q + 3]};
              pp::split4(x, hi[q], lo[q]);
            }
#pragma unroll
            for (int p = 0; p < 2; ++p) {
              pp::store_b128(R, rowb + 32u * p, pp::widen(hi[2 * p], hi[2 * p + 1]));
              if (Pl) pp::store_b128(Rl, rowb + 32u * p, pp::widen(lo[2 * p], lo[2 * p + 1]));
            }
          }
        }
      } else {
        const BufRsrc R = make_rsrc(vt16, vt_bytes);
        const BufRsrc Rl = make_rsrc(vt16_lo ? vt16_lo : vt16, vt16_lo ? vt_bytes : 0u);
#pragma unroll
        for (int j = 0; j < TM; ++j) {
          const bool ok = m_w + 32 * j + r < M;
          const int tokp = pos_off + pos[j];
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const float x[4] = {__builtin_fmaf(acc[j][i][4 * q], al[q].x, b[q].x), __builtin_fmaf(acc[j][i][4 * q + 1], al[q].y, b[q].y),
                                __builtin_fmaf(acc[j][i][4 * q + 2], al[q].z, b[q].z), __builtin_fmaf(acc[j][i][4 * q + 3], al[q].w, b[q].w)};
            uint32_t hv[2], lv[2];
            pp::split4(x, hv, lv);
            const int d = d0 + 8 * q + 4 * h;
            const uint32_t base = (uint32_t)((bp[j] * heads + hh) * 64 + d) * (uint32_t)ldvt;  // element index of (channel d, token 0)
            if (pair_ok) {
              // lane pair (even, odd) = tokens (t, t + 1) of one sequence, t even: the even lane writes channels d, d+1 of both tokens, the
              // odd lane channels d+2, d+3 — two 4-byte stores each instead of four 2-byte ones
              const bool odd = lane & 1;
              const uint32_t got = pp::xchg1(odd ? hv[0] : hv[1]);    // even receives the partner's (d, d+1); odd the partner's (d+2, d+3)
              const uint32_t gotl = pp::xchg1(odd ? lv[0] : lv[1]);
              const uint32_t mine = odd ? hv[1] : hv[0], minel = odd ? lv[1] : lv[0];
              const uint32_t first = odd ? got : mine, second = odd ? mine : got;  // token t, token t + 1
              const uint32_t firstl = odd ? gotl : minel, secondl = odd ? minel : gotl;
              const uint32_t w0 = (first & 0xffffu) | (second << 16), w1 = (first >> 16) | (second & 0xffff0000u);
              const uint32_t w0l = (firstl & 0xffffu) | (secondl << 16), w1l = (firstl >> 16) | (secondl & 0xffff0000u);
              const uint32_t o = ok ? (base + (uint32_t)(odd ? 2 : 0) * (uint32_t)ldvt + (uint32_t)(tokp - (odd ? 1 : 0))) * 2u : OOB_ROW;
              pp::store_b32(R, o, w0);
              pp::store_b32(R, o + (uint32_t)ldvt * 2u, w1);
              if (vt16_lo) { pp::store_b32(Rl, o, w0l); pp::store_b32(Rl, o + (uint32_t)ldvt * 2u, w1l); }
            } else {
              const uint32_t o = ok ? (base + (uint32_t)tokp) * 2u : OOB_ROW;
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                pp::store_b16(R, o + (uint32_t)e * (uint32_t)ldvt * 2u, (uint16_t)(hv[e >> 1] >> (16 * (e & 1))));
                if (vt16_lo) pp::store_b16(Rl, o + (uint32_t)e * (uint32_t)ldvt * 2u, (uint16_t)(lv[e >> 1] >> (16 * (e & 1))));
              }
            }
          }
        }
      }
    }
  }
};

// the weight conditioning undone on a wave's accumulator tiles (GemmCore::w_alpha): channels n_w + 32 i + 8 q + 4 (lane >> 5) + 0..3
template <int TM, int TN>
__device__ __forceinline__ void pp_unscale(f32x16 (&acc)[TM][TN], const GemmCore& g, int n_w, int lane) {
  if (!g.w_alpha) return;  // wave-uniform
  const int h = lane >> 5;
#pragma unroll
  for (int i = 0; i < TN; ++i) {
    const int nb = n_w + 32 * i;
    if (nb >= g.N) continue;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float4 a = *reinterpret_cast<const float4*>(g.w_alpha + nb + 8 * q + 4 * h);
#pragma unroll
      for (int j = 0; j < TM; ++j) {
        acc[j][i][4 * q] *= a.x; acc[j][i][4 * q + 1] *= a.y; acc[j][i][4 * q + 2] *= a.z; acc[j][i][4 * q + 3] *= a.w;
      }
    }
  }
}

// the wave-tile epilogue of a finished accumulator tile: the conditioning's 2^-e[n] first, inside the epilogue's own arithmetic where it fuses
template <typename Epi, typename = void>
struct pp_fuses_unscale : std::false_type {};
template <typename Epi>
struct pp_fuses_unscale<Epi, std::enable_if_t<Epi::FUSES_UNSCALE>> : std::true_type {};
// FUSE = false: the scale as a pass of its own even where the epilogue could fuse it — the MX form of the ping-pong kernel sits at 256 registers,
// and the 16 more that hold the scale vectors next to the biases sent it to scratch (52 / 144 bytes; tests/test_isa_hazards.py)
template <int TM, int TN, typename Epi, bool FUSE = true>
__device__ __forceinline__ void pp_finish(f32x16 (&acc)[TM][TN], const GemmCore& g, const Epi& epi, int m_w, int n_w, int lane) {
  if constexpr (pp_fuses_unscale<Epi>::value && FUSE) {
    epi.template tile<TM, TN>(acc, m_w, n_w, lane, g.w_alpha);
  } else if constexpr (pp_fuses_unscale<Epi>::value) {
    pp_unscale<TM, TN>(acc, g, n_w, lane);
    epi.template tile<TM, TN>(acc, m_w, n_w, lane, nullptr);
  } else {
    pp_unscale<TM, TN>(acc, g, n_w, lane);
    epi.template tile<TM, TN>(acc, m_w, n_w, lane);
  }
}

// ---- the kernel ---------------------------------------------------------------------------------------------------------------------
template <int TM, int TN, int WGM, int WGN, int NS, int KSP = 1>
constexpr int gemm_pp_lds_bytes() {
  return NS * KSP * 32 * (WGM * TM + WGN * TN) * GEMM_KTB;
}
// TM x TN 32x32 tiles per wave, WGM x WGN waves, ring of NS stages, JG activation tiles per fragment slot (TM % JG == 0).
// KSP = 2 ("k-split"): TWO groups of WGM x WGN waves work on the SAME output tile, group g on the k-tiles 2 s + g — two waves per SIMD with
// the large wave tiles of a 4-wave workgroup: one group's LDS-DMA issue, fragment reads and waits hide behind the other's MFMAs, and the
// epilogue is shared (the groups exchange the partial sums of the tiles they do not finish through the idle ring).  For the one-round
// launches of a single utterance, where a CU holds one workgroup and nothing else covers those gaps.
// KSS = 2 ("k-step split"): two groups as above, but on the SAME k-tiles — group g multiplies the 16-wide k-steps g, g + 2, .. of every
// k-tile (fp16x3 lines hold two k-steps, plain fp16 lines four).  No extra LDS: one ring, filled by all the waves together, read by both
// groups; for the tiles whose ring already fills the LDS (192x192).  The partial sums meet in the epilogue exactly as with KSP.
// ABL (microbenchmark ablations): bit 0 = no epilogue, bit 2 = no LDS-DMA after the prologue, bit 3 = no MFMAs.
template <typename T, int NSPLIT, int TM, int TN, int WGM, int WGN, int NS, int JG, typename Epi, int ABL = 0, int KSP = 1, int KSS = 1>
__global__ __launch_bounds__(64 * WGM * WGN * KSP * KSS) void gemm_pp_kernel(GemmCore g, Epi epi) {
  constexpr int NW = WGM * WGN;  // waves of one group
  constexpr int DW = NW * KSS;   // waves that fill one (sub-)stage
  constexpr int BM = 32 * WGM * TM, BN = 32 * WGN * TN;
  constexpr bool MX = NSPLIT == 2;                 // fp16m lines: 32 hi halves | P_0 | P_1 (common.h) — 2 fp16 MFMAs + 1 fp6 MFMA per 32 k
  constexpr int NPL = (NSPLIT == 3 || MX) ? 2 : 1;
  constexpr int NFR = MX ? 3 : NPL;                // 16-byte fragments per tile: hi | lo, or hi | the two halves of the lane's P words
  constexpr int KS = NPL == 2 ? 2 : 4;             // fragment steps per k-tile: 16-wide MFMA k-steps
  constexpr int KSL = KS / KSS;                    // k-steps one group multiplies per k-tile
  constexpr int PA = BM / 8 / DW, PW = BN / 8 / DW;  // DMA pieces (8 rows x 128 B) per wave per k-tile
  constexpr int LPT = PA + PW;
  constexpr int TILE_A = BM * GEMM_KTB;
  constexpr int SUB = (BM + BN) * GEMM_KTB, STAGE = KSP * SUB;  // a stage holds the k-tiles of all groups
  constexpr int NSLOT = TM / JG;                   // fragment slots per k-step
  static_assert(PA * 8 * DW == BM && PW * 8 * DW == BN, "tile rows must split evenly into 8-row DMA pieces over the waves");
  static_assert(TM % JG == 0 && NS >= 2 && NS <= 5 && (NS - 1) * LPT <= 63 && (KSP == 1 || KSP == 2) && (KSS == 1 || KSS == 2) && KSP * KSS <= 2,
                "slot / ring shape (the counted waits are 6-bit immediates)");
  static_assert(sizeof(T) == 2, "fp16 operands (plain or hi/lo packed)");
  // MX lines under the k-step split: group g multiplies hi k-step g of every line, and the line's fp6 correction belongs to group 1 on even
  // k-tiles and to group 0 on odd ones (the tiles run as even / odd pairs anyway): 2 fp16 + 1 fp6 MFMA per group per pair of k-tiles
  F5_DYN_LDS(char, smem_all);

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave_all = pp::uniform(tid >> 6);
  const int grp = KSP * KSS == 1 ? 0 : wave_all / NW, wave = KSP * KSS == 1 ? wave_all : wave_all % NW;  // group, wave inside the group
  const int dwave = KSS == 1 ? wave : wave_all;    // this wave's place among those that fill a (sub-)stage
  char* smem = smem_all + (KSP == 1 ? 0 : grp * SUB);  // k-split: this group's half of every stage
  const int wm = wave % WGM, wn = wave / WGM;
  int m0, n0;
  {  // tile order as gemm_kernel: XCD-contiguous runs, channel tiles fastest, optional groups of row tiles
    const int nt = (g.N + BN - 1) / BN, nwg = gridDim.x;
    const int bid = blockIdx.x, q8 = nwg >> 3, r8 = nwg & 7, xcd = bid & 7, slot = bid >> 3;
    const int L = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + slot;
    int mt, ntile;
    if (g.group_m > 1) {
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
  const int kbytes = g.K * 2 * NPL;  // bytes of the hi (| lo | P) part of one operand row; a multiple of 128 (launcher)
  const int rowbytes = kbytes;
  const int nkt = kbytes / GEMM_KTB / KSP;  // k-tiles of this group (a multiple of KSP in total: launcher)
  const BufRsrc Ar = make_rsrc(g.A, (uint32_t)((int64_t)(g.a_rows - 1) * g.lda * 2 + rowbytes));
  const BufRsrc Wr = make_rsrc(g.W, (uint32_t)((int64_t)(g.w_rows - 1) * g.ldw * 2 + rowbytes));
  // DMA piece P of an operand = rows 8P .. 8P+7 -> LDS bytes [1024 P, +1024); lane l brings row 8P + l/8, logical chunk (l%8) ^ swz(row)
  uint32_t a_off[PA], w_off[PW];
#pragma unroll
  for (int p = 0; p < PA; ++p) {
    const int row = 8 * (dwave * PA + p) + (lane >> 3), lc = (lane & 7) ^ ((row >> 1) & 7);
    a_off[p] = (m0 + row) < g.a_rows ? (uint32_t)((int64_t)(m0 + row) * g.lda * 2 + lc * 16) : OOB_ROW;
  }
#pragma unroll
  for (int p = 0; p < PW; ++p) {
    const int row = 8 * (dwave * PW + p) + (lane >> 3), lc = (lane & 7) ^ ((row >> 1) & 7);
    w_off[p] = (n0 + row) < g.w_rows ? (uint32_t)((int64_t)(n0 + row) * g.ldw * 2 + lc * 16) : OOB_ROW;
  }
  auto issue = [&](int kt, int stage) {
    char* base = smem + stage * STAGE;
    const uint32_t kb = (uint32_t)(KSP * kt + (KSP == 1 ? 0 : grp)) * GEMM_KTB;
#pragma unroll
    for (int p = 0; p < PA; ++p) pp::dma_b128(Ar, base + (dwave * PA + p) * 1024, a_off[p], kb);
#pragma unroll
    for (int p = 0; p < PW; ++p) pp::dma_b128(Wr, base + TILE_A + (dwave * PW + p) * 1024, w_off[p], kb);
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int j = 0; j < TM; ++j)
#pragma unroll
    for (int i = 0; i < TN; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[j][i][r] = 0.f;

  // fragment addressing: lane (row = lane & 31, hi = lane >> 5) reads logical chunk 2 ks + hi (+4 for the lo plane) of its row
  const uint32_t lds0 = pp::lds_base(smem);
  const int frow = (lane & 31) * GEMM_KTB, fswz = ((lane & 31) >> 1) & 7, fhi = lane >> 5;
  // MX lines: chunks 4 + 2 fhi and 5 + 2 fhi are the lane's P words, read with the line's LAST k-step
  uint32_t fa_addr[NFR][KSL], fw_addr[NFR][KSL];  // + stage offset (updated per k-tile) + 4096 * tile index (immediate)
#pragma unroll
  for (int p = 0; p < NFR; ++p)
#pragma unroll
    for (int ksl = 0; ksl < KSL; ++ksl) {
      const int ks = KSS == 1 ? ksl : ksl * KSS + grp;  // k-step split: this group's k-steps of the line
      const int chunk = (MX && p > 0) ? 4 + 2 * fhi + (p - 1) : p * 4 + 2 * ks + fhi;
      const uint32_t o = (uint32_t)(frow + ((chunk ^ fswz) << 4));
      fa_addr[p][ksl] = lds0 + o + (uint32_t)(wm * 32 * TM) * GEMM_KTB;
      fw_addr[p][ksl] = lds0 + o + TILE_A + (uint32_t)(wn * 32 * TN) * GEMM_KTB;
    }
  // The k-loop as a function of the GROUP when MX lines meet the k-step split (whose turn a k-tile's fp6 correction is depends on the
  // group): two copies with the roles resolved at compile time — a run-time test put a scalar branch in front of every fp6 MFMA (round 4).
  auto k_loop = [&](auto GC) {
  constexpr int G = decltype(GC)::value;  // this wave's group (meaningful for MX && KSS == 2 only)
    Frag fa[2][NFR][JG], fw[2][NFR][TN];

    // slot s of a k-tile = (k-step s / NSLOT, activation tiles JG * (s % NSLOT) ..); its weight fragments are read with the k-step's first slot.
    // Slots are numbered v = parity * SLOTS + s over a PAIR of k-tiles: the fragment buffers alternate with v, so tiles with an odd number
    // of slots or k-steps (the k-step split) run as even / odd tiles; with even counts the parity is always 0.
    constexpr int SLOTS = KSL * NSLOT;
    constexpr bool PAIRED = (SLOTS % 2 != 0) || (KSL % 2 != 0);
    auto read_slot = [&](auto SC, uint32_t soff, auto WHICHC) {  // SC = integral_constant<int, v>; WHICH: bit 0 the activation fragments, bit 1 the weight fragments
      constexpr int v = decltype(SC)::value, s = v % SLOTS, ks = s / NSLOT, jg = s % NSLOT, buf = v & 1, wbuf = ((v / SLOTS) * KSL + ks) & 1;
      constexpr int WHICH = decltype(WHICHC)::value;
      static_for<NFR>([&](auto P) {
        constexpr int p = decltype(P)::value;
        auto reads = [&] {
          constexpr int TS = 4096;  // bytes of 32 rows of a line
          if constexpr (jg == 0 && (WHICH & 2)) {
            const uint32_t wb = fw_addr[p][ks] + soff;
            static_for<TN>([&](auto I) { fw[wbuf][p][decltype(I)::value].u = pp::lds_read_b128<decltype(I)::value * TS>(wb); });
          }
          if constexpr (WHICH & 1) {
            const uint32_t ab = fa_addr[p][ks] + soff;
            static_for<JG>([&](auto J) { fa[buf][p][decltype(J)::value].u = pp::lds_read_b128<(jg * JG + decltype(J)::value) * TS>(ab); });
          }
        };
        if constexpr (!MX || p == 0) reads();
        else if constexpr (KSS == 1) { if constexpr (ks == KSL - 1) reads(); }  // the lane's MX words: with the line's last k-step
        else { if constexpr (G != v / SLOTS) reads(); }                          // k-step split: the group whose turn this k-tile is
      });
      // (no scheduling fence here: pinning the reads ahead of the slot's MFMAs measured -20 % on the 8-wave tiles and 0 on the 4-wave ones —
      // hipcc's own interleaving of a slot's first MFMAs with the next slot's reads is the better one; profiles/r02b_kernel_bench.md)
    };
    auto mma_slot = [&](auto SC) {
      constexpr int v = decltype(SC)::value, s = v % SLOTS, ks = s / NSLOT, jg = s % NSLOT, buf = v & 1, wbuf = ((v / SLOTS) * KSL + ks) & 1;
      if constexpr (MX && (KSS == 1 ? ks == KSL - 1 : G != v / SLOTS)) {  // this slot multiplies MX words: re-define them behind the wait that preceded this call
        static_for<JG>([&](auto J) { pin_after_wait(fa[buf][1][decltype(J)::value].u); pin_after_wait(fa[buf][2][decltype(J)::value].u); });
        if constexpr (jg == 0) static_for<TN>([&](auto I) { pin_after_wait(fw[wbuf][1][decltype(I)::value].u); pin_after_wait(fw[wbuf][2][decltype(I)::value].u); });
      }
  #pragma unroll
      for (int jj = 0; jj < JG; ++jj)
  #pragma unroll
        for (int i = 0; i < TN; ++i) {
          if constexpr (ABL & 8) {  // keep the fragments alive without the MFMAs
  #ifndef F5_HIPEMU
            asm volatile("" ::"v"(fw[wbuf][0][i].u.x), "v"(fw[wbuf][MX ? 0 : NPL - 1][i].u.w), "v"(fa[buf][0][jj].u.x), "v"(fa[buf][MX ? 0 : NPL - 1][jj].u.w));
  #endif
            continue;
          }
          Mma32<T>::mma(acc[jg * JG + jj][i], fw[wbuf][0][i], fa[buf][0][jj]);
          if constexpr (MX && KSS == 1) {
            if constexpr (ks == KSL - 1) mx_mma(acc[jg * JG + jj][i], fw[wbuf][1][i].u, fw[wbuf][2][i].u, fa[buf][1][jj].u, fa[buf][2][jj].u);  // both correction terms of the line
          } else if constexpr (MX) {
            if constexpr (G != v / SLOTS) mx_mma(acc[jg * JG + jj][i], fw[wbuf][1][i].u, fw[wbuf][2][i].u, fa[buf][1][jj].u, fa[buf][2][jj].u);
          } else if constexpr (NPL == 2) {
            Mma32<T>::mma(acc[jg * JG + jj][i], fw[wbuf][0][i], fa[buf][1][jj]);  // W_hi . A_lo
            Mma32<T>::mma(acc[jg * JG + jj][i], fw[wbuf][1][i], fa[buf][0][jj]);  // W_lo . A_hi
          }
        }
    };

    // One k-tile.  On entry: slot 0's reads are in flight (or landed); tiles t+1 .. t+NS-1 are issued.  MODE 0: steady state (refill tile
    // t+NS when it exists), 1: next-to-last tile (nothing left to issue, everything outstanding is waited for), 2: last tile.
    auto ktile = [&](auto MODE, auto PAR, int t, uint32_t soff, uint32_t soff_next, int stage) {
      constexpr int mode = decltype(MODE)::value, v0 = decltype(PAR)::value * SLOTS, v0_next = PAIRED ? (1 - decltype(PAR)::value) * SLOTS : 0;
      // slots 0 .. SLOTS-2: wait for this slot's fragments, read the next slot's, multiply
      using BOTH = std::integral_constant<int, 3>;
      static_for<SLOTS - 1>([&](auto S) {
        pp::lds_wait();
        read_slot(std::integral_constant<int, v0 + decltype(S)::value + 1>{}, soff, BOTH{});
        mma_slot(std::integral_constant<int, v0 + decltype(S)::value>{});
      });
      pp::lds_wait();  // the last slot's fragments: every read of this tile by this wave has landed
      if constexpr (mode != 2) {
        if constexpr (mode == 0) pp::wait_vmcnt<(NS - 2) * LPT>();  // tile t+1 landed (this wave's pieces); NS-2 tiles stay in flight
        else pp::wait_vmcnt<0>();
        pp::wg_barrier();  // tile t+1 visible to all; nobody reads this tile's stage any more
        if constexpr (mode == 0) {
          if constexpr (!(ABL & 4)) {
            if (t + NS < nkt) issue(t + NS, stage);
          }
        }
        read_slot(std::integral_constant<int, v0_next>{}, soff_next, BOTH{});
      }
      mma_slot(std::integral_constant<int, v0 + SLOTS - 1>{});
    };

    // prologue: tiles 0 .. NS-1 in flight, tile 0 landed and visible, its first fragments requested
  #pragma unroll
    for (int s = 0; s < NS; ++s) issue(s, s);
    pp::wait_vmcnt<(NS - 1) * LPT>();
    pp::wg_barrier();
    read_slot(std::integral_constant<int, 0>{}, 0u, std::integral_constant<int, 3>{});
    int stage = 0;
    uint32_t soff = 0;
    auto next_soff = [&](uint32_t so) { return so + STAGE == (uint32_t)(NS * STAGE) ? 0u : so + STAGE; };
    using P0 = std::integral_constant<int, 0>;
    using P1 = std::integral_constant<int, PAIRED ? 1 : 0>;  // parity of the odd tiles (an even number of k-tiles: launcher)
    int t = 0;
    auto steady = [&](auto PAR) {
      const uint32_t sn = next_soff(soff);
      ktile(std::integral_constant<int, 0>{}, PAR, t, soff, sn, stage);
      soff = sn;
      stage = stage == NS - 1 ? 0 : stage + 1;
      ++t;
    };
    if constexpr (PAIRED) {
      while (t < nkt - 2) { steady(P0{}); steady(P1{}); }
    } else {
      while (t < nkt - 2) steady(P0{});
    }
    {
      const uint32_t sn = next_soff(soff);
      ktile(std::integral_constant<int, 1>{}, P0{}, t, soff, sn, stage);
      soff = sn;
      ++t;
    }
    ktile(std::integral_constant<int, 2>{}, P1{}, t, soff, 0u, 0);
  };
  if constexpr (MX && KSS == 2) {
    if (grp == 0) k_loop(std::integral_constant<int, 0>{});
    else k_loop(std::integral_constant<int, 1>{});
  } else {
    k_loop(std::integral_constant<int, 0>{});
  }

  if constexpr (ABL & 1) {
#ifndef F5_HIPEMU
#pragma unroll
    for (int j = 0; j < TM; ++j)
#pragma unroll
      for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int r = 0; r < 16; r += 4) asm volatile("" ::"v"(acc[j][i][r]), "v"(acc[j][i][r + 1]), "v"(acc[j][i][r + 2]), "v"(acc[j][i][r + 3]));
#endif
  } else if constexpr (KSP * KSS == 1) {
    // (MX lines on the 8-tile wave layout run at the 256-register limit: the scale stays a pass of its own there, see pp_finish)
    pp_finish<TM, TN, Epi, !(NSPLIT == 2 && TM * TN >= 8)>(acc, g, epi, m0 + wm * 32 * TM, n0 + wn * 32 * TN, lane);
  } else {
    // The two groups hold partial sums of the same tiles.  Tile t = j * TN + i is FINISHED by group (t < NT0 ? 0 : 1): every wave parks the
    // tiles it does not finish in the (now idle) ring — [wave][tile][quad][lane] float4, one conflict-free 1 KB run per store — and adds its
    // partner's copy of the tiles it does finish, always partner + own in group order 0 + 1 (one summation order whoever finishes).
    constexpr int NT = TM * TN, NT0 = (NT + 1) / 2;
    pp::wg_barrier();  // every wave has read its last fragments: the ring is free
    float4* xch = reinterpret_cast<float4*>(smem_all);
    static_for<NT>([&](auto TI) {
      constexpr int ti = decltype(TI)::value, j = ti / TN, i = ti % TN;
      if (grp != (ti < NT0 ? 0 : 1)) {
#pragma unroll
        for (int q = 0; q < 4; ++q)
          xch[((wave * NT + ti) * 4 + q) * 64 + lane] = make_float4(acc[j][i][4 * q], acc[j][i][4 * q + 1], acc[j][i][4 * q + 2], acc[j][i][4 * q + 3]);
      }
    });
    __syncthreads();
    // The partner's copies are read with the inline-asm fragment reads: hipcc puts `s_waitcnt vmcnt(0)` in front of every LDS read it can see
    // (an LDS-DMA might still be in flight, for all it knows), and on gfx9 that counter also holds the STORES of the tile finished just
    // before — every tile of a wave would wait for the previous tile's global stores to be acknowledged (round 3: found in the disassembly).
    const uint32_t xbase = pp::lds_base(smem_all) + (uint32_t)((wave * NT * 4) * 64 + lane) * 16u;
    static_for<NT>([&](auto TI) {
      constexpr int ti = decltype(TI)::value, j = ti / TN, i = ti % TN;
      if (grp == (ti < NT0 ? 0 : 1)) {
        f32x16 one[1][1];
        uint4 ou[4];
        static_for<4>([&](auto Q) { ou[decltype(Q)::value] = pp::lds_read_b128<(ti * 4 + decltype(Q)::value) * 1024>(xbase); });
        pp::lds_wait();
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          union { uint4 u; float4 f; } cv;
          cv.u = ou[q];
          const float4 o = cv.f;
          if (grp == 0) {  // own (group 0) + partner (group 1)
            one[0][0][4 * q] = acc[j][i][4 * q] + o.x; one[0][0][4 * q + 1] = acc[j][i][4 * q + 1] + o.y;
            one[0][0][4 * q + 2] = acc[j][i][4 * q + 2] + o.z; one[0][0][4 * q + 3] = acc[j][i][4 * q + 3] + o.w;
          } else {         // partner (group 0) + own
            one[0][0][4 * q] = o.x + acc[j][i][4 * q]; one[0][0][4 * q + 1] = o.y + acc[j][i][4 * q + 1];
            one[0][0][4 * q + 2] = o.z + acc[j][i][4 * q + 2]; one[0][0][4 * q + 3] = o.w + acc[j][i][4 * q + 3];
          }
        }
        pp_finish<1, 1>(one, g, epi, m0 + wm * 32 * TM + 32 * j, n0 + wn * 32 * TN + 32 * i, lane);
      }
    });
  }
}
