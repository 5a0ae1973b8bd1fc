// attention_kernel.h — the flash attention kernel of attention.hip as a header (so that tests/hipemu can compile the same source for
// the host).  See attention.hip for the design notes.
#pragma once
#include <math.h>

#include "kernels.h"

namespace {

constexpr int QB = 128;              // query rows per workgroup (4 waves); the 6-wave variant takes 192
constexpr int KT = 64;               // keys per tile
constexpr int K_ROWB = 144;          // K tile LDS row: 64 halves + 16 B pad (conflict-free 32-row ds_read_b128)
constexpr int V_ROWB = 136;          // V^T tile LDS row: 64 halves + 8 B pad (conflict-free 32-row ds_read_b64)
constexpr int K_PLANE = KT * K_ROWB;
constexpr int V_PLANE = 64 * V_ROWB;
constexpr float LOG2E = 1.4426950408889634f;

struct FlashArgs {
  const f16 *q, *q_lo, *k, *k_lo, *vt, *vt_lo;
  f16 *o, *o_lo;
  const int32_t* kvlen;  // per batch' or null
  const int32_t* kvlen2; // MMDiT joint attention with a key mask (modules.py:643-657): a second run of valid keys [seg2_off, seg2_off +
  int seg2_off;          // kvlen2[batch']) behind the first one [0, kvlen[batch']) — audio frames, then the text tokens; null = one run
  int n, ldv, heads, nqb, nwg, o_packed;
  // key-split variant (SPLIT): every (batch', head, query block) is cut into kv_split workgroups that each walk a contiguous range of
  // key tiles and leave UNNORMALISED partial results — part_o [BH * n, kv_split, 64] (O relative to the split's own running maximum),
  // part_ml [BH * n, kv_split, 2] (that maximum, the row sum) — which flash_combine_kernel merges.  Appended last: the layout the
  // unsplit kernels see is unchanged.
  int kv_split;
  float* part_o;
  float* part_ml;
  // 1: q already carries log2(e) (the QKV epilogue folded it into the 1/sqrt(dh) scale): scores are base-2 logarithms and the exponentials
  // need no multiply.  0: natural-log scores (tests and microbenchmarks that prepare q themselves).
  int log2q;
  // Packed rows (engine option "packed_rows", with kvlen): the output row of query q of sequence b' is cu_rows[b'] + q instead of
  // b' * n + q, queries >= kvlen[b'] do not exist (their blocks exit at once) — the reference's varlen path, modules.py:522-543.
  const int32_t* cu_rows;
};

// Scores of keys that do not exist (>= kv_end) or lie in the masked hole [hole_lo, hole_hi) -> -inf, for the tile of keys t KT .. t KT + 63:
// register r of block kb of a lane (row, hi) is key t KT + 32 kb + (r & 3) + 8 (r >> 2) + 4 hi.  Branch-free — one compare and one select per
// score against the lane's own limits; the `if (...) s = -inf` form compiled to an exec-mask save / restore per element (~430 instructions
// for the one tail tile of a launch with n % 64 != 0).
__device__ __forceinline__ void flash_mask_tile(f32x16 (&s)[2], int t, int hi, int kv_end, int hole_lo, int hole_hi) {
  const int base = t * KT + 4 * hi, lim = kv_end - base, hlo = hole_lo - base, hhi = hole_hi - base;
  const bool hole = hole_hi > hole_lo;  // wave-uniform
#pragma unroll
  for (int kb = 0; kb < 2; ++kb)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int off = kb * 32 + (r & 3) + 8 * (r >> 2);
      bool dead = off >= lim;
      if (hole) dead = dead | ((off >= hlo) & (off < hhi));
      s[kb][r] = dead ? -INFINITY : s[kb][r];
    }
}

// The operand row of the out-projection from a lane's share of a normalised output row: lane (row, hi) owns O[32 db + 8 c + 4 hi + e] =
// o[db][4 c + e].  o_packed 0: plain fp16 rows, 1: packed hi | lo lines (fp16x3), 2: MX lines (fp16m, common.h) — the lane's 16 features of
// a 32-block are exactly the k-set of P_hi, so the pack needs no lane exchange.
__device__ __forceinline__ void flash_store_row(const FlashArgs& a, const f32x16 (&o)[2], float inv, int64_t row, int hh, int hi, bool live) {
  const int64_t orow = row * ((int64_t)a.heads * 64 * (a.o_packed ? 2 : 1));
  if (a.o_packed == 2) {  // MX lines: the P word is the lane's own; the hi halves trade quads with the partner lane (l ^ 32) for 16-byte stores (whole waves)
#pragma unroll
    for (int db = 0; db < 2; ++db) {
      float x[16];
#pragma unroll
      for (int t = 0; t < 16; ++t) x[t] = o[db][t] * inv;
      uint32_t hv[8], pw[8];
      mx_pack16<false>(x, hv, pw);
#pragma unroll
      for (int j = 0; j < 4; ++j) f5_swap32(hv[j], hv[4 + j]);  // half-wave 0: channels 0-15 (own quads 0, 1 + the partner's), half-wave 1: 16-31
      if (live) {
        char* line = reinterpret_cast<char*>(a.o + orow) + (int64_t)(hh * 2 + db) * 128;
        *reinterpret_cast<uint4*>(line + 32 * hi) = make_uint4(hv[0], hv[1], hv[4], hv[5]);
        *reinterpret_cast<uint4*>(line + 32 * hi + 16) = make_uint4(hv[2], hv[3], hv[6], hv[7]);
        *reinterpret_cast<uint4*>(line + 64 + 32 * hi) = make_uint4(pw[0], pw[1], pw[2], pw[3]);
        *reinterpret_cast<uint4*>(line + 80 + 32 * hi) = make_uint4(pw[4], pw[5], pw[6], pw[7]);
      }
    }
    return;
  }
  if (!live) return;
#pragma unroll
  for (int db = 0; db < 2; ++db)
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int d = pk_off(hh * 64 + db * 32 + 8 * c + 4 * hi, a.o_packed);
      f16x4 oh, ol;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float v = o[db][4 * c + e] * inv;
        f16 h, l;
        split_f16(v, h, l);
        oh[e] = h;
        ol[e] = l;
      }
      *reinterpret_cast<f16x4*>(a.o + orow + d) = oh;
      if (a.o_lo) *reinterpret_cast<f16x4*>(a.o_lo + orow + d) = ol;
    }
}

// NSPLIT: operand split of S = QK^T (1, 2 or 3); PVSPLIT: of O = PV (1, 2 or 3).  The scores feed an exponential, so
// their rounding matters ~10x more than that of P and V: NSPLIT = 3 with PVSPLIT = 1 keeps near-fp32 scores at 4 instead of 6
// MFMAs per key-query pair.  PVSPLIT = 2 (round 6): V as hi + lo halves, P plain — O = V_hi P + V_lo P, one more fp16 MFMA per product.
// The sharpness sweep showed that the error of plain fp16 P . V GROWS with the logits — under near one-hot rows the
// output is one V row and nothing averages its 2^-12 rounding away — and the CPU decomposition (oracle/operand_scheme_emulation.py
// --attn v16 / p16 / vs) that V carries 2.6x the error of P; PVSPLIT = 3 adds V_hi P_lo as well.  NSPLIT = 2 (round 5): the two correction products K_hi . Q_lo + K_lo . Q_hi of the split as ONE MX-fp6 MFMA per
// 32 head channels (common.h, "fp16 + MX-fp6 corrections": the block GEMMs' scheme) — 6 MFMAs per 32 keys x 32 queries where NSPLIT = 3
// takes 12 and plain fp16 takes 4.  The second plane of q and of k then holds, per row, the P words of its two 32-channel blocks:
// [block 0: P_0 | P_1][block 1: P_0 | P_1], 32 bytes each, q packed as the activation and k as the weight (the QKV epilogue writes both
// from the lane's own 16 channels, gemm_pp.h PpEpiQKV::mx_qk).
template <int NSPLIT, int PVSPLIT>
constexpr int flash_lds_bytes() {
  return 2 * ((NSPLIT >= 2 ? 2 : 1) * K_PLANE + (PVSPLIT >= 2 ? 2 : 1) * V_PLANE);
}

// NW: waves per workgroup = 32-row query groups per block.  4 (128 query rows, two workgroups per CU) everywhere except where 6
// (192 rows, one workgroup per CU) makes the grid fit the chip in ONE round: B = 1, N = 1406 gives 11 x 32 = 352 blocks of 128 rows
// (CUs with 2 and CUs with 1 workgroup: 69 % balance) but 8 x 32 = 256 blocks of 192 rows.  Only the first 4 waves stage K / V tiles.
// LAZY (round 3; needs log2q, not with SPLIT): the running maximum is a REFERENCE, not the exact maximum — it is raised only when a tile
// exceeds it by more than LAZY_TAU (8: P <= 256, far inside fp16 and fp32 range), which after the first tiles is rare.  The scores then leave the
// matrix pipe already relative to the reference (the first MFMA of a chain accumulates onto a register tile holding -reference), so the
// steady-state tile needs no subtraction, no exp of the correction and no rescale of O and the row sum: 32 FMA + 32 multiplies + the
// lane exchange of the maximum fewer per 64-key tile and wave (of ~215 VALU instructions against 20 MFMAs: the kernel is VALU-bound).
constexpr float LAZY_TAU = 8.0f;
template <int NSPLIT, int PVSPLIT, int NW = 4, bool SPLIT = false, bool VSUM = false, bool LAZY = false>  // VSUM: A/B switch — row sums on the VALU (round 1)
__global__ __launch_bounds__(64 * NW, NW == 4 ? 2 : 1) void flash_attn_kernel(FlashArgs a) {
  static_assert(!(LAZY && SPLIT), "the key-split partial results carry exact maxima");
  static_assert(NSPLIT == 1 || NSPLIT == 2 || NSPLIT == 3, "operands of the scores: plain, fp16 + MX corrections, hi/lo split");
  constexpr int NPL = NSPLIT >= 2 ? 2 : 1;    // planes of q and k
  constexpr int NPV = PVSPLIT >= 2 ? 2 : 1;   // planes of v
  constexpr int NPP = PVSPLIT == 3 ? 2 : 1;   // planes of P
  constexpr bool VADAPT = PVSPLIT == 2 && LAZY;  // V_lo . P tile by tile, where the row's weights are concentrated (below)
  constexpr bool MXQK = NSPLIT == 2;          // plane 1 = P words
  constexpr int STAGE = NPL * K_PLANE + NPV * V_PLANE;
  F5_DYN_LDS(char, smem);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, hi = lane >> 5, ql = lane & 31;
  // XCD-aware placement: block b runs on XCD b % 8 (observed; speed only) -> give each XCD a contiguous range of
  // (batch', head) so the query blocks sharing one K/V slab hit the same L2.  Bijective for any grid size.
  const int bid = blockIdx.x;
  const int q8 = a.nwg >> 3, r8 = a.nwg & 7, xcd = bid & 7, slot = bid >> 3;
  int L = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + slot;
  int ks = 0;  // which part of the keys (SPLIT): the parts of one query block are neighbours in the order, so they share Q and an L2
  if constexpr (SPLIT) { ks = L % a.kv_split; L /= a.kv_split; }
  const int bh = L / a.nqb, qb = L - bh * a.nqb;
  const int bp = bh / a.heads, hh = bh - bp * a.heads;
  const int n = a.n;
  // valid keys: [0, hole_lo) and [hole_hi, kv_end); the hole is empty unless a second run is given
  int kv_end = a.kvlen ? min(a.kvlen[bp], n) : n, hole_lo = 0, hole_hi = 0;
  if (a.kvlen && a.kvlen2) {
    hole_lo = kv_end;
    hole_hi = min(a.seg2_off, n);
    kv_end = min(a.seg2_off + a.kvlen2[bp], n);
    if (hole_lo >= hole_hi) hole_lo = hole_hi = 0;
  }
  const int q_end = a.cu_rows ? (a.kvlen ? min(a.kvlen[bp], n) : n) : n;  // queries of this sequence that exist
  if (qb * (32 * NW) >= q_end) return;                                    // (whole workgroup: no barrier has been reached yet)
  const int ntile_all = (kv_end + KT - 1) / KT;
  // tiles [t0, t0 + ntile) of the key range belong to this workgroup
  const int t0 = SPLIT ? (int)((int64_t)ks * ntile_all / a.kv_split) : 0;
  const int ntile = SPLIT ? (int)((int64_t)(ks + 1) * ntile_all / a.kv_split) - t0 : ntile_all;

  // K / V^T slabs of this (batch', head) through buffer descriptors: rows or keys past the end read as zeros (hardware
  // bounds check), so the tile loads are unconditional and nothing waits on them until the matching LDS store.
  // V^T pad columns [n, ldv) are zero-initialised once at allocation and never written; a key >= kv_end gets P = 0 exactly.
  const uint32_t k_bytes = (uint32_t)n * 128u, v_bytes = (uint32_t)(64 * a.ldv) * 2u;
  BufRsrc Kr[NPL], Vr[NPV];
  const f16* Qp[NPL];
  Kr[0] = make_rsrc(a.k + (int64_t)bh * n * 64, k_bytes);
  Vr[0] = make_rsrc(a.vt + (int64_t)bh * 64 * a.ldv, v_bytes);
  Qp[0] = a.q + (int64_t)bh * n * 64;
  if constexpr (NPL == 2) {
    Kr[1] = make_rsrc(a.k_lo + (int64_t)bh * n * 64, k_bytes);
    Qp[1] = a.q_lo + (int64_t)bh * n * 64;
  }
  if constexpr (NPV == 2) Vr[1] = make_rsrc(a.vt_lo + (int64_t)bh * 64 * a.ldv, v_bytes);

  // Q rows of this wave stay in registers for the whole kernel: fq[p][ks] = Q[q][16 ks + 8 hi .. +7]; MXQK: plane 1 is read as the P words
  // of the lane's half-wave, fq[1][2 blk], fq[1][2 blk + 1] = the 32 bytes of P_hi of channel block blk
  const int qrow = qb * (32 * NW) + wave * 32 + ql;
  Frag fq[NPL][4];
#pragma unroll
  for (int p = 0; p < NPL; ++p)
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const int off = MXQK && p == 1 ? (ks >> 1) * 32 + hi * 16 + (ks & 1) * 8 : ks * 16 + hi * 8;
      fq[p][ks].u = qrow < n ? *reinterpret_cast<const uint4*>(Qp[p] + (int64_t)qrow * 64 + off) : make_uint4(0, 0, 0, 0);
    }

  // thread -> 16-byte chunk c = tid + 256 i of a tile: K tile row = key (c >> 3), V^T tile row = d (c >> 3), 8 chunks per row
  uint32_t k_off[2], v_off[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int c = tid + i * 256, row = c >> 3, col = c & 7;
    k_off[i] = (uint32_t)(row * 128 + col * 16);                 // + key0 * 128
    v_off[i] = (uint32_t)(row * a.ldv * 2 + col * 16);           // + key0 * 2
  }
  uint4 rk0[NPL][2], rv0[NPV][2], rk1[NPL][2], rv1[NPV][2];
  auto load_global = [&](int t, uint4 (&rk)[NPL][2], uint4 (&rv)[NPV][2]) {
    if (NW > 4 && wave >= 4) return;  // waves 4.. only compute
    const uint32_t key0 = (uint32_t)t * KT;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const uint32_t vkey_off = key0 * 2 + (uint32_t)((tid + i * 256) & 7) * 16;  // byte offset of the chunk within a V^T row
      const uint32_t voff = vkey_off < (uint32_t)a.ldv * 2 ? v_off[i] + key0 * 2 : OOB_OFF;
#pragma unroll
      for (int p = 0; p < NPL; ++p) rk[p][i] = buffer_load_b128(Kr[p], k_off[i] + key0 * 128);
#pragma unroll
      for (int p = 0; p < NPV; ++p) rv[p][i] = buffer_load_b128(Vr[p], voff);
    }
  };
  auto store_lds = [&](int stage, const uint4 (&rk)[NPL][2], const uint4 (&rv)[NPV][2]) {
    if (NW > 4 && wave >= 4) return;
    char* base = smem + stage * STAGE;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int c = tid + i * 256, row = c >> 3, col = c & 7;
#pragma unroll
      for (int p = 0; p < NPL; ++p) *reinterpret_cast<uint4*>(base + p * K_PLANE + row * K_ROWB + col * 16) = rk[p][i];
#pragma unroll
      for (int p = 0; p < NPV; ++p) {
        char* vd = base + NPL * K_PLANE + p * V_PLANE + row * V_ROWB + col * 16;  // 8-byte aligned rows
        *reinterpret_cast<uint2*>(vd) = make_uint2(rv[p][i].x, rv[p][i].y);
        *reinterpret_cast<uint2*>(vd + 8) = make_uint2(rv[p][i].z, rv[p][i].w);
      }
    }
  };

  f32x16 o[2];
#pragma unroll
  for (int r = 0; r < 16; ++r) { o[0][r] = 0.f; o[1][r] = 0.f; }
  float m_run = -INFINITY, l_run = 0.f;
  // Row sums on the matrix pipe (plain-fp16 P): the kernel is VALU-bound (33 exp + ~130 other VALU against 16 MFMAs per 64-key tile and
  // wave), and 32 of those VALU are the adds of the row sum.  A V^T fragment of ones turns them into 4 MFMAs per tile on the pipe that
  // has the slack: every accumulator row then holds sum_k P[q][k] over the 16 keys of a group — of BOTH half-waves, so the final
  // lane ^ 32 exchange goes too.  The sum is that of the fp16 P the O product uses (numerator and denominator see the same weights).
  constexpr bool MSUM = NPP == 1 && !VSUM;
  Frag ones;
#pragma unroll
  for (int e = 0; e < 8; ++e) ones.h[e] = (f16)1.0f;

  const float dom = a.log2q ? 1.0f : LOG2E;  // score units -> base-2 exponent
  f32x16 negm;                               // LAZY: -reference maximum of this lane's query row in every register (the srcC of a tile's first MFMAs)
#pragma unroll
  for (int r = 0; r < 16; ++r) negm[r] = 0.f;
  bool first = true;                         // LAZY: the first processed tile fixes the reference at its exact row maximum

  // one 64-key tile held in LDS stage `stage`: scores, online softmax, O update
  auto process = [&](int t, int stage) {
    const char* base = smem + stage * STAGE;
    const char* sK = base + ql * K_ROWB + hi * 16;
    const char* sV = base + NPL * K_PLANE + ql * V_ROWB + hi * 8;

    // ---- S^T = K . Q^T for the 64 keys of the tile --------------------------------------------------
    f32x16 s[2];
#pragma unroll
    for (int r = 0; r < 16; ++r) { s[0][r] = LAZY ? negm[r] : 0.f; s[1][r] = LAZY ? negm[r] : 0.f; }
    Frag fkp;  // (MXQK) the first half of a block's P words
    fkp.u = make_uint4(0, 0, 0, 0);
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        Frag fk[NPL];
#pragma unroll
        for (int p = 0; p < NPL; ++p) {
          // MXQK, plane 1: the half-wave's P words of block ks >> 1 — fetched with the fragment of the same index, used at the odd one
          const int off = MXQK && p == 1 ? (ks >> 1) * 64 + hi * 16 + (ks & 1) * 16 : ks * 32;
          fk[p].u = *reinterpret_cast<const uint4*>(sK + p * K_PLANE + kb * 32 * K_ROWB + off);
        }
        Mma32<f16>::mma(s[kb], fk[0], fq[0][ks]);
        if constexpr (NSPLIT == 3) {
          Mma32<f16>::mma(s[kb], fk[0], fq[1][ks]);  // K_hi . Q_lo
          Mma32<f16>::mma(s[kb], fk[1], fq[0][ks]);  // K_lo . Q_hi
        }
        if constexpr (MXQK) {
          if (ks & 1) mx_mma(s[kb], fkp.u, fk[1].u, fq[1][ks - 1].u, fq[1][ks].u);  // both correction products of channels 32 (ks >> 1) .. + 31
          else fkp = fk[1];
        }
      }

    // ---- online softmax (fp32), lane-local per query row ----------------------------------------------
    if ((t + 1) * KT > kv_end || (t * KT < hole_hi && (t + 1) * KT > hole_lo)) {  // tail tile (keys >= kv_end do not exist / are
                                                                                      // masked) or a tile touching the masked hole
      flash_mask_tile(s, t, hi, kv_end, hole_lo, hole_hi);
    }
    float mx = s[0][0];
#pragma unroll
    for (int r = 1; r < 16; ++r) mx = fmaxf(mx, s[0][r]);
#pragma unroll
    for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s[1][r]);
    float alpha = 1.0f, rs = 0.f;
    float raised = 0.f;  // (VADAPT) what this tile's reference raise took off the scores
    if constexpr (LAZY) {
      // s holds score - reference (base-2 units).  Raise the reference where a row went more than LAZY_TAU above it (and fix it on the first tile)
      if (f5_wave_any(first || mx > LAZY_TAU)) {
        const float mxr = fmaxf(mx, __shfl_xor(mx, 32, 64));  // the row's maximum over both half-waves' keys
        const float delta = first ? mxr : (mxr > LAZY_TAU ? mxr : 0.f);
        raised = delta;
        if (!first) {
          alpha = __builtin_amdgcn_exp2f(-delta);
#pragma unroll
          for (int r = 0; r < 16; ++r) { o[0][r] *= alpha; o[1][r] *= alpha; }
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) { s[0][r] -= delta; s[1][r] -= delta; negm[r] -= delta; }
        first = false;
      }
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float p = __builtin_amdgcn_exp2f(s[kb][r]);
          s[kb][r] = p;
          if constexpr (!MSUM) rs += p;
        }
      if constexpr (!MSUM) l_run = l_run * alpha + rs;
    } else {
      mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
      const float m_new = fmaxf(m_run, mx);
      alpha = __builtin_amdgcn_exp2f((m_run - m_new) * dom);
      const float mb = m_new * dom;
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float p = __builtin_amdgcn_exp2f(fmaf(s[kb][r], dom, -mb));
          s[kb][r] = p;
          if constexpr (!MSUM) rs += p;
        }
      if constexpr (!MSUM) l_run = l_run * alpha + rs;
      m_run = m_new;
#pragma unroll
      for (int r = 0; r < 16; ++r) { o[0][r] *= alpha; o[1][r] *= alpha; }
    }

    // ---- O^T += V^T . P^T ----------------------------------------------------------------------------
    // VADAPT (PVSPLIT 2 in the lazy forms): the V_lo . P product only for tiles that can matter.  The rounding of V shows in the output
    // where a FEW keys carry a row (under a near one-hot row the output is one V row; over many comparable keys the 2^-12 errors average
    // away), so a tile whose largest weight is below 1 / 16 of the mass the row has gathered so far — for every row of the wave — skips
    // its eight V_lo MFMAs and fragment reads: every tile of a flat row after the first, none that holds a dominant key (the first tile,
    // l = 0, always runs; a late dominant key makes its tile's weight large against the mass before it).  One exponential, one compare and
    // one ballot per tile.
    bool vlo = true;
    // (l_run: with the row sums on the matrix pipe it is the mass BEFORE this tile, not yet rescaled by a raise; on the VALU this lane's half of
    // the row including this tile — smaller than the row's, so the test only errs towards running the product)
    if constexpr (VADAPT) vlo = f5_wave_any(__builtin_amdgcn_exp2f(mx - raised) * 16.0f > (MSUM ? l_run * alpha : l_run));
    f32x16 rsum;
#pragma unroll
    for (int r = 0; r < 16; ++r) rsum[r] = 0.f;
#pragma unroll
    for (int g = 0; g < 4; ++g) {  // 16-key groups of the tile; P registers 8*(g&1) .. +7 of s[g>>1]
      Frag fp[NPP];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float p = s[g >> 1][8 * (g & 1) + e];
        const f16 ph = (f16)p;
        fp[0].h[e] = ph;
        if constexpr (NPP == 2) fp[1].h[e] = (f16)(p - (float)ph);
      }
      if constexpr (MSUM) Mma32<f16>::mma(rsum, ones, fp[0]);
#pragma unroll
      for (int db = 0; db < 2; ++db) {
        Frag fv[NPV];
#pragma unroll
        for (int p = 0; p < NPV; ++p) {
          if (VADAPT && p == 1 && !vlo) continue;  // (wave-uniform)
          const char* src = sV + p * V_PLANE + db * 32 * V_ROWB + g * 32;
          const uint2 v0 = *reinterpret_cast<const uint2*>(src);       // keys 16g + 4hi + 0..3
          const uint2 v1 = *reinterpret_cast<const uint2*>(src + 16);  // keys 16g + 8 + 4hi + 0..3
          fv[p].u = make_uint4(v0.x, v0.y, v1.x, v1.y);
        }
        Mma32<f16>::mma(o[db], fv[0], fp[0]);
        if constexpr (NPP == 2) Mma32<f16>::mma(o[db], fv[0], fp[1]);  // V_hi . P_lo
        if constexpr (NPV == 2) {
          if (!VADAPT || vlo) Mma32<f16>::mma(o[db], fv[1], fp[0]);     // V_lo . P_hi
        }
      }
    }
    if constexpr (MSUM) l_run = l_run * alpha + rsum[0];  // the whole row: both half-waves' keys
  };

  // Two register sets, as in gemm.h: the loads of tile t+2 are issued at the top of iteration t and written to LDS at the end
  // of iteration t+1.  Tiles past the end read zeros / stale finite data and are never processed.
  load_global(t0, rk0, rv0);
  load_global(t0 + 1, rk1, rv1);
  store_lds(0, rk0, rv0);
  __syncthreads();
  int t = 0;
  for (; t + 1 < ntile; t += 2) {
    load_global(t0 + t + 2, rk0, rv0);
    __builtin_amdgcn_sched_barrier(0);
    process(t0 + t, 0);
    store_lds(1, rk1, rv1);
    __syncthreads();
    load_global(t0 + t + 3, rk1, rv1);
    __builtin_amdgcn_sched_barrier(0);
    process(t0 + t + 1, 1);
    store_lds(0, rk0, rv0);
    __syncthreads();
  }
  if (t < ntile) process(t0 + t, 0);

  // ---- normalise and store: lane (q, hi) owns O[q][32 db + 8 c + 4 hi + 0..3] --------------------------
  const float l_tot = MSUM ? l_run : l_run + __shfl_xor(l_run, 32, 64);
  if constexpr (SPLIT) {  // unnormalised partial result of this key range; flash_combine_kernel finishes the row
    if (qrow < n) {
      const int64_t prow = ((int64_t)bh * n + qrow) * a.kv_split + ks;
      float* po = a.part_o + prow * 64;
#pragma unroll
      for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int c = 0; c < 4; ++c)
          *reinterpret_cast<float4*>(po + db * 32 + 8 * c + 4 * hi) = make_float4(o[db][4 * c], o[db][4 * c + 1], o[db][4 * c + 2], o[db][4 * c + 3]);
      if (hi == 0) { a.part_ml[prow * 2] = m_run; a.part_ml[prow * 2 + 1] = l_tot; }
    }
    return;
  }
  flash_store_row(a, o, 1.0f / l_tot, a.cu_rows ? (int64_t)a.cu_rows[bp] + qrow : (int64_t)bp * n + qrow, hh, hi, qrow < q_end);
}

// ---- LDS fragment reads with hand-placed waits (flash_pipe_kernel) -------------------------------------------------------------------
// hipcc issues a fragment read right before the MFMA that consumes it, so every MFMA pair waits out an LDS round trip (~100+ cycles under
// load).  As in gemm_pp.h the reads are inline asm — issued where the source says, early — and each wait names the registers it guards (an
// in/out operand), so their consumers cannot be scheduled above it.  LDS operations of one wave complete in order: lgkmcnt(N) = all but the
// N youngest have landed.  Under the host shim these are plain memory reads.
namespace fa {
#ifdef F5_HIPEMU
inline uint32_t lds_addr(const char* p) { return (uint32_t)(p - hipemu::dyn_lds()); }
template <int OFF>
inline void read_b128(Frag& f, uint32_t addr) { memcpy(&f, hipemu::dyn_lds() + addr + OFF, 16); }
template <int OFF>
inline void read_2b64(Frag& f, uint32_t addr) {  // 8 bytes at OFF and 8 bytes at OFF + 16
  memcpy(&f, hipemu::dyn_lds() + addr + OFF, 8);
  memcpy(reinterpret_cast<char*>(&f) + 8, hipemu::dyn_lds() + addr + OFF + 16, 8);
}
template <int N>
inline void landed(Frag&, Frag&) {}
template <int N>
inline void landed(Frag (&)[8]) {}
template <int N>
inline void landed_behind(Frag (&)[8], f32x16&, f32x16&) {}
#else
__device__ __forceinline__ uint32_t lds_addr(const char* p) { return (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const char*)p; }
template <int OFF>
__device__ __forceinline__ void read_b128(Frag& f, uint32_t addr) {
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(f.f) : "v"(addr), "n"(OFF));
}
template <int OFF>
__device__ __forceinline__ void read_2b64(Frag& f, uint32_t addr) {
  static_assert(OFF % 8 == 0 && OFF / 8 + 2 < 256, "ds_read2_b64 offsets are 8-bit counts of 8 bytes");
  asm volatile("ds_read2_b64 %0, %1 offset0:%2 offset1:%3" : "=v"(f.f) : "v"(addr), "n"(OFF / 8), "n"(OFF / 8 + 2));
}
template <int N>
__device__ __forceinline__ void landed(Frag& a, Frag& b) {
  asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(a.f), "+v"(b.f) : "n"(N));
}
template <int N>
__device__ __forceinline__ void landed(Frag (&k)[8]) {
  asm volatile("s_waitcnt lgkmcnt(%8)" : "+v"(k[0].f), "+v"(k[1].f), "+v"(k[2].f), "+v"(k[3].f), "+v"(k[4].f), "+v"(k[5].f), "+v"(k[6].f), "+v"(k[7].f) : "n"(N));
}
// the same wait, additionally ordered behind the MFMAs that produce x and y (pure intrinsics otherwise float above a volatile asm)
template <int N>
__device__ __forceinline__ void landed_behind(Frag (&k)[8], f32x16& x, f32x16& y) {
  asm volatile("s_waitcnt lgkmcnt(%10)"
               : "+v"(k[0].f), "+v"(k[1].f), "+v"(k[2].f), "+v"(k[3].f), "+v"(k[4].f), "+v"(k[5].f), "+v"(k[6].f), "+v"(k[7].f), "+v"(x), "+v"(y)
               : "n"(N));
}
#endif
}  // namespace fa

// Software-pipelined form of the production configuration (round 3): plain fp16 q, k, P, V; base-2 scores (log2q); lazy reference maximum;
// unsplit keys.  In flash_attn_kernel a tile is three DEPENDENT phases per wave — score MFMAs, softmax VALU, PV MFMAs — so a wave alone on
// its SIMD (B = 1: 6 waves on 4 SIMDs) leaves the matrix pipe idle during the exponentials and the VALU idle during the products, and two
// waves only overlap as far as their phases happen to differ.  Here the scores of tile t + 1 are issued INSIDE the softmax of tile t: one
// score MFMA (32 cycles of the matrix pipe), then four exponentials + their fp16 conversions (VALU, independent of that MFMA), eight times.
// LDS staging is skewed to match: ring slot j ("bundle j") holds the K tile j + 1 and the V^T tile j, so an iteration still reads one slot
// and fills the other; the prologue brings K tile 0 alone.  Same arithmetic as flash_attn_kernel<1, 1, NW, false, VSUM, true> except for
// the one extra (discarded) score tile past the end.
// Measured (profiles/r03j_attn_pipe_ab.log): the overlap alone changed nothing (32.6 -> 32.9 us at B' = 2), the early fragment reads on top
// of it -3 % there and nothing with two workgroups per CU: neither a wave's phase order nor its LDS round trips is what the ~80 cycles
// per MFMA slot of this kernel are spent on.  The launcher uses this form for the one-round 192-row launch only.
// ABL (microbenchmark only, results are garbage): 1 no exponentials, 2 no score MFMAs, 3 no PV / row-sum MFMAs, 4 no K / V traffic (global
// loads, LDS stores; barriers kept), 5 = 4 without the barriers, 6 no row maximum / reference check, 7 = 1 + 6 (no softmax VALU at all).
template <int NW, bool VSUM, int ABL = 0>
__global__ __launch_bounds__(64 * NW, NW == 4 ? 2 : 1) void flash_pipe_kernel(FlashArgs a) {
  constexpr int STAGE = K_PLANE + V_PLANE;
  F5_DYN_LDS(char, smem);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, hi = lane >> 5, ql = lane & 31;
  const int bid = blockIdx.x;  // XCD-aware placement as flash_attn_kernel
  const int q8 = a.nwg >> 3, r8 = a.nwg & 7, xcd = bid & 7, slot = bid >> 3;
  const int L = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + slot;
  const int bh = L / a.nqb, qb = L - bh * a.nqb;
  const int bp = bh / a.heads, hh = bh - bp * a.heads;
  const int n = a.n;
  int kv_end = a.kvlen ? min(a.kvlen[bp], n) : n, hole_lo = 0, hole_hi = 0;
  if (a.kvlen && a.kvlen2) {
    hole_lo = kv_end;
    hole_hi = min(a.seg2_off, n);
    kv_end = min(a.seg2_off + a.kvlen2[bp], n);
    if (hole_lo >= hole_hi) hole_lo = hole_hi = 0;
  }
  const int q_end = a.cu_rows ? (a.kvlen ? min(a.kvlen[bp], n) : n) : n;
  if (qb * (32 * NW) >= q_end) return;
  const int ntile = (kv_end + KT - 1) / KT;

  const uint32_t k_bytes = (uint32_t)n * 128u, v_bytes = (uint32_t)(64 * a.ldv) * 2u;
  const BufRsrc Kr = make_rsrc(a.k + (int64_t)bh * n * 64, k_bytes);
  const BufRsrc Vr = make_rsrc(a.vt + (int64_t)bh * 64 * a.ldv, v_bytes);
  const f16* Qp = a.q + (int64_t)bh * n * 64;
  const int qrow = qb * (32 * NW) + wave * 32 + ql;
  Frag fq[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks)
    fq[ks].u = qrow < n ? *reinterpret_cast<const uint4*>(Qp + (int64_t)qrow * 64 + ks * 16 + hi * 8) : make_uint4(0, 0, 0, 0);

  uint32_t k_off[2], v_off[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int c = tid + i * 256, row = c >> 3, col = c & 7;
    k_off[i] = (uint32_t)(row * 128 + col * 16);
    v_off[i] = (uint32_t)(row * a.ldv * 2 + col * 16);
  }
  uint4 rk0[2], rv0[2], rk1[2], rv1[2];
  int t_now = 0;  // (ablations 4, 5)
  auto load_bundle = [&](int j, uint4 (&rk)[2], uint4 (&rv)[2]) {  // K tile j + 1 and V^T tile j (j = -1: K tile 0 alone)
    if (NW > 4 && wave >= 4) return;
    if ((ABL == 4 || ABL == 5) && j > 0) return;
    const uint32_t keyk = (uint32_t)(j + 1) * KT, keyv = (uint32_t)(j < 0 ? 0 : j) * KT;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const uint32_t vkey_off = keyv * 2 + (uint32_t)((tid + i * 256) & 7) * 16;
      const uint32_t voff = (j >= 0 && vkey_off < (uint32_t)a.ldv * 2) ? v_off[i] + keyv * 2 : OOB_OFF;
      rk[i] = buffer_load_b128(Kr, k_off[i] + keyk * 128);  // past the last key: zeros (descriptor bounds)
      rv[i] = buffer_load_b128(Vr, voff);
    }
  };
  auto store_lds = [&](int stage, const uint4 (&rk)[2], const uint4 (&rv)[2]) {
    if ((ABL == 4 || ABL == 5) && t_now > 0) return;
    if (NW > 4 && wave >= 4) return;
    char* base = smem + stage * STAGE;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int c = tid + i * 256, row = c >> 3, col = c & 7;
      *reinterpret_cast<uint4*>(base + row * K_ROWB + col * 16) = rk[i];
      char* vd = base + K_PLANE + row * V_ROWB + col * 16;
      *reinterpret_cast<uint2*>(vd) = make_uint2(rv[i].x, rv[i].y);
      *reinterpret_cast<uint2*>(vd + 8) = make_uint2(rv[i].z, rv[i].w);
    }
  };

  f32x16 o[2], negm, sa[2], sb[2];
#pragma unroll
  for (int r = 0; r < 16; ++r) { o[0][r] = 0.f; o[1][r] = 0.f; negm[r] = 0.f; }
  float l_run = 0.f;
  bool first = true;
  Frag ones;
#pragma unroll
  for (int e = 0; e < 8; ++e) ones.h[e] = (f16)1.0f;

  // one iteration: softmax of tile t (its scores `cur`, relative to the reference, were produced by the previous iteration), the scores
  // of tile t + 1 into `nxt` from the K plane of `stage`, and O += V^T . P^T from its V^T plane.  Fragment reads: all eight K fragments
  // first (they land under the row maximum), the V^T fragments of key groups 0 and 1 before the score MFMAs, those of groups 2 and 3
  // as groups 0 and 1 retire.
  const uint32_t lds_k = fa::lds_addr(smem) + ql * K_ROWB + hi * 16, lds_v = fa::lds_addr(smem) + K_PLANE + ql * V_ROWB + hi * 8;
  auto step = [&](int t, int stage, f32x16 (&cur)[2], f32x16 (&nxt)[2]) {
    const uint32_t aK = lds_k + stage * STAGE, aV0 = lds_v + stage * STAGE, aV1 = aV0 + 32 * V_ROWB;
    Frag fk[8], fv[4][2];
    static_for<8>([&](auto I) {
      constexpr int i = decltype(I)::value;
      fa::read_b128<(i >> 2) * 32 * K_ROWB + (i & 3) * 32>(fk[i], aK);
    });
    if ((t + 1) * KT > kv_end || (t * KT < hole_hi && (t + 1) * KT > hole_lo)) {
      flash_mask_tile(cur, t, hi, kv_end, hole_lo, hole_hi);
    }
    float mx = cur[0][0];
    if constexpr (ABL != 6 && ABL != 7) {
#pragma unroll
      for (int r = 1; r < 16; ++r) mx = fmaxf(mx, cur[0][r]);
#pragma unroll
      for (int r = 0; r < 16; ++r) mx = fmaxf(mx, cur[1][r]);
    }
    float alpha = 1.0f, rs = 0.f;
    if ((ABL == 6 || ABL == 7) ? first : f5_wave_any(first || mx > LAZY_TAU)) {  // raise the reference (rare after the first tiles), as flash_attn_kernel's LAZY branch
      const float mxr = fmaxf(mx, __shfl_xor(mx, 32, 64));
      const float delta = first ? mxr : (mxr > LAZY_TAU ? mxr : 0.f);
      if (!first) {
        alpha = __builtin_amdgcn_exp2f(-delta);
#pragma unroll
        for (int r = 0; r < 16; ++r) { o[0][r] *= alpha; o[1][r] *= alpha; }
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) { cur[0][r] -= delta; cur[1][r] -= delta; negm[r] -= delta; }
      first = false;
    }
    fa::read_2b64<0>(fv[0][0], aV0);
    fa::read_2b64<0>(fv[0][1], aV1);
    fa::read_2b64<32>(fv[1][0], aV0);
    fa::read_2b64<32>(fv[1][1], aV1);
    Frag fp[4];
    fa::landed<4>(fk);  // the K fragments (older than the four V^T reads)
    // scores of the next tile (matrix pipe) under the exponentials of this one (VALU)
#pragma unroll
    for (int r = 0; r < 16; ++r) { nxt[0][r] = negm[r]; nxt[1][r] = negm[r]; }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if constexpr (ABL != 2) Mma32<f16>::mma(nxt[i >> 2], fk[i], fq[i & 3]);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float p = (ABL == 1 || ABL == 7) ? cur[i >> 2][4 * (i & 3) + e] : __builtin_amdgcn_exp2f(cur[i >> 2][4 * (i & 3) + e]);
        fp[i >> 1].h[4 * (i & 1) + e] = (f16)p;
      }
      if constexpr (VSUM) {  // row sum over the rounded values the P.V product multiplies: one v_dot2_f32_f16 per packed pair
        rs = f5_sum2_f16((i & 1) ? fp[i >> 1].u.z : fp[i >> 1].u.x, rs);
        rs = f5_sum2_f16((i & 1) ? fp[i >> 1].u.w : fp[i >> 1].u.y, rs);
      }
    }
    // O^T += V^T . P^T (+ the row sums on the matrix pipe unless VSUM)
    f32x16 rsum;
#pragma unroll
    for (int r = 0; r < 16; ++r) rsum[r] = 0.f;
    static_for<4>([&](auto G) {  // reads in flight: group g + 1's two (none behind the last group)
      constexpr int g = decltype(G)::value;
      fa::landed<(g < 3 ? 2 : 0)>(fv[g][0], fv[g][1]);
      if constexpr (g < 2) {  // the fragments of group g + 2 behind the MFMAs of group g
        fa::read_2b64<(g + 2) * 32>(fv[g + 2][0], aV0);
        fa::read_2b64<(g + 2) * 32>(fv[g + 2][1], aV1);
      }
      if constexpr (ABL == 3) {  // keep the operands alive without the matrix pipe
        o[0][g] += fv[g][0].f[0] + (float)fp[g].h[0];
        o[1][g] += fv[g][1].f[0];
      } else {
        if constexpr (!VSUM) Mma32<f16>::mma(rsum, ones, fp[g]);
        Mma32<f16>::mma(o[0], fv[g][0], fp[g]);
        Mma32<f16>::mma(o[1], fv[g][1], fp[g]);
      }
    });
    l_run = l_run * alpha + (VSUM ? rs : rsum[0]);
  };

  load_bundle(-1, rk0, rv0);
  load_bundle(0, rk1, rv1);
  store_lds(1, rk0, rv0);  // K tile 0 -> slot 1
  store_lds(0, rk1, rv1);  // bundle 0 -> slot 0
  load_bundle(1, rk1, rv1);
  __syncthreads();
  {
    const char* sK = smem + STAGE + ql * K_ROWB + hi * 16;
#pragma unroll
    for (int r = 0; r < 16; ++r) { sa[0][r] = 0.f; sa[1][r] = 0.f; }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      Frag fk;
      fk.u = *reinterpret_cast<const uint4*>(sK + (i >> 2) * 32 * K_ROWB + (i & 3) * 32);
      Mma32<f16>::mma(sa[i >> 2], fk, fq[i & 3]);
    }
  }
  __syncthreads();  // slot 1 is refilled at the end of the first iteration
  int t = 0;
  for (; t + 1 < ntile; t += 2) {
    t_now = t + 1;
    load_bundle(t + 2, rk0, rv0);
    __builtin_amdgcn_sched_barrier(0);
    step(t, 0, sa, sb);
    store_lds(1, rk1, rv1);
    if constexpr (ABL != 5) __syncthreads();
    load_bundle(t + 3, rk1, rv1);
    __builtin_amdgcn_sched_barrier(0);
    step(t + 1, 1, sb, sa);
    store_lds(0, rk0, rv0);
    if constexpr (ABL != 5) __syncthreads();
  }
  if (t < ntile) step(t, 0, sa, sb);

  const float l_tot = VSUM ? l_run + __shfl_xor(l_run, 32, 64) : l_run;
  flash_store_row(a, o, 1.0f / l_tot, a.cu_rows ? (int64_t)a.cu_rows[bp] + qrow : (int64_t)bp * n + qrow, hh, hi, qrow < q_end);
}

// merge the kv_split partial results of every query row: O = sum_s e^(m_s - m) O_s / sum_s e^(m_s - m) l_s with m = max_s m_s (the
// same base-2 exponentials as the kernel), then the operand store of the unsplit kernel.  One thread = 4 output features of one row.
__global__ __launch_bounds__(256) void flash_combine_kernel(FlashArgs a, int64_t rows) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= rows * 16) return;
  const int64_t row = idx >> 4;  // bh * n + qrow
  const int d4 = (int)(idx & 15) * 4;
  const int bh = (int)(row / a.n), qrow = (int)(row - (int64_t)bh * a.n);
  const int bp = bh / a.heads, hh = bh - bp * a.heads;
  const float* ml = a.part_ml + row * a.kv_split * 2;
  float m = ml[0];
  for (int s = 1; s < a.kv_split; ++s) m = fmaxf(m, ml[2 * s]);
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  float l = 0.f;
  for (int s = 0; s < a.kv_split; ++s) {
    const float w = __builtin_amdgcn_exp2f((ml[2 * s] - m) * (a.log2q ? 1.0f : LOG2E));  // 0 for a part without valid keys (m_s = -inf)
    const float4 v = *reinterpret_cast<const float4*>(a.part_o + (row * a.kv_split + s) * 64 + d4);
    acc.x = fmaf(w, v.x, acc.x); acc.y = fmaf(w, v.y, acc.y); acc.z = fmaf(w, v.z, acc.z); acc.w = fmaf(w, v.w, acc.w);
    l = fmaf(w, ml[2 * s + 1], l);
  }
  const float inv = 1.0f / l;
  const float x[4] = {acc.x * inv, acc.y * inv, acc.z * inv, acc.w * inv};
  f16x4 oh, ol;
#pragma unroll
  for (int e = 0; e < 4; ++e) { f16 h, w; split_f16(x[e], h, w); oh[e] = h; ol[e] = w; }
  const int64_t orow = ((int64_t)bp * a.n + qrow) * ((int64_t)a.heads * 64 * (a.o_packed ? 2 : 1));
  const int d = pk_off(hh * 64 + d4, a.o_packed);
  *reinterpret_cast<f16x4*>(a.o + orow + d) = oh;
  if (a.o_lo) *reinterpret_cast<f16x4*>(a.o_lo + orow + d) = ol;
}

}  // namespace
