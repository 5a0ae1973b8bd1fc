// attention.hip — flash-style non-causal self-attention over mel frames for gfx950 (dh = 64).
//
// Replaces F.scaled_dot_product_attention at reference src/f5_tts/model/modules.py:511-520 (the q/k/v
// head split + rope of :481-509 are fused into the QKV GEMM epilogue, gemm.h EpiQKV).
//
// One workgroup = one 128-row query block of one (batch', head): 4 waves x 32 query rows.  Keys/values are
// walked in 64-key tiles, double-buffered in LDS, with the next tile's global loads in flight in registers
// during the MFMAs of the current one.  Everything a query row needs lives in ONE lane pair (l, l^32):
//   S^T = K . Q^T   (v_mfma_f32_32x32x16_f16, A = K tile rows, B = Q rows held in registers)  -> lane (q = l&31, hi = l>>5)
//                    owns scores of keys (r&3) + 8*(r>>2) + 4*hi of each 32-key block: the row max / row sum are
//                    in-lane reductions plus one lane^32 exchange; no LDS round trip for P.
//   O^T = V^T . P^T (A = V^T tile rows, B = P of the lane's own row)                          -> lane owns O[q][d-subset]:
//                    the online-softmax rescale is a lane-local multiply.
// The dot product over keys is order-free, so P's registers are used as the B fragment as they are and the V^T
// fragment is read in the matching key order (two ds_read_b64 per fragment) — no cross-lane shuffles for P.
// V arrives transposed ([BH, 64, ldv], written by the QKV epilogue) so both tiles are plain 16-byte row copies.
//
// NSPLIT/PVSPLIT == 3: fp16 hi/lo split operands (q, k / v, P), 3 MFMAs per product, ~fp32 accuracy (parity mode
// "fp16x3"); == 1: plain fp16 operands.  Softmax statistics, P and O accumulate in fp32 in all variants.
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
};

// NSPLIT: operand split of S = QK^T (1 or 3); PVSPLIT: of O = PV (1 or 3, <= NSPLIT).  The scores feed an exponential, so
// their rounding matters ~10x more than that of P and V: NSPLIT = 3 with PVSPLIT = 1 keeps near-fp32 scores at 4 instead of 6
// MFMAs per key-query pair.
template <int NSPLIT, int PVSPLIT>
constexpr int flash_lds_bytes() {
  return 2 * ((NSPLIT == 3 ? 2 : 1) * K_PLANE + (PVSPLIT == 3 ? 2 : 1) * V_PLANE);
}

// NW: waves per workgroup = 32-row query groups per block.  4 (128 query rows, two workgroups per CU) everywhere except where 6
// (192 rows, one workgroup per CU) makes the grid fit the chip in ONE round: B = 1, N = 1406 gives 11 x 32 = 352 blocks of 128 rows
// (CUs with 2 and CUs with 1 workgroup: 69 % balance) but 8 x 32 = 256 blocks of 192 rows.  Only the first 4 waves stage K / V tiles.
template <int NSPLIT, int PVSPLIT, int NW = 4>
__global__ __launch_bounds__(64 * NW, NW == 4 ? 2 : 1) void flash_attn_kernel(FlashArgs a) {
  constexpr int NPL = NSPLIT == 3 ? 2 : 1;    // planes of q and k
  constexpr int NPV = PVSPLIT == 3 ? 2 : 1;   // planes of v and P
  constexpr int STAGE = NPL * K_PLANE + NPV * V_PLANE;
  extern __shared__ __attribute__((aligned(16))) char smem[];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, hi = lane >> 5, ql = lane & 31;
  // XCD-aware placement: block b runs on XCD b % 8 (observed; speed only) -> give each XCD a contiguous range of
  // (batch', head) so the query blocks sharing one K/V slab hit the same L2.  Bijective for any grid size.
  const int bid = blockIdx.x;
  const int q8 = a.nwg >> 3, r8 = a.nwg & 7, xcd = bid & 7, slot = bid >> 3;
  const int L = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + slot;
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
  const int ntile = (kv_end + KT - 1) / KT;

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

  // Q rows of this wave stay in registers for the whole kernel: fq[p][ks] = Q[q][16 ks + 8 hi .. +7]
  const int qrow = qb * (32 * NW) + wave * 32 + ql;
  Frag fq[NPL][4];
#pragma unroll
  for (int p = 0; p < NPL; ++p)
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
      fq[p][ks].u = qrow < n ? *reinterpret_cast<const uint4*>(Qp[p] + (int64_t)qrow * 64 + ks * 16 + hi * 8) : make_uint4(0, 0, 0, 0);

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

  // one 64-key tile held in LDS stage `stage`: scores, online softmax, O update
  auto process = [&](int t, int stage) {
    const char* base = smem + stage * STAGE;
    const char* sK = base + ql * K_ROWB + hi * 16;
    const char* sV = base + NPL * K_PLANE + ql * V_ROWB + hi * 8;

    // ---- S^T = K . Q^T for the 64 keys of the tile --------------------------------------------------
    f32x16 s[2];
#pragma unroll
    for (int r = 0; r < 16; ++r) { s[0][r] = 0.f; s[1][r] = 0.f; }
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        Frag fk[NPL];
#pragma unroll
        for (int p = 0; p < NPL; ++p) fk[p].u = *reinterpret_cast<const uint4*>(sK + p * K_PLANE + kb * 32 * K_ROWB + ks * 32);
        Mma32<f16>::mma(s[kb], fk[0], fq[0][ks]);
        if constexpr (NPL == 2) {
          Mma32<f16>::mma(s[kb], fk[0], fq[1][ks]);  // K_hi . Q_lo
          Mma32<f16>::mma(s[kb], fk[1], fq[0][ks]);  // K_lo . Q_hi
        }
      }

    // ---- online softmax (fp32), lane-local per query row ----------------------------------------------
    if ((t + 1) * KT > kv_end || (t * KT < hole_hi && (t + 1) * KT > hole_lo)) {  // tail tile (keys >= kv_end do not exist / are
                                                                                      // masked) or a tile touching the masked hole
#pragma unroll
      for (int kb = 0; kb < 2; ++kb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int key = t * KT + kb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
          if (key >= kv_end || (key >= hole_lo && key < hole_hi)) s[kb][r] = -INFINITY;
        }
    }
    float mx = s[0][0];
#pragma unroll
    for (int r = 1; r < 16; ++r) mx = fmaxf(mx, s[0][r]);
#pragma unroll
    for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s[1][r]);
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float m_new = fmaxf(m_run, mx);
    const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * LOG2E);
    const float mb = m_new * LOG2E;
    float rs = 0.f;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float p = __builtin_amdgcn_exp2f(fmaf(s[kb][r], LOG2E, -mb));
        s[kb][r] = p;
        rs += p;
      }
    l_run = l_run * alpha + rs;
    m_run = m_new;
#pragma unroll
    for (int r = 0; r < 16; ++r) { o[0][r] *= alpha; o[1][r] *= alpha; }

    // ---- O^T += V^T . P^T ----------------------------------------------------------------------------
#pragma unroll
    for (int g = 0; g < 4; ++g) {  // 16-key groups of the tile; P registers 8*(g&1) .. +7 of s[g>>1]
      Frag fp[NPV];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float p = s[g >> 1][8 * (g & 1) + e];
        const f16 ph = (f16)p;
        fp[0].h[e] = ph;
        if constexpr (NPV == 2) fp[1].h[e] = (f16)(p - (float)ph);
      }
#pragma unroll
      for (int db = 0; db < 2; ++db) {
        Frag fv[NPV];
#pragma unroll
        for (int p = 0; p < NPV; ++p) {
          const char* src = sV + p * V_PLANE + db * 32 * V_ROWB + g * 32;
          const uint2 v0 = *reinterpret_cast<const uint2*>(src);       // keys 16g + 4hi + 0..3
          const uint2 v1 = *reinterpret_cast<const uint2*>(src + 16);  // keys 16g + 8 + 4hi + 0..3
          fv[p].u = make_uint4(v0.x, v0.y, v1.x, v1.y);
        }
        Mma32<f16>::mma(o[db], fv[0], fp[0]);
        if constexpr (NPV == 2) {
          Mma32<f16>::mma(o[db], fv[0], fp[1]);  // V_hi . P_lo
          Mma32<f16>::mma(o[db], fv[1], fp[0]);  // V_lo . P_hi
        }
      }
    }
  };

  // Two register sets, as in gemm.h: the loads of tile t+2 are issued at the top of iteration t and written to LDS at the end
  // of iteration t+1.  Tiles past the end read zeros / stale finite data and are never processed.
  load_global(0, rk0, rv0);
  load_global(1, rk1, rv1);
  store_lds(0, rk0, rv0);
  __syncthreads();
  int t = 0;
  for (; t + 1 < ntile; t += 2) {
    load_global(t + 2, rk0, rv0);
    __builtin_amdgcn_sched_barrier(0);
    process(t, 0);
    store_lds(1, rk1, rv1);
    __syncthreads();
    load_global(t + 3, rk1, rv1);
    __builtin_amdgcn_sched_barrier(0);
    process(t + 1, 1);
    store_lds(0, rk0, rv0);
    __syncthreads();
  }
  if (t < ntile) process(t, 0);

  // ---- normalise and store: lane (q, hi) owns O[q][32 db + 8 c + 4 hi + 0..3] --------------------------
  const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
  if (qrow < n) {
    const float inv = 1.0f / l_tot;
    const int64_t orow = ((int64_t)bp * n + qrow) * ((int64_t)a.heads * 64 * (a.o_packed ? 2 : 1));
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
}

template <int NSPLIT, int PVSPLIT>
hipError_t launch(FlashArgs a, int bh, int co, hipStream_t s) {
  constexpr int lds = flash_lds_bytes<NSPLIT, PVSPLIT>();
  static const int nw_env = [] { const char* v = getenv("F5HIP_ATTN_WAVES"); return v ? atoi(v) : 0; }();  // tuning knob: 4 or 6
  const int nqb6 = (a.n + 191) / 192, nqb4 = (a.n + QB - 1) / QB;
  // 192-row blocks when they (times the `co` identical launches that run concurrently: the cond / uncond chains) fit the chip in one
  // round of one workgroup per CU and the 128-row blocks neither fit one round nor fill two per CU
  const int e4 = bh * co * nqb4, e6 = bh * co * nqb6;
  const bool six = nw_env ? nw_env == 6 : (e6 <= 256 && e4 > 256 && e4 < 512);
  if (six) {
    a.nqb = nqb6; a.nwg = bh * nqb6;
    hipLaunchKernelGGL((flash_attn_kernel<NSPLIT, PVSPLIT, 6>), dim3(a.nwg), dim3(384), lds, s, a);
  } else {
    a.nqb = nqb4; a.nwg = bh * nqb4;
    hipLaunchKernelGGL((flash_attn_kernel<NSPLIT, PVSPLIT, 4>), dim3(a.nwg), dim3(256), lds, s, a);
  }
  return hipGetLastError();
}
template <int NSPLIT, int PVSPLIT>
hipError_t set_attr() {
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(flash_attn_kernel<NSPLIT, PVSPLIT, 4>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                     flash_lds_bytes<NSPLIT, PVSPLIT>());
  if (e != hipSuccess) return e;
  return hipFuncSetAttribute(reinterpret_cast<const void*>(flash_attn_kernel<NSPLIT, PVSPLIT, 6>), hipFuncAttributeMaxDynamicSharedMemorySize,
                             flash_lds_bytes<NSPLIT, PVSPLIT>());
}

}  // namespace

bool flash_attn_available() { return true; }

hipError_t init_attention_kernels() {
  hipError_t e;
  if ((e = set_attr<1, 1>()) != hipSuccess) return e;
  if ((e = set_attr<3, 1>()) != hipSuccess) return e;
  return set_attr<3, 3>();
}

hipError_t launch_flash_attn(int nsplit, const f16* q, const f16* q_lo, const f16* k, const f16* k_lo, const f16* vt, const f16* vt_lo,
                             int ldv, int Bp, int heads, int n, const int32_t* kvlen, f16* o16, f16* o16_lo, hipStream_t s, int o_packed,
                             const int32_t* kvlen2, int seg2_off, int co_launches) {
  FlashArgs a{};
  a.o_packed = o_packed;
  a.kvlen2 = kvlen2; a.seg2_off = seg2_off;
  a.q = q; a.q_lo = q_lo; a.k = k; a.k_lo = k_lo; a.vt = vt; a.vt_lo = vt_lo;
  a.o = o16; a.o_lo = o16_lo; a.kvlen = kvlen;
  a.n = n; a.ldv = ldv; a.heads = heads;
  const int bh = Bp * heads;
  if (nsplit == 3) {  // hi/lo q, k, v, P
    if (!q_lo || !k_lo || !vt_lo) return hipErrorInvalidValue;
    return launch<3, 3>(a, bh, co_launches < 1 ? 1 : co_launches, s);
  }
  if (nsplit == 2) {  // hi/lo q and k (scores), plain fp16 P and V
    if (!q_lo || !k_lo) return hipErrorInvalidValue;
    return launch<3, 1>(a, bh, co_launches < 1 ? 1 : co_launches, s);
  }
  return launch<1, 1>(a, bh, co_launches < 1 ? 1 : co_launches, s);
}
