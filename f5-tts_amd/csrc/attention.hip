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
// NSPLIT/PVSPLIT == 3: fp16 hi/lo split operands (q, k / v, P), 3 MFMAs per product, ~fp32 accuracy; == 1: plain fp16 operands;
// NSPLIT == 2: fp16 hi . hi + MX-fp6 corrections for the scores (the default of the parity modes since round 5).  Softmax statistics, P and
// O accumulate in fp32 in all variants.
#include "attention_kernel.h"

namespace {

template <int NSPLIT, int PVSPLIT>
hipError_t launch(FlashArgs a, int bh, int co, hipStream_t s) {
  constexpr int lds = flash_lds_bytes<NSPLIT, PVSPLIT>();
  const int nqb6 = (a.n + 191) / 192, nqb4 = (a.n + QB - 1) / QB;
  // 192-row blocks when they (times the `co` identical launches that run concurrently: the cond / uncond chains) fit the chip in one
  // round of one workgroup per CU and the 128-row blocks neither fit one round nor fill two per CU
  const int e4 = bh * co * nqb4, e6 = bh * co * nqb6;
  const bool six = e6 <= 256 && e4 > 256 && e4 < 512;
  if (a.kv_split > 1) {  // key-split variant: kv_split workgroups per query block + the merge (attention_kernel.h)
    if (!a.part_o || !a.part_ml || a.kv_split > 8) return hipErrorInvalidValue;
    a.nqb = nqb4; a.nwg = bh * nqb4 * a.kv_split;
    hipLaunchKernelGGL((flash_attn_kernel<NSPLIT, PVSPLIT, 4, true>), dim3(a.nwg), dim3(256), lds, s, a);
    const int64_t rows = (int64_t)bh * a.n;
    hipLaunchKernelGGL(flash_combine_kernel, dim3((unsigned)((rows * 16 + 255) / 256)), dim3(256), 0, s, a, rows);
    return hipGetLastError();
  }
  // Row sums of P: on the matrix pipe (a V^T fragment of ones, attention_kernel.h) where the kernel is VALU-bound — many workgroups per
  // CU: -9 % at B' = 16, -4 % at B' = 64, -6 % at n = 3000 — and on the VALU for the one-round 192-row launch of a single utterance, which is
  // latency-bound and pays +3 % for the extra dependent MFMAs at the end of a tile (profiles/r02h_attn_rowsum_ab.log).
  const bool vsum = six;
  // lazy reference maximum (attention_kernel.h LAZY): whenever q carries log2(e) (profiles/r03e_attn_lazy_ab.log)
  const bool lazy = a.log2q != 0;
  // software-pipelined form of the plain-fp16 lazy configuration (attention_kernel.h flash_pipe_kernel) for the one-round 192-row launch,
  // where SIMDs hold one or two waves and a wave's own phases are all the overlap there is: 32.8 -> 31.8 us at B' = 2 (12.8 against 13.4 ms
  // per B = 1 sample); with two workgroups per CU it ties or loses (B' = 16: 197.9 against 192.0 us, B' = 64: 829 against 825), so the
  // 128-row launches keep the phase-by-phase kernel (profiles/r03j_attn_pipe_ab.log).
  if constexpr (NSPLIT == 1 && PVSPLIT == 1) {
    if (lazy && six) {
      a.nqb = nqb6; a.nwg = bh * a.nqb;
      hipLaunchKernelGGL((flash_pipe_kernel<6, true>), dim3(a.nwg), dim3(384), lds, s, a);  // (row sums on the VALU: vsum == six)
      return hipGetLastError();
    }
  }
#define F5_FLASH(NWV, VS, LZ, THREADS) hipLaunchKernelGGL((flash_attn_kernel<NSPLIT, PVSPLIT, NWV, false, VS, LZ>), dim3(a.nwg), dim3(THREADS), lds, s, a)
  if (six) {
    a.nqb = nqb6; a.nwg = bh * nqb6;
    if (vsum && PVSPLIT != 3) { if (lazy) F5_FLASH(6, true, true, 384); else F5_FLASH(6, true, false, 384); }
    else { if (lazy) F5_FLASH(6, false, true, 384); else F5_FLASH(6, false, false, 384); }
  } else {
    a.nqb = nqb4; a.nwg = bh * nqb4;
    if (vsum && PVSPLIT != 3) { if (lazy) F5_FLASH(4, true, true, 256); else F5_FLASH(4, true, false, 256); }
    else { if (lazy) F5_FLASH(4, false, true, 256); else F5_FLASH(4, false, false, 256); }
  }
#undef F5_FLASH
  return hipGetLastError();
}
template <int NSPLIT, int PVSPLIT, int NW, bool SPLIT, bool VS, bool LZ>
hipError_t set_attr_one() {
  return hipFuncSetAttribute(reinterpret_cast<const void*>(flash_attn_kernel<NSPLIT, PVSPLIT, NW, SPLIT, VS, LZ>), hipFuncAttributeMaxDynamicSharedMemorySize,
                             flash_lds_bytes<NSPLIT, PVSPLIT>());
}
template <int NSPLIT, int PVSPLIT>
hipError_t set_attr() {  // every instantiation launch() can pick
  hipError_t e;
  if ((e = set_attr_one<NSPLIT, PVSPLIT, 4, true, false, false>()) != hipSuccess) return e;
  if ((e = set_attr_one<NSPLIT, PVSPLIT, 4, false, false, false>()) != hipSuccess) return e;
  if ((e = set_attr_one<NSPLIT, PVSPLIT, 4, false, false, true>()) != hipSuccess) return e;
  if ((e = set_attr_one<NSPLIT, PVSPLIT, 6, false, false, false>()) != hipSuccess) return e;
  if ((e = set_attr_one<NSPLIT, PVSPLIT, 6, false, false, true>()) != hipSuccess) return e;
  if constexpr (PVSPLIT != 3) {
    if ((e = set_attr_one<NSPLIT, PVSPLIT, 4, false, true, false>()) != hipSuccess) return e;
    if ((e = set_attr_one<NSPLIT, PVSPLIT, 4, false, true, true>()) != hipSuccess) return e;
    if ((e = set_attr_one<NSPLIT, PVSPLIT, 6, false, true, false>()) != hipSuccess) return e;
    if ((e = set_attr_one<NSPLIT, PVSPLIT, 6, false, true, true>()) != hipSuccess) return e;
  }
  return hipSuccess;
}

}  // namespace

bool flash_attn_available() { return true; }

template <int NW, bool VS>
hipError_t set_attr_pipe() {
  return hipFuncSetAttribute(reinterpret_cast<const void*>(flash_pipe_kernel<NW, VS>), hipFuncAttributeMaxDynamicSharedMemorySize, flash_lds_bytes<1, 1>());
}

hipError_t init_attention_kernels() {
  hipError_t e;
  if ((e = set_attr_pipe<6, true>()) != hipSuccess) return e;
  if ((e = set_attr<1, 1>()) != hipSuccess) return e;
  if ((e = set_attr<2, 1>()) != hipSuccess) return e;
  if ((e = set_attr<2, 2>()) != hipSuccess) return e;
  if ((e = set_attr<2, 3>()) != hipSuccess) return e;
  if ((e = set_attr<3, 1>()) != hipSuccess) return e;
  return set_attr<3, 3>();
}

hipError_t launch_flash_attn(int nsplit, const f16* q, const f16* q_lo, const f16* k, const f16* k_lo, const f16* vt, const f16* vt_lo,
                             int ldv, int Bp, int heads, int n, const int32_t* kvlen, f16* o16, f16* o16_lo, hipStream_t s, int o_packed,
                             const int32_t* kvlen2, int seg2_off, int co_launches, int kv_split, float* part_o, float* part_ml, int log2q, const int32_t* cu_rows) {
  FlashArgs a{};
  a.log2q = log2q;
  a.cu_rows = cu_rows;
  if (cu_rows && (kv_split > 1 || !kvlen)) return hipErrorInvalidValue;  // packed rows: the unsplit kernel, lengths required
  if (o_packed >= 2 && (kv_split > 1 || (reinterpret_cast<uintptr_t>(o16) & 15))) return hipErrorInvalidValue;  // MX lines: the kernels' own epilogue only (not the merge kernel)
  a.kv_split = kv_split < 1 ? 1 : kv_split; a.part_o = part_o; a.part_ml = part_ml;
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
  if (nsplit == 4) {  // scores = fp16 hi . hi + the two correction products as MX-fp6 (q_lo, k_lo hold P words), plain fp16 P and V
    if (!q_lo || !k_lo || ((reinterpret_cast<uintptr_t>(q_lo) | reinterpret_cast<uintptr_t>(k_lo)) & 15)) return hipErrorInvalidValue;
    return launch<2, 1>(a, bh, co_launches < 1 ? 1 : co_launches, s);
  }
  if (nsplit == 5 || nsplit == 6) {  // MX-corrected scores; V as hi + lo halves (5), P as well (6)
    if (!q_lo || !k_lo || !vt_lo || ((reinterpret_cast<uintptr_t>(q_lo) | reinterpret_cast<uintptr_t>(k_lo)) & 15)) return hipErrorInvalidValue;
    return nsplit == 5 ? launch<2, 2>(a, bh, co_launches < 1 ? 1 : co_launches, s) : launch<2, 3>(a, bh, co_launches < 1 ? 1 : co_launches, s);
  }
  if (nsplit == 2) {  // hi/lo q and k (scores), plain fp16 P and V
    if (!q_lo || !k_lo) return hipErrorInvalidValue;
    return launch<3, 1>(a, bh, co_launches < 1 ? 1 : co_launches, s);
  }
  return launch<1, 1>(a, bh, co_launches < 1 ? 1 : co_launches, s);
}
