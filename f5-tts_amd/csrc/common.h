// common.h — shared device helpers for libf5hip (gfx950 / CDNA4 only; wave = 64).
#pragma once
#include <type_traits>
#include <utility>
#ifdef F5_HIPEMU  // tests/hipemu: the same kernel source compiled for the host (one std::thread per HIP thread)
#include "hipemu.h"
#else
#include <hip/hip_runtime.h>
#endif
#include <stdint.h>

// the workgroup's dynamic LDS allocation as an array of `type`
#ifdef F5_HIPEMU
#define F5_DYN_LDS(type, name) type* name = reinterpret_cast<type*>(hipemu::dyn_lds())
#else
#define F5_DYN_LDS(type, name) extern __shared__ __attribute__((aligned(16))) type name[]
#endif

typedef _Float16 f16;
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef f16 f16x4 __attribute__((ext_vector_type(4)));
typedef f16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// compile-time loop: f(integral_constant<int, 0>) ... f(integral_constant<int, N-1>) (indices usable as immediates / template arguments)
template <typename F, int... I>
__host__ __device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, I...>) {
  (f(std::integral_constant<int, I>{}), ...);
}
template <int N, typename F>
__host__ __device__ __forceinline__ void static_for(F&& f) {
  static_for_impl(f, std::make_integer_sequence<int, N>{});
}

// One 16-byte MFMA operand fragment: 8 halves (k = 8*(lane>>5)+0..7 of a 16-wide k-step of
// v_mfma_f32_32x32x16_f16) or 4 floats (k = 4*(lane>>5)+j, j-th of four v_mfma_f32_32x32x2_f32:
// the dot product is order-free in k, so a lane may own 4 consecutive k as long as A and B agree).
union Frag {
  uint4 u;
  f16x8 h;
  f32x4 f;
};

// acc + lo + hi of a packed pair of halves in ONE instruction (v_dot2_f32_f16 against (1, 1), fp32 accumulate): the row sums of the flash
// kernels' P, taken over the values the P.V product multiplies (the fp16-rounded ones) at half the adds
__device__ __forceinline__ float f5_sum2_f16(uint32_t pair, float acc) {
#ifdef F5_HIPEMU
  union { uint32_t u; f16 h[2]; } v;
  v.u = pair;
  return acc + ((float)v.h[0] + (float)v.h[1]);
#else
  typedef _Float16 f5_half2 __attribute__((ext_vector_type(2)));
  union { uint32_t u; f5_half2 v; } a;
  a.u = pair;
  const f5_half2 ones = {(_Float16)1.0f, (_Float16)1.0f};
  return __builtin_amdgcn_fdot2(a.v, ones, acc, false);
#endif
}

template <typename T>
struct Mma32;
template <>
struct Mma32<f16> {
  static __device__ __forceinline__ void mma(f32x16& acc, const Frag& a, const Frag& b) {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a.h, b.h, acc, 0, 0, 0);
  }
};
template <>
struct Mma32<float> {
  static __device__ __forceinline__ void mma(f32x16& acc, const Frag& a, const Frag& b) {
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.f[0], b.f[0], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.f[1], b.f[1], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.f[2], b.f[2], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.f[3], b.f[3], acc, 0, 0, 0);
  }
};

// 32x32 MFMA accumulator layout (dtype independent): element r of lane l is
//   D[i][j],  j = l & 31,  i = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5).
// All GEMM-like kernels here put the WEIGHT/output-channel axis on i and the ROW (frame) axis on j,
// so a lane owns 4 consecutive output channels of one row per register quad -> 16-byte stores,
// and rope's (2i, 2i+1) pairs are lane-local.

// buffer-descriptor loads: out-of-range offsets return zeros (hardware bounds check), no predication needed
typedef __amdgpu_buffer_rsrc_t BufRsrc;
constexpr uint32_t OOB_OFF = 0xFFFFFFF0u;  // >= any num_records used here
constexpr uint32_t OOB_ROW = 0x80000000u;  // base offset of a non-existent row: stays out of range after adding a k offset
                                           // (every operand matrix is < 2 GiB; checked by the launchers)
__device__ __forceinline__ BufRsrc make_rsrc(const void* p, uint32_t bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), /*stride*/ (short)0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ uint4 buffer_load_b128(BufRsrc r, uint32_t byte_off) {
  typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
  const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)byte_off, 0, 0);
  return make_uint4(v[0], v[1], v[2], v[3]);
}

// Packed fp16x3 operand row: k-blocks of 32 elements stored as [32 hi | 32 lo] halves (one 128-byte line per block), so
// hi(k) sits at pk_off(k) and lo(k) 32 halves later.  pk == 0: plain row (offset k).
__device__ __host__ __forceinline__ int pk_off(int k, int pk) { return pk ? ((k >> 5) << 6) + (k & 31) : k; }

// activations -------------------------------------------------------------------------------------
enum { ACT_NONE = 0, ACT_SILU = 1, ACT_GELU_ERF = 2, ACT_GELU_TANH = 3, ACT_MISH = 4 };

// exp / reciprocal on the hardware transcendental unit (v_exp_f32 / v_rcp_f32, ~1 ulp each): the activation epilogues run once
// per GEMM output, so libm-grade expf/tanhf/division (30-40 VALU instructions each) would cost as much as a K=1024 main loop.
__device__ __forceinline__ float fast_exp(float x) { return __builtin_amdgcn_exp2f(x * 1.4426950408889634f); }
__device__ __forceinline__ float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ float act_silu(float x) { return x * fast_rcp(1.0f + fast_exp(-x)); }
__device__ __forceinline__ float act_gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }
__device__ __forceinline__ float act_gelu_tanh(float x) {
  // torch: 0.5*x*(1+tanh(u)), u = sqrt(2/pi)*(x+0.044715*x^3);  0.5*(1+tanh(u)) = sigmoid(2u) = 1 / (1 + 2^z), z = -2 u log2(e) = x (c0 + c1 x^2).
  // 1 / (1 + 2^z) has no cancellation on either side (z -> +inf: 2^z = inf, rcp = 0, result -0 like torch; z -> -inf: 2^z = 0, result x), so the
  // form needs no |u| / select dance: 4 full-rate VALU instructions + the two quarter-rate transcendentals per element (round 6: the epilogue of
  // FeedForward's first linear is VALU-bound - 128 outputs per lane at 90 cycles each were 30 % of a many-round launch; now 58)
  const float c0 = -2.0f * 0.7978845608028654f * 1.4426950408889634f, c1 = c0 * 0.044715f;
  const float p = __builtin_fmaf(x * x, c1, c0);
  const float e = __builtin_amdgcn_exp2f(x * p);
  return x * fast_rcp(1.0f + e);
}
__device__ __forceinline__ float act_mish(float x) {
  // x * tanh(softplus(x)); with e = e^x: tanh(log(1+e)) = ((1+e)^2 - 1) / ((1+e)^2 + 1) = n / (n + 2), n = e*(e+2).
  // torch's softplus threshold (x > 20 -> softplus = x, tanh = 1 in fp32) is kept.
  const float e = fast_exp(fminf(x, 20.0f));
  const float n = e * (e + 2.0f);
  return x > 20.0f ? x : x * n * fast_rcp(n + 2.0f);
}
__device__ __forceinline__ float apply_act(int act, float x) {
  switch (act) {
    case ACT_SILU: return act_silu(x);
    case ACT_GELU_ERF: return act_gelu_erf(x);
    case ACT_GELU_TANH: return act_gelu_tanh(x);
    case ACT_MISH: return act_mish(x);
    default: return x;
  }
}

// Pin two values to fp32 registers before halves and remainders are taken off them.  Why (round 6): where a value is a product x * r, hipcc
// is free (-ffp-contract=fast) to form a half as v_fma_mixlo_f16(x, r, 0) — the EXACT product rounded once — in one place and as the
// conversion of the fp32-ROUNDED product in another: for the stored half and for the half a remainder v - (float)hi is taken from, or in one
// unrolled copy of an epilogue and not in the next.  The two agree except where the product sits within an fp32 ulp of a half-way point
// (tools/probes/cvt_pk_f16_probe.hip: 144 of 2 M products).  Seen on the GPU, never on the host shim (one conversion): hi + lo off by one
// fp16 ulp in 0.007 % of the operand elements (the stored hi and the remainder's hi derived apart), and identical utterances of one batch
// 6e-5 apart (rows in different 32-row tiles of a wave rounded by different instructions).  Behind the pin every half is RNE16(fl32(value))
// wherever and however often it is derived, and two of them convert with one v_cvt_pk_f16_f32.
__device__ __forceinline__ void pin_f32(float& a, float& b) {
#ifndef F5_HIPEMU
  asm volatile("" : "+v"(a), "+v"(b));
#endif
}

// fp16 hi/lo split: v ~= hi + lo with 22 significant bits
__device__ __forceinline__ void split_f16(float v, f16& hi, f16& lo) {
#ifndef F5_HIPEMU
  asm volatile("" : "+v"(v));  // v as ONE fp32 value: the stored half and the half the remainder is taken from cannot be derived apart (pin_f32)
#endif
  hi = (f16)v;
  lo = (f16)(v - (float)hi);
}

// half-wave exchange (v_permlane32_swap): lanes 32-63 of a trade places with lanes 0-31 of b
__device__ __forceinline__ void f5_swap32(uint32_t& a, uint32_t& b) {
#ifdef F5_HIPEMU
  hipemu::permlane32_swap(a, b);
#else
  const auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
  a = r[0];
  b = r[1];
#endif
}
// ---- fp16 + MX-fp6 corrections (operand mode fp16m, round 4) -----------------------------------------------------------------------------
// The three-term product A_hi W_hi + A_hi W_lo + A_lo W_hi costs three fp16 MFMAs.  Its two correction terms are 2^-11 of the first and
// only need ~4 significant bits per factor: on gfx950 they fit ONE v_mfma_scale_f32_32x32x64_f8f6f4 with e2m3 operands per 32 k, which runs
// at 4x the fp16 rate (tools/probes/mx6_probe.hip: 2 fp16 + 1 fp6 MFMA take 128 ns where 6 fp16 MFMAs take 268, every CU busy) — 1.5
// MFMA-equivalents per product instead of 3, at the accuracy of the split (oracle/operand_scheme_emulation.py: 3.07e-4 against 2.67e-4 on
// the full-size golden with all four block GEMMs in this form).
//
// Operand line of the mode = one 32-k block of one row, 128 bytes like the fp16x3 line:
//   [ 32 hi halves (k order) | P_0 : 32 bytes | P_1 : 32 bytes ],   P_h = 24 bytes of 32 e2m3 elements, 1 scale word, 1 zero word.
// P_h covers the 16 k-values an accumulator lane of half-wave h owns in a 32-channel tile, k_t = 8 (t / 4) + 4 h + (t % 4), t = 0..15 —
// so the GEMM and attention epilogues pack it from registers, no lane exchange.  Its elements 2 t, 2 t + 1 are, for an ACTIVATION row,
// (c6(x[k_t]), l6(x[k_t])) and for a WEIGHT row (l6(w[k_t]), c6(w[k_t])), where c6(v) = e2m3(v / S) is the coarse value, l6(v) =
// e2m3((v - fp16(v)) 2^11 / S) the fp16 rounding remainder, and S = 2^(E - 2) the block scale (E = exponent of the largest |v| of the 16).
// The fp6 MFMA multiplies element by element, so it sums c6(x) l6(w) + l6(x) c6(w) over the 32 k of the block, with the per-lane scale
// bytes 2^(E_x - 2 - 11) and 2^(E_w - 2): exactly the two correction terms.  MFMA lane (row, half g) reads P_g: 32 bytes = the 8-register
// operand (registers 6, 7 are ignored by the instruction for fp6; register 6 is the scale word, its byte 0 the E8M0 exponent).
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x6 __attribute__((ext_vector_type(6)));
constexpr int MX_MIN_EXP = 16;  // smallest biased exponent a block scale is computed from (keeps both scale bytes > 0; such blocks are ~0)

// 16 values (index t = 4 q + e <-> k = 8 q + 4 h + e) -> their 16 fp16 `hi` halves (two per dword, t order) and the P_h words
template <bool WEIGHT>
__device__ __forceinline__ void mx_pack16(const float (&v)[16], uint32_t (&hi)[8], uint32_t (&p)[8]) {
  float amax = 0.f;
#pragma unroll
  for (int t = 0; t < 16; ++t) amax = fmaxf(amax, fabsf(v[t]));
  union { float f; uint32_t u; } cv;
  cv.f = amax;
  int ex = (int)((cv.u >> 23) & 0xffu);
  ex = ex < MX_MIN_EXP ? MX_MIN_EXP : ex;
  const int sb = ex - 2;  // biased exponent of S: the largest element lands in [4, 8)
  cv.u = (uint32_t)sb << 23;
  const float S = cv.f;
  f32x16 c, l;
#pragma unroll
  for (int t = 0; t < 16; t += 2) {
    float va = v[t], vb = v[t + 1];
    pin_f32(va, vb);  // halves, coarse values and remainders all come off the same two fp32 values
    const f16x2 hp = {(f16)va, (f16)vb};  // one v_cvt_pk_f16_f32
    hi[t >> 1] = __builtin_bit_cast(uint32_t, hp);
    c[t] = va; c[t + 1] = vb;
    l[t] = (va - (float)hp[0]) * 2048.0f;  // exact: the remainder has <= 13 significant bits, 2^11 is a power of two
    l[t + 1] = (vb - (float)hp[1]) * 2048.0f;
  }
#ifdef F5_HIPEMU
  const u32x6 r = WEIGHT ? __builtin_amdgcn_cvt_scalef32_2xpk16_fp6_f32(l, c, S) : __builtin_amdgcn_cvt_scalef32_2xpk16_fp6_f32(c, l, S);
#else
  // v_cvt_scalef32_2xpk16_fp6_f32 converts its 16 + 16 values in passes of 4 + 4 (1.5 destination dwords each) and reads scale and sources
  // in every pass, so its DESTINATION must overlap neither.  Two hazards of the register allocator's freedom, both found on the GPU with the
  // host shim (no register file) clean:
  //  * round 5: the scale in the first destination dword (it is dead behind the instruction) — everything behind the first pass scaled by
  //    payload bits: the q|k|v epilogue's P words decoded to noise in all 53 kernels (profiles/r05h_mxqk_check_scale_overlap.log);
  //  * round 6: the destination INSIDE a source tuple at a positive offset (v[6:11] <- v[2:17], v[18:33]: layernorm_mx_kernel, whenever the
  //    code around the pack changed a little) — destination dword k is written in pass k / 1.5, source dword j is read in pass j / 4, so
  //    source dwords 4, 5 are payload by the time their pass reads them: one or two of a block's 16 coarse values wrong, every format check
  //    green (2 of 32 fp6 elements at 2^-11 of the product), the trained-like golden at 1.0e-3 instead of 4.9e-4 (bisected over five library
  //    builds, profiles/r06f_fp6_dst_in_src_bisect.log).  A destination at offset 0 of a source is safe and was the common case.
  // Keeping the inputs alive behind the builtin (an empty asm that reads them) stopped the first but not the second: the allocator still
  // split the source tuple.  The instruction is therefore issued as inline asm with an EARLY-CLOBBER destination — no overlap with any input by
  // construction; tests/test_isa_hazards.py scans every instance in the built library for both forms.
  u32x6 r;
  if constexpr (WEIGHT) asm("v_cvt_scalef32_2xpk16_fp6_f32 %0, %1, %2, %3" : "=&v"(r) : "v"(l), "v"(c), "v"(S));
  else asm("v_cvt_scalef32_2xpk16_fp6_f32 %0, %1, %2, %3" : "=&v"(r) : "v"(c), "v"(l), "v"(S));
#endif
#pragma unroll
  for (int i = 0; i < 6; ++i) p[i] = r[i];
  p[6] = (uint32_t)(WEIGHT ? sb : sb - 11);
  p[7] = 0u;
}

// A fragment that an inline-asm ds_read delivered, re-defined BEHIND the wait for it.  The compiler takes an asm read's destination as
// written when the read is issued; what it derives from it on its own — the copies that assemble an MX operand's register tuple, which
// loop-invariant code motion hoists to the read itself; the reuse of a destination dword nobody reads (the zero word) — then happens
// before the data has landed (round 4: NaNs on the GPU only, both forms; tests/test_isa_hazards.py walks the disassembly for them).  An
// empty volatile asm with the fragment as an in/out operand stays behind the (volatile) wait, and everything downstream hangs off its result.
__device__ __forceinline__ void pin_after_wait(uint4& f) {
#ifndef F5_HIPEMU
  typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
  u32x4 v = {f.x, f.y, f.z, f.w};
  asm volatile("" : "+v"(v));
  f = make_uint4(v[0], v[1], v[2], v[3]);
#endif
}

// acc += (the two correction terms of one 32-k block): W's P words in (w0, w1), the activation's in (a0, a1), 16 bytes each (pinned behind
// their wait by the caller: pin_after_wait)
__device__ __forceinline__ void mx_mma(f32x16& acc, const uint4& w0, const uint4& w1, const uint4& a0, const uint4& a1) {
  const i32x8 wv = {(int)w0.x, (int)w0.y, (int)w0.z, (int)w0.w, (int)w1.x, (int)w1.y, (int)w1.z, (int)w1.w};
  const i32x8 av = {(int)a0.x, (int)a0.y, (int)a0.z, (int)a0.w, (int)a1.x, (int)a1.y, (int)a1.z, (int)a1.w};
  acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(wv, av, acc, 2, 2, 0, (int)w1.z, 0, (int)a1.z);
}

// A wave's own LDS writes become readable by its own lanes: LDS operations of one wave are processed in order, so nothing has to be
// waited for — the compiler only must not move the reads above the writes.  (The host shim runs the lanes of a wave as fibers that switch
// at barriers: there it is a real wave rendezvous.)
__device__ __forceinline__ void wave_lds_sync() {
#ifdef F5_HIPEMU
  hipemu::barrier_wait(hipemu::wv().bar);
#else
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#endif
}

// wave-uniform: does any lane hold the predicate?
__device__ __forceinline__ bool f5_wave_any(bool pred) {
#ifdef F5_HIPEMU
  return hipemu::wave_any(pred);
#else
  return __builtin_amdgcn_ballot_w64(pred) != 0ull;
#endif
}

// Wave-wide reductions, the result in every lane.  On the GPU: data-parallel-primitive moves inside the VALU (quad swaps, the two mirrors of a
// 16-lane row, then the row totals handed down the rows with row_bcast 15 / 31 and read from lane 63) instead of six `ds_bpermute` round
// trips through the LDS pipe (what __shfl_xor compiles to): the reductions sit on the critical path of the latency-bound LayerNorm launches.
#ifndef F5_HIPEMU
template <int CTRL, int ROW_MASK = 0xf>
__device__ __forceinline__ float f5_dpp(float v, float identity) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, identity), __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xf, false));
}
#endif
__device__ __forceinline__ float wave_sum(float v) {
#ifdef F5_HIPEMU
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
#else
  v += f5_dpp<0xB1>(v, 0.f);        // quad_perm [1, 0, 3, 2]
  v += f5_dpp<0x4E>(v, 0.f);        // quad_perm [2, 3, 0, 1]: every lane holds its quad's sum
  v += f5_dpp<0x141>(v, 0.f);       // row_half_mirror: 8 lanes
  v += f5_dpp<0x140>(v, 0.f);       // row_mirror: the 16-lane row
  v += f5_dpp<0x142, 0xa>(v, 0.f);  // row_bcast 15 into rows 1 and 3: rows 0 + 1, rows 2 + 3
  v += f5_dpp<0x143, 0xc>(v, 0.f);  // row_bcast 31 into rows 2 and 3: lane 63 holds the total
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
#endif
}
__device__ __forceinline__ float wave_max(float v) {
#ifdef F5_HIPEMU
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
#else
  v = fmaxf(v, f5_dpp<0xB1>(v, v));
  v = fmaxf(v, f5_dpp<0x4E>(v, v));
  v = fmaxf(v, f5_dpp<0x141>(v, v));
  v = fmaxf(v, f5_dpp<0x140>(v, v));
  v = fmaxf(v, f5_dpp<0x142, 0xa>(v, v));  // (rows outside the mask keep their own value: max(v, v))
  v = fmaxf(v, f5_dpp<0x143, 0xc>(v, v));
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
#endif
}
