// mx6_probe.hip — what gfx950 does with MX-fp6 operands (round 4): the facts the fp16 + MX6-correction GEMM (csrc/gemm_pp.h, operand mode
// fp16m) is built on, measured before the kernel was written.
//
//   A  v_cvt_scalef32_2xpk16_fp6_f32: where the 32 elements of the two sources land in the 192-bit result, what the scale does
//      (divide), rounding (nearest even), saturation
//   B  v_mfma_scale_f32_32x32x64_f8f6f4 with cbsz = blgp = 2 (e2m3): element e of lane l of src0 is A[i = l % 32][k = 32 (l / 32) + e],
//      same for src1 with j; result layout of the fp16 MFMAs; the per-lane scale is byte 0 of the scale register, value 2^(b - 127);
//      checked against a host model on random codes and random scales.  Also: which (half, element) of src0 meets which of src1.
//   C  issue rates per SIMD with every CU busy: 6 fp16 MFMAs (today's three-term product of two 16-wide k-steps) against
//      4 fp16 + 2 fp6 (k = 64) and against 2 x (2 fp16 + 1 fp6) per 32-k block, dependent on one accumulator like the GEMM's loop
//   D  the conversion's VALU cost
// Build: hipcc --offload-arch=gfx950 -O3 -o mx6_probe mx6_probe.hip ; run: ./mx6_probe
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

typedef int v8i __attribute__((ext_vector_type(8)));
typedef unsigned v6u __attribute__((ext_vector_type(6)));
typedef float v16f __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

// ---- A ------------------------------------------------------------------------------------------------------------------------------
__global__ void cvt_kernel(const float* in, float scale, unsigned* out) {
  v16f a, b;
  for (int i = 0; i < 16; ++i) { a[i] = in[i]; b[i] = in[16 + i]; }
  const v6u r = __builtin_amdgcn_cvt_scalef32_2xpk16_fp6_f32(a, b, scale);
  if (threadIdx.x == 0)
    for (int i = 0; i < 6; ++i) out[i] = r[i];
}

static float fp6_val(unsigned c) {  // e2m3
  const unsigned s = c >> 5, e = (c >> 3) & 3, m = c & 7;
  const float v = e == 0 ? m / 8.0f : ldexpf(1.0f + m / 8.0f, (int)e - 1);
  return s ? -v : v;
}
static unsigned get6(const unsigned* w, int e) {  // contiguous little-endian 6-bit fields
  const int bit = 6 * e;
  uint64_t two = w[bit >> 5] | ((uint64_t)(bit / 32 + 1 < 8 ? w[bit / 32 + 1] : 0u) << 32);
  return (unsigned)(two >> (bit & 31)) & 63u;
}

// ---- E (round 4, second half): v_cvt_scalef32_pk32_fp6_f16 — 32 halves -> 32 e2m3 codes; is element i the code of source i? ---------------
typedef _Float16 v32h __attribute__((ext_vector_type(32)));
__global__ void cvt32_kernel(const float* in, float scale, unsigned* out) {
  v32h a;
  for (int i = 0; i < 32; ++i) a[i] = (_Float16)in[i];
  const v6u r = __builtin_amdgcn_cvt_scalef32_pk32_fp6_f16(a, scale);
  if (threadIdx.x == 0)
    for (int i = 0; i < 6; ++i) out[i] = r[i];
}

// ---- B ------------------------------------------------------------------------------------------------------------------------------
__global__ void mfma_kernel(const unsigned* A, const unsigned* B, const int* sa, const int* sb, float* D) {
  const int l = threadIdx.x;
  v8i a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (int)A[l * 8 + i]; b[i] = (int)B[l * 8 + i]; }
  v16f c = {};
  c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 2, 2, 0, sa[l], 0, sb[l]);
  for (int r = 0; r < 16; ++r) D[l * 16 + r] = c[r];
}
// pairing: src0 has 1.0 in (half g0, element e0) of every row, src1 1.0 in (g1, e1) of every column: D = 1 iff the two meet
__global__ void pair_kernel(unsigned char* meet) {
  const int l = threadIdx.x, g = l >> 5, c0 = blockIdx.x, g0 = c0 >> 5, e0 = c0 & 31;
  for (int c1 = 0; c1 < 64; ++c1) {
    const int g1 = c1 >> 5, e1 = c1 & 31;
    unsigned wa[8] = {0}, wb[8] = {0};
    if (g == g0) { const int bit = 6 * e0; uint64_t v = (uint64_t)8u << (bit & 31); wa[bit >> 5] |= (unsigned)v; wa[(bit >> 5) + 1] |= (unsigned)(v >> 32); }
    if (g == g1) { const int bit = 6 * e1; uint64_t v = (uint64_t)8u << (bit & 31); wb[bit >> 5] |= (unsigned)v; wb[(bit >> 5) + 1] |= (unsigned)(v >> 32); }
    v8i a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (int)wa[i]; b[i] = (int)wb[i]; }
    v16f c = {};
    c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 2, 2, 0, 127, 0, 127);
    if (l == 0) meet[c0 * 64 + c1] = c[0] != 0.f;
  }
}

// ---- C / D --------------------------------------------------------------------------------------------------------------------------
template <int MODE>
__global__ __launch_bounds__(512, 1) void pace_kernel(int iters, float* sink) {
  extern __shared__ char smem[];
  const int lane = threadIdx.x & 63;
  v16f acc[8];
  for (int i = 0; i < 8; ++i)
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  f16x8 ha, hb;
  for (int e = 0; e < 8; ++e) { ha[e] = (_Float16)(0.01f * (lane + e)); hb[e] = (_Float16)(0.02f * (lane - e)); }
  v8i qa, qb;
  for (int i = 0; i < 8; ++i) { qa[i] = 0x01041041 * (lane + i); qb[i] = 0x00820820 * (lane + 3 * i); }
  int sca = 120 + (lane & 3), scb = 118 + (lane & 7);
  v16f x;
  for (int i = 0; i < 16; ++i) x[i] = 0.37f * (lane + i);
  unsigned keep = 0;
  for (int it = 0; it < iters; ++it) {
    if constexpr (MODE == 0) {  // 2 k-steps x (hi.hi, hi.lo, lo.hi) on two tiles, dependent triples as in gemm_pp's loop
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int s = 0; s < 2; ++s) {
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ha, hb, acc[t], 0, 0, 0);
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ha, hb, acc[t], 0, 0, 0);
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(hb, ha, acc[t], 0, 0, 0);
        }
    } else if constexpr (MODE == 1) {  // the same 32 k: 2 fp16 + 1 fp6 per tile
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ha, hb, acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(hb, ha, acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(qa, qb, acc[t], 2, 2, 0, sca, 0, scb);
      }
    } else if constexpr (MODE == 2) {  // fp6 only, 6 per iteration on two accumulators
#pragma unroll
      for (int t = 0; t < 6; ++t) acc[t & 1] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(qa, qb, acc[t & 1], 2, 2, 0, sca, 0, scb);
    } else if constexpr (MODE == 3) {  // mode 1 on 6 independent accumulators (no dependent issue)
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        acc[3 * t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ha, hb, acc[3 * t], 0, 0, 0);
        acc[3 * t + 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(hb, ha, acc[3 * t + 1], 0, 0, 0);
        acc[3 * t + 2] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(qa, qb, acc[3 * t + 2], 2, 2, 0, sca, 0, scb);
      }
    } else if constexpr (MODE == 4) {  // fp16 only, 6 per iteration, 6 independent accumulators
#pragma unroll
      for (int t = 0; t < 6; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ha, hb, acc[t], 0, 0, 0);
    } else if constexpr (MODE == 5) {  // 8 conversions of 32 values
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        const v6u r = __builtin_amdgcn_cvt_scalef32_2xpk16_fp6_f32(x, x, 1.0f + t);
        keep ^= r[0] ^ r[5];
        x[t] += 1.0f;
      }
    } else if constexpr (MODE == 6) {  // fp8 (e4m3) operands through the same instruction: 2 per iteration
#pragma unroll
      for (int t = 0; t < 2; ++t) acc[t] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(qa, qb, acc[t], 0, 0, 0, sca, 0, scb);
    }
    asm volatile("" : "+v"(ha), "+v"(hb));
  }
  float s = (float)keep;
  for (int i = 0; i < 8; ++i)
    for (int r = 0; r < 16; ++r) s += acc[i][r];
  if (s == 12345.678f) sink[0] = s;
}

template <int MODE>
static void pace(const char* what, int waves, double mfma_units) {
  float* sink;
  CHECK(hipMalloc(&sink, 4));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  const int iters = 20000, lds = 100 * 1024;
  CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(pace_kernel<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  for (int cus : {1, 256}) {
    hipLaunchKernelGGL(pace_kernel<MODE>, dim3(cus), dim3(64 * waves), lds, 0, 200, sink);
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL(pace_kernel<MODE>, dim3(cus), dim3(64 * waves), lds, 0, iters, sink);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double ns_it = ms * 1e6 / iters;
    printf("pace mode %d %-44s waves/CU %d CUs %3d: %8.1f ns per iteration per wave-slot", MODE, what, waves, cus, ns_it);
    if (mfma_units > 0) printf("  (%.1f ns per fp16-MFMA-equivalent of 32 cycles; per SIMD %.1f ns per unit issued)", ns_it / mfma_units, ns_it / (mfma_units * (waves / 4.0)));
    printf("\n");
  }
  CHECK(hipFree(sink));
}

int main() {
  // ---- A
  {
    float h[32];
    for (int i = 0; i < 16; ++i) h[i] = 0.125f * i;
    const float t[16] = {2.f, 2.5f, 3.f, 3.5f, 4.f, 5.f, 6.f, 7.f, 7.5f, 8.f, 9.f, 100.f, -1.f, -0.125f, 0.0625f, 0.1875f};
    memcpy(h + 16, t, sizeof(t));
    float* din;
    unsigned* dout;
    CHECK(hipMalloc(&din, sizeof(h)));
    CHECK(hipMalloc(&dout, 32));
    CHECK(hipMemcpy(din, h, sizeof(h), hipMemcpyHostToDevice));
    for (float sc : {1.0f, 2.0f, 0.5f, 3.0f}) {
      hipLaunchKernelGGL(cvt_kernel, dim3(1), dim3(64), 0, 0, din, sc, dout);
      unsigned w[8] = {0};
      CHECK(hipMemcpy(w, dout, 24, hipMemcpyDeviceToHost));
      printf("cvt scale %.2f raw %08x %08x %08x %08x %08x %08x\n  decoded (contiguous 6-bit fields):", sc, w[0], w[1], w[2], w[3], w[4], w[5]);
      for (int e = 0; e < 32; ++e) printf(" %g", fp6_val(get6(w, e)));
      printf("\n  inputs                            :");
      for (int e = 0; e < 32; ++e) printf(" %g", h[e]);
      printf("\n");
    }
  }
  // ---- E
  {
    float h[32];
    for (int i = 0; i < 32; ++i) h[i] = (i & 1 ? -1.f : 1.f) * 0.25f * (i % 30);  // 0 .. 7.25 in quarter steps, alternating sign
    float* din;
    unsigned* dout;
    CHECK(hipMalloc(&din, sizeof(h)));
    CHECK(hipMalloc(&dout, 32));
    CHECK(hipMemcpy(din, h, sizeof(h), hipMemcpyHostToDevice));
    for (float sc : {1.0f, 4.0f}) {
      hipLaunchKernelGGL(cvt32_kernel, dim3(1), dim3(64), 0, 0, din, sc, dout);
      unsigned w[8] = {0};
      CHECK(hipMemcpy(w, dout, 24, hipMemcpyDeviceToHost));
      printf("cvt_pk32_f16 scale %.2f raw %08x %08x %08x %08x %08x %08x\n  decoded:", sc, w[0], w[1], w[2], w[3], w[4], w[5]);
      for (int e = 0; e < 32; ++e) printf(" %g", fp6_val(get6(w, e)));
      printf("\n  inputs :");
      for (int e = 0; e < 32; ++e) printf(" %g", h[e]);
      printf("\n");
    }
  }
  // ---- B
  {
    std::vector<unsigned> A(64 * 8, 0), B(64 * 8, 0);
    std::vector<int> sa(64), sb(64);
    srand(7);
    auto put = [](unsigned* w, int e, unsigned c) { const int bit = 6 * e; uint64_t v = (uint64_t)c << (bit & 31); w[bit >> 5] |= (unsigned)v; if ((bit >> 5) + 1 < 8) w[(bit >> 5) + 1] |= (unsigned)(v >> 32); };
    for (int l = 0; l < 64; ++l) {
      for (int e = 0; e < 32; ++e) { put(&A[l * 8], e, rand() & 63); put(&B[l * 8], e, rand() & 63); }
      A[l * 8 + 6] = 0xdeadbeef; A[l * 8 + 7] = 0x12345678;  // must be ignored
      sa[l] = (120 + rand() % 12) | 0x5a3c7700;                 // only byte 0 may count
      sb[l] = (122 + rand() % 9) | 0x11223300;
    }
    unsigned *dA, *dB;
    int *dsa, *dsb;
    float* dD;
    CHECK(hipMalloc(&dA, 2048)); CHECK(hipMalloc(&dB, 2048)); CHECK(hipMalloc(&dsa, 256)); CHECK(hipMalloc(&dsb, 256)); CHECK(hipMalloc(&dD, 4096));
    CHECK(hipMemcpy(dA, A.data(), 2048, hipMemcpyHostToDevice)); CHECK(hipMemcpy(dB, B.data(), 2048, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(dsa, sa.data(), 256, hipMemcpyHostToDevice)); CHECK(hipMemcpy(dsb, sb.data(), 256, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(mfma_kernel, dim3(1), dim3(64), 0, 0, dA, dB, dsa, dsb, dD);
    std::vector<float> D(1024);
    CHECK(hipMemcpy(D.data(), dD, 4096, hipMemcpyDeviceToHost));
    double worst = 0, big = 0;
    for (int l = 0; l < 64; ++l)
      for (int r = 0; r < 16; ++r) {
        const int j = l & 31, i = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
        double ref = 0;
        for (int g = 0; g < 2; ++g) {
          const int la = i + 32 * g, lb = j + 32 * g;
          double part = 0;
          for (int e = 0; e < 32; ++e) part += (double)fp6_val(get6(&A[la * 8], e)) * fp6_val(get6(&B[lb * 8], e));
          ref += part * ldexp(1.0, (sa[la] & 255) - 127) * ldexp(1.0, (sb[lb] & 255) - 127);
        }
        worst = fmax(worst, fabs(ref - D[l * 16 + r]));
        big = fmax(big, fabs(ref));
      }
    printf("mfma_scale fp6 x fp6 against the host model (src0 lane l = row l%%32, k = 32 (l/32) + e; scale = byte 0, 2^(b-127)): max |diff| %.3g of |D| max %.3g  -> %s\n",
           worst, big, worst <= 1e-5 * big ? "MODEL HOLDS" : "MODEL WRONG");
    unsigned char* dm;
    CHECK(hipMalloc(&dm, 4096));
    hipLaunchKernelGGL(pair_kernel, dim3(64), dim3(64), 0, 0, dm);
    std::vector<unsigned char> m(4096);
    CHECK(hipMemcpy(m.data(), dm, 4096, hipMemcpyDeviceToHost));
    int diag = 0, off = 0;
    for (int a = 0; a < 64; ++a)
      for (int b = 0; b < 64; ++b) (a == b ? diag : off) += m[a * 64 + b];
    printf("pairing: (half, element) of src0 meets the same (half, element) of src1 in %d of 64 cases, any other in %d\n", diag, off);
    if (off)
      for (int a = 0; a < 64; ++a) { printf("  src0 (%d,%2d) meets:", a >> 5, a & 31); for (int b = 0; b < 64; ++b) if (m[a * 64 + b]) printf(" (%d,%d)", b >> 5, b & 31); printf("\n"); }
  }
  // ---- C / D
  if (getenv("MX6_PROBE_NO_PACE")) return 0;
  for (int waves : {4, 8}) {
    pace<0>("6 fp16 MFMA (x3 product, 2 k-steps, 2 tiles /2)", waves, 12);
    pace<1>("2 x (2 fp16 + 1 fp6), dependent", waves, 6);
    pace<3>("2 x (2 fp16 + 1 fp6), independent", waves, 6);
    pace<2>("6 fp6 MFMA", waves, 6);
    pace<4>("6 fp16 MFMA, independent", waves, 6);
    pace<6>("2 fp8 MFMA through f8f6f4", waves, 4);
    pace<5>("8 x cvt_scalef32_2xpk16_fp6_f32", waves, 0);
  }
  return 0;
}
