// cvt_pk_f16_probe — does gfx950's v_cvt_pk_f16_f32 (what hipcc emits for a two-element fp32 -> fp16 vector conversion) round like
// v_cvt_f16_f32?  Round 6: packed-hi / remainder pairs built from a vector conversion disagreed with the scalar form on the GPU by one
// fp16 ulp in ~3 % of the values while the host shim (round to nearest even for both) agreed everywhere.
//   hipcc --offload-arch=gfx950 -O2 -o cvt_pk_f16_probe cvt_pk_f16_probe.hip && ./cvt_pk_f16_probe
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>
typedef _Float16 f16;
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
// fp16(x * r) two ways: the fp32 product converted (v_mul_f32 + v_cvt_f16_f32) and v_fma_mixlo_f16 x, r, 0 (what hipcc fuses a multiply and its
// conversion into under -ffp-contract=fast)
__global__ void kmix(const float* a, uint16_t* conv, uint16_t* mix, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (2 * i + 1 >= n) return;
  const float x = a[2 * i], r = a[2 * i + 1];
  float prod;
  uint32_t c, m;
  asm volatile("v_mul_f32 %0, %1, %2" : "=v"(prod) : "v"(x), "v"(r));
  asm volatile("v_cvt_f16_f32 %0, %1" : "=v"(c) : "v"(prod));
  asm volatile("v_fma_mixlo_f16 %0, %1, %2, 0" : "=v"(m) : "v"(x), "v"(r));
  conv[i] = (uint16_t)c; mix[i] = (uint16_t)m;
}
__global__ void k(const float* a, uint16_t* scalar, uint16_t* packed, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (2 * i + 1 >= n) return;
  const float x = a[2 * i], y = a[2 * i + 1];
  uint32_t s0, s1, p;
  asm volatile("v_cvt_f16_f32 %0, %1" : "=v"(s0) : "v"(x));
  asm volatile("v_cvt_f16_f32 %0, %1" : "=v"(s1) : "v"(y));
  asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(p) : "v"(x), "v"(y));
  scalar[2 * i] = (uint16_t)s0; scalar[2 * i + 1] = (uint16_t)s1;
  packed[2 * i] = (uint16_t)(p & 0xffff); packed[2 * i + 1] = (uint16_t)(p >> 16);
}
static float h2f(uint16_t h) { f16 v; memcpy(&v, &h, 2); return (float)v; }
int main() {
  const int n = 1 << 22;
  std::vector<float> a(n);
  uint32_t s = 12345;
  for (int i = 0; i < n; ++i) {  // magnitudes 2^-30 .. 2^18, both signs, random mantissas; a block of exact ties
    s = s * 1664525u + 1013904223u;
    const int e = (int)((s >> 8) % 49) - 30;
    s = s * 1664525u + 1013904223u;
    float m = 1.0f + (float)(s >> 9) / 8388608.0f;
    if (i % 64 == 0) m = 1.0f + (float)((s >> 22) * 2 + 1) / 2048.0f;  // exactly between two halves
    a[i] = ldexpf(m, e) * ((s & 256) ? -1.f : 1.f);
  }
  float* da; uint16_t *ds, *dp;
  hipMalloc(&da, n * 4); hipMalloc(&ds, n * 2); hipMalloc(&dp, n * 2);
  hipMemcpy(da, a.data(), n * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(n / 2 / 256), dim3(256), 0, 0, da, ds, dp, n);
  std::vector<uint16_t> hs(n), hp(n);
  hipMemcpy(hs.data(), ds, n * 2, hipMemcpyDeviceToHost); hipMemcpy(hp.data(), dp, n * 2, hipMemcpyDeviceToHost);
  long diff = 0, toward_zero = 0, away = 0, tie = 0, denorm = 0, shown = 0;
  for (int i = 0; i < n; ++i) {
    if (hs[i] == hp[i]) continue;
    ++diff;
    const float fs = h2f(hs[i]), fp = h2f(hp[i]);
    if (fabsf(fp) < fabsf(fs)) ++toward_zero; else ++away;
    if (fabsf(a[i]) < 6.2e-5f) ++denorm;
    if (i % 64 == 0) ++tie;
    if (shown++ < 12) printf("  x = %.9g (%a): v_cvt_f16_f32 -> %.9g, v_cvt_pk_f16_f32 -> %.9g\n", a[i], a[i], fs, fp);
  }
  printf("%d values: %ld differ (%.3f %%); packed result nearer zero %ld, farther %ld; of them below the fp16 normal range %ld, exact ties %ld\n", n, diff,
         100.0 * diff / n, toward_zero, away, denorm, tie);
  // second question: v_fma_mixlo_f16
  for (int i = 0; i < n; i += 2) {  // x in +-[2^-6, 8), r in (0, 1]: a GELU gate
    s = s * 1664525u + 1013904223u;
    a[i] = ldexpf(1.0f + (float)(s >> 9) / 8388608.0f, (int)((s >> 3) % 9) - 6) * ((s & 4) ? -1.f : 1.f);
    s = s * 1664525u + 1013904223u;
    a[i + 1] = (float)((s >> 8) + 1) / 16777216.0f;
  }
  hipMemcpy(da, a.data(), n * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(kmix, dim3(n / 2 / 256), dim3(256), 0, 0, da, ds, dp, n);
  hipMemcpy(hs.data(), ds, n, hipMemcpyDeviceToHost); hipMemcpy(hp.data(), dp, n, hipMemcpyDeviceToHost);
  diff = toward_zero = away = shown = 0;
  for (int i = 0; i < n / 2; ++i) {
    if (hs[i] == hp[i]) continue;
    ++diff;
    const float fs = h2f(hs[i]), fp = h2f(hp[i]);
    if (fabsf(fp) < fabsf(fs)) ++toward_zero; else ++away;
    if (shown++ < 8) printf("  x = %a, r = %a, x r = %.9g: v_cvt_f16_f32(v_mul_f32) -> %.9g, v_fma_mixlo_f16 -> %.9g\n", a[2 * i], a[2 * i + 1], (double)a[2 * i] * a[2 * i + 1], fs, fp);
  }
  printf("v_fma_mixlo_f16 x, r, 0 against v_cvt_f16_f32(v_mul_f32 x, r): %d products, %ld differ (%.3f %%); mix result nearer zero %ld, farther %ld\n", n / 2, diff,
         200.0 * diff / n, toward_zero, away);
  return 0;
}
