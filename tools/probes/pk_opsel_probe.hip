// pk_opsel_probe.hip — minimal reproducer attempt for the fault behind round 2's "co-residency race" (DESIGN.md section 4):
// v_pk_mul_f32 with op_sel:[0,1] (low result = src0.lo * src1.HI) returned src0.lo * 0 in lanes 48-63 now and then when two waves shared
// a SIMD.  Every wave repeats the instruction sequence of the failing epilogue (two v_pk_add_f32 that produce the pair, then the
// v_pk_mul_f32 that reads its high half for the low product) on pseudo-random operands and compares with single-lane v_mul_f32.
//   mode bit 0 / 2 / 3: the partner waves (4-7 of a 512-thread workgroup: the ones that share a SIMD with the test waves 0-3) run MFMAs /
//                       LDS reads / global loads instead of the test sequence
//   mode bit 1: the test waves also stream global loads between the iterations (as the epilogue does)
// build: hipcc --offload-arch=gfx950 -O2 -o pk_opsel_probe pk_opsel_probe.hip ; run: ./pk_opsel_probe [waves_per_simd] [mode] [iters]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ uint32_t lcg(uint32_t& s) { s = s * 1664525u + 1013904223u; return s; }
__device__ __forceinline__ float rnd(uint32_t& s) { return (float)(int)(lcg(s) >> 8) * (1.0f / 8388608.0f) - 1.0f + 1.0e-3f; }  // never 0

__global__ __launch_bounds__(512) void probe(int iters, int mode, const float* table, uint32_t table_mask, unsigned long long* counts, float* sink) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  uint32_t s = (blockIdx.x * 512u + threadIdx.x) * 2654435761u + 12345u;
  // waves w and w + 4 of a 512-thread workgroup share a SIMD: waves 4-7 are the partners
  if ((mode & (1 | 4 | 8)) && wave >= 4) {
    __shared__ uint4 lds[1024];
    f32x16 acc = {};
    f16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)rnd(s); b[i] = (_Float16)rnd(s); }
    lds[threadIdx.x] = make_uint4(s, s + 1, s + 2, s + 3);
    lds[threadIdx.x + 512] = make_uint4(s, s + 5, s + 6, s + 7);
    uint32_t x = 0;
    for (int it = 0; it < iters * 2; ++it) {
      if (mode & 1) {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(b, a, acc, 0, 0, 0);
      }
      if (mode & 4) {
        const uint4 v = lds[(threadIdx.x + it * 64) & 1023];
        const uint4 w = lds[(threadIdx.x + it * 64 + 512) & 1023];
        x += v.x ^ v.w ^ w.y;
      }
      if (mode & 8) {
        const float4 v = *reinterpret_cast<const float4*>(table + ((lcg(s) & table_mask) & ~3u));
        x += __float_as_uint(v.x) ^ __float_as_uint(v.w);
      }
    }
    float t = (float)x;
    for (int i = 0; i < 16; ++i) t += acc[i];
    if (t == 123.456f) sink[0] = t;
    return;
  }
  unsigned long long bad_lo = 0, bad_hi = 0, zero_lo = 0;
  float keep = 0.f;
  for (int it = 0; it < iters; ++it) {
    f32x2 bias = {rnd(s), rnd(s)}, acc = {rnd(s), rnd(s)}, bias2 = {rnd(s), rnd(s)}, acc2 = {rnd(s), rnd(s)};
    f32x2 sc = {rnd(s), rnd(s)}, sc2 = {rnd(s), rnd(s)};
    if (mode & 2) {
      const float4 v = *reinterpret_cast<const float4*>(table + ((lcg(s) & table_mask) & ~3u));
      sc[0] += v.x * 1e-9f; sc2[0] += v.y * 1e-9f; keep += v.z + v.w;
    }
    f32x2 x, x2, t, t2;
    float r_lo, r_hi, r2_lo;
    asm volatile(
        "v_pk_add_f32 %0, %7, %8\n\t"
        "v_pk_add_f32 %1, %9, %10\n\t"
        "v_pk_mul_f32 %2, %11, %0 op_sel:[0,1]\n\t"
        "v_pk_mul_f32 %3, %12, %1 op_sel:[0,1]\n\t"
        "s_nop 4"
        : "=&v"(x), "=&v"(x2), "=&v"(t), "=&v"(t2), "=&v"(r_lo), "=&v"(r_hi), "=&v"(r2_lo)
        : "v"(bias), "v"(acc), "v"(bias2), "v"(acc2), "v"(sc), "v"(sc2));
    // reference products from the pairs as the program sees them afterwards
    {
      const float s0 = sc[0], s1 = sc[1], s20 = sc2[0], xh = x[1], x2h = x2[1];
      asm volatile("v_mul_f32 %0, %3, %5\n\tv_mul_f32 %1, %4, %5\n\tv_mul_f32 %2, %6, %7" : "=&v"(r_lo), "=&v"(r_hi), "=&v"(r2_lo) : "v"(s0), "v"(s1), "v"(xh), "v"(s20), "v"(x2h));
    }
    if (__float_as_uint(t[0]) != __float_as_uint(r_lo)) { ++bad_lo; if (t[0] == 0.f) ++zero_lo; }
    if (__float_as_uint(t2[0]) != __float_as_uint(r2_lo)) { ++bad_lo; if (t2[0] == 0.f) ++zero_lo; }
    if (__float_as_uint(t[1]) != __float_as_uint(r_hi)) ++bad_hi;
  }
  if (keep == 1.2345f) sink[1] = keep;
  if (bad_lo) atomicAdd(&counts[lane >> 4], bad_lo);
  if (bad_hi) atomicAdd(&counts[4 + (lane >> 4)], bad_hi);
  if (zero_lo) atomicAdd(&counts[8 + (lane >> 4)], zero_lo);
}

int main(int argc, char** argv) {
  const int wps = argc > 1 ? atoi(argv[1]) : 2, mode = argc > 2 ? atoi(argv[2]) : 0, iters = argc > 3 ? atoi(argv[3]) : 20000;
  hipDeviceProp_t p;
  hipGetDeviceProperties(&p, 0);
  const int cus = p.multiProcessorCount;
  unsigned long long* counts;
  float *sink, *table;
  const uint32_t tn = 1u << 22;
  hipMalloc(&counts, 12 * 8); hipMalloc(&sink, 16); hipMalloc(&table, tn * 4);
  hipMemset(table, 0, tn * 4);
  for (int rep = 0; rep < 3; ++rep) {
    hipMemset(counts, 0, 12 * 8);
    // 512-thread workgroups = 2 waves per SIMD inside one workgroup; wps 1 -> 256 threads (launch_bounds still 512)
    hipLaunchKernelGGL(probe, dim3(cus * 2), dim3(wps >= 2 ? 512 : 256), 0, 0, iters, mode, table, tn - 1, counts, sink);
    hipDeviceSynchronize();
    unsigned long long h[12];
    hipMemcpy(h, counts, sizeof(h), hipMemcpyDeviceToHost);
    printf("pk_opsel_probe waves/SIMD %d mode %d iters %d: low-product mismatches by lane group [%llu %llu %llu %llu] (of them exactly 0: [%llu %llu %llu %llu]), high-product mismatches [%llu %llu %llu %llu]\n",
           wps, mode, iters, h[0], h[1], h[2], h[3], h[8], h[9], h[10], h[11], h[4], h[5], h[6], h[7]);
  }
  return 0;
}
