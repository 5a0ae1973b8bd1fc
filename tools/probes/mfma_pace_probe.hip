// mfma_pace_probe.hip — what one SIMD of gfx950 sustains for the instruction mixes of the flash-attention loop (round 3).
//
// Why: every arrangement of csrc/attention_kernel.h (two independent 4-wave workgroups per CU, intra-wave software pipelining, 8-wave
// ping-pong phases) lands on the same ~1600 shader cycles per (32 query rows x 64 keys) wave-tile, of which the 20 MFMAs should need 640.
// This probe times the building blocks alone, per SIMD, with 1 or 2 waves on it:
//   mode 0  20 MFMAs per iteration, operands in registers, accumulators alternating (no LDS, no VALU)
//   mode 1  mode 0 + the 16 fragment reads of a tile (8 ds_read_b128 + 8 ds_read2_b64), issued ahead, counted waits
//   mode 2  the softmax VALU block alone (17 max, 32 v_exp_f32, 16 v_cvt_pk_f16_f32)
//   mode 3  mode 1 followed by mode 2 in the same wave (what a lone wave does per tile)
//   mode 7  mode 3 with a scheduling barrier between the MFMA block and the VALU block (hipcc interleaves them otherwise)
//   mode 4  waves 0-3: mode 1, waves 4-7: mode 2 (the ping-pong pairing: matrix beside VALU on every SIMD; needs 8 waves)
//   mode 5  mode 4 with the VALU waves at s_setprio 2;  mode 6  mode 4 with the roles swapped (the OLDER waves do the VALU block)
// Build: hipcc --offload-arch=gfx950 -O3 -o mfma_pace_probe mfma_pace_probe.hip ; run: ./mfma_pace_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ void mfma(f32x16& acc, f16x8 a, f16x8 b) {
  acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
  __builtin_amdgcn_sched_barrier(0);
}

template <int MODE>
__global__ __launch_bounds__(512, 1) void pace_kernel(int iters, float* sink, long long* cycles, int waves) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < 36 * 1024 / 4; i += blockDim.x) reinterpret_cast<float*>(smem)[i] = 0.001f * (i & 63);
  __syncthreads();
  f32x16 acc[4];
  f16x8 qa, qb;
#pragma unroll
  for (int r = 0; r < 16; ++r) { acc[0][r] = 0.f; acc[1][r] = 0.f; acc[2][r] = 0.f; acc[3][r] = 0.f; }
#pragma unroll
  for (int e = 0; e < 8; ++e) { qa[e] = (_Float16)(0.01f * (lane + e)); qb[e] = (_Float16)(0.02f * (lane - e)); }
  const unsigned lds = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem + (lane & 31) * 144 + (lane >> 5) * 16;
  f32x16 sc[2];
#pragma unroll
  for (int r = 0; r < 16; ++r) { sc[0][r] = 0.01f * r - lane * 0.001f; sc[1][r] = -0.02f * r; }
  const bool matrix = MODE == 0 || MODE == 1 || MODE == 3 || MODE == 7 || ((MODE == 4 || MODE == 5) && wave < 4) || (MODE == 6 && wave >= 4);
  const bool valu = MODE == 2 || MODE == 3 || MODE == 7 || ((MODE == 4 || MODE == 5) && wave >= 4) || (MODE == 6 && wave < 4);
  const bool reads = MODE == 1 || MODE == 3 || MODE == 7 || MODE == 4 || MODE == 5 || MODE == 6;
  if (MODE == 5 && wave >= 4) __builtin_amdgcn_s_setprio(2);
  float mxs = 0.f;
  const long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
    if (matrix) {
      f32x4 fk[8], fv[8];
      if (reads) {
#pragma unroll
        for (int i = 0; i < 8; ++i) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(fk[i]) : "v"(lds), "n"((i & 1) * 4608 + (i >> 1) * 32));
#pragma unroll
        for (int i = 0; i < 8; ++i) asm volatile("ds_read2_b64 %0, %1 offset0:%2 offset1:%3" : "=v"(fv[i]) : "v"(lds + 9216 + (i & 1) * 4352), "n"((i >> 1) * 4), "n"((i >> 1) * 4 + 2));
        asm volatile("s_waitcnt lgkmcnt(8)" : "+v"(fk[0]), "+v"(fk[1]), "+v"(fk[2]), "+v"(fk[3]), "+v"(fk[4]), "+v"(fk[5]), "+v"(fk[6]), "+v"(fk[7]));
      } else {
#pragma unroll
        for (int i = 0; i < 8; ++i) { fk[i] = __builtin_bit_cast(f32x4, qa); fv[i] = __builtin_bit_cast(f32x4, qb); }
      }
      __builtin_amdgcn_sched_barrier(0);
      // 4 row-sum-like + 8 score-like + 8 PV-like MFMAs, neighbours on different accumulators
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        mfma(acc[0], __builtin_bit_cast(f16x8, fk[2 * i]), qa);
        mfma(acc[1], __builtin_bit_cast(f16x8, fk[2 * i + 1]), qa);
        mfma(acc[2], qb, qa);
      }
      if (reads) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(fv[0]), "+v"(fv[1]), "+v"(fv[2]), "+v"(fv[3]), "+v"(fv[4]), "+v"(fv[5]), "+v"(fv[6]), "+v"(fv[7]));
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        mfma(acc[2], __builtin_bit_cast(f16x8, fv[2 * i]), qb);
        mfma(acc[3], __builtin_bit_cast(f16x8, fv[2 * i + 1]), qb);
      }
    }
    if (MODE == 7) __builtin_amdgcn_sched_barrier(0);  // the MFMA block complete before the first VALU instruction
    if (valu) {
      float mx = sc[0][0];
#pragma unroll
      for (int r = 1; r < 16; ++r) mx = fmaxf(mx, sc[0][r]);
#pragma unroll
      for (int r = 0; r < 16; ++r) mx = fmaxf(mx, sc[1][r]);
      mxs += mx;
      f16x8 p[4];
#pragma unroll
      for (int i = 0; i < 32; ++i) p[i >> 3][i & 7] = (_Float16)__builtin_amdgcn_exp2f(sc[i >> 4][i & 15]);
      // feed the result back so that nothing is hoisted out of the loop
#pragma unroll
      for (int i = 0; i < 32; ++i) sc[i >> 4][i & 15] = (float)p[i >> 3][i & 7] * 0.5f - 1.0f - 0.01f * i;
      qb = p[0] + p[1] + p[2] + p[3];
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  const long long t1 = clock64();
  float s = mxs;
#pragma unroll
  for (int r = 0; r < 16; ++r) s += acc[0][r] + acc[1][r] + acc[2][r] + acc[3][r] + sc[0][r];
  sink[blockIdx.x * blockDim.x + tid] = s;
  if (lane == 0) cycles[blockIdx.x * waves + wave] = t1 - t0;
}

template <int MODE>
void run(const char* what, int waves, int blocks, int iters) {
  float* sink; long long* cyc;
  CHECK(hipMalloc(&sink, (size_t)blocks * 512 * 4));
  CHECK(hipMalloc(&cyc, (size_t)blocks * 8 * 8));
  CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(pace_kernel<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, 40 * 1024));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  hipLaunchKernelGGL(pace_kernel<MODE>, dim3(blocks), dim3(64 * waves), 40 * 1024, 0, 10, sink, cyc, waves);
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(e0, 0));
  hipLaunchKernelGGL(pace_kernel<MODE>, dim3(blocks), dim3(64 * waves), 40 * 1024, 0, iters, sink, cyc, waves);
  CHECK(hipEventRecord(e1, 0));
  CHECK(hipDeviceSynchronize());
  float ms = 0;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  std::vector<long long> h((size_t)blocks * waves);
  CHECK(hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost));
  double lo = 0, hi = 0;  // waves 0-3 and 4-7 of block 0
  for (int w = 0; w < waves; ++w) (w < 4 ? lo : hi) += (double)h[w] / iters / (w < 4 ? (waves < 4 ? waves : 4) : waves - 4);
  // clock64() = s_memtime: a constant 100 MHz counter on gfx9; report wall time per iteration too
  printf("%-58s waves/WG %d  blocks %4d: %8.1f ns per iteration (wall, whole launch / iters); s_memtime ticks per iteration waves 0-3 %.2f  waves 4-7 %.2f\n", what, waves,
         blocks, ms * 1e6 / iters, lo, hi);
  CHECK(hipFree(sink)); CHECK(hipFree(cyc));
}

int main() {
  const int iters = 20000;
  for (int blocks : {1, 256}) {
    run<0>("20 MFMAs, register operands", 4, blocks, iters);
    run<0>("20 MFMAs, register operands", 8, blocks, iters);
    run<1>("20 MFMAs + 16 fragment reads", 4, blocks, iters);
    run<1>("20 MFMAs + 16 fragment reads", 8, blocks, iters);
    run<2>("softmax VALU block (17 max, 32 exp, 16 cvt_pk)", 4, blocks, iters);
    run<2>("softmax VALU block (17 max, 32 exp, 16 cvt_pk)", 8, blocks, iters);
    run<3>("MFMAs + reads, then the VALU block (one wave's tile)", 4, blocks, iters);
    run<3>("MFMAs + reads, then the VALU block (one wave's tile)", 8, blocks, iters);
    run<7>("... the same, blocks kept apart (sched_barrier)", 4, blocks, iters);
    run<7>("... the same, blocks kept apart (sched_barrier)", 8, blocks, iters);
    run<4>("waves 0-3 MFMAs + reads, waves 4-7 VALU block", 8, blocks, iters);
    run<5>("... the VALU waves at s_setprio 2", 8, blocks, iters);
    run<6>("waves 0-3 VALU block, waves 4-7 MFMAs + reads", 8, blocks, iters);
  }
  return 0;
}
