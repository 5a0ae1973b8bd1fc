// l2bw.hip — how fast can a CU pull L2-resident data?  (a) buffer_load_dwordx4 -> VGPR, (b) global_load_lds_dwordx4 -> LDS.
// Build: hipcc --offload-arch=gfx950 -O3 l2bw.hip -o l2bw ; run on the GPU box.  Not part of the product.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int ROWB>  // bytes per "row" segment read contiguously by ROWB/16 lanes (64 or 128), rows 2 KB apart like a K=1024 f16 matrix
__global__ __launch_bounds__(256) void reg_kernel(const char* base, size_t region, int iters, unsigned* sink) {
  __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, (int)region, 0x00020000);
  const int tid = threadIdx.x;
  constexpr int LPR = ROWB / 16;
  const unsigned rowoff = (tid / LPR) * 2048u + (tid % LPR) * 16u;  // 256 threads cover 256/LPR rows
  unsigned acc = 0;
  unsigned start = (blockIdx.x * 131072u) % (unsigned)region;
  for (int it = 0; it < iters; ++it) {
    u32x4 v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      unsigned off = (start + rowoff + j * ROWB + (unsigned)it * 1024u) % (unsigned)(region - 4096 * 256);
      off &= ~15u;
      v[j] = __builtin_amdgcn_raw_buffer_load_b128(r, (int)off, 0, 0);
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) acc ^= v[j][0] ^ v[j][3];
  }
  if (acc == 0x12345678u) sink[0] = acc;
}

__global__ __launch_bounds__(256) void lds_kernel(const char* base, size_t region, int iters, unsigned* sink) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  unsigned start = (blockIdx.x * 131072u) % (unsigned)region;
  unsigned acc = 0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      unsigned off = (start + (unsigned)(tid / 8) * 2048u + (tid % 8) * 16u + j * 128 + (unsigned)it * 1024u) % (unsigned)(region - 4096 * 256);
      off &= ~15u;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(base + off),
                                       (__attribute__((address_space(3))) void*)(smem + (j * 4 + wave) * 1024), 16, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    acc ^= *(unsigned*)(smem + lane * 16 + (it & 7) * 1024);
  }
  if (acc == 0x12345678u) sink[0] = acc;
}

int main() {
  const size_t region = 24u << 20;  // 24 MB: fits the 32 MB aggregate L2 poorly, MALL fully; also test 3 MB (per-XCD L2 resident)
  char* buf; unsigned* sink;
  hipMalloc(&buf, region); hipMalloc(&sink, 64); hipMemset(buf, 1, region);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (size_t reg : {(size_t)3 << 20, (size_t)24 << 20}) {
    for (int wgs : {256, 512, 1024, 2048}) {
      const int iters = 2000;
      auto run = [&](int which) {
        for (int rep = 0; rep < 2; ++rep) {
          hipEventRecord(e0);
          if (which == 0) hipLaunchKernelGGL(reg_kernel<128>, dim3(wgs), dim3(256), 0, 0, buf, reg, iters, sink);
          if (which == 1) hipLaunchKernelGGL(reg_kernel<64>, dim3(wgs), dim3(256), 0, 0, buf, reg, iters, sink);
          if (which == 2) hipLaunchKernelGGL(lds_kernel, dim3(wgs), dim3(256), 32768, 0, buf, reg, iters, sink);
          hipEventRecord(e1); hipEventSynchronize(e1);
        }
        float ms; hipEventElapsedTime(&ms, e0, e1);
        double bytes = (double)wgs * 256 * 8 * 16 * iters;
        return bytes / ms / 1e9;
      };
      printf("region %2zu MB  wgs %4d : reg128 %6.2f TB/s  reg64 %6.2f TB/s  glds %6.2f TB/s\n", reg >> 20, wgs, run(0), run(1), run(2));
    }
  }
  return 0;
}
