// fillrate.hip — how fast can ONE CU move L2-resident lines into LDS?  (a) LDS-DMA (buffer_load ... lds, 16 B/lane), (b) buffer_load -> VGPR ->
// ds_write_b128, (c) a mix of both.  The GEMM k-loop at B=1 runs at ~56 KB/us per CU whatever the tile / occupancy;
// this probe separates the paths.  Build: hipcc --offload-arch=gfx950 -O3 fillrate.hip -o fillrate.  Not part of the product.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// Per iteration every wave moves PIECES x 1 KB (64 lanes x 16 B).  DMA_P of the pieces go global -> LDS directly, the rest through VGPRs.
// Two iterations are kept in flight (software pipelined by hand for the VGPR path, counted vmcnt for the DMA path).
template <int PIECES, int DMA_P>
__global__ __launch_bounds__(256) void fill_kernel(const char* base, unsigned region, int iters, unsigned* sink) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int REG_P = PIECES - DMA_P;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, (int)region, 0x00020000);
  char* my = smem + wave * (3 * PIECES * 1024);                       // 3-stage ring per wave
  const unsigned lane_off = (unsigned)(lane >> 3) * 2048u + (unsigned)(lane & 7) * 16u;  // 8 rows of 128 B per piece, rows 2 KB apart
  unsigned pos = ((blockIdx.x * 4u + wave) * 65536u) % (region - 262144u);
  unsigned acc = 0;
  u32x4 regs[2][REG_P > 0 ? REG_P : 1];
  auto issue = [&](int it, int stage, int set) {
    const unsigned p0 = (pos + (unsigned)it * 128u) % (region - 262144u);
#pragma unroll
    for (int j = 0; j < DMA_P; ++j)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)(my + (stage * PIECES + j) * 1024), 16,
                                               (int)((p0 + lane_off + j * 16384u) & ~15u), 0, 0, 0);
#pragma unroll
    for (int j = 0; j < REG_P; ++j)
      regs[set][j] = __builtin_amdgcn_raw_buffer_load_b128(r, (int)((p0 + lane_off + (DMA_P + j) * 16384u) & ~15u), 0, 0);
  };
  issue(0, 0, 0);
  issue(1, 1, 1);
  for (int it = 0; it < iters; it += 2) {
    // stage of iteration `it` is complete when at most one iteration's loads are outstanding
    if constexpr (REG_P > 0) {
#pragma unroll
      for (int j = 0; j < REG_P; ++j) *reinterpret_cast<u32x4*>(my + ((it % 3) * PIECES + DMA_P + j) * 1024 + lane * 16) = regs[0][j];
    } else {
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PIECES) : "memory");
    }
    acc ^= *reinterpret_cast<unsigned*>(my + (it % 3) * PIECES * 1024 + lane * 4);
    issue(it + 2, (it + 2) % 3, 0);
    if constexpr (REG_P > 0) {
#pragma unroll
      for (int j = 0; j < REG_P; ++j) *reinterpret_cast<u32x4*>(my + (((it + 1) % 3) * PIECES + DMA_P + j) * 1024 + lane * 16) = regs[1][j];
    } else {
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PIECES) : "memory");
    }
    acc ^= *reinterpret_cast<unsigned*>(my + ((it + 1) % 3) * PIECES * 1024 + lane * 4);
    issue(it + 3, (it + 3) % 3, 1);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (acc == 0x12345678u) sink[0] = acc;
}

template <int PIECES, int DMA_P>
double run(const char* buf, unsigned region, int wgs, int iters, unsigned* sink) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int lds = 4 * 3 * PIECES * 1024;
  hipFuncSetAttribute(reinterpret_cast<const void*>(fill_kernel<PIECES, DMA_P>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  float ms = 0;
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((fill_kernel<PIECES, DMA_P>), dim3(wgs), dim3(256), lds, 0, buf, region, iters, sink);
    hipEventRecord(e1); hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
  }
  return (double)wgs * 4 * PIECES * 1024.0 * iters / ms / 1e9;  // TB/s
}

int main() {
  char* buf; unsigned* sink;
  const unsigned big = 64u << 20;
  hipMalloc(&buf, big); hipMalloc(&sink, 64); hipMemset(buf, 1, big);
  for (unsigned region : {2u << 20, 24u << 20}) {
    for (int wgs : {256, 512}) {
      const int iters = 2000;
      const double a = run<6, 6>(buf, region, wgs, iters, sink), b = run<6, 0>(buf, region, wgs, iters, sink), c = run<6, 2>(buf, region, wgs, iters, sink),
                   d = run<6, 3>(buf, region, wgs, iters, sink), e = run<6, 4>(buf, region, wgs, iters, sink);
      auto per = [](double tbs) { return tbs * 1e3 / 256.0; };  // KB/us per CU
      printf("region %2u MB wgs %3d | all-DMA %5.2f TB/s (%5.1f KB/us/CU) | all-VGPR %5.2f (%5.1f) | DMA 2/6 %5.2f (%5.1f) | DMA 3/6 %5.2f (%5.1f) | DMA 4/6 %5.2f (%5.1f)\n",
             region >> 20, wgs, a, per(a), b, per(b), c, per(c), d, per(d), e, per(e));
    }
  }
  return 0;
}
