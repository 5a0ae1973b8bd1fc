// wave_simd_probe — which SIMD of its CU does wave w of a workgroup run on?  (HW_REG_HW_ID: wave [3:0], SIMD [5:4], CU [11:8], SE [15:13])
// The one-round flash attention launch holds 6 (or 8) waves per CU; which of them share a SIMD decides the pace of every tile
// (attention_kernel.h, "NW = 8").   hipcc --offload-arch=gfx950 -O2 -o wave_simd_probe wave_simd_probe.hip && ./wave_simd_probe
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>

__global__ void probe(uint32_t* out, int spin) {
  uint32_t id;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(id));
  float x = threadIdx.x;
  for (int i = 0; i < spin; ++i) x = x * 1.0001f + 0.5f;  // keep every wave resident while the others start
  if ((threadIdx.x & 63) == 0) out[blockIdx.x * 16 + (threadIdx.x >> 6)] = id | (x == 7.f ? 1u << 31 : 0u);
}

int main() {
  uint32_t* d;
  const int nwg = 256;
  hipMalloc(&d, nwg * 16 * 4);
  for (int threads : {256, 384, 512}) {
    hipMemset(d, 0xff, nwg * 16 * 4);
    hipLaunchKernelGGL(probe, dim3(nwg), dim3(threads), 55296, 0, d, 20000);
    hipDeviceSynchronize();
    std::vector<uint32_t> h(nwg * 16);
    hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost);
    const int nw = threads / 64;
    int hist[8][4] = {};
    for (int b = 0; b < nwg; ++b)
      for (int w = 0; w < nw; ++w) hist[w][(h[b * 16 + w] >> 4) & 3]++;
    printf("%d threads per workgroup, %d workgroups (54 KB of LDS each): workgroups whose wave w ran on SIMD 0 1 2 3\n", threads, nwg);
    for (int w = 0; w < nw; ++w) printf("  wave %d: %4d %4d %4d %4d\n", w, hist[w][0], hist[w][1], hist[w][2], hist[w][3]);
    printf("  first workgroups, SIMD of wave 0..%d:", nw - 1);
    for (int b = 0; b < 6; ++b) {
      printf("  [");
      for (int w = 0; w < nw; ++w) printf("%u", (h[b * 16 + w] >> 4) & 3);
      printf(" cu%u se%u]", (h[b * 16] >> 8) & 15, (h[b * 16] >> 13) & 7);
    }
    printf("\n");
    fflush(stdout);
  }
  return 0;
}
