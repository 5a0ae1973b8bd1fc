// l2stride.hip — does the row stride of a K-contiguous operand (2 KB for K=1024 f16) hot-spot L2 channels when every workgroup
// walks k in lock step?  Mimics the GEMM's tile loads: 256 rows x SEG bytes per k-tile per workgroup, k advancing by SEG.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int SEG>
__global__ __launch_bounds__(256) void tile_kernel(const char* base, unsigned bytes, unsigned stride, int rows_total, int kbytes, int stagger,
                                                  unsigned* sink) {
  __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, (int)bytes, 0x00020000);
  const int tid = threadIdx.x;
  constexpr int LPR = SEG / 16, RPP = 256 / LPR, NL = 256 / RPP;  // lanes per row, rows per pass, loads per thread for 256 rows
  const int row0 = (blockIdx.x * 256) % rows_total;
  const int nkt = kbytes / SEG;
  const int shift = stagger ? (blockIdx.x * 7) % nkt : 0;
  unsigned acc = 0;
  for (int rep = 0; rep < 4; ++rep)
    for (int kt = 0; kt < nkt; ++kt) {
      const int k = ((kt + shift) % nkt) * SEG;
      u32x4 v[NL];
#pragma unroll
      for (int j = 0; j < NL; ++j) {
        const unsigned row = row0 + j * RPP + tid / LPR;
        v[j] = __builtin_amdgcn_raw_buffer_load_b128(r, (int)(row * stride + k + (tid % LPR) * 16), 0, 0);
      }
#pragma unroll
      for (int j = 0; j < NL; ++j) acc ^= v[j][0] ^ v[j][2];
    }
  if (acc == 0x12345678u) sink[0] = acc;
}

int main() {
  const int rows = 22528, kbytes = 2048;
  char* buf; unsigned* sink;
  const size_t cap = (size_t)rows * 4096;
  (void)hipMalloc(&buf, cap); (void)hipMalloc(&sink, 64); (void)hipMemset(buf, 1, cap);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int rt : {2816, 22528})
    for (unsigned stride : {2048u, 2048u + 64, 2048u + 128, 2048u + 256, 2048u + 512})
      for (int stagger : {0, 1}) {
        auto run = [&](int seg) {
          float ms = 0;
          for (int rep = 0; rep < 2; ++rep) {
            (void)hipEventRecord(e0);
            if (seg == 128) hipLaunchKernelGGL(tile_kernel<128>, dim3(2048), dim3(256), 0, 0, buf, (unsigned)(rt * stride), stride, rt, kbytes, stagger, sink);
            else hipLaunchKernelGGL(tile_kernel<64>, dim3(2048), dim3(256), 0, 0, buf, (unsigned)(rt * stride), stride, rt, kbytes, stagger, sink);
            (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
            (void)hipEventElapsedTime(&ms, e0, e1);
          }
          return 2048.0 * 256 * kbytes * 4 / ms / 1e9;
        };
        printf("rows %5d stride %4u stagger %d : seg128 %6.2f TB/s   seg64 %6.2f TB/s\n", rt, stride, stagger, run(128), run(64));
      }
  return 0;
}
