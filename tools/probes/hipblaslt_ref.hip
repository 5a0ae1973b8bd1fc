// hipblaslt_ref.hip — calibration probe (NOT part of the product): what the vendor library reaches on a PLAIN fp16 GEMM
// out[M,N] = A[M,K] . W[N,K]^T (fp32 accumulate, fp16 out) at the DiT block shapes, next to which libf5hip's fused kernels can be read.
// The product cannot use it (the operands are hi/lo-split and the epilogues fused), but it separates "what the part can do at this
// shape" from "what our k-loop loses".   Build + run on the GPU box:
//   hipcc --offload-arch=gfx950 -O2 tools/probes/hipblaslt_ref.hip -lhipblaslt -o tools/probes/hipblaslt_ref && tools/probes/hipblaslt_ref
#include <hip/hip_fp16.h>
#include <hip/hip_runtime.h>
#include <hipblaslt/hipblaslt.h>

#include <cstdio>
#include <vector>

#define CK(x)                                                                   \
  do {                                                                          \
    auto _s = (x);                                                              \
    if ((int)_s != 0) { printf("%s failed: %d (line %d)\n", #x, (int)_s, __LINE__); return 1; } \
  } while (0)

__global__ void fill_kernel(__half* x, size_t n, unsigned seed, float scale) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    unsigned h = (unsigned)i * 2654435761u ^ seed;
    h ^= h >> 16; h *= 0x7feb352du; h ^= h >> 15; h *= 0x846ca68bu; h ^= h >> 16;
    x[i] = __float2half(scale * ((float)(h & 0xffffff) * (1.0f / 8388608.0f) - 1.0f));
  }
}

static int run(hipblasLtHandle_t h, int M, int N, int K, const char* tag) {
  // row-major out[M,N] = A[M,K] W[N,K]^T  ==  column-major out^T[N,M] = W^T-as-stored(op T)[N,K] . A-as-stored[K,M]
  __half *A, *W, *C;
  CK(hipMalloc(&A, (size_t)M * K * 2));
  CK(hipMalloc(&W, (size_t)N * K * 2));
  CK(hipMalloc(&C, (size_t)M * N * 2));
  fill_kernel<<<1024, 256>>>(A, (size_t)M * K, 1u, 1.0f);  // full-range random operands: zero fill runs at a higher clock (DVFS) and flatters the library
  fill_kernel<<<1024, 256>>>(W, (size_t)N * K, 2u, 0.05f);
  CK(hipDeviceSynchronize());
  hipblasLtMatmulDesc_t desc;
  hipblasLtMatrixLayout_t la, lb, lc;
  CK(hipblasLtMatmulDescCreate(&desc, HIPBLAS_COMPUTE_32F, HIP_R_32F));
  hipblasOperation_t opT = HIPBLAS_OP_T, opN = HIPBLAS_OP_N;
  CK(hipblasLtMatmulDescSetAttribute(desc, HIPBLASLT_MATMUL_DESC_TRANSA, &opT, sizeof(opT)));
  CK(hipblasLtMatmulDescSetAttribute(desc, HIPBLASLT_MATMUL_DESC_TRANSB, &opN, sizeof(opN)));
  CK(hipblasLtMatrixLayoutCreate(&la, HIP_R_16F, K, N, K));  // W stored [N,K] row-major = [K,N] column-major, transposed in the op
  CK(hipblasLtMatrixLayoutCreate(&lb, HIP_R_16F, K, M, K));  // A stored [M,K] row-major = [K,M] column-major
  CK(hipblasLtMatrixLayoutCreate(&lc, HIP_R_16F, N, M, N));  // out^T
  hipblasLtMatmulPreference_t pref;
  CK(hipblasLtMatmulPreferenceCreate(&pref));
  size_t wsz = 64 << 20;
  void* ws;
  CK(hipMalloc(&ws, wsz));
  CK(hipblasLtMatmulPreferenceSetAttribute(pref, HIPBLASLT_MATMUL_PREF_MAX_WORKSPACE_BYTES, &wsz, sizeof(wsz)));
  hipblasLtMatmulHeuristicResult_t heur[8];
  int found = 0;
  CK(hipblasLtMatmulAlgoGetHeuristic(h, desc, la, lb, lc, lc, pref, 8, heur, &found));
  if (!found) { printf("%s: no algorithm\n", tag); return 0; }
  const float alpha = 1.f, beta = 0.f;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  double best = 1e30;
  for (int a = 0; a < found; ++a) {
    for (int w = 0; w < 3; ++w) CK(hipblasLtMatmul(h, desc, &alpha, W, la, A, lb, &beta, C, lc, C, lc, &heur[a].algo, ws, wsz, nullptr));
    const int iters = 20;
    CK(hipEventRecord(e0, nullptr));
    for (int i = 0; i < iters; ++i) CK(hipblasLtMatmul(h, desc, &alpha, W, la, A, lb, &beta, C, lc, C, lc, &heur[a].algo, ws, wsz, nullptr));
    CK(hipEventRecord(e1, nullptr));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    best = ms / iters < best ? ms / iters : best;
  }
  printf("hipblaslt fp16 %-10s M=%6d N=%5d K=%5d: %8.1f us  %7.1f TFLOP/s (best of %d heuristics)\n", tag, M, N, K, best * 1e3, 2.0 * M * N * K / best / 1e9, found);
  (void)hipFree(A); (void)hipFree(W); (void)hipFree(C); (void)hipFree(ws);
  return 0;
}

int main() {
  hipblasLtHandle_t h;
  CK(hipblasLtCreate(&h));
  const int shapes[][3] = {{2812, 3072, 1024}, {2812, 1024, 1024}, {2812, 2048, 1024}, {2812, 1024, 2048}, {1406, 1024, 1024}, {22496, 3072, 1024},
                           {22496, 1024, 2048}, {89984, 2048, 1024}, {89984, 3072, 1024}};
  const char* tags[] = {"QKV B=1", "out B=1", "FF1 B=1", "FF2 B=1", "out chain", "QKV B=8", "FF2 B=8", "FF1 B=32", "QKV B=32"};
  for (int i = 0; i < 9; ++i)
    if (run(h, shapes[i][0], shapes[i][1], shapes[i][2], tags[i])) return 1;
  return 0;
}
