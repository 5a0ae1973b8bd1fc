// What a dependent chain of kernel launches costs on this GPU when the kernels do (almost) nothing: the floor under the 3168 launches of
// one B = 1 sample (16 steps x 22 blocks x 9 kernels), i.e. the most a persistent per-block / per-step kernel could remove.
//   empty   : 256 workgroups x 256 threads, no memory access
//   touch   : every workgroup reads a line the previous kernel wrote and writes one for the next (a real dependence through memory)
//   args    : as touch, with a 200-byte argument block (the size of GemmCore + an epilogue) read by every wave
// each as eager launches on one stream and as a captured graph replayed.  Build: hipcc --offload-arch=gfx950 -O2 -o launch_floor launch_floor.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

struct Args { float a[50]; };

__global__ void empty_kernel() {}
__global__ void touch_kernel(const float* in, float* out) {
  if (threadIdx.x == 0) out[blockIdx.x * 32] = in[((blockIdx.x + 1) & 255) * 32] + 1.0f;
}
__global__ void args_kernel(const float* in, float* out, Args a) {
  if (threadIdx.x == 0) out[blockIdx.x * 32] = in[((blockIdx.x + 1) & 255) * 32] + a.a[blockIdx.x % 50];
}

template <typename F>
static double run(const char* name, int n, hipStream_t s, F launch) {
  for (int i = 0; i < 64; ++i) launch(i);
  CK(hipStreamSynchronize(s));
  auto t0 = std::chrono::steady_clock::now();
  for (int i = 0; i < n; ++i) launch(i);
  CK(hipStreamSynchronize(s));
  const double eager = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / n;
  hipGraph_t g;
  hipGraphExec_t ge;
  CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
  for (int i = 0; i < n; ++i) launch(i);
  CK(hipStreamEndCapture(s, &g));
  CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
  CK(hipGraphLaunch(ge, s));
  CK(hipStreamSynchronize(s));
  double best = 1e30;
  for (int rep = 0; rep < 5; ++rep) {
    t0 = std::chrono::steady_clock::now();
    CK(hipGraphLaunch(ge, s));
    CK(hipStreamSynchronize(s));
    const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / n;
    if (us < best) best = us;
  }
  printf("%-6s chain of %d launches: eager %.2f us per launch, graph replay %.2f us per launch (= %.1f ms per chain)\n", name, n, eager, best, best * n / 1e3);
  CK(hipGraphExecDestroy(ge));
  CK(hipGraphDestroy(g));
  return best;
}

int main(int argc, char** argv) {
  const int n = argc > 1 ? atoi(argv[1]) : 3168;
  hipStream_t s;
  CK(hipStreamCreate(&s));
  float *a, *b;
  CK(hipMalloc(&a, 256 * 32 * 4));
  CK(hipMalloc(&b, 256 * 32 * 4));
  CK(hipMemset(a, 0, 256 * 32 * 4));
  CK(hipMemset(b, 0, 256 * 32 * 4));
  Args args{};
  run("empty", n, s, [&](int) { hipLaunchKernelGGL(empty_kernel, dim3(256), dim3(256), 0, s); });
  run("touch", n, s, [&](int i) { hipLaunchKernelGGL(touch_kernel, dim3(256), dim3(256), 0, s, (i & 1) ? b : a, (i & 1) ? a : b); });
  run("args", n, s, [&](int i) { hipLaunchKernelGGL(args_kernel, dim3(256), dim3(256), 0, s, (i & 1) ? b : a, (i & 1) ? a : b, args); });
  // the same with the dynamic LDS and the block size of the pipelined GEMM (one workgroup per CU): allocation / wave launch cost
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(touch_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 144 * 1024));
  run("lds144", n, s, [&](int i) { hipLaunchKernelGGL(touch_kernel, dim3(240), dim3(512), 144 * 1024, s, (i & 1) ? b : a, (i & 1) ? a : b); });
  return 0;
}
