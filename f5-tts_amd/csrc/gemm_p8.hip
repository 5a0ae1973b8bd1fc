// gemm_p8.hip — instantiations of the ping-pong block GEMM (gemm_p8.h): a translation unit of its own (every instantiation is about a
// minute of hipcc; gemm.hip already takes four)
#include <cstdio>
#include <cstdlib>

#include "kernels.h"
#include "gemm_p8.h"

namespace {
template <int NSPLIT, typename Epi, int ABL>
hipError_t launch_abl(const GemmCore& g, const Epi& e, hipStream_t s) {
  auto kern = gemm_p8_kernel<NSPLIT, Epi, ABL>;
  if constexpr (ABL != 0) {  // microbenchmark ablations only: the production instantiations get their limit in init_p8_kernels()
    const hipError_t err = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, P8_LDS_BYTES);
    if (err != hipSuccess) return err;
  }
  const dim3 grid(((g.M + 255) / 256) * ((g.N + 255) / 256), 1, 1);
  static const bool trace = getenv("F5HIP_GEMM_TRACE") != nullptr;
  if (trace) fprintf(stderr, "gemm_p8 nsplit %d abl %d M=%d N=%d K=%d grid %u\n", NSPLIT, ABL, g.M, g.N, g.K, grid.x);
  hipLaunchKernelGGL(kern, grid, dim3(512), P8_LDS_BYTES, s, g, e);
  return hipGetLastError();
}
template <int NSPLIT, typename Epi>
hipError_t set_attr() {
  return hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_p8_kernel<NSPLIT, Epi, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, P8_LDS_BYTES);
}
}  // namespace

bool p8_applies(int nsplit, const GemmCore& g) {  // an even number of whole k-tiles (the loop body is a pair), the operand modes built here
  if (nsplit != 1 && nsplit != 2) return false;
  const int64_t kbytes = (int64_t)g.K * 2 * (nsplit == 2 ? 2 : 1);
  return kbytes % (2 * GEMM_KTB) == 0 && kbytes >= 2 * GEMM_KTB && g.N % 32 == 0 && g.M >= 1 && 256 * g.lda * 2 < (int64_t)0x7ff00000 && 256 * g.ldw * 2 < (int64_t)0x7ff00000;  // (31-bit offsets inside a tile)
}

template <int NSPLIT, typename Epi>
hipError_t launch_p8(const GemmCore& g, const Epi& e, int abl, hipStream_t s) {
  if (!p8_applies(NSPLIT, g)) return hipErrorNotSupported;
  if constexpr (std::is_same<Epi, PpEpiAct16<NSPLIT == 2 ? 2 : 0, ACT_GELU_TANH>>::value) {  // ablations: the FF1 launch only
    switch (abl) {
      case 0: break;
      case 1: return launch_abl<NSPLIT, Epi, 1>(g, e, s);
      case 4: return launch_abl<NSPLIT, Epi, 4>(g, e, s);
      case 8: return launch_abl<NSPLIT, Epi, 8>(g, e, s);
      case 9: return launch_abl<NSPLIT, Epi, 9>(g, e, s);
      default: return hipErrorNotSupported;
    }
  } else if (abl != 0) {
    return hipErrorNotSupported;
  }
  return launch_abl<NSPLIT, Epi, 0>(g, e, s);
}

#define F5_P8_INST(NSPLIT, ...) template hipError_t launch_p8<NSPLIT, __VA_ARGS__>(const GemmCore&, const __VA_ARGS__&, int, hipStream_t);
F5_P8_INST(1, PpEpiAct16<0, ACT_GELU_TANH>)
F5_P8_INST(1, PpEpiAct16<0, ACT_NONE>)
F5_P8_INST(1, PpEpiGateRes<true>)
F5_P8_INST(1, PpEpiGateRes<false>)
F5_P8_INST(1, PpEpiQKV)
F5_P8_INST(2, PpEpiAct16<2, ACT_GELU_TANH>)
F5_P8_INST(2, PpEpiAct16<2, ACT_NONE>)
F5_P8_INST(2, PpEpiGateRes<true>)
F5_P8_INST(2, PpEpiGateRes<false>)
F5_P8_INST(2, PpEpiQKV)
#undef F5_P8_INST

hipError_t init_p8_kernels() {
  hipError_t e;
  if ((e = set_attr<1, PpEpiAct16<0, ACT_GELU_TANH>>()) != hipSuccess || (e = set_attr<1, PpEpiAct16<0, ACT_NONE>>()) != hipSuccess ||
      (e = set_attr<1, PpEpiGateRes<true>>()) != hipSuccess || (e = set_attr<1, PpEpiGateRes<false>>()) != hipSuccess || (e = set_attr<1, PpEpiQKV>()) != hipSuccess)
    return e;
  if ((e = set_attr<2, PpEpiAct16<2, ACT_GELU_TANH>>()) != hipSuccess || (e = set_attr<2, PpEpiAct16<2, ACT_NONE>>()) != hipSuccess ||
      (e = set_attr<2, PpEpiGateRes<true>>()) != hipSuccess || (e = set_attr<2, PpEpiGateRes<false>>()) != hipSuccess || (e = set_attr<2, PpEpiQKV>()) != hipSuccess)
    return e;
  return hipSuccess;
}
