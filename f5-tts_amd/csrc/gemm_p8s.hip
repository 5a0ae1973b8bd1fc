// gemm_p8s.hip — instantiations of the one-round ping-pong block GEMM (gemm_p8s.h): tile ids 90 .. 94
//   90  192x192  (q|k|v at B = 1: 15 x 16 = 240 workgroups)   91  192x128  (FF1)   92  192x64  (out / FF2)   — 3-stage rings
//   93  192x128, 4 stages (160 KB)   94  192x64, 4 stages (128 KB)
#include <cstdio>
#include <cstdlib>

#include "kernels.h"
#include "gemm_p8s.h"

namespace {
template <int ID>
struct P8sV;
#define F5_P8SV(ID, TM_, TN_, NS_)                      \
  template <>                                           \
  struct P8sV<ID> {                                     \
    static constexpr int TM = TM_, TN = TN_, NS = NS_;  \
  }
F5_P8SV(90, 3, 3, 3);
F5_P8SV(91, 3, 2, 3);
F5_P8SV(92, 3, 1, 3);
F5_P8SV(93, 3, 2, 4);
F5_P8SV(94, 3, 1, 4);
#undef F5_P8SV

template <int NSPLIT, int ID, typename Epi, int ABL>
hipError_t launch_one(const GemmCore& g, const Epi& e, hipStream_t s) {
  using C = P8sV<ID>;
  constexpr int lds = gemm_p8s_lds_bytes<C::TM, C::TN, C::NS>();
  const int64_t kt = (int64_t)g.K * 2 * (NSPLIT == 2 ? 2 : 1) / GEMM_KTB;
  if (kt % 2 != 0 || kt < C::NS + 1) return hipErrorNotSupported;
  auto kern = gemm_p8s_kernel<NSPLIT, C::TM, C::TN, C::NS, Epi, ABL>;
  if constexpr (ABL != 0) {  // microbenchmark ablations only: the production instantiations get their limit in init_p8s_kernels()
    const hipError_t err = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (err != hipSuccess) return err;
  }
  const dim3 grid(((g.M + 64 * C::TM - 1) / (64 * C::TM)) * ((g.N + 64 * C::TN - 1) / (64 * C::TN)), 1, 1);
  static const bool trace = getenv("F5HIP_GEMM_TRACE") != nullptr;
  if (trace) fprintf(stderr, "gemm_p8s tile %d (%dx%d, %d stages) nsplit %d abl %d M=%d N=%d K=%d grid %u\n", ID, 64 * C::TM, 64 * C::TN, C::NS, NSPLIT, ABL, g.M, g.N, g.K, grid.x);
  hipLaunchKernelGGL(kern, grid, dim3(512), lds, s, g, e);
  return hipGetLastError();
}
template <int NSPLIT, int ID, typename Epi>
hipError_t set_attr() {
  using C = P8sV<ID>;
  return hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_p8s_kernel<NSPLIT, C::TM, C::TN, C::NS, Epi, 0>), hipFuncAttributeMaxDynamicSharedMemorySize,
                             gemm_p8s_lds_bytes<C::TM, C::TN, C::NS>());
}
template <int NSPLIT, typename Epi>
hipError_t set_attrs() {
  hipError_t e;
  if ((e = set_attr<NSPLIT, 90, Epi>()) != hipSuccess || (e = set_attr<NSPLIT, 91, Epi>()) != hipSuccess || (e = set_attr<NSPLIT, 92, Epi>()) != hipSuccess ||
      (e = set_attr<NSPLIT, 93, Epi>()) != hipSuccess || (e = set_attr<NSPLIT, 94, Epi>()) != hipSuccess)
    return e;
  return hipSuccess;
}
}  // namespace

bool p8s_applies(int nsplit, const GemmCore& g) {  // whole k-tiles, an even number of them, the operand modes built here, 31-bit offsets inside a tile's descriptors
  if (nsplit != 1 && nsplit != 2) return false;
  const int64_t kbytes = (int64_t)g.K * 2 * (nsplit == 2 ? 2 : 1);
  return kbytes % (2 * GEMM_KTB) == 0 && kbytes >= 6 * GEMM_KTB && g.N % 32 == 0 && g.M >= 1 && 256 * g.lda * 2 < (int64_t)0x7ff00000 && 256 * g.ldw * 2 < (int64_t)0x7ff00000;
}

// abl: 0, or the ablation code of tools/kernel_bench.py (1 no epilogue, 4 no LDS-DMA, 8 no MFMAs, 9 neither epilogue nor MFMAs) — FF1 launch only
template <int NSPLIT, typename Epi>
hipError_t launch_p8s(const GemmCore& g, const Epi& e, int tile, int abl, hipStream_t s) {
  if (!p8s_applies(NSPLIT, g)) return hipErrorNotSupported;
  if constexpr (std::is_same<Epi, PpEpiAct16<NSPLIT == 2 ? 2 : 0, ACT_GELU_TANH>>::value) {
    if (abl != 0) {
      switch (1000 * abl + tile) {
#define F5_ABL(A, ID) case 1000 * A + ID: return launch_one<NSPLIT, ID, Epi, A>(g, e, s);
        F5_ABL(1, 90) F5_ABL(4, 90) F5_ABL(8, 90) F5_ABL(9, 90)
        F5_ABL(1, 91) F5_ABL(4, 91) F5_ABL(8, 91) F5_ABL(9, 91)
#undef F5_ABL
        default: return hipErrorNotSupported;
      }
    }
  } else if (abl != 0) {
    return hipErrorNotSupported;
  }
  switch (tile) {
    case 90: return launch_one<NSPLIT, 90, Epi, 0>(g, e, s);
    case 91: return launch_one<NSPLIT, 91, Epi, 0>(g, e, s);
    case 92: return launch_one<NSPLIT, 92, Epi, 0>(g, e, s);
    case 93: return launch_one<NSPLIT, 93, Epi, 0>(g, e, s);
    case 94: return launch_one<NSPLIT, 94, Epi, 0>(g, e, s);
    default: return hipErrorNotSupported;
  }
}

#define F5_P8S_INST(NSPLIT, ...) template hipError_t launch_p8s<NSPLIT, __VA_ARGS__>(const GemmCore&, const __VA_ARGS__&, int, int, hipStream_t);
F5_P8S_INST(1, PpEpiAct16<0, ACT_GELU_TANH>)
F5_P8S_INST(1, PpEpiAct16<0, ACT_NONE>)
F5_P8S_INST(1, PpEpiGateRes<true>)
F5_P8S_INST(1, PpEpiGateRes<false>)
F5_P8S_INST(1, PpEpiQKV)
F5_P8S_INST(2, PpEpiAct16<2, ACT_GELU_TANH>)
F5_P8S_INST(2, PpEpiAct16<2, ACT_NONE>)
F5_P8S_INST(2, PpEpiGateRes<true>)
F5_P8S_INST(2, PpEpiGateRes<false>)
F5_P8S_INST(2, PpEpiQKV)
#undef F5_P8S_INST

hipError_t init_p8s_kernels() {
  hipError_t e;
  if ((e = set_attrs<1, PpEpiAct16<0, ACT_GELU_TANH>>()) != hipSuccess || (e = set_attrs<1, PpEpiAct16<0, ACT_NONE>>()) != hipSuccess ||
      (e = set_attrs<1, PpEpiGateRes<true>>()) != hipSuccess || (e = set_attrs<1, PpEpiGateRes<false>>()) != hipSuccess || (e = set_attrs<1, PpEpiQKV>()) != hipSuccess)
    return e;
  if ((e = set_attrs<2, PpEpiAct16<2, ACT_GELU_TANH>>()) != hipSuccess || (e = set_attrs<2, PpEpiAct16<2, ACT_NONE>>()) != hipSuccess ||
      (e = set_attrs<2, PpEpiGateRes<true>>()) != hipSuccess || (e = set_attrs<2, PpEpiGateRes<false>>()) != hipSuccess || (e = set_attrs<2, PpEpiQKV>()) != hipSuccess)
    return e;
  return hipSuccess;
}
