// gemm.hip — instantiations and launch heuristics of the MFMA GEMM (gemm.h)
#include "kernels.h"

namespace {

template <typename T, int NSPLIT, int TM, int TN, typename Epi>
hipError_t set_attr() {
  constexpr int lds = gemm_lds_bytes<T, NSPLIT, TM, TN>();
  return hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_kernel<T, NSPLIT, TM, TN, Epi>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
}

template <typename Epi>
hipError_t set_attrs_epi() {
  hipError_t e;
  if ((e = set_attr<float, 1, 1, 2, Epi>()) != hipSuccess) return e;
  if ((e = set_attr<float, 1, 2, 2, Epi>()) != hipSuccess) return e;
  if ((e = set_attr<f16, 1, 1, 2, Epi>()) != hipSuccess) return e;
  if ((e = set_attr<f16, 1, 2, 2, Epi>()) != hipSuccess) return e;
  if ((e = set_attr<f16, 3, 1, 2, Epi>()) != hipSuccess) return e;
  if ((e = set_attr<f16, 3, 2, 2, Epi>()) != hipSuccess) return e;
  return hipSuccess;
}

template <typename T, int NSPLIT, int TM, int TN, typename Epi>
hipError_t launch_one(const GemmCore& g, const Epi& e, int batch, hipStream_t s) {
  constexpr int lds = gemm_lds_bytes<T, NSPLIT, TM, TN>();
  auto kern = gemm_kernel<T, NSPLIT, TM, TN, Epi>;
  dim3 grid((g.M + 64 * TM - 1) / (64 * TM), (g.N + 64 * TN - 1) / (64 * TN), batch);
  hipLaunchKernelGGL(kern, grid, dim3(256), lds, s, g, e);
  return hipGetLastError();
}

template <typename T, int NSPLIT, typename Epi>
hipError_t launch_tiled(const GemmCore& g, const Epi& e, int batch, hipStream_t s) {
  // 128x128 tiles unless that leaves most of the 256 CUs idle (B=1: M=2812, N=1024 -> 176 tiles)
  const int64_t big = (int64_t)((g.M + 127) / 128) * ((g.N + 127) / 128) * batch;
  if (big >= 384 || g.M <= 64) {
    if (g.M <= 64) return launch_one<T, NSPLIT, 1, 2, Epi>(g, e, batch, s);
    return launch_one<T, NSPLIT, 2, 2, Epi>(g, e, batch, s);
  }
  return launch_one<T, NSPLIT, 1, 2, Epi>(g, e, batch, s);
}

template <typename Epi>
hipError_t dispatch(int op, const GemmCore& g, const Epi& e, int batch, hipStream_t s) {
  switch (op) {
    case OP_F32: return launch_tiled<float, 1, Epi>(g, e, batch, s);
    case OP_F16: return launch_tiled<f16, 1, Epi>(g, e, batch, s);
    case OP_F16X3: return launch_tiled<f16, 3, Epi>(g, e, batch, s);
    default: return hipErrorInvalidValue;
  }
}

}  // namespace

hipError_t launch_gemm_store(int op, const GemmCore& g, const EpiStore& e, int batch, hipStream_t s) {
  return dispatch<EpiStore>(op, g, e, batch, s);
}
hipError_t launch_gemm_qkv(int op, const GemmCore& g, const EpiQKV& e, hipStream_t s) { return dispatch<EpiQKV>(op, g, e, 1, s); }

hipError_t init_gemm_kernels() {
  hipError_t e = set_attrs_epi<EpiStore>();
  if (e != hipSuccess) return e;
  return set_attrs_epi<EpiQKV>();
}
