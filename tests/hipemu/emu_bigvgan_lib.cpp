// emu_bigvgan_lib.cpp — the BigVGAN half of libf5hip built FOR THE HOST through hipemu.h (TEST INFRASTRUCTURE).
// A unity build of the product sources bigvgan.hip (kernels + launchers) and bigvgan_api.cpp (context, weight layouts, the enqueue
// logic of forward()) with "device" memory = host memory and every kernel launch executed thread for thread on the CPU.  Exports the
// same f5hip_bigvgan_* C ABI, so tests/test_hipemu.py drives the real orchestration code end to end against the oracle.
// Only what lives in translation units that cannot be built for the host (inline asm: gemm.hip's dispatcher, elementwise.hip) is
// replaced: the GEMM launcher below runs the SAME gemm_kernel template (gemm.h) with a fixed tile choice, the two one-time weight
// conversions are plain loops.
#include "hipemu.h"

#include "../../f5-tts_amd/csrc/bigvgan.hip"
#include "../../f5-tts_amd/csrc/bigvgan_api.cpp"

namespace {
template <typename T, int NSPLIT, int TM, int TN>
hipError_t emu_gemm(const GemmCore& g, const EpiStore& e, int batch) {
  constexpr int BM = 64 * TM, BN = 64 * TN;
  const int lds = gemm_lds_bytes<T, NSPLIT, TM, TN, 2, 2>();
  dim3 grid(((g.M + BM - 1) / BM) * ((g.N + BN - 1) / BN), 1, batch);
  hipemu::launch(grid, dim3(256), lds, [=] { gemm_kernel<T, NSPLIT, TM, TN, EpiStore, 2, 2>(g, e); });
  return hipSuccess;
}
template <typename T, int NSPLIT>
hipError_t emu_gemm_tile(const GemmCore& g, const EpiStore& e, int batch, int variant) {
  return variant == 1 ? emu_gemm<T, NSPLIT, 2, 1>(g, e, batch) : emu_gemm<T, NSPLIT, 2, 2>(g, e, batch);
}
}  // namespace

hipError_t launch_gemm_store_variant(int op, const GemmCore& g0, const EpiStore& e, int batch, int variant, hipStream_t) {
  GemmCore g = g0;
  g.group_m = 1;
  switch (op) {
    case OP_F32: return emu_gemm_tile<float, 1>(g, e, batch, variant);
    case OP_F16: return emu_gemm_tile<f16, 1>(g, e, batch, variant);
    case OP_F16X3: return emu_gemm_tile<f16, 3>(g, e, batch, variant);
    default: return hipErrorInvalidValue;
  }
}
hipError_t init_gemm_kernels() { return hipSuccess; }
hipError_t launch_split_f16(const float* src, int64_t n, float prescale, f16* hi, f16* lo, hipStream_t) {
  for (int64_t i = 0; i < n; ++i) {
    f16 h, l;
    split_f16(src[i] * prescale, h, l);
    hi[i] = h;
    if (lo) lo[i] = l;
  }
  return hipSuccess;
}
hipError_t launch_split_f16_packed(const float* src, int64_t rows, int K, f16* dst, hipStream_t) {
  for (int64_t r = 0; r < rows; ++r)
    for (int k = 0; k < K; ++k) {
      f16 h, l;
      split_f16(src[r * K + k], h, l);
      dst[r * 2 * K + pk_off(k, 1)] = h;
      dst[r * 2 * K + pk_off(k, 1) + 32] = l;
    }
  return hipSuccess;
}
