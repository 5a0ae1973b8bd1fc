// bigvgan.hip — the HBM-bound kernels of the BigVGAN-v2 generator path (mel -> waveform), gfx950.
//
// Replaces (together with bigvgan.cpp, which strings them and the MFMA GEMM together) `vocoder(mel)` for mel_spec_type "bigvgan"
// (reference src/f5_tts/infer/utils_infer.py:130-144,512-513).  The generator's source is an un-vendored submodule of the reference
// (.gitmodules:1-3, NVIDIA/BigVGAN); the arithmetic follows the published upstream modules named at each kernel and is restated on
// the CPU in oracle/bigvgan_oracle.py (parity unpinned: no reference artefact exists to check either against).
//
// Layout: activations are channels-last fp32 [b, L, C] (a time step is one contiguous row, so a wave reads 64 channels of a row as
// one 256-byte segment, and a tap-gathered GEMM operand row is a concatenation of whole rows).  tests/bigvgan_model.py restates the
// index arithmetic of every kernel here with torch gathers; tests/test_bigvgan_oracle.py checks that against the oracle.
#include <math.h>

#include "bigvgan_kernels.h"
#include "conv_gemm.h"

hipError_t launch_aa_snake(const float* x, float* y, const float* alpha, const float* beta, const float* filt12, int B, int L, int C,
                           int logscale, hipStream_t s, void* oper, int op, int cpad) {
  constexpr int TL = 32;
  if (oper && (cpad % 32 || cpad < C || (op != OP_F32 && op != OP_F16 && op != OP_F16X3))) return hipErrorInvalidValue;
  AaArgs a{};
  a.x = x; a.y = y; a.alpha = alpha; a.beta = beta; a.L = L; a.C = C; a.logscale = logscale;
  a.oper = oper; a.op = oper ? op : OP_F32; a.cpad = oper ? cpad : C;
  for (int j = 0; j < 12; ++j) a.f[j] = filt12[j];
  const int chunks = (L + TL - 1) / TL;
  dim3 grid(((oper ? cpad : C) + 63) / 64, (chunks + 3) / 4, B);
  hipLaunchKernelGGL(aa_snake_kernel<TL>, grid, dim3(64, 4), 0, s, a);
  return hipGetLastError();
}

hipError_t launch_im2col_taps(const float* src, int64_t sb, int64_t sl, int64_t sc, int B, int L, int C, int ntaps, int shift0, int dstep,
                              int cpad, int op, void* out, int64_t ldo, int64_t ob, hipStream_t s) {
  if (cpad % 32 || C > cpad || (op != OP_F32 && op != OP_F16 && op != OP_F16X3)) return hipErrorInvalidValue;
  ColArgs a{};
  a.src = src; a.sb = sb; a.sl = sl; a.sc = sc; a.L = L; a.C = C; a.ntaps = ntaps; a.shift0 = shift0; a.dstep = dstep; a.cpad = cpad;
  a.op = op; a.out = out; a.ldo = ldo; a.ob = ob;
  const int64_t n = (int64_t)L * (ntaps * cpad / 4);
  dim3 grid((unsigned)((n + 255) / 256), B);
  hipLaunchKernelGGL(im2col_kernel, grid, dim3(256), 0, s, a);
  return hipGetLastError();
}

hipError_t launch_mean_streams(const float* const* r, int nk, int64_t n, float* out, hipStream_t s) {
  if (nk < 1 || nk > 4 || n % 4) return hipErrorInvalidValue;
  const int64_t n4 = n / 4;
  hipLaunchKernelGGL(mean_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, s, r[0], nk > 1 ? r[1] : nullptr, nk > 2 ? r[2] : nullptr,
                     nk > 3 ? r[3] : nullptr, nk, (float)nk, n4, out);
  return hipGetLastError();
}

hipError_t launch_conv_post(const float* y, const float* w7, const float* bias, int B, int L, int C, int use_tanh, float* out, hipStream_t s) {
  if (C % 4 || 7 * C * 4 > 48 * 1024) return hipErrorInvalidValue;
  hipLaunchKernelGGL(conv_post_kernel, dim3((L + 255) / 256, B), dim3(256), 7 * C * sizeof(float), s, y, w7, bias, L, C, use_tanh, out);
  return hipGetLastError();
}

// ---- implicit-GEMM conv (conv_gemm.h) ------------------------------------------------------------------------------------------------
namespace {
template <typename T, int NSPLIT, int TM, int TN>
hipError_t launch_conv_one(const GemmCore& g, const ConvTaps& tp, const EpiStore& e, int batch, hipStream_t s) {
  constexpr int lds = gemm_lds_bytes<T, NSPLIT, TM, TN, 2, 2>();
  constexpr int BM = 64 * TM, BN = 64 * TN;
  auto kern = conv_gemm_kernel<T, NSPLIT, TM, TN, EpiStore, 2, 2>;  // (dynamic-LDS limit: init_bigvgan_kernels, per device)
  if ((int64_t)g.a_rows * g.lda * (int64_t)sizeof(T) >= (int64_t)0x7ff00000 || (int64_t)g.w_rows * g.ldw * (int64_t)sizeof(T) >= (int64_t)0x7ff00000)
    return hipErrorInvalidValue;
  dim3 grid(((g.M + BM - 1) / BM) * ((g.N + BN - 1) / BN), 1, batch);
  hipLaunchKernelGGL(kern, grid, dim3(256), lds, s, g, tp, e);
  return hipGetLastError();
}
template <typename T, int NSPLIT, int TM, int TN>
hipError_t set_conv_attr() {
  return hipFuncSetAttribute(reinterpret_cast<const void*>(conv_gemm_kernel<T, NSPLIT, TM, TN, EpiStore, 2, 2>), hipFuncAttributeMaxDynamicSharedMemorySize,
                             gemm_lds_bytes<T, NSPLIT, TM, TN, 2, 2>());
}
template <typename T, int NSPLIT>
hipError_t set_conv_attrs() {
  const hipError_t e = set_conv_attr<T, NSPLIT, 2, 1>();
  return e != hipSuccess ? e : set_conv_attr<T, NSPLIT, 2, 2>();
}
template <typename T, int NSPLIT>
hipError_t launch_conv_op(const GemmCore& g, const ConvTaps& tp, const EpiStore& e, int batch, hipStream_t s) {
  return g.N <= 64 ? launch_conv_one<T, NSPLIT, 2, 1>(g, tp, e, batch, s) : launch_conv_one<T, NSPLIT, 2, 2>(g, tp, e, batch, s);
}
}  // namespace

// Dynamic-LDS limits of the implicit-GEMM conv kernels for the CURRENT device: called at context creation (per device; never inside a capture)
hipError_t init_bigvgan_kernels() {
  hipError_t e;
  if ((e = set_conv_attrs<float, 1>()) != hipSuccess || (e = set_conv_attrs<f16, 1>()) != hipSuccess) return e;
  return set_conv_attrs<f16, 3>();
}

hipError_t launch_conv_gemm(int op, const GemmCore& g, int ntaps, int shift0, int dstep, int cpad, const EpiStore& e, int batch, hipStream_t s) {
  const int seg = cpad * (op == OP_F16 ? 2 : 4);  // bytes of one tap segment of an operand row (packed fp16x3: 2 planes x 2 bytes)
  if (seg % GEMM_KTB || g.K != ntaps * cpad) return hipErrorInvalidValue;
  ConvTaps tp{ntaps, shift0, dstep, seg / GEMM_KTB};
  switch (op) {
    case OP_F32: return launch_conv_op<float, 1>(g, tp, e, batch, s);
    case OP_F16: return launch_conv_op<f16, 1>(g, tp, e, batch, s);
    case OP_F16X3: return launch_conv_op<f16, 3>(g, tp, e, batch, s);
    default: return hipErrorInvalidValue;
  }
}
