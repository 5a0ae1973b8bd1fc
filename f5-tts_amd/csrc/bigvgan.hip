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

#include "conv_gemm.h"
#include "kernels.h"

namespace {

// ---- Activation1d: x2 kaiser-sinc upsample -> Snake / SnakeBeta -> x2 low-pass downsample ------------------------------------------
// upstream alias_free_activation/torch/{act,resample,filter}.py + activations.py.  With the 12-tap filter f, replicate padding of
// x by 5 (upsampler) and of v by (5, 6) (downsampler), the crops of UpSample1d and the stride-2 correlation of LowPassFilter1d:
//   u[2q]   = 2 * sum_{t<6} x[clamp(q - 3 + t, 0, L-1)] * f[11 - 2t]
//   u[2q+1] = 2 * sum_{t<6} x[clamp(q - 2 + t, 0, L-1)] * f[10 - 2t]
//   v[m]    = u[m] + 1/(beta + 1e-9) * sin^2(alpha * u[m])
//   z[l]    = sum_{j<12} f[j] * v[clamp(2l + j - 5, 0, 2L-1)]
// One lane = one channel of a run of TL consecutive time steps: a 12-entry window of v slides by two per step, so every v (one
// sinf) is evaluated once per run plus 10 for the run's prologue.  Lanes of a wave are adjacent channels: every global access is a
// contiguous row segment.  HBM-bound: reads each x row (through L1 for the 11-row neighbourhood) and writes each z row once.
struct AaArgs {
  const float* x;
  float* y;
  const float* alpha;
  const float* beta;
  int L, C, logscale;
  float f[12];
  // optional fused operand emission (conv_impl 2): instead of y, write the GEMM operand copy [L, cpad] in layout `op` (channels
  // [C, cpad) zero) — the conv that follows reads it directly (conv_gemm.h), no fp32 round trip, no separate launch_im2col_taps
  void* oper;
  int op, cpad;
};

__device__ __forceinline__ float aa_v(const float* __restrict__ xc, int ldx, int L, int m, const float (&f)[12], float a, float invb) {
  const int q = m >> 1, odd = m & 1;  // m is already clamped to [0, 2L-1]
  float acc = 0.f;
#pragma unroll
  for (int t = 0; t < 6; ++t) {
    const int i = min(max(q - 3 + odd + t, 0), L - 1);
    acc = fmaf(xc[(int64_t)i * ldx], odd ? f[10 - 2 * t] : f[11 - 2 * t], acc);
  }
  const float u = 2.0f * acc;
  const float s = sinf(u * a);
  return fmaf(invb, s * s, u);
}

template <int TL>
__global__ __launch_bounds__(256) void aa_snake_kernel(AaArgs a) {
  const int c = blockIdx.x * 64 + threadIdx.x;
  const int l0 = (blockIdx.y * 4 + threadIdx.y) * TL;
  if (l0 >= a.L) return;
  // operand addressing of channel c (fused emission): element offset inside a row, row stride, elements are floats or halves
  const int mul = a.op == OP_F16X3 ? 2 : 1;
  const int64_t ldo = (int64_t)a.cpad * mul;
  const int64_t ooff = (int64_t)blockIdx.z * a.L * ldo + pk_off(c, a.op == OP_F16X3);
  auto emit = [&](int l, float z) {
    if (a.op == OP_F32) {
      reinterpret_cast<float*>(a.oper)[ooff + (int64_t)l * ldo] = z;
    } else {
      f16 h, w;
      split_f16(z, h, w);
      f16* o = reinterpret_cast<f16*>(a.oper) + ooff + (int64_t)l * ldo;
      o[0] = h;
      if (a.op == OP_F16X3) o[32] = w;
    }
  };
  if (c >= a.C) {
    if (a.oper && c < a.cpad)
      for (int l = l0; l < min(l0 + TL, a.L); ++l) emit(l, 0.f);
    return;
  }
  const int64_t boff = (int64_t)blockIdx.z * a.L * a.C + c;
  const float* xc = a.x + boff;
  float* yc = a.y + boff;
  float al = a.alpha[c], be = a.beta[c];
  if (a.logscale) { al = expf(al); be = expf(be); }
  const float invb = 1.0f / (be + 1e-9f);
  const int mmax = 2 * a.L - 1;
  float vw[12];
#pragma unroll
  for (int j = 0; j < 12; ++j) vw[j] = aa_v(xc, a.C, a.L, min(max(2 * l0 + j - 5, 0), mmax), a.f, al, invb);
  const int lend = min(l0 + TL, a.L);
#pragma unroll 1
  for (int l = l0; l < lend; ++l) {
    float z = 0.f;
#pragma unroll
    for (int j = 0; j < 12; ++j) z = fmaf(a.f[j], vw[j], z);
    if (a.oper) emit(l, z);
    else yc[(int64_t)l * a.C] = z;
#pragma unroll
    for (int j = 0; j < 10; ++j) vw[j] = vw[j + 2];
    vw[10] = aa_v(xc, a.C, a.L, min(2 * l + 7, mmax), a.f, al, invb);  // window of step l+1: v[2(l+1) + j - 5]
    vw[11] = aa_v(xc, a.C, a.L, min(2 * l + 8, mmax), a.f, al, invb);
  }
}

// ---- tap-gathered GEMM operand ("im2col") ------------------------------------------------------------------------------------------
// out row (b, l), column j * cpad + c  =  src[b, l + shift0 + j * dstep, c]   (0 when that row is outside [0, L) or c >= C)
// in the GEMM's operand layout (kernels.h OP_*): fp32 rows, fp16 rows, or packed fp16 hi/lo rows ([K/32][32 hi | 32 lo], gemm.h).
// Conv1d(k, dilation d, "same" padding): ntaps = k, shift0 = -(k/2) d, dstep = d.  ConvTranspose1d(k, stride u, padding (k-u)/2):
// ntaps = 3, shift0 = -1, dstep = 1 against a weight matrix with u * Cout rows (bigvgan.cpp convt_matrix).  One thread = 4 columns.
struct ColArgs {
  const float* src;
  int64_t sb, sl, sc;  // element strides of src: batch, time step, channel (channel-major mel input: sl = 1, sc = T)
  int L, C, ntaps, shift0, dstep, cpad, op;
  void* out;
  int64_t ldo, ob;     // row / batch stride of out in elements of its type
};

__global__ __launch_bounds__(256) void im2col_kernel(ColArgs a) {
  const int k4 = a.ntaps * a.cpad / 4;
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (int64_t)a.L * k4) return;
  const int l = (int)(idx / k4), col = (int)(idx - (int64_t)l * k4) * 4;
  const int j = col / a.cpad, c = col - j * a.cpad;
  const int srow = l + a.shift0 + j * a.dstep;
  float v[4] = {0.f, 0.f, 0.f, 0.f};
  if (srow >= 0 && srow < a.L) {
    const float* p = a.src + (int64_t)blockIdx.y * a.sb + (int64_t)srow * a.sl;
#pragma unroll
    for (int e = 0; e < 4; ++e)
      if (c + e < a.C) v[e] = p[(int64_t)(c + e) * a.sc];
  }
  const int64_t row = (int64_t)blockIdx.y * a.ob + (int64_t)l * a.ldo;
  if (a.op == OP_F32) {
    *reinterpret_cast<float4*>(reinterpret_cast<float*>(a.out) + row + col) = make_float4(v[0], v[1], v[2], v[3]);
  } else {
    f16x4 hi, lo;
#pragma unroll
    for (int e = 0; e < 4; ++e) { f16 h, w; split_f16(v[e], h, w); hi[e] = h; lo[e] = w; }
    f16* o = reinterpret_cast<f16*>(a.out) + row + pk_off(col, a.op == OP_F16X3);
    *reinterpret_cast<f16x4*>(o) = hi;
    if (a.op == OP_F16X3) *reinterpret_cast<f16x4*>(o + 32) = lo;
  }
}

// ---- mean of the parallel resblocks: x = (r0 + r1 + ...) / nk   (bigvgan.py BigVGAN.forward: xs / self.num_kernels) -----------------
__global__ __launch_bounds__(256) void mean_kernel(const float* r0, const float* r1, const float* r2, const float* r3, int nk, float div,
                                                   int64_t n4, float* out) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n4) return;
  float4 s = reinterpret_cast<const float4*>(r0)[i];
  const float* rs[3] = {r1, r2, r3};
#pragma unroll
  for (int j = 0; j < 3; ++j)
    if (j + 1 < nk) {
      const float4 t = reinterpret_cast<const float4*>(rs[j])[i];
      s.x += t.x; s.y += t.y; s.z += t.z; s.w += t.w;
    }
  // torch divides by num_kernels; x * (1/3) differs from x / 3 in the last bit, so divide
  reinterpret_cast<float4*>(out)[i] = make_float4(s.x / div, s.y / div, s.z / div, s.w / div);
}

// ---- conv_post: Conv1d(C -> 1, k = 7, padding 3) + tanh / clamp(-1, 1) ---------------------------------------------------------------
// y [b, L, C] (the output of activation_post), w7 [7, C] tap-major, out [b, L].  One output sample per thread, fp32 throughout.
__global__ __launch_bounds__(256) void conv_post_kernel(const float* y, const float* w7, const float* bias, int L, int C, int use_tanh, float* out) {
  extern __shared__ float sw[];
  for (int i = threadIdx.x; i < 7 * C; i += 256) sw[i] = w7[i];
  __syncthreads();
  const int l = blockIdx.x * 256 + threadIdx.x;
  if (l >= L) return;
  const float* yb = y + (int64_t)blockIdx.y * L * C;
  float acc = 0.f;
#pragma unroll
  for (int j = 0; j < 7; ++j) {
    const int s = l + j - 3;
    if (s < 0 || s >= L) continue;
    const float4* row = reinterpret_cast<const float4*>(yb + (int64_t)s * C);
    float part = 0.f;
    for (int c4 = 0; c4 < C / 4; ++c4) {
      const float4 t = row[c4];
      const float* w = sw + j * C + c4 * 4;
      part = fmaf(t.x, w[0], part); part = fmaf(t.y, w[1], part); part = fmaf(t.z, w[2], part); part = fmaf(t.w, w[3], part);
    }
    acc += part;
  }
  if (bias) acc += bias[0];
  out[(int64_t)blockIdx.y * L + l] = use_tanh ? tanhf(acc) : fminf(fmaxf(acc, -1.0f), 1.0f);
}

}  // namespace

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
  auto kern = conv_gemm_kernel<T, NSPLIT, TM, TN, EpiStore, 2, 2>;
  static bool attr_done = false;
  if (!attr_done) {
    hipError_t err = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (err != hipSuccess) return err;
    attr_done = true;
  }
  if ((int64_t)g.a_rows * g.lda * (int64_t)sizeof(T) >= (int64_t)0x7ff00000 || (int64_t)g.w_rows * g.ldw * (int64_t)sizeof(T) >= (int64_t)0x7ff00000)
    return hipErrorInvalidValue;
  dim3 grid(((g.M + BM - 1) / BM) * ((g.N + BN - 1) / BN), 1, batch);
  hipLaunchKernelGGL(kern, grid, dim3(256), lds, s, g, tp, e);
  return hipGetLastError();
}
template <typename T, int NSPLIT>
hipError_t launch_conv_op(const GemmCore& g, const ConvTaps& tp, const EpiStore& e, int batch, hipStream_t s) {
  return g.N <= 64 ? launch_conv_one<T, NSPLIT, 2, 1>(g, tp, e, batch, s) : launch_conv_one<T, NSPLIT, 2, 2>(g, tp, e, batch, s);
}
}  // namespace

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
