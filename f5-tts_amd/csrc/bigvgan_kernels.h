// bigvgan_kernels.h — the HBM-bound kernels of the BigVGAN path (see bigvgan.hip for the launchers and the context).  A header so that
// tests/hipemu can compile the same source for the host and run it thread for thread against tests/bigvgan_model.py.
#pragma once
#include <math.h>

#include "kernels.h"

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
  F5_DYN_LDS(float, sw);
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

