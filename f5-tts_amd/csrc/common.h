// common.h — shared device helpers for libf5hip (gfx950 / CDNA4 only; wave = 64).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef _Float16 f16;
typedef f16 f16x4 __attribute__((ext_vector_type(4)));
typedef f16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// One 16-byte MFMA operand fragment: 8 halves (k = 8*(lane>>5)+0..7 of a 16-wide k-step of
// v_mfma_f32_32x32x16_f16) or 4 floats (k = 4*(lane>>5)+j, j-th of four v_mfma_f32_32x32x2_f32:
// the dot product is order-free in k, so a lane may own 4 consecutive k as long as A and B agree).
union Frag {
  uint4 u;
  f16x8 h;
  f32x4 f;
};

template <typename T>
struct Mma32;
template <>
struct Mma32<f16> {
  static __device__ __forceinline__ void mma(f32x16& acc, const Frag& a, const Frag& b) {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a.h, b.h, acc, 0, 0, 0);
  }
};
template <>
struct Mma32<float> {
  static __device__ __forceinline__ void mma(f32x16& acc, const Frag& a, const Frag& b) {
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.f[0], b.f[0], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.f[1], b.f[1], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.f[2], b.f[2], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.f[3], b.f[3], acc, 0, 0, 0);
  }
};

// 32x32 MFMA accumulator layout (dtype independent): element r of lane l is
//   D[i][j],  j = l & 31,  i = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5).
// All GEMM-like kernels here put the WEIGHT/output-channel axis on i and the ROW (frame) axis on j,
// so a lane owns 4 consecutive output channels of one row per register quad -> 16-byte stores,
// and rope's (2i, 2i+1) pairs are lane-local.

// activations -------------------------------------------------------------------------------------
enum { ACT_NONE = 0, ACT_SILU = 1, ACT_GELU_ERF = 2, ACT_GELU_TANH = 3, ACT_MISH = 4 };

__device__ __forceinline__ float act_silu(float x) { return x / (1.0f + expf(-x)); }
__device__ __forceinline__ float act_gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }
__device__ __forceinline__ float act_gelu_tanh(float x) {
  // torch: 0.5*x*(1+tanh(sqrt(2/pi)*(x+0.044715*x^3)))
  const float k0 = 0.7978845608028654f, k1 = 0.044715f;
  float inner = k0 * (x + k1 * x * x * x);
  return 0.5f * x * (1.0f + tanhf(inner));
}
__device__ __forceinline__ float act_mish(float x) {
  // x * tanh(softplus(x)); softplus with torch's threshold=20
  float sp = x > 20.0f ? x : log1pf(expf(x));
  return x * tanhf(sp);
}
__device__ __forceinline__ float apply_act(int act, float x) {
  switch (act) {
    case ACT_SILU: return act_silu(x);
    case ACT_GELU_ERF: return act_gelu_erf(x);
    case ACT_GELU_TANH: return act_gelu_tanh(x);
    case ACT_MISH: return act_mish(x);
    default: return x;
  }
}

// fp16 hi/lo split: v ~= hi + lo with 22 significant bits
__device__ __forceinline__ void split_f16(float v, f16& hi, f16& lo) {
  hi = (f16)v;
  lo = (f16)(v - (float)hi);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
