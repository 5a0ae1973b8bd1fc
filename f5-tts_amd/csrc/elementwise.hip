// elementwise.hip — HBM-bound kernels of the hot path: LayerNorm/AdaLN-modulate, token embedding,
// depthwise conv + LN, GRN, masks, CFG+Euler update, tables and one-time weight repacking.
// All are one-wave-per-row or grid-stride kernels with 16-byte accesses along the channel axis.
#include "kernels.h"

namespace {

constexpr int WAVES_PER_BLOCK = 4;

// ---------------------------------------------------------------------------------------------
// LayerNorm (+ affine | AdaLN modulation).  One wave per row; the row lives in registers.
// ---------------------------------------------------------------------------------------------
template <int VPL, bool EARLY = true>  // float4 vectors per lane: D <= 64*4*VPL
__global__ __launch_bounds__(256) void layernorm_kernel(const float* __restrict__ x, int64_t ldx, int M, int D, float eps,
                                                         const float* __restrict__ weight, const float* __restrict__ bias,
                                                         const float* __restrict__ scale, const float* __restrict__ shift,
                                                         float* out32, f16* out16, f16* out16_lo, int64_t ldo, int pk16,
                                                         int64_t ldo16, int mode, int pair16) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * WAVES_PER_BLOCK + (threadIdx.x >> 6);
  if (row >= M) return;
  const float* xr = x + (int64_t)row * ldx;
  float4 v[VPL];
#pragma unroll
  for (int i = 0; i < VPL; ++i) {
    const int c = (i * 64 + lane) * 4;
    v[i] = c < D ? *reinterpret_cast<const float4*>(xr + c) : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  // the per-channel parameters do not depend on the row statistics: requested here, they arrive while the row is being reduced (a second
  // dependent round trip to L2 otherwise — with < 3 waves per SIMD at one utterance nothing else hides it)
  float4 pa[VPL], pb[VPL];  // (weight, bias) or (scale, shift); both pairs only in the general path below
  const bool two = !EARLY || (weight && scale);
  if (!two) {
    const float* A = weight ? weight : scale;
    const float* Bp = weight ? bias : shift;
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
      const int c = (i * 64 + lane) * 4;
      pa[i] = (A && c < D) ? *reinterpret_cast<const float4*>(A + c) : make_float4(0.f, 0.f, 0.f, 0.f);
      pb[i] = (Bp && c < D) ? *reinterpret_cast<const float4*>(Bp + c) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  }
  float sum = 0.f;
#pragma unroll
  for (int i = 0; i < VPL; ++i) sum += (v[i].x + v[i].y) + (v[i].z + v[i].w);
  const float mean = mode == 0 ? wave_sum(sum) / (float)D : 0.f;  // RMSNorm / copy: no centring
  float sq = 0.f;
#pragma unroll
  for (int i = 0; i < VPL; ++i) {
    const int c = (i * 64 + lane) * 4;
    if (c < D) {
      const float a = v[i].x - mean, b = v[i].y - mean, cc = v[i].z - mean, d = v[i].w - mean;
      sq += (a * a + b * b) + (cc * cc + d * d);
    }
  }
  float rstd;
  if (mode == 0) rstd = 1.0f / sqrtf(wave_sum(sq) / (float)D + eps);
  else if (mode == 1) rstd = sqrtf((float)D) / fmaxf(sqrtf(wave_sum(sq)), 1e-12f);  // F.normalize(x, dim=-1) * sqrt(D)
  else rstd = 1.0f;
#pragma unroll
  for (int i = 0; i < VPL; ++i) {
    const int c = (i * 64 + lane) * 4;
    const bool in = c < D;  // (no early exit: the lane exchange below is executed by whole waves)
    float y[4] = {(v[i].x - mean) * rstd, (v[i].y - mean) * rstd, (v[i].z - mean) * rstd, (v[i].w - mean) * rstd};
    if (weight) {
      const float4 w = two ? (in ? *reinterpret_cast<const float4*>(weight + c) : make_float4(0.f, 0.f, 0.f, 0.f)) : pa[i];
      const float4 b = two ? (bias && in ? *reinterpret_cast<const float4*>(bias + c) : make_float4(0.f, 0.f, 0.f, 0.f)) : pb[i];
      y[0] = y[0] * w.x + b.x; y[1] = y[1] * w.y + b.y; y[2] = y[2] * w.z + b.z; y[3] = y[3] * w.w + b.w;
    }
    if (scale) {
      const float4 sc = two ? (in ? *reinterpret_cast<const float4*>(scale + c) : make_float4(0.f, 0.f, 0.f, 0.f)) : pa[i];
      const float4 sh = two ? (in ? *reinterpret_cast<const float4*>(shift + c) : make_float4(0.f, 0.f, 0.f, 0.f)) : pb[i];
      y[0] = y[0] * (1.0f + sc.x) + sh.x; y[1] = y[1] * (1.0f + sc.y) + sh.y;
      y[2] = y[2] * (1.0f + sc.z) + sh.z; y[3] = y[3] * (1.0f + sc.w) + sh.w;
    }
    const int64_t o = (int64_t)row * ldo + c;
    if (out32 && in) *reinterpret_cast<float4*>(out32 + o) = make_float4(y[0], y[1], y[2], y[3]);
    if (out16) {
      f16x4 hi, lo;
#pragma unroll
      for (int e = 0; e < 4; ++e) { f16 h, l; split_f16(y[e], h, l); hi[e] = h; lo[e] = l; }
      const int64_t o16 = (int64_t)row * ldo16 + pk_off(c, pk16);
      if (pair16) {
        // neighbouring lanes hold channels c .. c+3 and c+4 .. c+7: the even lane collects the 8 hi halves, the odd lane the 8 lo halves —
        // one 16-byte store per lane instead of two 8-byte ones (the kernel is store-issue bound once its loads overlap)
        union { f16x4 h; uint32_t u[2]; } H, L;
        H.h = hi; L.h = lo;
        const bool odd = lane & 1;
        const uint32_t r0 = (uint32_t)__shfl_xor((int)(odd ? H.u[0] : L.u[0]), 1, 64), r1 = (uint32_t)__shfl_xor((int)(odd ? H.u[1] : L.u[1]), 1, 64);
        if (!in) {
        } else if (!odd) *reinterpret_cast<uint4*>(out16 + o16) = make_uint4(H.u[0], H.u[1], r0, r1);
        else if (out16_lo) *reinterpret_cast<uint4*>(out16_lo + o16 - 4) = make_uint4(r0, r1, L.u[0], L.u[1]);
      } else if (in) {
        *reinterpret_cast<f16x4*>(out16 + o16) = hi;
        if (out16_lo) *reinterpret_cast<f16x4*>(out16_lo + o16) = lo;
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// fp16m operand rows (common.h "fp16 + MX-fp6 corrections"): LayerNorm producer and one-time / test packing
// ---------------------------------------------------------------------------------------------
// LayerNorm / RMSNorm / copy (+ affine | AdaLN modulation) -> MX operand rows.  One wave per row.  The row is loaded, reduced and normalised
// in the coalesced layout of layernorm_kernel (lane L, piece i: the float4 group G = 64 i + L of a 1024-channel pass), then transposed
// through the wave's own 4 KB of LDS so that lane L = 2 blk + h owns the 16 channels 32 blk + 8 q + 4 h + e — exactly one P_h of common.h —
// and the pack needs no lane exchange.  (First form of round 4: the P_h pieces loaded directly, 16 bytes at a 32-byte stride — every
// instruction touched all 32 lines of the row; 332 against 188 ms of LayerNorm per B = 32 step.  profiles/r04c_*.)  LDS image: group G at
// slot (G & ~7) | ((G & 7) ^ 2 ((G >> 3) & 3)): the natural writes (8 lanes = one 128-byte line) and the transposed reads (a lane pair per
// line, four lines per 8 lanes) are both bank-conflict free.
template <int NB, bool EARLY>  // passes: D <= 1024 NB; EARLY: per-channel parameters requested before the row reduction (few rows: latency; many
                              // rows: the registers cost occupancy and with it bandwidth — as layernorm_kernel, profiles/r04d_*: 4.0 against 5.7 TB/s)
__global__ __launch_bounds__(256) void layernorm_mx_kernel(const float* __restrict__ x, int64_t ldx, int M, int D, float eps,
                                                            const float* __restrict__ weight, const float* __restrict__ bias,
                                                            const float* __restrict__ scale, const float* __restrict__ shift, f16* out16,
                                                            int64_t ldo16, int mode) {
  __shared__ float4 xpose[WAVES_PER_BLOCK][NB * 288];  // 4 KB of values per 1024 channels, re-used as the 32 x 144-byte image of the packed lines
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, h = lane & 1;
  const int row = blockIdx.x * WAVES_PER_BLOCK + wave;
  if (row >= M) return;  // (whole waves: the transposes below synchronise the wave only)
  const float* xr = x + (int64_t)row * ldx;
  const float* A = weight ? weight : scale;
  const float* Bp = weight ? bias : shift;
  float4 v[NB][4], pa[NB][4], pb[NB][4];
#pragma unroll
  for (int b = 0; b < NB; ++b)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int c = (b * 256 + i * 64 + lane) * 4;
      const bool in = c < D;
      v[b][i] = in ? *reinterpret_cast<const float4*>(xr + c) : make_float4(0.f, 0.f, 0.f, 0.f);
      if constexpr (EARLY) {
        pa[b][i] = (A && in) ? *reinterpret_cast<const float4*>(A + c) : make_float4(0.f, 0.f, 0.f, 0.f);
        pb[b][i] = (Bp && in) ? *reinterpret_cast<const float4*>(Bp + c) : make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
  float sum = 0.f;
#pragma unroll
  for (int b = 0; b < NB; ++b)
#pragma unroll
    for (int i = 0; i < 4; ++i) sum += (v[b][i].x + v[b][i].y) + (v[b][i].z + v[b][i].w);
  const float mean = mode == 0 ? wave_sum(sum) / (float)D : 0.f;
  float sq = 0.f;
#pragma unroll
  for (int b = 0; b < NB; ++b)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int c = (b * 256 + i * 64 + lane) * 4;
      if (c < D) {
        const float a0 = v[b][i].x - mean, a1 = v[b][i].y - mean, a2 = v[b][i].z - mean, a3 = v[b][i].w - mean;
        sq += (a0 * a0 + a1 * a1) + (a2 * a2 + a3 * a3);
      }
    }
  float rstd;
  if (mode == 0) rstd = 1.0f / sqrtf(wave_sum(sq) / (float)D + eps);
  else if (mode == 1) rstd = sqrtf((float)D) / fmaxf(sqrtf(wave_sum(sq)), 1e-12f);
  else rstd = 1.0f;
  float4* tp = xpose[wave];
#pragma unroll
  for (int b = 0; b < NB; ++b)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float t[4] = {(v[b][i].x - mean) * rstd, (v[b][i].y - mean) * rstd, (v[b][i].z - mean) * rstd, (v[b][i].w - mean) * rstd};
      if constexpr (!EARLY) {
        const int c = (b * 256 + i * 64 + lane) * 4;
        const bool in = c < D;
        pa[b][i] = (A && in) ? *reinterpret_cast<const float4*>(A + c) : make_float4(0.f, 0.f, 0.f, 0.f);
        pb[b][i] = (Bp && in) ? *reinterpret_cast<const float4*>(Bp + c) : make_float4(0.f, 0.f, 0.f, 0.f);
      }
      if (weight) { t[0] = t[0] * pa[b][i].x + pb[b][i].x; t[1] = t[1] * pa[b][i].y + pb[b][i].y; t[2] = t[2] * pa[b][i].z + pb[b][i].z; t[3] = t[3] * pa[b][i].w + pb[b][i].w; }
      else if (scale) { t[0] = t[0] * (1.0f + pa[b][i].x) + pb[b][i].x; t[1] = t[1] * (1.0f + pa[b][i].y) + pb[b][i].y;
                        t[2] = t[2] * (1.0f + pa[b][i].z) + pb[b][i].z; t[3] = t[3] * (1.0f + pa[b][i].w) + pb[b][i].w; }
      const int G = b * 256 + i * 64 + lane;
      tp[(G & ~7) | ((G & 7) ^ (2 * ((G >> 3) & 3)))] = make_float4(t[0], t[1], t[2], t[3]);
    }
  wave_lds_sync();  // the wave's own 4 KB: no block barrier (a block-wide one cost bandwidth at many rows: 4.0 against 5.7 TB/s, profiles/r04d_*)
  uint32_t hi[NB][8], p[NB][8];
#pragma unroll
  for (int b = 0; b < NB; ++b) {
    const int blk = 32 * b + (lane >> 1);
    float y[16];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int G = blk * 8 + 2 * q + h;  // float4 group 2 q + h of the block (channels 8 q + 4 h + e)
      const float4 t = tp[(G & ~7) | ((G & 7) ^ (2 * ((G >> 3) & 3)))];
      y[4 * q] = t.x; y[4 * q + 1] = t.y; y[4 * q + 2] = t.z; y[4 * q + 3] = t.w;
    }
    mx_pack16<false>(y, hi[b], p[b]);
  }
  // The packed lines leave through LDS as well: a lane holds 64 scattered bytes of its line (8-byte hi pieces, its 32-byte P words), and
  // stores of 16 bytes at a 64-byte stride reach 4.0 TB/s where the fp16x3 kernel's contiguous ones reach 5.7 (profiles/r04d_*).  Image of
  // line blk at 144 blk (the 16 bytes of padding walk the banks), then every lane stores consecutive 16-byte pieces of the row.
  wave_lds_sync();  // every lane has taken its values out of tp
  char* img = reinterpret_cast<char*>(tp);
#pragma unroll
  for (int b = 0; b < NB; ++b) {
    const int blk = 32 * b + (lane >> 1);
    char* line = img + blk * 144;
#pragma unroll
    for (int q = 0; q < 4; ++q) *reinterpret_cast<uint2*>(line + 16 * q + 8 * h) = make_uint2(hi[b][2 * q], hi[b][2 * q + 1]);
    *reinterpret_cast<uint4*>(line + 64 + 32 * h) = make_uint4(p[b][0], p[b][1], p[b][2], p[b][3]);
    *reinterpret_cast<uint4*>(line + 80 + 32 * h) = make_uint4(p[b][4], p[b][5], p[b][6], p[b][7]);
  }
  wave_lds_sync();
  char* orow = reinterpret_cast<char*>(out16 + (int64_t)row * ldo16);
#pragma unroll
  for (int b = 0; b < NB; ++b)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int g = (b * 4 + i) * 64 + lane, ln = g >> 3, pc = g & 7;  // 16-byte piece pc of line ln
      if (32 * ln < D) *reinterpret_cast<uint4*>(orow + (int64_t)ln * 128 + pc * 16) = *reinterpret_cast<const uint4*>(img + ln * 144 + pc * 16);
    }
}

// [rows, K] fp32 (row stride ld) x rowscale[r] -> MX operand rows [rows, 2K halves]; WEIGHT selects which of (coarse, remainder) leads in P
// (common.h).  One thread per (row, 32-k block, half): finalize (weights) and tests / microbenchmarks (activations), not a hot path.
template <bool WEIGHT>
__global__ __launch_bounds__(256) void pack_mx_rows_kernel(const float* src, int64_t ld, int64_t rows, int K, const float* rowscale, f16* dst) {
  const int upr = K / 16;  // units per row
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;  // (the grid covers whole lane pairs: K / 16 is even)
  if (i >= rows * upr) return;
  const int64_t r = i / upr;
  const int u = (int)(i - r * upr), blk = u >> 1, h = u & 1;
  const float rs = rowscale ? rowscale[r] : 1.0f;
  float v[16];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const float4 t = *reinterpret_cast<const float4*>(src + r * ld + 32 * blk + 8 * q + 4 * h);
    v[4 * q] = t.x * rs; v[4 * q + 1] = t.y * rs; v[4 * q + 2] = t.z * rs; v[4 * q + 3] = t.w * rs;
  }
  uint32_t hi[8], p[8];
  mx_pack16<WEIGHT>(v, hi, p);
  char* line = reinterpret_cast<char*>(dst + r * 2 * K) + (int64_t)blk * 128;
#pragma unroll
  for (int q = 0; q < 4; ++q) *reinterpret_cast<uint2*>(line + (8 * q + 4 * h) * 2) = make_uint2(hi[2 * q], hi[2 * q + 1]);
  *reinterpret_cast<uint4*>(line + 64 + 32 * h) = make_uint4(p[0], p[1], p[2], p[3]);
  *reinterpret_cast<uint4*>(line + 80 + 32 * h) = make_uint4(p[4], p[5], p[6], p[7]);
}

// ---------------------------------------------------------------------------------------------
// text embedding (reference model/backbones/dit.py:86-127)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void text_embed_kernel(const int32_t* __restrict__ tok, const uint8_t* __restrict__ valid,
                                                          const float* __restrict__ table, const float* __restrict__ freqs,
                                                          int B, int n, int T, int mask_padding, int has_pos, float* out) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * WAVES_PER_BLOCK + (threadIdx.x >> 6);  // over 2B*n
  if (row >= (int64_t)2 * B * n) return;
  const int s = (int)(row / n), pos = (int)(row - (int64_t)s * n);
  const int b = s % B;
  const bool uncond = s >= B;
  const int id = tok[(int64_t)b * n + pos];
  const bool ok = valid[(int64_t)b * n + pos] != 0;
  const bool filler = mask_padding && id == 0;  // text_mask is computed BEFORE drop_text zeroes the ids (dit.py:103-107)
  const float* e = table + (int64_t)(uncond ? 0 : id) * T;
  const float* f = freqs + (int64_t)pos * T;
  float* o = out + row * T;
  for (int c = lane * 4; c < T; c += 256) {
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (ok && !(filler && has_pos)) {
      v = *reinterpret_cast<const float4*>(e + c);
      if (has_pos) {
        const float4 p = *reinterpret_cast<const float4*>(f + c);
        v.x += p.x; v.y += p.y; v.z += p.z; v.w += p.w;
      }
    }
    *reinterpret_cast<float4*>(o + c) = v;
  }
}

// ---------------------------------------------------------------------------------------------
// depthwise conv k=7 + bias + LayerNorm(affine).  One wave per (sequence, frame) row.
// ---------------------------------------------------------------------------------------------
template <int VPL>
__global__ __launch_bounds__(256) void dwconv7_ln_kernel(const float* __restrict__ x, int S, int n, int C,
                                                          const float* __restrict__ w7, const float* __restrict__ cbias,
                                                          const float* __restrict__ ln_w, const float* __restrict__ ln_b, float eps,
                                                          float* out) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * WAVES_PER_BLOCK + (threadIdx.x >> 6);
  if (row >= (int64_t)S * n) return;
  const int pos = (int)(row % n);
  float4 v[VPL];
  float sum = 0.f;
#pragma unroll
  for (int i = 0; i < VPL; ++i) {
    const int c = (i * 64 + lane) * 4;
    if (c < C) {
      float4 acc = *reinterpret_cast<const float4*>(cbias + c);
#pragma unroll
      for (int t = 0; t < 7; ++t) {
        const int p = pos + t - 3;
        if (p >= 0 && p < n) {
          const float4 xv = *reinterpret_cast<const float4*>(x + (row + (t - 3)) * C + c);
          const float4 wv = *reinterpret_cast<const float4*>(w7 + (int64_t)t * C + c);
          acc.x += xv.x * wv.x; acc.y += xv.y * wv.y; acc.z += xv.z * wv.z; acc.w += xv.w * wv.w;
        }
      }
      v[i] = acc;
      sum += (acc.x + acc.y) + (acc.z + acc.w);
    } else {
      v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
  }
  const float mean = wave_sum(sum) / (float)C;
  float sq = 0.f;
#pragma unroll
  for (int i = 0; i < VPL; ++i) {
    const int c = (i * 64 + lane) * 4;
    if (c < C) {
      const float a = v[i].x - mean, b = v[i].y - mean, cc = v[i].z - mean, d = v[i].w - mean;
      sq += (a * a + b * b) + (cc * cc + d * d);
    }
  }
  const float rstd = 1.0f / sqrtf(wave_sum(sq) / (float)C + eps);
#pragma unroll
  for (int i = 0; i < VPL; ++i) {
    const int c = (i * 64 + lane) * 4;
    if (c >= C) continue;
    const float4 w = *reinterpret_cast<const float4*>(ln_w + c);
    const float4 b = *reinterpret_cast<const float4*>(ln_b + c);
    *reinterpret_cast<float4*>(out + row * C + c) =
        make_float4((v[i].x - mean) * rstd * w.x + b.x, (v[i].y - mean) * rstd * w.y + b.y, (v[i].z - mean) * rstd * w.z + b.z,
                    (v[i].w - mean) * rstd * w.w + b.w);
  }
}

// ---------------------------------------------------------------------------------------------
// GRN
// ---------------------------------------------------------------------------------------------
// Round 6 (grn_apply at 22 % / 46 % of HBM peak, the finish kernel 8.8 us for 300 KB).  Three kernels per GRN:
//   grn_sumsq_kernel   grid (slices, S): a workgroup sums h^2 over its slice of rows for ALL channels — whole 4 KB rows per load instruction
//                      (the round-1 form read 256-byte pieces of a row per workgroup) — into part[(s * RS + z) * C + c];
//   grn_finish_kernel  grid S: adds the slices in slice order (all loads issued before the first add: the serial form paid one L2 round trip
//                      per slice), takes Gx = sqrt, the channel mean once per SEQUENCE (the round-1 apply kernel re-reduced it in every wave,
//                      4 KB of L2 reads and C square roots per row) and writes nx[s, c] = Gx / (mean_c(Gx) + 1e-6);
//   grn_apply_kernel   a wave keeps nx, gamma, beta of its channels in registers over GRN_ROWS rows of one sequence: per row 4 KB in, 4 KB out.
constexpr int GRN_MAX_SLICES = 32;
constexpr int GRN_ROWS = 8;  // rows per wave of grn_apply_kernel
__global__ __launch_bounds__(256) void grn_sumsq_kernel(const float* __restrict__ h, int n, int C, float* part) {
  const int s = blockIdx.y, RS = gridDim.x, z = blockIdx.x;
  const int r0 = (int)((int64_t)n * z / RS), r1 = (int)((int64_t)n * (z + 1) / RS);
  for (int c = threadIdx.x * 4; c < C; c += 1024) {
    const float* p = h + ((int64_t)s * n + r0) * C + c;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    int r = r0;
    for (; r + 4 <= r1; r += 4, p += 4 * (int64_t)C) {  // four rows in flight
      const float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + C);
      const float4 d = *reinterpret_cast<const float4*>(p + 2 * (int64_t)C), e = *reinterpret_cast<const float4*>(p + 3 * (int64_t)C);
      acc.x += a.x * a.x; acc.y += a.y * a.y; acc.z += a.z * a.z; acc.w += a.w * a.w;
      acc.x += b.x * b.x; acc.y += b.y * b.y; acc.z += b.z * b.z; acc.w += b.w * b.w;
      acc.x += d.x * d.x; acc.y += d.y * d.y; acc.z += d.z * d.z; acc.w += d.w * d.w;
      acc.x += e.x * e.x; acc.y += e.y * e.y; acc.z += e.z * e.z; acc.w += e.w * e.w;
    }
    for (; r < r1; ++r, p += C) {
      const float4 a = *reinterpret_cast<const float4*>(p);
      acc.x += a.x * a.x; acc.y += a.y * a.y; acc.z += a.z * a.z; acc.w += a.w * a.w;
    }
    *reinterpret_cast<float4*>(part + ((int64_t)s * RS + z) * C + c) = acc;
  }
}
// one workgroup per sequence: nx[s, c] = sqrt(sum_z part) / (mean_c sqrt(.) + 1e-6)
__global__ __launch_bounds__(256) void grn_finish_kernel(const float* __restrict__ part, int RS, int C, float* nx) {
  __shared__ float red[4];
  const int s = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float local = 0.f;
  for (int c = threadIdx.x * 4; c < C; c += 1024) {
    float4 v[GRN_MAX_SLICES];
#pragma unroll
    for (int z = 0; z < GRN_MAX_SLICES; ++z)
      v[z] = z < RS ? *reinterpret_cast<const float4*>(part + ((int64_t)s * RS + z) * C + c) : make_float4(0.f, 0.f, 0.f, 0.f);
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int z = 0; z < GRN_MAX_SLICES; ++z) { acc.x += v[z].x; acc.y += v[z].y; acc.z += v[z].z; acc.w += v[z].w; }
    const float4 g = make_float4(sqrtf(acc.x), sqrtf(acc.y), sqrtf(acc.z), sqrtf(acc.w));
    *reinterpret_cast<float4*>(nx + (int64_t)s * C + c) = g;  // Gx for now: divided below, by the threads that wrote it
    local += (g.x + g.y) + (g.z + g.w);
  }
  local = wave_sum(local);
  if (lane == 0) red[wave] = local;
  __syncthreads();
  const float denom = ((red[0] + red[1]) + (red[2] + red[3])) / (float)C + 1e-6f;
  for (int c = threadIdx.x * 4; c < C; c += 1024) {
    float4 g = *reinterpret_cast<const float4*>(nx + (int64_t)s * C + c);
    g.x /= denom; g.y /= denom; g.z /= denom; g.w /= denom;
    *reinterpret_cast<float4*>(nx + (int64_t)s * C + c) = g;
  }
}

// grid (ceil(n / (4 GRN_ROWS)), S): wave w of a block takes rows [GRN_ROWS (4 blockIdx.x + w), + GRN_ROWS) of sequence blockIdx.y.
// VPL float4 per lane cover C <= 1024 VPL / 4 channels.
template <int VPL>
__global__ __launch_bounds__(256) void grn_apply_kernel(const float* __restrict__ h, const float* __restrict__ nx,
                                                         const float* __restrict__ gamma, const float* __restrict__ beta, int n, int C, float* out) {
  const int lane = threadIdx.x & 63, s = blockIdx.y;
  const int r0 = (blockIdx.x * WAVES_PER_BLOCK + (threadIdx.x >> 6)) * GRN_ROWS;
  if (r0 >= n) return;
  float4 sv[VPL], g[VPL], b[VPL];
#pragma unroll
  for (int i = 0; i < VPL; ++i) {
    const int c = (i * 64 + lane) * 4;
    const bool in = c < C;
    sv[i] = in ? *reinterpret_cast<const float4*>(nx + (int64_t)s * C + c) : make_float4(0.f, 0.f, 0.f, 0.f);
    g[i] = in ? *reinterpret_cast<const float4*>(gamma + c) : make_float4(0.f, 0.f, 0.f, 0.f);
    b[i] = in ? *reinterpret_cast<const float4*>(beta + c) : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  const int r1 = r0 + GRN_ROWS < n ? r0 + GRN_ROWS : n;
  for (int r = r0; r < r1; ++r) {
    const int64_t row = (int64_t)s * n + r;
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
      const int c = (i * 64 + lane) * 4;
      if (c >= C) continue;
      const float4 hv = *reinterpret_cast<const float4*>(h + row * C + c);
      float4 o;
      o.x = g[i].x * (hv.x * sv[i].x) + b[i].x + hv.x;
      o.y = g[i].y * (hv.y * sv[i].y) + b[i].y + hv.y;
      o.z = g[i].z * (hv.z * sv[i].z) + b[i].z + hv.z;
      o.w = g[i].w * (hv.w * sv[i].w) + b[i].w + hv.w;
      *reinterpret_cast<float4*>(out + row * C + c) = o;
    }
  }
}

// ---------------------------------------------------------------------------------------------
// masks / selects  (C % 4 == 0)
// ---------------------------------------------------------------------------------------------
__global__ void zero_rows_kernel(float* x, const uint8_t* mask, int64_t rows, int C4) {
  const int64_t total = rows * C4;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    if (mask[i / C4]) reinterpret_cast<float4*>(x)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
}
__global__ void where_rows_kernel(const uint8_t* mask, const float* a, const float* b, int64_t rows, int C4, float* out) {
  const int64_t total = rows * C4;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const bool m = mask[i / C4] != 0;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (m) v = reinterpret_cast<const float4*>(a)[i];
    else if (b) v = reinterpret_cast<const float4*>(b)[i];
    reinterpret_cast<float4*>(out)[i] = v;
  }
}

// v [2*half]: first half cond prediction, second half uncond.  (reference cfm.py:190-191; euler step)
__global__ void cfg_euler_kernel(const float* __restrict__ base, float* dst, const float* __restrict__ v, int64_t half4, int has_uncond,
                                 const float* coef_ptr, const float* cfg_ptr, float* traj_next, float* vel_dbg) {
  const float dt = *coef_ptr, cfg = *cfg_ptr;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < half4; i += (int64_t)gridDim.x * blockDim.x) {
    const float4 c = reinterpret_cast<const float4*>(v)[i];
    float4 g = c;
    if (has_uncond) {
      const float4 u = reinterpret_cast<const float4*>(v)[half4 + i];
      g.x = c.x + (c.x - u.x) * cfg; g.y = c.y + (c.y - u.y) * cfg; g.z = c.z + (c.z - u.z) * cfg; g.w = c.w + (c.w - u.w) * cfg;
    }
    float4 yy = reinterpret_cast<const float4*>(base)[i];
    yy.x += dt * g.x; yy.y += dt * g.y; yy.z += dt * g.z; yy.w += dt * g.w;
    reinterpret_cast<float4*>(dst)[i] = yy;
    if (traj_next) reinterpret_cast<float4*>(traj_next)[i] = yy;
    if (vel_dbg) reinterpret_cast<float4*>(vel_dbg)[i] = g;
  }
}

// ---------------------------------------------------------------------------------------------
// tables
// ---------------------------------------------------------------------------------------------
// reference model/modules.py:157-169: emb = 1000 * t * exp(-k * ln(1e4)/(half-1)); cat(sin, cos)
__global__ void time_sinus_kernel(const float* t, int S, int dim, float* out) {
  const int half = dim / 2;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= S * half) return;
  const int s = i / half, k = i - s * half;
  const float e = logf(10000.0f) / (float)(half - 1);
  const float f = expf((float)k * -e);
  const float a = 1000.0f * t[s] * f;
  out[(int64_t)s * dim + k] = sinf(a);
  out[(int64_t)s * dim + half + k] = cosf(a);
}
__global__ void rope_table_kernel(const float* inv_freq, int n, int half, float* out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n * half) return;
  const int pos = i / half, k = i - pos * half;
  const float a = (float)pos * inv_freq[k];
  out[2 * (int64_t)i] = cosf(a);
  out[2 * (int64_t)i + 1] = sinf(a);
}
// [rows, K] fp32 -> packed fp16x3 operand rows [rows, 2K] ([K/32][32 hi | 32 lo]), K % 32 == 0
__global__ void split_f16_packed_kernel(const float* src, int64_t rows, int K, f16* dst) {
  const int64_t n = rows * K;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = i / K;
    const int k = (int)(i - r * K);
    f16 h, l;
    split_f16(src[i], h, l);
    f16* d = dst + r * 2 * K + pk_off(k, 1);
    d[0] = h;
    d[32] = l;
  }
}
// Weight conditioning (GemmCore::w_alpha): one wave per row of W [rows, K]: scale[r] = 2^e with the row's largest |entry| * 2^e in
// [2^12, 2^13), alpha[r] = 2^-e; an all-zero row keeps 1.  Then the plain fp16 copy and the packed hi | lo copy of W * scale.
__global__ __launch_bounds__(256) void row_pow2_scale_kernel(const float* src, int rows, int K, float* scale, float* alpha) {
  const int lane = threadIdx.x & 63, r = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= rows) return;
  float m = 0.f;
  for (int k = lane; k < K; k += 64) m = fmaxf(m, fabsf(src[(int64_t)r * K + k]));
  m = wave_max(m);
  if (lane == 0) {
    int e = 0;
    if (m > 0.f && m < INFINITY) {
      int ex;
      (void)frexpf(m, &ex);  // m = f * 2^ex, f in [0.5, 1)
      e = 13 - ex;           // m * 2^e in [2^12, 2^13)
      e = e > 100 ? 100 : e < -100 ? -100 : e;
    }
    scale[r] = ldexpf(1.0f, e);
    alpha[r] = ldexpf(1.0f, -e);
  }
}
__global__ void split_f16_rows_kernel(const float* src, int64_t rows, int K, const float* rowscale, f16* hi, f16* pk) {
  const int64_t n = rows * K;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = i / K;
    const int k = (int)(i - r * K);
    f16 h, l;
    split_f16(src[i] * rowscale[r], h, l);
    hi[i] = h;
    f16* d = pk + r * 2 * K + pk_off(k, 1);
    d[0] = h;
    d[32] = l;
  }
}
__global__ void split_f16_kernel(const float* src, int64_t n, float prescale, f16* hi, f16* lo) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    f16 h, l;
    split_f16(src[i] * prescale, h, l);
    hi[i] = h;
    if (lo) lo[i] = l;
  }
}
// w [D, cpg, K] (out-channel, in-channel-in-group, tap) -> [G][K][co][ci]
__global__ void convpos_pack_kernel(const float* w, int D, int cpg, int K, float* w32, f16* whi, f16* wlo) {
  const int64_t total = (int64_t)D * cpg * K;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int ci = (int)(i % cpg);
    const int co = (int)((i / cpg) % cpg);
    const int t = (int)((i / ((int64_t)cpg * cpg)) % K);
    const int g = (int)(i / ((int64_t)cpg * cpg * K));
    const float v = w[((int64_t)(g * cpg + co) * cpg + ci) * K + t];
    w32[i] = v;
    f16 h, l;
    split_f16(v, h, l);
    whi[i] = h;
    wlo[i] = l;
  }
}
__global__ void dw_pack_kernel(const float* w, int C, float* w7) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= 7 * C) return;
  const int t = i / C, c = i - t * C;
  w7[i] = w[c * 7 + t];
}

// exact attention: softmax over the first kv columns of each score row, zero the rest (incl. padding to ld).
// stats (engine option "attn_stats", f5hip_attention_stats): the row's LARGEST probability is 1 / sum here (exp(max - max) = 1) — how sharp a
// checkpoint's attention is, which is what decides the half-precision attention form it needs (sharpness sweep);
// stats = {max over rows, sum over rows, rows, rows above 1/2} as doubles, accumulated with atomics (a diagnostic pass, not a hot path)
__global__ __launch_bounds__(256) void softmax_rows_kernel(float* S, int64_t rows, int ld, int nseq, int heads,
                                                            const int32_t* kvlen, int kv_default, const int32_t* kvlen2, int seg2_off, double* stats) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * WAVES_PER_BLOCK + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int bp = (int)(row / ((int64_t)heads * nseq));
  // valid columns: [0, hlo) and [hhi, kv) (the hole is empty unless a second key run is given: MMDiT audio + text mask)
  int kv = kvlen ? kvlen[bp] : kv_default, hlo = 0, hhi = 0;
  if (kvlen && kvlen2) {
    hlo = kv; hhi = seg2_off; kv = seg2_off + kvlen2[bp];
    if (hlo >= hhi) hlo = hhi = 0;
  }
  float* p = S + row * ld;
  float mx = -INFINITY;
  for (int c = lane; c < kv; c += 64)
    if (c < hlo || c >= hhi) mx = fmaxf(mx, p[c]);
  mx = wave_max(mx);
  float sum = 0.f;
  for (int c = lane; c < kv; c += 64) {
    const float e = (c < hlo || c >= hhi) ? expf(p[c] - mx) : 0.f;
    p[c] = e;
    sum += e;
  }
  sum = wave_sum(sum);
  const float inv = 1.0f / sum;
  for (int c = lane; c < ld; c += 64) p[c] = c < kv ? p[c] * inv : 0.f;
  if (stats && lane == 0 && kv > 0) {
    const double pm = (double)inv;
    unsigned long long bits;
    memcpy(&bits, &pm, 8);  // (positive doubles order like their bit patterns)
    atomicMax(reinterpret_cast<unsigned long long*>(stats), bits);
    atomicAdd(stats + 1, pm);
    atomicAdd(stats + 2, 1.0);
    if (inv > 0.5f) atomicAdd(stats + 3, 1.0);
  }
}

// mel [B, T, Cin] (or [B, Cin, T]) -> col [B*T, ldc], k = ci*7 + tap (matches weight [Cout, Cin, 7] flattened)
__global__ void im2col7_kernel(const float* mel, int B, int T, int Cin, int channel_major, float* col, int64_t ldc) {
  const int64_t total = (int64_t)B * T * ldc;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int k = (int)(i % ldc);
    const int64_t row = i / ldc;
    float v = 0.f;
    if (k < Cin * 7) {
      const int ci = k / 7, tap = k - ci * 7;
      const int b = (int)(row / T), pos = (int)(row - (int64_t)b * T) + tap - 3;
      if (pos >= 0 && pos < T) v = channel_major ? mel[((int64_t)b * Cin + ci) * T + pos] : mel[((int64_t)b * T + pos) * Cin + ci];
    }
    col[i] = v;
  }
}

inline int grid_1d(int64_t total, int block = 256, int cap = 8192) {
  int64_t g = (total + block - 1) / block;
  if (g > cap) g = cap;
  if (g < 1) g = 1;
  return (int)g;
}

}  // namespace

hipError_t launch_layernorm(const float* x, int64_t ldx, int M, int D, float eps, const float* weight, const float* bias,
                            const float* scale, const float* shift, float* out32, f16* out16, f16* out16_lo, int64_t ldo,
                            hipStream_t s, int pk16, int64_t ldo16, int mode) {
  if (D % 4 || D > 2048 || M <= 0) return hipErrorInvalidValue;
  if (ldo16 == 0) ldo16 = ldo;
  if (pk16 == 2) {  // MX operand rows (fp16m lines): a kernel of its own lane layout
    if (D % 32 || out32 || !out16 || ldo16 % 8 || (reinterpret_cast<uintptr_t>(out16) & 15) || (weight && scale)) return hipErrorInvalidValue;
    dim3 grid((M + WAVES_PER_BLOCK - 1) / WAVES_PER_BLOCK);
    const bool early = M < 8192;
#define F5_LNMX(NB, E) hipLaunchKernelGGL((layernorm_mx_kernel<NB, E>), grid, dim3(256), 0, s, x, ldx, M, D, eps, weight, bias, scale, shift, out16, ldo16, mode)
    if (D <= 1024) { if (early) F5_LNMX(1, true); else F5_LNMX(1, false); }
    else { if (early) F5_LNMX(2, true); else F5_LNMX(2, false); }
#undef F5_LNMX
    return hipGetLastError();
  }
  // paired 16-byte half stores: whole pairs of 4-channel groups per row, 16-byte aligned rows
  const int pair16 = out16 && D % 8 == 0 && ldo16 % 8 == 0 && (reinterpret_cast<uintptr_t>(out16) & 15) == 0 &&
                     (!out16_lo || (reinterpret_cast<uintptr_t>(out16_lo) & 15) == 0);
  dim3 grid((M + WAVES_PER_BLOCK - 1) / WAVES_PER_BLOCK);
  // Two shapes of the same arithmetic.  Few rows (one utterance: < 3 waves per SIMD, latency-bound): parameters requested before the
  // reduction and paired 16-byte stores, 7.24 against 7.70 ms per B = 1 sample.  Many rows (bandwidth-bound, 8 waves per SIMD hide the
  // second round trip): the lean kernel — the early parameters cost registers and the lane exchange LDS-pipe slots, 229 against 187 ms per
  // B = 32 sample (same box, profiles/r02d_*).
  const bool late = M >= 8192;
#define F5_LN(V, E, P) hipLaunchKernelGGL((layernorm_kernel<V, E>), grid, dim3(256), 0, s, x, ldx, M, D, eps, weight, bias, scale, shift, out32, out16, out16_lo, ldo, pk16, ldo16, mode, P)
  if (late) {
    if (D <= 256) F5_LN(1, false, 0); else if (D <= 512) F5_LN(2, false, 0); else if (D <= 1024) F5_LN(4, false, 0); else F5_LN(8, false, 0);
  } else {
    if (D <= 256) F5_LN(1, true, pair16); else if (D <= 512) F5_LN(2, true, pair16); else if (D <= 1024) F5_LN(4, true, pair16); else F5_LN(8, true, pair16);
  }
#undef F5_LN
  return hipGetLastError();
}

hipError_t launch_text_embed(const int32_t* tok, const uint8_t* valid, const float* table, const float* freqs_cis, int B, int n,
                             int T, int mask_padding, int has_pos, float* out, hipStream_t s) {
  if (T % 4) return hipErrorInvalidValue;
  const int64_t rows = (int64_t)2 * B * n;
  hipLaunchKernelGGL(text_embed_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, s, tok, valid, table, freqs_cis, B, n, T,
                     mask_padding, has_pos, out);
  return hipGetLastError();
}

hipError_t launch_dwconv7_ln(const float* x, int S, int n, int C, const float* w7, const float* cbias, const float* ln_w,
                             const float* ln_b, float eps, float* out, hipStream_t s) {
  if (C % 4 || C > 2048) return hipErrorInvalidValue;
  const int64_t rows = (int64_t)S * n;
  dim3 grid((unsigned)((rows + 3) / 4));
  if (C <= 256) hipLaunchKernelGGL(dwconv7_ln_kernel<1>, grid, dim3(256), 0, s, x, S, n, C, w7, cbias, ln_w, ln_b, eps, out);
  else if (C <= 512) hipLaunchKernelGGL(dwconv7_ln_kernel<2>, grid, dim3(256), 0, s, x, S, n, C, w7, cbias, ln_w, ln_b, eps, out);
  else if (C <= 1024) hipLaunchKernelGGL(dwconv7_ln_kernel<4>, grid, dim3(256), 0, s, x, S, n, C, w7, cbias, ln_w, ln_b, eps, out);
  else hipLaunchKernelGGL(dwconv7_ln_kernel<8>, grid, dim3(256), 0, s, x, S, n, C, w7, cbias, ln_w, ln_b, eps, out);
  return hipGetLastError();
}

int grn_sumsq_slices(int n) { return n >= 256 ? GRN_MAX_SLICES : 1; }
// nx [S, C] <- Gx / (mean_c Gx + 1e-6); part: S * slices * C floats of scratch
hipError_t launch_grn_stats(const float* h, int S, int n, int C, float* nx, float* part, hipStream_t s) {
  if (C % 4) return hipErrorInvalidValue;
  const int RS = grn_sumsq_slices(n);
  hipLaunchKernelGGL(grn_sumsq_kernel, dim3(RS, S), dim3(256), 0, s, h, n, C, part);
  hipLaunchKernelGGL(grn_finish_kernel, dim3(S), dim3(256), 0, s, part, RS, C, nx);
  return hipGetLastError();
}
hipError_t launch_grn_apply(const float* h, const float* nx, const float* gamma, const float* beta, int S, int n, int C,
                            float* out, hipStream_t s) {
  if (C % 4 || C > 4096) return hipErrorInvalidValue;
  const dim3 grid((n + WAVES_PER_BLOCK * GRN_ROWS - 1) / (WAVES_PER_BLOCK * GRN_ROWS), S);
  if (C <= 256) hipLaunchKernelGGL(grn_apply_kernel<1>, grid, dim3(256), 0, s, h, nx, gamma, beta, n, C, out);
  else if (C <= 512) hipLaunchKernelGGL(grn_apply_kernel<2>, grid, dim3(256), 0, s, h, nx, gamma, beta, n, C, out);
  else if (C <= 1024) hipLaunchKernelGGL(grn_apply_kernel<4>, grid, dim3(256), 0, s, h, nx, gamma, beta, n, C, out);
  else if (C <= 2048) hipLaunchKernelGGL(grn_apply_kernel<8>, grid, dim3(256), 0, s, h, nx, gamma, beta, n, C, out);
  else hipLaunchKernelGGL(grn_apply_kernel<16>, grid, dim3(256), 0, s, h, nx, gamma, beta, n, C, out);
  return hipGetLastError();
}
hipError_t launch_zero_rows(float* x, const uint8_t* mask, int64_t rows, int C, hipStream_t s) {
  if (C % 4) return hipErrorInvalidValue;
  hipLaunchKernelGGL(zero_rows_kernel, dim3(grid_1d(rows * (C / 4))), dim3(256), 0, s, x, mask, rows, C / 4);
  return hipGetLastError();
}
hipError_t launch_mask_select(const float* a, const uint8_t* mask, int64_t rows, int C, float* out, hipStream_t s) {
  return launch_where_rows(mask, a, nullptr, rows, C, out, s);
}
hipError_t launch_where_rows(const uint8_t* mask, const float* a, const float* b, int64_t rows, int C, float* out, hipStream_t s) {
  if (C % 4) return hipErrorInvalidValue;
  hipLaunchKernelGGL(where_rows_kernel, dim3(grid_1d(rows * (C / 4))), dim3(256), 0, s, mask, a, b, rows, C / 4, out);
  return hipGetLastError();
}
hipError_t launch_cfg_euler(const float* base, float* dst, const float* v, int64_t half_elems, int has_uncond, const float* coef_ptr,
                            const float* cfg_ptr, float* traj_next, float* vel_dbg, hipStream_t s) {
  if (half_elems % 4) return hipErrorInvalidValue;
  hipLaunchKernelGGL(cfg_euler_kernel, dim3(grid_1d(half_elems / 4)), dim3(256), 0, s, base, dst, v, half_elems / 4, has_uncond, coef_ptr, cfg_ptr,
                     traj_next, vel_dbg);
  return hipGetLastError();
}
hipError_t launch_time_sinus(const float* t, int S, int dim, float* out, hipStream_t s) {
  hipLaunchKernelGGL(time_sinus_kernel, dim3(grid_1d((int64_t)S * dim / 2)), dim3(256), 0, s, t, S, dim, out);
  return hipGetLastError();
}
hipError_t launch_rope_table(const float* inv_freq, int n, int half, float* out, hipStream_t s) {
  hipLaunchKernelGGL(rope_table_kernel, dim3(grid_1d((int64_t)n * half)), dim3(256), 0, s, inv_freq, n, half, out);
  return hipGetLastError();
}
namespace {
__global__ void set_token_rows_kernel(float* x, const float* t, int S, int64_t seq_stride, int D4) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= S * D4) return;
  const int sq = i / D4, c = i - sq * D4;
  reinterpret_cast<float4*>(x + (int64_t)sq * seq_stride)[c] = reinterpret_cast<const float4*>(t)[c];
}
}  // namespace
hipError_t launch_set_token_rows(float* x, const float* t, int S, int nseq, int D, hipStream_t s) {
  if (D % 4) return hipErrorInvalidValue;
  hipLaunchKernelGGL(set_token_rows_kernel, dim3((S * (D / 4) + 255) / 256), dim3(256), 0, s, x, t, S, (int64_t)nseq * D, D / 4);
  return hipGetLastError();
}
hipError_t launch_split_f16_packed(const float* src, int64_t rows, int K, f16* dst, hipStream_t s) {
  if (K % 32) return hipErrorInvalidValue;
  hipLaunchKernelGGL(split_f16_packed_kernel, dim3(grid_1d(rows * K)), dim3(256), 0, s, src, rows, K, dst);
  return hipGetLastError();
}
namespace {
template <bool SCATTER>
__global__ void move_rows_kernel(const float4* src, const int32_t* rowmap, int64_t rows, int C4, float4* dst) {
  const int64_t total = rows * C4;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = i / C4;
    const int c = (int)(i - r * C4);
    const int64_t o = (int64_t)rowmap[r] * C4 + c;
    if (SCATTER) dst[o] = src[i];
    else dst[i] = src[o];
  }
}
}  // namespace
hipError_t launch_gather_rows(const float* src, const int32_t* rowmap, int64_t rows, int C, float* dst, hipStream_t s) {
  if (C % 4) return hipErrorInvalidValue;
  hipLaunchKernelGGL(move_rows_kernel<false>, dim3(grid_1d(rows * (C / 4))), dim3(256), 0, s, reinterpret_cast<const float4*>(src), rowmap, rows, C / 4,
                     reinterpret_cast<float4*>(dst));
  return hipGetLastError();
}
hipError_t launch_scatter_rows(const float* src, const int32_t* rowmap, int64_t rows, int C, float* dst, hipStream_t s) {
  if (C % 4) return hipErrorInvalidValue;
  hipLaunchKernelGGL(move_rows_kernel<true>, dim3(grid_1d(rows * (C / 4))), dim3(256), 0, s, reinterpret_cast<const float4*>(src), rowmap, rows, C / 4,
                     reinterpret_cast<float4*>(dst));
  return hipGetLastError();
}
hipError_t launch_condition_weight(const float* src, int rows, int K, float* scale, float* alpha, f16* hi, f16* pk, hipStream_t s) {
  if (K % 32) return hipErrorInvalidValue;
  hipLaunchKernelGGL(row_pow2_scale_kernel, dim3((rows + 3) / 4), dim3(256), 0, s, src, rows, K, scale, alpha);
  hipLaunchKernelGGL(split_f16_rows_kernel, dim3(grid_1d((int64_t)rows * K)), dim3(256), 0, s, src, (int64_t)rows, K, scale, hi, pk);
  return hipGetLastError();
}
hipError_t launch_pack_mx_rows(const float* src, int64_t ld, int64_t rows, int K, const float* rowscale, f16* dst, int weight, hipStream_t s) {
  if (K % 32 || ld % 4 || (reinterpret_cast<uintptr_t>(src) & 15) || (reinterpret_cast<uintptr_t>(dst) & 15)) return hipErrorInvalidValue;
  const int64_t units = rows * (K / 16);
  const dim3 grid((unsigned)((units + 255) / 256));
  if (weight) hipLaunchKernelGGL(pack_mx_rows_kernel<true>, grid, dim3(256), 0, s, src, ld, rows, K, rowscale, dst);
  else hipLaunchKernelGGL(pack_mx_rows_kernel<false>, grid, dim3(256), 0, s, src, ld, rows, K, rowscale, dst);
  return hipGetLastError();
}
hipError_t launch_split_f16(const float* src, int64_t n, float prescale, f16* hi, f16* lo, hipStream_t s) {
  hipLaunchKernelGGL(split_f16_kernel, dim3(grid_1d(n)), dim3(256), 0, s, src, n, prescale, hi, lo);
  return hipGetLastError();
}
hipError_t launch_convpos_pack(const float* w, int D, int cpg, int K, float* w32, f16* whi, f16* wlo, hipStream_t s) {
  hipLaunchKernelGGL(convpos_pack_kernel, dim3(grid_1d((int64_t)D * cpg * K)), dim3(256), 0, s, w, D, cpg, K, w32, whi, wlo);
  return hipGetLastError();
}
hipError_t launch_dw_pack(const float* w, int C, float* w7, hipStream_t s) {
  hipLaunchKernelGGL(dw_pack_kernel, dim3(grid_1d(7 * C)), dim3(256), 0, s, w, C, w7);
  return hipGetLastError();
}
hipError_t launch_softmax_rows(float* S, int64_t rows, int ld, int nseq, int heads, const int32_t* kvlen_per_batch, int kv_default,
                               hipStream_t s, const int32_t* kvlen2, int seg2_off, double* stats) {
  hipLaunchKernelGGL(softmax_rows_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, s, S, rows, ld, nseq, heads, kvlen_per_batch,
                     kv_default, kvlen2, seg2_off, stats);
  return hipGetLastError();
}
hipError_t launch_im2col7(const float* mel, int B, int T, int Cin, int channel_major, float* col, int64_t ldc, hipStream_t s) {
  hipLaunchKernelGGL(im2col7_kernel, dim3(grid_1d((int64_t)B * T * ldc)), dim3(256), 0, s, mel, B, T, Cin, channel_major, col, ldc);
  return hipGetLastError();
}

// ---- optional DiT variants (no shipped config enables them; reference dit.py:181-189) ----------------------------------------------
namespace {
// q/k RMSNorm over dim_head (reference modules.py:286-305 with eps 1e-6, applied modules.py:493-496) -> rope on the first pe_heads
// heads (modules.py:498-509) -> SDPA scale on q.  The QKV GEMM left q and k as fp32 rows [BH, n, dh] holding only Wx + b.
// One thread owns 4 consecutive channels of one row; the dh/4 threads of a row sit in one wavefront and reduce with shuffles.
__global__ __launch_bounds__(256) void qk_norm_rope_kernel(float* q32, float* k32, const float* __restrict__ wq, const float* __restrict__ wk,
                                                           const float* __restrict__ rope_cs, int64_t rows, int nseq, int heads, int dh,
                                                           int pe_heads, float qscale, float eps, f16* q16, f16* q16_lo, f16* k16, f16* k16_lo,
                                                           int n1, const float* __restrict__ wq2, const float* __restrict__ wk2) {
  const int tpr = dh >> 2;                                     // threads per row (power of two, <= 64)
  const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int64_t row = t / tpr;
  const bool live = row < rows;
  const int d = (int)(t - row * tpr) << 2;
  const int which = blockIdx.y;                                // 0 = q, 1 = k
  float* src = which ? k32 : q32;
  float4 v = live ? *reinterpret_cast<const float4*>(src + row * dh + d) : make_float4(0.f, 0.f, 0.f, 0.f);
  float ss = v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
  for (int o = tpr >> 1; o > 0; o >>= 1) ss += __shfl_xor(ss, o);
  if (!live) return;
  const float r = rsqrtf(ss / (float)dh + eps);
  const int64_t bh = row / nseq;
  int pos = (int)(row - bh * nseq);
  const int hh = (int)(bh % heads);
  const bool second = n1 > 0 && pos >= n1;  // text-stream token of an MMDiT joint slab: own gains, positions restart at 0
  if (second) pos -= n1;
  const float4 g = *reinterpret_cast<const float4*>((second ? (which ? wk2 : wq2) : (which ? wk : wq)) + d);
  float x[4] = {v.x * r * g.x, v.y * r * g.y, v.z * r * g.z, v.w * r * g.w};
  if (pe_heads < 0 || hh < pe_heads) {
    const float4 cs = *reinterpret_cast<const float4*>(rope_cs + ((int64_t)pos * (dh / 2) + d / 2) * 2);
    const float a0 = x[0] * cs.x - x[1] * cs.y, a1 = x[1] * cs.x + x[0] * cs.y;
    const float a2 = x[2] * cs.z - x[3] * cs.w, a3 = x[3] * cs.z + x[2] * cs.w;
    x[0] = a0; x[1] = a1; x[2] = a2; x[3] = a3;
  }
  if (which == 0) { x[0] *= qscale; x[1] *= qscale; x[2] *= qscale; x[3] *= qscale; }
  f16* o16 = which ? k16 : q16;
  if (o16) {
    f16* o16l = which ? k16_lo : q16_lo;
    f16x4 hv, lv;
#pragma unroll
    for (int e = 0; e < 4; ++e) { f16 h, l; split_f16(x[e], h, l); hv[e] = h; lv[e] = l; }
    *reinterpret_cast<f16x4*>(o16 + row * dh + d) = hv;
    if (o16l) *reinterpret_cast<f16x4*>(o16l + row * dh + d) = lv;
  } else {
    *reinterpret_cast<float4*>(src + row * dh + d) = make_float4(x[0], x[1], x[2], x[3]);
  }
}

// out[s, pos, :] = idx[s % B, pos] >= 0 ? src[s, idx[s % B, pos], :] : 0   (average upsampling of the text tokens, dit.py:55-84)
__global__ void gather_seq_rows_kernel(const float* __restrict__ src, const int32_t* __restrict__ idx, int S, int B, int n, int C4, float* out) {
  const int64_t total = (int64_t)S * n * C4;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t row = i / C4;
    const int c = (int)(i - row * C4);
    const int s = (int)(row / n), pos = (int)(row - (int64_t)s * n);
    const int j = idx[(int64_t)(s % B) * n + pos];
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (j >= 0) v = reinterpret_cast<const float4*>(src + ((int64_t)s * n + j) * C4 * 4)[c];
    reinterpret_cast<float4*>(out)[i] = v;
  }
}

__global__ void add_inplace_kernel(float* x, const float* __restrict__ y, int64_t n4) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
    float4 a = reinterpret_cast<float4*>(x)[i];
    const float4 b = reinterpret_cast<const float4*>(y)[i];
    a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
    reinterpret_cast<float4*>(x)[i] = a;
  }
}
}  // namespace

hipError_t launch_qk_norm_rope(float* q32, float* k32, const float* wq, const float* wk, const float* rope_cs, int64_t rows, int nseq,
                               int heads, int dh, int pe_heads, float qscale, float eps, f16* q16, f16* q16_lo, f16* k16, f16* k16_lo,
                               hipStream_t s, int n1, const float* wq2, const float* wk2) {
  const int tpr = dh / 4;
  if (dh % 4 || tpr < 1 || tpr > 64 || (tpr & (tpr - 1))) return hipErrorInvalidValue;
  const int64_t threads = rows * tpr;
  hipLaunchKernelGGL(qk_norm_rope_kernel, dim3((unsigned)((threads + 255) / 256), 2), dim3(256), 0, s, q32, k32, wq, wk, rope_cs, rows, nseq,
                     heads, dh, pe_heads, qscale, eps, q16, q16_lo, k16, k16_lo, n1, wq2, wk2);
  return hipGetLastError();
}
hipError_t launch_gather_seq_rows(const float* src, const int32_t* idx, int S, int B, int n, int C, float* out, hipStream_t s) {
  if (C % 4) return hipErrorInvalidValue;
  hipLaunchKernelGGL(gather_seq_rows_kernel, dim3(grid_1d((int64_t)S * n * (C / 4))), dim3(256), 0, s, src, idx, S, B, n, C / 4, out);
  return hipGetLastError();
}
hipError_t launch_add_inplace(float* x, const float* y, int64_t n, hipStream_t s) {
  if (n % 4) return hipErrorInvalidValue;
  hipLaunchKernelGGL(add_inplace_kernel, dim3(grid_1d(n / 4)), dim3(256), 0, s, x, y, n / 4);
  return hipGetLastError();
}
