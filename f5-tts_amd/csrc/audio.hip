// audio.hip — Vocos mel front-end and iSTFT head: real 1024-point transforms as HALF-SIZE complex FFTs that live in LDS.
//   mel   : reference model/modules.py:80-109 (torchaudio MelSpectrogram power=1, center=True, HTK, norm=None) -> log(clamp 1e-5),
//           and the BigVGAN-type variant model/modules.py:35-77 (no centring, 384-sample reflect pad, +1e-9, slaney filterbank)
//   istft : vocos ISTFTHead (exp, clip 1e2, mag*(cos p + i sin p)) + torch.istft(n_fft=1024, hop=256, hann, center=True);
//           head math restated in-repo at runtime/triton_trtllm/scripts/export_vocoder_to_onnx.py:45-59
//
// Round 6 (mel ran at 1-4 % and the inverse transform at 8-14 % of HBM peak).  What the round-1 kernels paid for:
// one workgroup per frame running a 1024-point complex radix-2 transform on real data (10 barrier-separated stages, twiddles from global
// memory), a DENSE 513 x 100 filterbank product per frame (205 KB of L2 reads for 4 KB of samples: the triangles are 98 % zeros), and for
// the inverse a frames buffer written and read back (8 KB per frame next to 4 KB of logits) by a second overlap-add kernel.  Now:
//   * a real frame of 1024 samples is ONE 512-point complex transform (z[n] = x[2n] + i x[2n+1], split / merged with the table's own
//     twiddles) = a radix-2 stage + four radix-4 stages of 128 butterflies, twiddle table and data in LDS: a workgroup of 256 threads
//     carries TWO frames side by side (128 threads each).  Frames are never paired into one complex transform: a quiet frame would inherit
//     the rounding noise of a loud neighbour, and the mel of silence is a logarithm;
//   * the filterbank product walks each mel bin's own support [lo, hi) — api.cpp builds the ranges and a compact [mel][support] copy of the
//     weights next to the table: same terms, same order as the dense sum;
//   * the inverse transform and the overlap-add are one kernel: a workgroup owns G consecutive output hops, transforms the G + 3 frames
//     that touch them (two at a time; 3 of them recomputed by the neighbour: 23 % at G = 13), adds them in ascending frame order into an LDS
//     strip (the summation order of the old two-kernel form) and writes finished samples — logits read once (x 1.23), samples written
//     once, no intermediate in HBM.
#include "kernels.h"

namespace {

constexpr int NFFT = 1024;
constexpr int HOP = 256;
constexpr int NBIN = NFFT / 2 + 1;
constexpr int NH = NFFT / 2;  // points of the complex transform

// LDS index of complex element i: one element of padding per 16 (128 bytes = every bank once), so the strided accesses of the later
// stages do not pile onto a few banks
__device__ __forceinline__ int zpad(int i) { return i + (i >> 4); }
constexpr int ZLEN = NH + (NH >> 4);

// Input position of element n of a 2 x 4 x 4 x 4 x 4 decimation-in-time transform: n = r0 + 4 r1 + 16 r2 + 64 r3 + 256 r4 (r0..r3 base-4
// digits, r4 one bit) sits at r0 128 + r1 32 + r2 8 + r3 2 + r4
__device__ __forceinline__ int perm512(int n) {
  return ((n & 3) << 7) | (((n >> 2) & 3) << 5) | (((n >> 4) & 3) << 3) | (((n >> 6) & 3) << 1) | (n >> 8);
}

__device__ __forceinline__ float2 cmul(float2 a, float2 b) { return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }

// In-place 512-point complex transform of z (perm512 order in, natural order out) by the 128 threads t = 0..127 of one half of the
// workgroup; both halves call it together (the barriers are the workgroup's).  tw[e] = (cos, sin)(2 pi e / 1024), e < 512, in LDS.
// SIGN = -1 forward (W = e^-i), +1 inverse, unnormalised.
template <int SIGN>
__device__ __forceinline__ void fft512(float2* z, const float2* tw, int t) {
  __syncthreads();
#pragma unroll
  for (int r = 0; r < 2; ++r) {  // radix 2 on neighbours, twiddle 1
    const int j = 2 * (t + 128 * r);
    const float2 a = z[zpad(j)], b = z[zpad(j + 1)];
    z[zpad(j)] = make_float2(a.x + b.x, a.y + b.y);
    z[zpad(j + 1)] = make_float2(a.x - b.x, a.y - b.y);
  }
#pragma unroll
  for (int lq = 1; lq <= 7; lq += 2) {  // radix 4, quarter sizes 2, 8, 32, 128: y_p = sum_r (SIGN i)^(r p) W_4q^(r k) x_r
    __syncthreads();
    const int q = 1 << lq, k = t & (q - 1), i0 = ((t - k) << 2) + k;
    const int e1 = k << (8 - lq);  // W_4q^k = W_1024^e1, e1 < 256
    float2 w1 = tw[e1], w2 = tw[2 * e1], w3;
    const int e3 = 3 * e1;         // < 768: W^(e + 512) = -W^e
    if (e3 < NH) w3 = tw[e3];
    else { w3 = tw[e3 - NH]; w3.x = -w3.x; w3.y = -w3.y; }
    w1.y *= (float)SIGN; w2.y *= (float)SIGN; w3.y *= (float)SIGN;
    const float2 x0 = z[zpad(i0)], x1 = cmul(w1, z[zpad(i0 + q)]), x2 = cmul(w2, z[zpad(i0 + 2 * q)]), x3 = cmul(w3, z[zpad(i0 + 3 * q)]);
    const float2 a = make_float2(x0.x + x2.x, x0.y + x2.y), b = make_float2(x0.x - x2.x, x0.y - x2.y);
    const float2 c = make_float2(x1.x + x3.x, x1.y + x3.y), d = make_float2(x1.x - x3.x, x1.y - x3.y);
    z[zpad(i0)] = make_float2(a.x + c.x, a.y + c.y);
    z[zpad(i0 + 2 * q)] = make_float2(a.x - c.x, a.y - c.y);
    z[zpad(i0 + q)] = make_float2(b.x - (float)SIGN * d.y, b.y + (float)SIGN * d.x);      // b + SIGN i d
    z[zpad(i0 + 3 * q)] = make_float2(b.x + (float)SIGN * d.y, b.y - (float)SIGN * d.x);  // b - SIGN i d
  }
  __syncthreads();
}

// pad = samples of reflect padding in front of frame 0: n_fft/2 for the centred Vocos-type STFT, (n_fft-hop)/2 for the BigVGAN-type
// one (reference model/modules.py:59-60); mag_eps = 1e-9 inside the square root for the BigVGAN type (modules.py:74), 0 otherwise.
// Workgroup = frames 2 blockIdx.x (threads 0-127) and 2 blockIdx.x + 1 (threads 128-255) of utterance blockIdx.y.
__global__ __launch_bounds__(256) void mel_kernel(const float* __restrict__ wav, int64_t nsamp, int frames, const float2* __restrict__ tw_g,
                                                   const float* __restrict__ window, const float* __restrict__ melfb,
                                                   const int2* __restrict__ melrange, int melw_ld, int nmel, int frame_major, int pad, float mag_eps,
                                                   float* out) {
  __shared__ float2 z[2][ZLEN];
  __shared__ float2 tw[NH];
  __shared__ float mag[2][NBIN + 7];
  const int tid = threadIdx.x, half = tid >> 7, t = tid & 127, b = blockIdx.y;
  const int f = 2 * blockIdx.x + half;
  const bool live = f < frames;
  for (int i = tid; i < NH; i += 256) tw[i] = tw_g[i];
  const float* w = wav + (int64_t)b * nsamp;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int n = t + 128 * j;
    float2 v = make_float2(0.f, 0.f);
    if (live) {
      int64_t i0 = (int64_t)f * HOP + 2 * n - pad, i1 = i0 + 1;  // pad_mode="reflect"
      if (i0 < 0) i0 = -i0;
      if (i0 >= nsamp) i0 = 2 * (nsamp - 1) - i0;
      if (i1 < 0) i1 = -i1;
      if (i1 >= nsamp) i1 = 2 * (nsamp - 1) - i1;
      const float2 wn = *reinterpret_cast<const float2*>(window + 2 * n);
      v = make_float2(w[i0] * wn.x, w[i1] * wn.y);
    }
    z[half][zpad(perm512(n))] = v;
  }
  fft512<-1>(z[half], tw, t);
  // X[k] = E[k] + W^k O[k] with E = (Z[k] + conj Z[512-k]) / 2 (the even samples' spectrum), O = (Z[k] - conj Z[512-k]) / 2i (the odd ones');
  // X[512] = E[0] - O[0].  Magnitude (power = 1) of bins 0..512.
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int k = t + 128 * j;
    const float2 zk = z[half][zpad(k)], zn = z[half][zpad((NH - k) & (NH - 1))];
    const float2 e = make_float2(0.5f * (zk.x + zn.x), 0.5f * (zk.y - zn.y)), o = make_float2(0.5f * (zk.y + zn.y), 0.5f * (zn.x - zk.x));
    const float2 wo = cmul(make_float2(tw[k].x, -tw[k].y), o);
    const float xr = e.x + wo.x, xi = e.y + wo.y;
    mag[half][k] = sqrtf(xr * xr + xi * xi + mag_eps);
    if (k == 0) {
      const float x512 = zk.x - zk.y;  // E[0], O[0] are real
      mag[half][NH] = sqrtf(x512 * x512 + mag_eps);
    }
  }
  __syncthreads();
  if (!live) return;
  for (int m = t; m < nmel; m += 128) {
    const int2 r = melrange[m];
    float acc = 0.f;
    // the triangle's weights are one contiguous run of melw (api.cpp: [nmel][melw_ld], bin r.x first, zero-padded to a multiple of 4): four per
    // load, every load of the run independent of the sum — the [bin][mel] table cost one dependent-latency global load per bin of the support
    const float4* w4 = reinterpret_cast<const float4*>(melfb + (size_t)m * melw_ld);
    for (int k = r.x, j = 0; k < r.y; k += 4, ++j) {
      const float4 w = w4[j];
      acc += mag[half][k] * w.x;
      if (k + 1 < r.y) acc += mag[half][k + 1] * w.y;
      if (k + 2 < r.y) acc += mag[half][k + 2] * w.z;
      if (k + 3 < r.y) acc += mag[half][k + 3] * w.w;
    }
    const float v = logf(fmaxf(acc, 1e-5f));
    if (frame_major) out[((int64_t)b * frames + f) * nmel + m] = v;
    else out[((int64_t)b * nmel + m) * frames + f] = v;
  }
}

// logits row = [log-mag (513) | phase (513) | pad].  One workgroup = NFR - 3 output hops of utterance blockIdx.y:
//   wav[b, j] = sum_f frame_f[j + 512 - 256 f] / sum_f window[j + 512 - 256 f]^2,  frame_f = irfft(spectrum_f) * window  (f ascending)
// hop h (samples 256 h .. + 255) is touched by the frames h - 1 .. h + 2.
template <int NFR>
__global__ __launch_bounds__(256) void istft_fused_kernel(const float* __restrict__ logits, int64_t ld, int T, const float2* __restrict__ tw_g,
                                                           const float* __restrict__ window_g, int64_t nout, float* wav) {
  static_assert(NFR % 2 == 0 && NFR >= 6, "frames are transformed two at a time");
  constexpr int G = NFR - 3, NSTRIP = (NFR + 3) * HOP;  // strip = the samples frames F0 .. F0 + NFR - 1 cover: (NFR - 1) hops + 1024
  __shared__ float2 z[2][ZLEN];
  __shared__ float2 tw[NH];
  __shared__ float win[NFFT];
  __shared__ float strip[NSTRIP];
  const int tid = threadIdx.x, half = tid >> 7, t = tid & 127, b = blockIdx.y;
  const int H0 = blockIdx.x * G, F0 = H0 - 1;
  for (int i = tid; i < NH; i += 256) tw[i] = tw_g[i];
  for (int i = tid; i < NFFT; i += 256) win[i] = window_g[i];
  for (int i = tid; i < NSTRIP; i += 256) strip[i] = 0.f;
  __syncthreads();
#pragma unroll 1
  for (int it = 0; it < NFR / 2; ++it) {
    const int f = F0 + 2 * it + half;
    const bool live = f >= 0 && f < T;
    const float* row = logits + ((int64_t)b * T + (live ? f : 0)) * ld;
    // spectrum bin k: mag (cos p, sin p), mag = min(exp(log-mag), 1e2); C2R ignores the imaginary part of DC / Nyquist
    auto spec = [&](int k) {
      float m = expf(row[k]);
      m = fminf(m, 1e2f);
      float s, c;
      sincosf(row[NBIN + k], &s, &c);
      return make_float2(m * c, (k == 0 || k == NH) ? 0.f : m * s);
    };
    // Z = E + i O with E = (X[k] + conj X[512-k]) / 2, O = W^-k (X[k] - conj X[512-k]) / 2; Z[512-k] = conj E + i conj O
    for (int k = t; k <= NH / 2; k += 128) {
      float2 zk = make_float2(0.f, 0.f), zn = zk;
      if (live) {
        const float2 x1 = spec(k), x2 = spec(NH - k);
        const float2 e = make_float2(0.5f * (x1.x + x2.x), 0.5f * (x1.y - x2.y)), d = make_float2(0.5f * (x1.x - x2.x), 0.5f * (x1.y + x2.y));
        const float2 o = cmul(tw[k], d);  // W^-k = (cos, +sin)
        zk = make_float2(e.x - o.y, e.y + o.x);
        zn = make_float2(e.x + o.y, o.x - e.y);
      }
      z[half][zpad(perm512(k))] = zk;
      if (k != 0 && k != NH / 2) z[half][zpad(perm512(NH - k))] = zn;
    }
    fft512<+1>(z[half], tw, t);
    // overlap-add in frame order: x[2n] = Re z[n] / 512, x[2n + 1] = Im z[n] / 512, times the window
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
      const int fc = F0 + 2 * it + hh;
      if (fc >= 0 && fc < T) {  // (uniform over the workgroup)
#pragma unroll
        for (int r = 0; r < 2; ++r) {
          const int n = tid + 256 * r;
          const float2 v = z[hh][zpad(n)];
          float* dst = strip + (2 * it + hh) * HOP + 2 * n;
          dst[0] += v.x * (1.0f / NH) * win[2 * n];
          dst[1] += v.y * (1.0f / NH) * win[2 * n + 1];
        }
      }
      __syncthreads();
    }
  }
  for (int idx = tid; idx < G * HOP; idx += 256) {
    const int64_t j = (int64_t)H0 * HOP + idx;
    if (j >= nout) break;
    const int h = H0 + idx / HOP;
    const int tpos = (idx & (HOP - 1)) + NFFT / 2;  // j + 512 relative to sample 256 h of the padded signal
    float env = 0.f;
    const int flo = h - 1 < 0 ? 0 : h - 1, fhi = h + 2 > T - 1 ? T - 1 : h + 2;
    for (int f = flo; f <= fhi; ++f) {
      const float wv = win[tpos - (f - h) * HOP];
      env += wv * wv;
    }
    wav[(int64_t)b * nout + j] = strip[idx + 3 * HOP] / env;
  }
}

}  // namespace

hipError_t launch_mel(const float* wav, int B, int64_t nsamp, int frames, const float* twiddle, const float* window,
                      const float* melfb, const int* melrange, int melw_ld, int nmel, int frame_major, int pad, float mag_eps, float* out, hipStream_t s) {
  if (nmel > 256 || nsamp < pad + 1 || frames <= 0 || !melrange || melw_ld <= 0 || (melw_ld & 3)) return hipErrorInvalidValue;
  hipLaunchKernelGGL(mel_kernel, dim3((frames + 1) / 2, B), dim3(256), 0, s, wav, nsamp, frames, reinterpret_cast<const float2*>(twiddle), window, melfb,
                     reinterpret_cast<const int2*>(melrange), melw_ld, nmel, frame_major, pad, mag_eps, out);
  return hipGetLastError();
}
// inverse STFT of B x T logits rows into B x 256 (T - 1) samples: workgroups of 13 hops (16 frames each) once they fill the chip twice,
// of 5 hops (8 frames) below that
hipError_t launch_istft(const float* logits, int64_t ld, int B, int T, const float* twiddle, const float* window, float* wav, hipStream_t s) {
  const int64_t nout = (int64_t)HOP * (T - 1);
  if (nout <= 0 || B <= 0) return hipErrorInvalidValue;
  const int hops = T - 1;
  const float2* tw = reinterpret_cast<const float2*>(twiddle);
  if ((int64_t)B * ((hops + 12) / 13) >= 512) {
    hipLaunchKernelGGL(istft_fused_kernel<16>, dim3((hops + 12) / 13, B), dim3(256), 0, s, logits, ld, T, tw, window, nout, wav);
  } else {
    hipLaunchKernelGGL(istft_fused_kernel<8>, dim3((hops + 4) / 5, B), dim3(256), 0, s, logits, ld, T, tw, window, nout, wav);
  }
  return hipGetLastError();
}
