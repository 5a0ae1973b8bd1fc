// audio.hip — Vocos mel front-end and iSTFT head as LDS-resident 1024-point FFT kernels.
//   mel   : reference model/modules.py:80-109 (torchaudio MelSpectrogram power=1, center=True, HTK, norm=None) -> log(clamp 1e-5),
//           and the BigVGAN-type variant model/modules.py:35-77 (no centring, 384-sample reflect pad, +1e-9, slaney filterbank)
//   istft : vocos ISTFTHead (exp, clip 1e2, mag*(cos p + i sin p)) + torch.istft(n_fft=1024, hop=256, hann, center=True);
//           head math restated in-repo at runtime/triton_trtllm/scripts/export_vocoder_to_onnx.py:45-59
// One workgroup per frame: the 1024-point radix-2 FFT runs entirely in LDS (8 KiB), 256 threads = 2 butterflies
// per thread per stage.  These kernels are HBM/latency-bound (~10 FLOP/B); the frame data is read once.
#include "kernels.h"

namespace {

constexpr int NFFT = 1024;
constexpr int HOP = 256;
constexpr int NBIN = NFFT / 2 + 1;

__device__ __forceinline__ int bitrev10(int i) { return (int)(__brev((unsigned)i) >> 22); }

// in-place radix-2 DIT over bit-reversed input; SIGN = -1 forward, +1 inverse (unnormalised)
template <int SIGN>
__device__ __forceinline__ void fft1024(float* re, float* im, const float* __restrict__ tw, int tid) {
#pragma unroll 1
  for (int half = 1; half < NFFT; half <<= 1) {
    __syncthreads();
    const int tstride = (NFFT / 2) / half;
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      const int j = tid + r * 256;
      const int k = j & (half - 1);
      const int i0 = ((j - k) << 1) + k;
      const int i1 = i0 + half;
      const float wr = tw[2 * (k * tstride)];
      const float wi = (float)SIGN * tw[2 * (k * tstride) + 1];
      const float xr = re[i1], xi = im[i1];
      const float tr = wr * xr - wi * xi, ti = wr * xi + wi * xr;
      const float ar = re[i0], ai = im[i0];
      re[i0] = ar + tr; im[i0] = ai + ti;
      re[i1] = ar - tr; im[i1] = ai - ti;
    }
  }
  __syncthreads();
}

// pad = samples of reflect padding in front of frame 0: n_fft/2 for the centred Vocos-type STFT, (n_fft-hop)/2 for the BigVGAN-type
// one (reference model/modules.py:59-60); mag_eps = 1e-9 inside the square root for the BigVGAN type (modules.py:74), 0 otherwise.
__global__ __launch_bounds__(256) void mel_kernel(const float* __restrict__ wav, int64_t nsamp, int frames,
                                                   const float* __restrict__ tw, const float* __restrict__ window,
                                                   const float* __restrict__ melfb, int nmel, int frame_major, int pad, float mag_eps,
                                                   float* out) {
  __shared__ float re[NFFT];
  __shared__ float im[NFFT];
  const int tid = threadIdx.x, f = blockIdx.x, b = blockIdx.y;
  const float* w = wav + (int64_t)b * nsamp;
  for (int i = tid; i < NFFT; i += 256) {
    int64_t idx = (int64_t)f * HOP + i - pad;  // pad_mode="reflect"
    if (idx < 0) idx = -idx;
    if (idx >= nsamp) idx = 2 * (nsamp - 1) - idx;
    const int br = bitrev10(i);
    re[br] = w[idx] * window[i];
    im[br] = 0.f;
  }
  fft1024<-1>(re, im, tw, tid);
  // magnitude (power = 1) of bins 0..512 -> re[0..512]
  float mag[3];
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    const int k = tid + r * 256;
    mag[r] = k < NBIN ? sqrtf(re[k] * re[k] + im[k] * im[k] + mag_eps) : 0.f;
  }
  __syncthreads();
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    const int k = tid + r * 256;
    if (k < NBIN) re[k] = mag[r];
  }
  __syncthreads();
  if (tid < nmel) {
    float acc = 0.f;
    for (int k = 0; k < NBIN; ++k) acc += re[k] * melfb[k * nmel + tid];
    const float v = logf(fmaxf(acc, 1e-5f));
    if (frame_major) out[((int64_t)b * frames + f) * nmel + tid] = v;
    else out[((int64_t)b * nmel + tid) * frames + f] = v;
  }
}

// logits row = [log-mag (513) | phase (513) | pad]; frames[b, f, :] = irfft(spectrum) * window
__global__ __launch_bounds__(256) void istft_frames_kernel(const float* __restrict__ logits, int64_t ld, int T,
                                                            const float* __restrict__ tw, const float* __restrict__ window,
                                                            float* frames) {
  __shared__ float re[NFFT];
  __shared__ float im[NFFT];
  const int tid = threadIdx.x, f = blockIdx.x, b = blockIdx.y;
  const float* row = logits + ((int64_t)b * T + f) * ld;
  for (int k = tid; k < NBIN; k += 256) {
    float mag = expf(row[k]);
    mag = fminf(mag, 1e2f);
    const float p = row[NBIN + k];
    float xr = mag * cosf(p), xi = mag * sinf(p);
    if (k == 0 || k == NFFT / 2) xi = 0.f;  // C2R ignores the imaginary part of DC / Nyquist
    const int br = bitrev10(k);
    re[br] = xr; im[br] = xi;
    if (k > 0 && k < NFFT / 2) {
      const int br2 = bitrev10(NFFT - k);
      re[br2] = xr; im[br2] = -xi;
    }
  }
  fft1024<+1>(re, im, tw, tid);
  float* o = frames + ((int64_t)b * T + f) * NFFT;
  for (int i = tid; i < NFFT; i += 256) o[i] = re[i] * (1.0f / NFFT) * window[i];
}

// wav[b, j] = sum_f frames[b, f, j + 512 - 256 f] / sum_f window[j + 512 - 256 f]^2
__global__ void istft_ola_kernel(const float* __restrict__ frames, const float* __restrict__ window, int T, int64_t nout, float* wav) {
  const int b = blockIdx.y;
  for (int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < nout; j += (int64_t)gridDim.x * blockDim.x) {
    const int64_t t = j + NFFT / 2;
    int f_hi = (int)(t / HOP);
    if (f_hi > T - 1) f_hi = T - 1;
    int64_t f_lo64 = (t - (NFFT - 1) + HOP - 1) / HOP;
    int f_lo = f_lo64 < 0 ? 0 : (int)f_lo64;
    float acc = 0.f, env = 0.f;
    for (int f = f_lo; f <= f_hi; ++f) {
      const int i = (int)(t - (int64_t)f * HOP);
      acc += frames[((int64_t)b * T + f) * NFFT + i];
      env += window[i] * window[i];
    }
    wav[(int64_t)b * nout + j] = acc / env;
  }
}

}  // namespace

hipError_t launch_mel(const float* wav, int B, int64_t nsamp, int frames, const float* twiddle, const float* window,
                      const float* melfb, int nmel, int frame_major, int pad, float mag_eps, float* out, hipStream_t s) {
  if (nmel > 256 || nsamp < pad + 1 || frames <= 0) return hipErrorInvalidValue;
  hipLaunchKernelGGL(mel_kernel, dim3(frames, B), dim3(256), 0, s, wav, nsamp, frames, twiddle, window, melfb, nmel, frame_major, pad,
                     mag_eps, out);
  return hipGetLastError();
}
hipError_t launch_istft_frames(const float* logits, int64_t ld, int B, int T, const float* twiddle, const float* window,
                               float* frames, hipStream_t s) {
  hipLaunchKernelGGL(istft_frames_kernel, dim3(T, B), dim3(256), 0, s, logits, ld, T, twiddle, window, frames);
  return hipGetLastError();
}
hipError_t launch_istft_ola(const float* frames, const float* window, int B, int T, float* wav, hipStream_t s) {
  const int64_t nout = (int64_t)HOP * (T - 1);
  if (nout <= 0) return hipErrorInvalidValue;
  int gx = (int)((nout + 255) / 256);
  if (gx > 4096) gx = 4096;
  hipLaunchKernelGGL(istft_ola_kernel, dim3(gx, B), dim3(256), 0, s, frames, window, T, nout, wav);
  return hipGetLastError();
}
