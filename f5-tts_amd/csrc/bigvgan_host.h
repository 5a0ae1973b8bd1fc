// bigvgan_host.h — host-only (no HIP) pieces of the BigVGAN path: the resampling filter and the GEMM weight matrices of the convs.
// Kept free of device code so that the CPU suite can compile them with g++ and check them against tests/bigvgan_model.py
// (tests/c_abi/bigvgan_host_test.cpp) — the only part of bigvgan.cpp whose arithmetic can be verified without a GPU.
#pragma once
#include <algorithm>
#include <cmath>
#include <vector>

// alias_free_activation/torch/filter.py kaiser_sinc_filter1d(cutoff 0.25, half_width 0.3, kernel_size 12) — the one filter both
// resamplers of Activation1d use (up_ratio = down_ratio = 2).  The Kaiser window is torch.kaiser_window(12, periodic=False, beta):
// I0(beta * sqrt(1 - ((2i - 11) / 11)^2)) / I0(beta), evaluated in double.
inline double bv_bessel_i0(double x) {
  double sum = 1.0, term = 1.0;
  const double q = x * x / 4.0;
  for (int k = 1; k < 64; ++k) {
    term *= q / ((double)k * (double)k);
    sum += term;
    if (term < 1e-18 * sum) break;
  }
  return sum;
}
inline void bv_kaiser_sinc_12(float (&f)[12]) {
  const int ks = 12, half = 6;
  const double cutoff = 0.25, half_width = 0.3, pi = 3.14159265358979323846;
  const double delta_f = 4 * half_width;
  const double A = 2.285 * (half - 1) * pi * delta_f + 7.95;
  double beta = 0.0;
  if (A > 50.0) beta = 0.1102 * (A - 8.7);
  else if (A >= 21.0) beta = 0.5842 * std::pow(A - 21.0, 0.4) + 0.07886 * (A - 21.0);
  double w[12], sum = 0.0;
  for (int i = 0; i < ks; ++i) {
    const double r = (2.0 * i - (ks - 1)) / (double)(ks - 1);
    const double win = bv_bessel_i0(beta * std::sqrt(std::max(0.0, 1.0 - r * r))) / bv_bessel_i0(beta);
    const double t = (double)(i - half) + 0.5, x = 2 * cutoff * t;
    const double sinc = x == 0.0 ? 1.0 : std::sin(pi * x) / (pi * x);
    w[i] = 2 * cutoff * win * sinc;
    sum += w[i];
  }
  for (int i = 0; i < ks; ++i) f[i] = (float)(w[i] / sum);
}

// taps of a transposed conv (k, stride u, padding (k-u)/2): output row l*u + r reads input rows l + s for every s with
// 0 <= r + pad - s*u < k for some phase r; they form the contiguous range [shift0, shift0 + ntaps)
inline void bv_convt_taps(int k, int u, int& shift0, int& ntaps) {
  const int pad = (k - u) / 2;
  int lo = 1 << 30, hi = -(1 << 30);
  for (int s = -k; s <= k; ++s)
    for (int r = 0; r < u; ++r) {
      const int j = r + pad - s * u;
      if (j >= 0 && j < k) { lo = std::min(lo, s); hi = std::max(hi, s); }
    }
  shift0 = lo;
  ntaps = hi - lo + 1;
}

// Conv1d weight [Cout, Cin, k] -> GEMM weight [Cout, k * cpad]: column j * cpad + ci
inline void bv_conv_matrix(const float* w, int cout, int cin, int k, int cpad, std::vector<float>& mat) {
  const size_t K = (size_t)k * cpad;
  mat.assign((size_t)cout * K, 0.f);
  for (int co = 0; co < cout; ++co)
    for (int ci = 0; ci < cin; ++ci)
      for (int j = 0; j < k; ++j) mat[(size_t)co * K + (size_t)j * cpad + ci] = w[((size_t)co * cin + ci) * k + j];
}

// ConvTranspose1d weight [Cin, Cout, k], stride u -> GEMM weight [u * Cout, ntaps * cpad]: row r * Cout + co (output phase r),
// column t * cpad + ci holds w[ci, co, r + pad - (shift0 + t) u] where that tap index exists, else 0
inline void bv_convt_matrix(const float* w, int cin, int cout, int k, int u, int cpad, int shift0, int ntaps, std::vector<float>& mat) {
  const int pad = (k - u) / 2;
  const size_t K = (size_t)ntaps * cpad;
  mat.assign((size_t)u * cout * K, 0.f);
  for (int r = 0; r < u; ++r)
    for (int t = 0; t < ntaps; ++t) {
      const int j = r + pad - (shift0 + t) * u;
      if (j < 0 || j >= k) continue;
      for (int co = 0; co < cout; ++co)
        for (int ci = 0; ci < cin; ++ci) mat[((size_t)r * cout + co) * K + (size_t)t * cpad + ci] = w[((size_t)ci * cout + co) * k + j];
    }
}
