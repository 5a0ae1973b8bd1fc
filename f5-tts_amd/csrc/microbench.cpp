// microbench.cpp — kernel-level measurement entry points (include/f5hip_bench.h, libf5hip_bench.so — NOT part of libf5hip.so): time ONE
// kernel of the hot path on synthetic operands with HIP events on the launch stream, check operand formats and tiles against each other.
// Used by tools/kernel_bench.py and the tile / format tests; reaches the engine's internal launchers through libf5hip.so; no model state.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>

#include "../../include/f5hip_bench.h"
#include "engine.h"

namespace {

__global__ void fill_f32_kernel(float* x, int64_t n, uint32_t seed, float scale) {
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    uint32_t h = (uint32_t)i * 2654435761u ^ seed;
    h ^= h >> 16; h *= 0x7feb352du; h ^= h >> 15; h *= 0x846ca68bu; h ^= h >> 16;
    x[i] = scale * ((float)(h & 0xffffff) * (1.0f / 8388608.0f) - 1.0f);  // uniform [-scale, scale): full-range random operands
  }
}

struct Tmp {
  std::vector<void*> ptrs;
  ~Tmp() { for (void* p : ptrs) (void)hipFree(p); }
  template <typename T>
  T* get(size_t n) {
    void* p = nullptr;
    if (hipMalloc(&p, n * sizeof(T) + 256) != hipSuccess) return nullptr;
    ptrs.push_back(p);
    return reinterpret_cast<T*>(p);
  }
};

hipError_t fill(float* x, int64_t n, uint32_t seed, float scale, hipStream_t s) {
  hipLaunchKernelGGL(fill_f32_kernel, dim3(1024), dim3(256), 0, s, x, n, seed, scale);
  return hipGetLastError();
}

template <typename F>
int time_it(F&& launch, int iters, hipStream_t s, double* avg_ms) {
  hipEvent_t e0, e1;
  if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) return F5HIP_ERR_HIP;
  for (int i = 0; i < 2; ++i)
    if (launch() != hipSuccess) return F5HIP_ERR_HIP;
  (void)hipEventRecord(e0, s);
  for (int i = 0; i < iters; ++i)
    if (launch() != hipSuccess) return F5HIP_ERR_HIP;
  (void)hipEventRecord(e1, s);
  if (hipEventSynchronize(e1) != hipSuccess) return F5HIP_ERR_HIP;
  float ms = 0.f;
  (void)hipEventElapsedTime(&ms, e0, e1);
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  *avg_ms = (double)ms / iters;
  return F5HIP_OK;
}

}  // namespace

extern "C" {

// Host-side reading of MX operand rows (common.h "fp16 + MX-fp6 corrections"): rows x K values as [K / 32] lines of 128 bytes, against the
// fp32 values they should encode.  Reports the largest coarse / remainder error in units of their rounding steps' scale (S and S 2^-11: the
// format guarantees <= 0.25, 0.5 where the top binade saturates), the count outside the given bounds and the hi halves that are not
// fp16(ref).  `weight`: the W-side element order (remainder first) and scale byte.
void mx_lines_check(const f16* lines, const float* ref, int64_t rows, int K, bool weight, double bound_c, double bound_l, double* worst_c, double* worst_l,
                    size_t* bad, size_t* hi_diff, double* maxv) {
  auto fp6 = [](unsigned c) { const unsigned ex = (c >> 3) & 3, m = c & 7; const float v = ex == 0 ? m / 8.0f : ldexpf(1.0f + m / 8.0f, (int)ex - 1); return (c & 32) ? -v : v; };
  for (int64_t r = 0; r < rows; ++r)
    for (int blk = 0; blk < K / 32; ++blk) {
      const f16* q = lines + (r * (K / 32) + blk) * 64;
      const float* x = ref + r * K + blk * 32;
      for (int k = 0; k < 32; ++k) { const f16 h = (f16)x[k]; *hi_diff += memcmp(&h, &q[k], 2) != 0; }
      for (int h = 0; h < 2; ++h) {
        uint32_t w[8];
        memcpy(w, reinterpret_cast<const char*>(q) + 64 + 32 * h, 32);
        const int b = (int)(w[6] & 255u);
        const double S = ldexp(1.0, b + (weight ? 0 : 11) - 127);  // block scale of the coarse values; the remainders carry S 2^-11
        for (int tt = 0; tt < 16; ++tt) {
          const int k = 8 * (tt / 4) + 4 * h + (tt % 4);
          auto code = [&](int el) { const int bit = 6 * el; const uint64_t two = w[bit >> 5] | ((uint64_t)w[(bit >> 5) + 1] << 32); return (unsigned)(two >> (bit & 31)) & 63u; };
          const double lo = (double)x[k] - (double)(float)q[k];  // what the line's remainder should say (against the hi half the line holds)
          const double c = fp6(code(2 * tt + (weight ? 1 : 0))) * S, l = fp6(code(2 * tt + (weight ? 0 : 1))) * S / 2048.0;
          const double ec = fabs(c - x[k]) / S, el = fabs(l - lo) / (S / 2048.0);
          *worst_c = std::max(*worst_c, ec);
          *worst_l = std::max(*worst_l, el);
          *bad += ec > bound_c || el > bound_l || w[7] != 0u || (w[6] >> 8) != 0u;
          *maxv = std::max(*maxv, fabs((double)x[k]));
        }
      }
    }
}

// out[M,N] = gelu_tanh(A[M,K] . W[N,K]^T + bias) written as fp32 (fp32 mode) or f16 hi(/lo) planes — the FF1 GEMM of a DiT block.
int f5hip_bench_gemm(f5hip_ctx* ctx, int precision, int variant, int epilogue, int M, int N, int K, int iters, double* avg_ms) {
  if (!ctx || !avg_ms || M <= 0 || N <= 0 || K <= 0 || iters <= 0 || (K % 8) || (N % 4)) return F5HIP_ERR_INVALID;
  std::lock_guard<std::mutex> lk(ctx->mu);
  if (hipSetDevice(ctx->device) != hipSuccess) return F5HIP_ERR_HIP;
  if (init_gemm_kernels() != hipSuccess) return F5HIP_ERR_HIP;
  hipStream_t s = nullptr;
  Tmp t;
  const int op = precision == F5HIP_PREC_FP32 ? OP_F32 : precision == F5HIP_PREC_FP16 ? OP_F16 : precision == F5HIP_PREC_FP16M ? OP_F16M : OP_F16X3;
  const bool mx = op == OP_F16M;  // MX lines: the pipelined tiles only (variant < 0 or >= 50)
  float* a32 = t.get<float>((size_t)M * K);
  float* w32 = t.get<float>((size_t)N * K);
  float* bias = t.get<float>(N);
  float* o32 = t.get<float>((size_t)M * N);
  float* res = t.get<float>((size_t)M * N);
  const bool x3 = op == OP_F16X3 || mx;
  if (x3 && (K % 32 || N % 32)) return F5HIP_ERR_INVALID;
  const size_t pl = x3 ? 2 : 1;  // packed hi/lo rows are twice as long
  f16 *ah = t.get<f16>((size_t)M * K * pl), *wh = t.get<f16>((size_t)N * K * pl), *oh = t.get<f16>((size_t)M * N * pl);
  if (!res || !a32 || !w32 || !bias || !o32 || !ah || !wh || !oh) return F5HIP_ERR_HIP;
  if (fill(a32, (int64_t)M * K, 1u, 1.0f, s) != hipSuccess || fill(w32, (int64_t)N * K, 2u, 0.05f, s) != hipSuccess ||
      fill(bias, N, 3u, 0.02f, s) != hipSuccess)
    return F5HIP_ERR_HIP;
  if (mx) {
    if (launch_pack_mx_rows(a32, K, M, K, nullptr, ah, 0, s) != hipSuccess || launch_pack_mx_rows(w32, K, N, K, nullptr, wh, 1, s) != hipSuccess) return F5HIP_ERR_HIP;
  } else if (x3) {
    if (launch_split_f16_packed(a32, M, K, ah, s) != hipSuccess || launch_split_f16_packed(w32, N, K, wh, s) != hipSuccess) return F5HIP_ERR_HIP;
  } else if (launch_split_f16(a32, (int64_t)M * K, 1.0f, ah, nullptr, s) != hipSuccess ||
             launch_split_f16(w32, (int64_t)N * K, 1.0f, wh, nullptr, s) != hipSuccess) {
    return F5HIP_ERR_HIP;
  }
  GemmCore g{};
  g.A = op == OP_F32 ? (const void*)a32 : (const void*)ah;
  g.W = op == OP_F32 ? (const void*)w32 : (const void*)wh;
  g.lda = (int64_t)K * pl; g.ldw = (int64_t)K * pl; g.M = M; g.N = N; g.K = K; g.a_rows = M; g.w_rows = N;
  EpiStore e{};
  e.alpha = 1.f; e.bias = bias; e.ldo = N; e.ldres = N;
  if (epilogue == 2) {  // out-proj / FF2: x += gate * (acc + bias), fp32 residual stream
    if (fill(res, (int64_t)M * N, 4u, 1.0f, s) != hipSuccess) return F5HIP_ERR_HIP;
    e.colscale = bias; e.res = res; e.out32 = res;
  } else {               // 0: bias only, 1: FF1 (tanh-GELU); operand rows of the next GEMM
    e.act = epilogue == 1 ? ACT_GELU_TANH : ACT_NONE;
    if (op == OP_F32) e.out32 = o32;
    else { e.out16 = oh; if (x3) { e.out16_lo = oh + 32; e.pk16 = mx ? 2 : 1; e.ldo16 = 2 * (int64_t)N; } }
  }
  if (getenv("KB_CHECK") && mx) {  // MX lines against the three-term product of the same fp32 operands (generic kernel, epilogue 2: fp32 results)
    if (epilogue != 2) {
      // the MX rows this launch writes (the next GEMM's operand) against the hi | lo rows of the fp16x3 generic kernel on the same fp32
      // operands, decoded on the host: hi halves, then per half-line the coarse values and the remainders within their rounding steps
      f16 *a3 = t.get<f16>((size_t)M * K * 2), *w3 = t.get<f16>((size_t)N * K * 2), *o3 = t.get<f16>((size_t)M * N * 2);
      if (!a3 || !w3 || !o3 || launch_split_f16_packed(a32, M, K, a3, s) != hipSuccess || launch_split_f16_packed(w32, N, K, w3, s) != hipSuccess) return F5HIP_ERR_HIP;
      GemmCore g3 = g;
      g3.A = a3; g3.W = w3; g3.lda = g3.ldw = 2 * (int64_t)K;
      EpiStore e3 = e;
      e3.out16 = o3; e3.out16_lo = o3 + 32; e3.pk16 = 1;
      const size_t nh = (size_t)M * N * 2;
      std::vector<f16> ref(nh), got(nh);
      if (hipMemsetAsync(oh, 0, nh * 2, s) != hipSuccess || launch_gemm_store_variant(OP_F16X3, g3, e3, 1, 1, s) != hipSuccess ||
          launch_gemm_store_variant(op, g, e, 1, variant, s) != hipSuccess || hipMemcpyAsync(ref.data(), o3, nh * 2, hipMemcpyDeviceToHost, s) != hipSuccess ||
          hipMemcpyAsync(got.data(), oh, nh * 2, hipMemcpyDeviceToHost, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess)
        return F5HIP_ERR_HIP;
      std::vector<float> xref(nh / 2);
      for (size_t ln = 0; ln < nh / 64; ++ln)
        for (int k = 0; k < 32; ++k) xref[ln * 32 + k] = (float)ref[ln * 64 + k] + (float)ref[ln * 64 + 32 + k];
      size_t hi_diff = 0, bad = 0;
      const size_t lines = nh / 64;
      double worst_c = 0, worst_l = 0, maxv = 0;
      // (the two GEMMs' own results differ by a few 1e-5 relative — up to a remainder step: bounds 0.51 / 1.5 still catch any layout slip,
      // the exact bounds are f5hip_bench_mx_pack's)
      mx_lines_check(got.data(), xref.data(), (int64_t)M, N, false, 0.51, 1.5, &worst_c, &worst_l, &bad, &hi_diff, &maxv);
      fprintf(stderr, "KB_CHECK fp16m variant %d epi %d: %zu lines, %zu hi halves differ from the fp16x3 rows, coarse / remainder errors %.3f / %.3f block steps, %zu out of bounds, max |value| %.3g\n",
              variant, epilogue, lines, hi_diff, worst_c, worst_l, bad, maxv);
    } else {
      f16 *a3 = t.get<f16>((size_t)M * K * 2), *w3 = t.get<f16>((size_t)N * K * 2);
      if (!a3 || !w3 || launch_split_f16_packed(a32, M, K, a3, s) != hipSuccess || launch_split_f16_packed(w32, N, K, w3, s) != hipSuccess) return F5HIP_ERR_HIP;
      GemmCore g3 = g;
      g3.A = a3; g3.W = w3; g3.lda = g3.ldw = 2 * (int64_t)K;
      std::vector<float> ref((size_t)M * N), got((size_t)M * N);
      for (int pass = 0; pass < 2; ++pass) {
        if (fill(res, (int64_t)M * N, 4u, 1.0f, s) != hipSuccess) return F5HIP_ERR_HIP;
        if ((pass == 0 ? launch_gemm_store_variant(OP_F16X3, g3, e, 1, 1, s) : launch_gemm_store_variant(op, g, e, 1, variant, s)) != hipSuccess ||
            hipMemcpyAsync(pass == 0 ? ref.data() : got.data(), res, (size_t)M * N * 4, hipMemcpyDeviceToHost, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess)
          return F5HIP_ERR_HIP;
      }
      double maxd = 0, maxv = 0, sumd = 0;
      for (size_t i = 0; i < ref.size(); ++i) { const double d = fabs((double)ref[i] - got[i]); maxd = std::max(maxd, d); sumd += d; maxv = std::max(maxv, (double)fabsf(ref[i])); }
      fprintf(stderr, "KB_CHECK fp16m variant %d: max |diff| %.3g mean %.3g of max |value| %.3g against the fp16x3 product\n", variant, maxd, sumd / ref.size(), maxv);
      if (fill(res, (int64_t)M * N, 4u, 1.0f, s) != hipSuccess) return F5HIP_ERR_HIP;
    }
  } else if (getenv("KB_CHECK")) {  // compare this variant's output with the generic 128x64 tiling (variant 1): bytes and numeric distance
    const bool r32 = epilogue == 2 || op == OP_F32;
    const size_t nb = r32 ? (size_t)M * N * 4 : (size_t)M * N * pl * 2;
    void* outp = epilogue == 2 ? (void*)res : op == OP_F32 ? (void*)o32 : (void*)oh;
    std::vector<unsigned char> ref(nb), got(nb);
    for (int rep = 0; rep < 3; ++rep) {
      for (int pass = 0; pass < 2; ++pass) {  // the same inputs for both kernels (epilogue 2 updates its residual in place)
        if (epilogue == 2 ? fill(res, (int64_t)M * N, 4u, 1.0f, s) != hipSuccess : hipMemsetAsync(outp, 0, nb, s) != hipSuccess) return F5HIP_ERR_HIP;
        if (launch_gemm_store_variant(op, g, e, 1, pass == 0 ? 1 : variant, s) != hipSuccess ||
            hipMemcpyAsync(pass == 0 ? ref.data() : got.data(), outp, nb, hipMemcpyDeviceToHost, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess)
          return F5HIP_ERR_HIP;
      }
      size_t bad = 0, first = 0;
      for (size_t i = 0; i < nb; ++i)
        if (ref[i] != got[i]) { if (!bad) first = i; ++bad; }
      double maxd = 0.0, maxv = 0.0;
      if (r32) {
        const float* a = reinterpret_cast<const float*>(ref.data());
        const float* b = reinterpret_cast<const float*>(got.data());
        for (size_t i = 0; i < nb / 4; ++i) { maxd = std::max(maxd, (double)fabsf(a[i] - b[i])); maxv = std::max(maxv, (double)fabsf(a[i])); }
      } else {
        const f16* a = reinterpret_cast<const f16*>(ref.data());
        const f16* b = reinterpret_cast<const f16*>(got.data());
        if (x3) {  // packed rows [32 hi | 32 lo] per 32 channels: the distance of the VALUES hi + lo (a flipped last bit of hi is absorbed by lo)
          for (size_t blk = 0; blk < nb / 2 / 64; ++blk)
            for (int c = 0; c < 32; ++c) {
              const double va = (double)(float)a[blk * 64 + c] + (double)(float)a[blk * 64 + 32 + c], vb = (double)(float)b[blk * 64 + c] + (double)(float)b[blk * 64 + 32 + c];
              maxd = std::max(maxd, fabs(va - vb));
              maxv = std::max(maxv, fabs(va));
            }
        } else {
          for (size_t i = 0; i < nb / 2; ++i) { maxd = std::max(maxd, (double)fabsf((float)a[i] - (float)b[i])); maxv = std::max(maxv, (double)fabsf((float)a[i])); }
        }
      }
      fprintf(stderr, "KB_CHECK variant %d epi %d rep %d: %zu of %zu bytes differ from variant 1 (first at %zu), max |diff| %.3g of max |value| %.3g\n",
              variant, epilogue, rep, bad, nb, first, maxd, maxv);
    }
    if (epilogue == 2 && fill(res, (int64_t)M * N, 4u, 1.0f, s) != hipSuccess) return F5HIP_ERR_HIP;
  }
  const int rc = time_it([&] { return launch_gemm_store_variant(op, g, e, 1, variant, s); }, iters, s, avg_ms);
  return rc;
}

// The MX operand format end to end on the device (tests): pack_mx_rows_kernel (both operand sides) and the LayerNorm producer against the
// fp32 values they encode, with the format's own bounds.  out[0..2] = worst coarse error, worst remainder error (block steps), values
// out of bounds + hi halves that are not fp16(value), for: activations, weights, LayerNorm rows (the last against the fp32 LayerNorm).
int f5hip_bench_mx_pack(f5hip_ctx* ctx, int rows, int K, double* out9) {
  if (!ctx || !out9 || rows <= 0 || K <= 0 || K % 32 || K > 2048) return F5HIP_ERR_INVALID;
  std::lock_guard<std::mutex> lk(ctx->mu);
  if (hipSetDevice(ctx->device) != hipSuccess) return F5HIP_ERR_HIP;
  hipStream_t s = nullptr;
  Tmp t;
  float *x = t.get<float>((size_t)rows * K), *y = t.get<float>((size_t)rows * K), *sc = t.get<float>(K), *sh = t.get<float>(K);
  f16* mxr = t.get<f16>((size_t)rows * K * 2);
  if (!x || !y || !sc || !sh || !mxr) return F5HIP_ERR_HIP;
  if (fill(x, (int64_t)rows * K, 21u, 3.0f, s) != hipSuccess || fill(sc, K, 22u, 0.5f, s) != hipSuccess || fill(sh, K, 23u, 0.3f, s) != hipSuccess) return F5HIP_ERR_HIP;
  std::vector<float> ref((size_t)rows * K);
  std::vector<f16> got((size_t)rows * K * 2);
  for (int which = 0; which < 3; ++which) {
    const float* src = x;
    if (which == 2) {  // AdaLN-modulated LayerNorm: fp32 rows from the fp32 kernel, MX rows from the MX kernel
      if (launch_layernorm(x, K, rows, K, 1e-6f, nullptr, nullptr, sc, sh, y, nullptr, nullptr, K, s) != hipSuccess ||
          launch_layernorm(x, K, rows, K, 1e-6f, nullptr, nullptr, sc, sh, nullptr, mxr, mxr + 32, K, s, 2, 2 * (int64_t)K) != hipSuccess)
        return F5HIP_ERR_HIP;
      src = y;
    } else if (launch_pack_mx_rows(x, K, rows, K, nullptr, mxr, which, s) != hipSuccess) {
      return F5HIP_ERR_HIP;
    }
    if (hipMemcpyAsync(ref.data(), src, ref.size() * 4, hipMemcpyDeviceToHost, s) != hipSuccess ||
        hipMemcpyAsync(got.data(), mxr, got.size() * 2, hipMemcpyDeviceToHost, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess)
      return F5HIP_ERR_HIP;
    double wc = 0, wl = 0, mv = 0;
    size_t bad = 0, hd = 0;
    // LayerNorm: the two kernels sum a row in different lane orders — values one fp32 ulp apart, a remainder step is 2^13 ulps: same bounds
    mx_lines_check(got.data(), ref.data(), rows, K, which == 1, 0.5001, which == 2 ? 0.27 : 0.2501, &wc, &wl, &bad, &hd, &mv);
    out9[3 * which] = wc; out9[3 * which + 1] = wl; out9[3 * which + 2] = (double)(bad + (which == 2 ? (hd > (size_t)rows * K / 1000 ? hd : 0) : hd));
  }
  return F5HIP_OK;
}

// The fused q|k|v projection of one block (bias, rope, head scatter into the flash layouts) for `seqs` sequences of nseq tokens, H = 16
// heads of 64: time of `variant`, and with check != 0 every output plane (q, k hi/lo, V^T) compared byte for byte with the generic kernel
// of gemm.h (variant 1).  Returns the number of differing bytes in *diff (negative status on errors).
// check == 2 (fp16x3 / fp16m): the MX-corrected scores end to end — the variant runs with EpiQKV::mx_qk, its P words are decoded on the host
// against the generic kernel's hi + lo values (the format's bounds, mx_lines_check), and the flash kernel's NSPLIT = 2 form on those planes
// must land on the split form's output where plain fp16 scores do not (*diff counts what is out of bounds).
int f5hip_bench_qkv(f5hip_ctx* ctx, int precision, int variant, int seqs, int nseq, int K, int iters, int check, double* avg_ms, int64_t* diff) {
  if (!ctx || !avg_ms || seqs <= 0 || nseq <= 1 || iters <= 0 || precision == F5HIP_PREC_FP32 || (K % 32)) return F5HIP_ERR_INVALID;
  std::lock_guard<std::mutex> lk(ctx->mu);
  if (hipSetDevice(ctx->device) != hipSuccess) return F5HIP_ERR_HIP;
  if (init_gemm_kernels() != hipSuccess) return F5HIP_ERR_HIP;
  hipStream_t s = nullptr;
  Tmp t;
  const bool mx = precision == F5HIP_PREC_FP16M;  // MX operand lines; the reference of the check is the fp16x3 generic kernel (values, to 1e-4)
  const int op = precision == F5HIP_PREC_FP16 ? OP_F16 : mx ? OP_F16M : OP_F16X3;
  const int H = 16, dh = 64, inner = H * dh, N = 3 * inner, M = seqs * nseq, ldv = (nseq + 7) & ~7;
  const bool x3 = op != OP_F16;
  const size_t pl = x3 ? 2 : 1;
  float* a32 = t.get<float>((size_t)M * K);
  float* w32 = t.get<float>((size_t)N * K);
  float* bias = t.get<float>(N);
  float* rope = t.get<float>((size_t)nseq * dh);
  float* invf = t.get<float>(dh / 2);
  f16 *ah = t.get<f16>((size_t)M * K * pl), *wh = t.get<f16>((size_t)N * K * pl);
  f16 *am = mx ? t.get<f16>((size_t)M * K * 2) : nullptr, *wm = mx ? t.get<f16>((size_t)N * K * 2) : nullptr;
  const size_t nq = (size_t)seqs * H * nseq * dh, nv = (size_t)seqs * H * dh * ldv;
  f16* out[2][6];  // [reference / variant][q, q_lo, k, k_lo, vt, vt_lo]
  for (auto& o : out)
    for (int i = 0; i < 6; ++i) {
      o[i] = t.get<f16>(i < 4 ? nq : nv);
      if (!o[i]) return F5HIP_ERR_HIP;
    }
  if (!a32 || !w32 || !bias || !rope || !invf || !ah || !wh) return F5HIP_ERR_HIP;
  std::vector<float> f(dh / 2);
  for (int k = 0; k < dh / 2; ++k) f[k] = 1.0f / powf(10000.0f, (float)(2 * k) / (float)dh);
  if (hipMemcpy(invf, f.data(), f.size() * 4, hipMemcpyHostToDevice) != hipSuccess || launch_rope_table(invf, nseq, dh / 2, rope, s) != hipSuccess) return F5HIP_ERR_HIP;
  if (fill(a32, (int64_t)M * K, 1u, 1.0f, s) != hipSuccess || fill(w32, (int64_t)N * K, 2u, 0.05f, s) != hipSuccess || fill(bias, N, 3u, 0.02f, s) != hipSuccess)
    return F5HIP_ERR_HIP;
  if (x3) {
    if (launch_split_f16_packed(a32, M, K, ah, s) != hipSuccess || launch_split_f16_packed(w32, N, K, wh, s) != hipSuccess) return F5HIP_ERR_HIP;
  } else if (launch_split_f16(a32, (int64_t)M * K, 1.0f, ah, nullptr, s) != hipSuccess || launch_split_f16(w32, (int64_t)N * K, 1.0f, wh, nullptr, s) != hipSuccess) {
    return F5HIP_ERR_HIP;
  }
  if (mx && (!am || !wm || launch_pack_mx_rows(a32, K, M, K, nullptr, am, 0, s) != hipSuccess || launch_pack_mx_rows(w32, K, N, K, nullptr, wm, 1, s) != hipSuccess)) return F5HIP_ERR_HIP;
  GemmCore g{};
  g.A = ah; g.W = wh; g.lda = (int64_t)K * pl; g.ldw = (int64_t)K * pl; g.M = M; g.N = N; g.K = K; g.a_rows = M; g.w_rows = N;
  GemmCore gm = g;  // the launch under test: MX lines in fp16m mode (same strides)
  if (mx) { gm.A = am; gm.W = wm; }
  const bool mxqk = check == 2;
  if (mxqk && !x3) return F5HIP_ERR_INVALID;
  auto epi = [&](int which) {
    EpiQKV e{};
    e.bias = bias; e.rope_cs = rope; e.nseq = nseq; e.heads = H; e.dh = dh; e.pe_heads = getenv("KB_QKV_PE") ? atoi(getenv("KB_QKV_PE")) : -1; e.ldvt = ldv;
    e.qscale = mxqk ? 6.0f : 0.125f;  // (mxqk: logits of tens, so that the rounding of the scores shows in the output)
    e.mx_qk = mxqk && which == 1;
    e.q16 = out[which][0]; e.k16 = out[which][2]; e.vt16 = out[which][4];
    if (x3) { e.q16_lo = out[which][1]; e.k16_lo = out[which][3]; e.vt16_lo = out[which][5]; }
    return e;
  };
  int64_t bad = 0;
  if (check) {
    for (int w = 0; w < 2; ++w) {
      for (int i = 0; i < 6; ++i)
        if (hipMemsetAsync(out[w][i], 0, (i < 4 ? nq : nv) * sizeof(f16), s) != hipSuccess) return F5HIP_ERR_HIP;
      if ((w == 0 ? launch_gemm_qkv_variant(mx ? OP_F16X3 : op, g, epi(w), 1, s) : launch_gemm_qkv_variant(op, gm, epi(w), variant, s)) != hipSuccess) return F5HIP_ERR_HIP;
    }
    if (hipStreamSynchronize(s) != hipSuccess) return F5HIP_ERR_HIP;
    // the VALUE of every output (hi + lo in fp16x3) must agree to fp32 rounding: the two kernels contract the rope arithmetic differently
    // (an fma here, a mul + add there), which flips a last bit of `hi` now and then and is absorbed by `lo`; anything above 4e-6 of the
    // plane's largest value is a real difference.  V (no rope) is compared byte for byte as well.
    if (mxqk) {
      for (int pl3 = 0; pl3 < 2; ++pl3) {  // q (packed as the activation), k (as the weight)
        std::vector<f16> ah(nq), al(nq), bh(nq), bp(nq), lines(nq * 2);
        if (hipMemcpy(ah.data(), out[0][2 * pl3], nq * 2, hipMemcpyDeviceToHost) != hipSuccess || hipMemcpy(al.data(), out[0][2 * pl3 + 1], nq * 2, hipMemcpyDeviceToHost) != hipSuccess ||
            hipMemcpy(bh.data(), out[1][2 * pl3], nq * 2, hipMemcpyDeviceToHost) != hipSuccess || hipMemcpy(bp.data(), out[1][2 * pl3 + 1], nq * 2, hipMemcpyDeviceToHost) != hipSuccess)
          return F5HIP_ERR_HIP;
        std::vector<float> xref(nq);
        for (size_t j = 0; j < nq; ++j) xref[j] = (float)ah[j] + (float)al[j];
        for (size_t r = 0; r < nq / 64; ++r)
          for (int blk = 0; blk < 2; ++blk) {  // the line mx_lines_check reads: [32 hi | P_0 | P_1]
            memcpy(&lines[(r * 2 + blk) * 64], &bh[r * 64 + blk * 32], 64);
            memcpy(&lines[(r * 2 + blk) * 64 + 32], &bp[r * 64 + blk * 32], 64);
          }
        double wc = 0, wl = 0, mv = 0;
        size_t nbad = 0, hd = 0;
        mx_lines_check(lines.data(), xref.data(), (int64_t)(nq / 64), 64, pl3 == 1, 0.51, 1.5, &wc, &wl, &nbad, &hd, &mv);
        fprintf(stderr, "QKV_CHECK variant %d %s P words: coarse / remainder errors %.3f / %.3f block steps, %zu out of bounds, %zu of %zu hi halves differ, max |value| %.3g\n",
                variant, pl3 == 0 ? "q" : "k", wc, wl, nbad, hd, nq, mv);
        bad += (int64_t)nbad + (hd > nq / 8 ? (int64_t)hd : 0);
      }
      if (init_attention_kernels() != hipSuccess) return F5HIP_ERR_HIP;
      f16* o[3];
      std::vector<f16> oh[3];
      for (int w = 0; w < 3; ++w) {  // split q, k (the reference's planes) / MX-corrected / plain fp16 (the variant's planes)
        o[w] = t.get<f16>(nq);
        if (!o[w]) return F5HIP_ERR_HIP;
        const int src = w == 0 ? 0 : 1;
        if (launch_flash_attn(w == 0 ? 2 : w == 1 ? 4 : 1, out[src][0], w < 2 ? out[src][1] : nullptr, out[src][2], w < 2 ? out[src][3] : nullptr, out[src][4], nullptr, ldv,
                              seqs, H, nseq, nullptr, o[w], nullptr, s, 0, nullptr, 0, 1, 1, nullptr, nullptr, /*log2q*/ 1) != hipSuccess)
          return F5HIP_ERR_HIP;
        oh[w].resize(nq);
        if (hipMemcpyAsync(oh[w].data(), o[w], nq * 2, hipMemcpyDeviceToHost, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess) return F5HIP_ERR_HIP;
      }
      double dm = 0, dp = 0, sm = 0, sp = 0, vmax = 0;
      for (size_t j = 0; j < nq; ++j) {
        const double a = (float)oh[0][j], m = (float)oh[1][j], pv = (float)oh[2][j];
        dm = std::max(dm, fabs(m - a)); dp = std::max(dp, fabs(pv - a)); sm += fabs(m - a); sp += fabs(pv - a); vmax = std::max(vmax, fabs(a));
      }
      fprintf(stderr, "QKV_CHECK variant %d attention on the MX planes against split q, k: max |diff| %.3g mean %.3g; plain fp16 scores: max %.3g mean %.3g; max |value| %.3g\n",
              variant, dm, sm / nq, dp, sp / nq, vmax);
      if (!(sm < 0.35 * sp) || !(dm < 0.02 * vmax)) ++bad;
      if (diff) *diff = bad;
      return time_it([&] { return launch_gemm_qkv_variant(op, gm, epi(1), variant, s); }, iters, s, avg_ms);
    }
    for (int pl3 = 0; pl3 < 3; ++pl3) {
      const size_t n = (pl3 < 2 ? nq : nv);
      std::vector<f16> ah(n), bh(n), al(x3 ? n : 0), bl(x3 ? n : 0);
      if (hipMemcpy(ah.data(), out[0][2 * pl3], n * 2, hipMemcpyDeviceToHost) != hipSuccess || hipMemcpy(bh.data(), out[1][2 * pl3], n * 2, hipMemcpyDeviceToHost) != hipSuccess)
        return F5HIP_ERR_HIP;
      if (x3 && (hipMemcpy(al.data(), out[0][2 * pl3 + 1], n * 2, hipMemcpyDeviceToHost) != hipSuccess || hipMemcpy(bl.data(), out[1][2 * pl3 + 1], n * 2, hipMemcpyDeviceToHost) != hipSuccess))
        return F5HIP_ERR_HIP;
      double vmax = 0;
      for (size_t j = 0; j < n; ++j) vmax = std::max(vmax, (double)fabsf((float)ah[j]));
      const double tol = (mx ? 1e-4 : x3 ? 4e-6 : 2e-3) * std::max(vmax, 1e-3);
      int64_t nb = 0, first = -1, nbytes = 0;
      double maxd = 0;
      for (size_t j = 0; j < n; ++j) {
        const double va = (double)(float)ah[j] + (x3 ? (double)(float)al[j] : 0.0), vb = (double)(float)bh[j] + (x3 ? (double)(float)bl[j] : 0.0);
        const double d = fabs(va - vb);
        if (d > tol) { if (first < 0) first = (int64_t)j; ++nb; }
        maxd = std::max(maxd, d);
        if (memcmp(&ah[j], &bh[j], 2) != 0) ++nbytes;
      }
      if (pl3 == 2 && nbytes && variant < 65 && !mx) { nb += nbytes; }  // (the k-split tiles, 65.., sum in another order: values only)
      if (nb) fprintf(stderr, "QKV_CHECK variant %d plane %s: %lld of %zu values differ by more than %.3g (first at %lld), max |diff| %.3g of max |value| %.3g\n", variant,
                      pl3 == 0 ? "q" : pl3 == 1 ? "k" : "v^T", (long long)nb, n, tol, (long long)first, maxd, vmax);
      bad += nb;
    }
  }
  if (diff) *diff = bad;
  return time_it([&] { return launch_gemm_qkv_variant(op, gm, epi(1), variant, s); }, iters, s, avg_ms);
}

// Reproducer of round 2's co-residency fault in the fused q|k|v epilogue (race_probe.hip): `reps` launches of tile `variant` with the
// epilogue form `expt` (ablation `abl`, `lds_pad` extra LDS per workgroup, an optional co-tenant kernel on a second stream: noise 1 =
// plain streaming loads, 2 = LDS-DMA), each compared with the generic kernel's q / k values and V^T bytes.  bad[r] = wrong outputs of
// launch r.  With dump_path the wrong outputs of all launches are appended as records {int32 rep, plane, int64 index, float ref, got, partner ref, partner got}.
int f5hip_bench_qkv_probe(f5hip_ctx* ctx, int variant, int expt, int abl, int lds_pad, int noise, int seqs, int nseq, int reps, int64_t* bad,
                          const char* dump_path) {
  if (!ctx || !bad || seqs <= 0 || nseq <= 1 || reps <= 0) return F5HIP_ERR_INVALID;
#ifdef F5_HIPEMU
  return F5HIP_ERR_UNSUPPORTED;  // the reproducer is gfx950 instructions (csrc/race_probe.hip is not part of the host build)
#else
  std::lock_guard<std::mutex> lk(ctx->mu);
  if (hipSetDevice(ctx->device) != hipSuccess) return F5HIP_ERR_HIP;
  if (init_gemm_kernels() != hipSuccess) return F5HIP_ERR_HIP;
  hipStream_t s = nullptr, s2 = nullptr;
  if (noise && hipStreamCreateWithFlags(&s2, hipStreamNonBlocking) != hipSuccess) return F5HIP_ERR_HIP;
  Tmp t;
  const int K = 1024, H = 16, dh = 64, inner = H * dh, N = 3 * inner, M = seqs * nseq, ldv = (nseq + 7) & ~7;
  float* a32 = t.get<float>((size_t)M * K);
  float* w32 = t.get<float>((size_t)N * K);
  float* bias = t.get<float>(N);
  float* rope = t.get<float>((size_t)nseq * dh);
  float* invf = t.get<float>(dh / 2);
  f16 *ah = t.get<f16>((size_t)M * K * 2), *wh = t.get<f16>((size_t)N * K * 2);
  const size_t nq = (size_t)seqs * H * nseq * dh, nv = (size_t)seqs * H * dh * ldv;
  f16* out[2][3];
  for (auto& o : out)
    for (int i = 0; i < 3; ++i) {
      o[i] = t.get<f16>(i < 2 ? nq : nv);
      if (!o[i]) return F5HIP_ERR_HIP;
    }
  const uint32_t noise_bytes = 64u << 20;
  void* nbuf = noise ? (void*)t.get<char>(noise_bytes) : nullptr;
  uint32_t* sink = t.get<uint32_t>(4);
  if (!a32 || !w32 || !bias || !rope || !invf || !ah || !wh || !sink || (noise && !nbuf)) return F5HIP_ERR_HIP;
  std::vector<float> f(dh / 2);
  for (int k = 0; k < dh / 2; ++k) f[k] = 1.0f / powf(10000.0f, (float)(2 * k) / (float)dh);
  if (hipMemcpy(invf, f.data(), f.size() * 4, hipMemcpyHostToDevice) != hipSuccess || launch_rope_table(invf, nseq, dh / 2, rope, s) != hipSuccess) return F5HIP_ERR_HIP;
  if (fill(a32, (int64_t)M * K, 1u, 1.0f, s) != hipSuccess || fill(w32, (int64_t)N * K, 2u, 0.05f, s) != hipSuccess || fill(bias, N, 3u, 0.02f, s) != hipSuccess)
    return F5HIP_ERR_HIP;
  if (noise && hipMemsetAsync(nbuf, 1, noise_bytes, s) != hipSuccess) return F5HIP_ERR_HIP;
  if (launch_split_f16_packed(a32, M, K, ah, s) != hipSuccess || launch_split_f16_packed(w32, N, K, wh, s) != hipSuccess) return F5HIP_ERR_HIP;
  GemmCore g{};
  g.A = ah; g.W = wh; g.lda = (int64_t)K * 2; g.ldw = (int64_t)K * 2; g.M = M; g.N = N; g.K = K; g.a_rows = M; g.w_rows = N;  // (group_m 0: the launcher's default rasterisation)
  auto epi = [&](int which) {
    EpiQKV e{};
    e.bias = bias; e.rope_cs = rope; e.nseq = nseq; e.heads = H; e.dh = dh; e.pe_heads = -1; e.qscale = 0.125f; e.ldvt = ldv;
    e.q16 = out[which][0]; e.k16 = out[which][1]; e.vt16 = out[which][2];
    return e;
  };
  for (int i = 0; i < 3; ++i)
    if (hipMemsetAsync(out[0][i], 0, (i < 2 ? nq : nv) * sizeof(f16), s) != hipSuccess) return F5HIP_ERR_HIP;
  if (launch_gemm_qkv_variant(OP_F16X3, g, epi(0), 1, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess) return F5HIP_ERR_HIP;
  std::vector<f16> ref[3], got[3];
  for (int i = 0; i < 3; ++i) {
    ref[i].resize(i < 2 ? nq : nv);
    got[i].resize(ref[i].size());
    if (hipMemcpy(ref[i].data(), out[0][i], ref[i].size() * 2, hipMemcpyDeviceToHost) != hipSuccess) return F5HIP_ERR_HIP;
  }
  double vmax[2] = {0, 0};
  for (int i = 0; i < 2; ++i)
    for (f16 v : ref[i]) vmax[i] = std::max(vmax[i], (double)fabsf((float)v));
  FILE* dump = dump_path && *dump_path ? fopen(dump_path, "ab") : nullptr;
  for (int rep = 0; rep < reps; ++rep) {
    for (int i = 0; i < 3; ++i)
      if (hipMemsetAsync(out[1][i], 0, (i < 2 ? nq : nv) * sizeof(f16), s) != hipSuccess) return F5HIP_ERR_HIP;
    if (hipStreamSynchronize(s) != hipSuccess) return F5HIP_ERR_HIP;
    if (noise && launch_noise(nbuf, noise_bytes, 512, 3, noise, 4096, sink, s2) != hipSuccess) return F5HIP_ERR_HIP;
    if (hipMemsetAsync(sink, 0, 16, s) != hipSuccess) return F5HIP_ERR_HIP;
    if (launch_pp_qkv_probe(g, epi(1), variant, expt, abl, lds_pad, sink + 1, s) != hipSuccess) return F5HIP_ERR_HIP;
    if (hipStreamSynchronize(s) != hipSuccess || (noise && hipStreamSynchronize(s2) != hipSuccess)) return F5HIP_ERR_HIP;
    int64_t nb = 0;
    for (int i = 0; i < 3; ++i) {
      if (hipMemcpy(got[i].data(), out[1][i], got[i].size() * 2, hipMemcpyDeviceToHost) != hipSuccess) return F5HIP_ERR_HIP;
      // q, k: the generic kernel contracts the rope arithmetic differently, so a last bit of an fp16 value may differ (2^-11 relative);
      // V^T carries no rope: byte for byte
      for (size_t j = 0; j < got[i].size(); ++j) {
        const float a = (float)ref[i][j], b = (float)got[i][j];
        const bool wrong = i < 2 ? fabsf(a - b) > 1.5e-3f * std::max(fabsf(a), 2e-3f * (float)vmax[i]) : memcmp(&ref[i][j], &got[i][j], 2) != 0;
        if (!wrong) continue;
        ++nb;
        if (dump && nb <= 20000) {
          const int32_t hd[2] = {rep, i};
          const int64_t idx = (int64_t)j;
          // + the reference and the launch's value of the rope partner (channel d ^ 1 of the same token; for V^T: the neighbouring token)
          const float vals[4] = {a, b, (float)ref[i][j ^ 1], (float)got[i][j ^ 1]};
          fwrite(hd, 4, 2, dump); fwrite(&idx, 8, 1, dump); fwrite(vals, 4, 4, dump);
        }
      }
    }
    bad[rep] = nb;
    uint32_t dbgv[4] = {0, 0, 0, 0};
    if (hipMemcpy(dbgv, sink, 16, hipMemcpyDeviceToHost) != hipSuccess) return F5HIP_ERR_HIP;
    if (dbgv[1]) fprintf(stderr, "QKV_PROBE launch %d: %u sine registers read differently right after the wait and 32 cycles later\n", rep, dbgv[1]);
  }
  if (dump) fclose(dump);
  if (s2) (void)hipStreamDestroy(s2);
  return F5HIP_OK;
#endif
}

// flash attention over [batch2 * heads, n, 64]; precision FP16 -> plain fp16 operands, FP16X3 -> hi/lo split
int f5hip_bench_attention(f5hip_ctx* ctx, int precision, int batch2, int heads, int n, int iters, double* avg_ms) {
  if (!ctx || !avg_ms || batch2 <= 0 || heads <= 0 || n <= 0 || iters <= 0 || precision == F5HIP_PREC_FP32) return F5HIP_ERR_INVALID;
  std::lock_guard<std::mutex> lk(ctx->mu);
  if (hipSetDevice(ctx->device) != hipSuccess) return F5HIP_ERR_HIP;
  if (init_attention_kernels() != hipSuccess) return F5HIP_ERR_HIP;
  hipStream_t s = nullptr;
  Tmp t;
  const int ldv = (n + 7) & ~7;
  const size_t BH = (size_t)batch2 * heads, nq = BH * n * 64, nv = BH * 64 * ldv;
  float* tmp = t.get<float>(nv > nq ? nv : nq);
  f16 *qh = t.get<f16>(nq), *ql = t.get<f16>(nq), *kh = t.get<f16>(nq), *kl = t.get<f16>(nq), *vh = t.get<f16>(nv), *vl = t.get<f16>(nv);
  f16 *oh = t.get<f16>(nq), *ol = t.get<f16>(nq);
  if (!tmp || !qh || !ql || !kh || !kl || !vh || !vl || !oh || !ol) return F5HIP_ERR_HIP;
  if (hipMemsetAsync(vh, 0, nv * sizeof(f16), s) != hipSuccess || hipMemsetAsync(vl, 0, nv * sizeof(f16), s) != hipSuccess) return F5HIP_ERR_HIP;
  if (fill(tmp, nq, 11u, 0.5f, s) != hipSuccess || launch_split_f16(tmp, nq, 1.0f, qh, ql, s) != hipSuccess) return F5HIP_ERR_HIP;
  if (fill(tmp, nq, 12u, 2.0f, s) != hipSuccess || launch_split_f16(tmp, nq, 1.0f, kh, kl, s) != hipSuccess) return F5HIP_ERR_HIP;
  if (fill(tmp, nv, 13u, 1.0f, s) != hipSuccess || launch_split_f16(tmp, nv, 1.0f, vh, vl, s) != hipSuccess) return F5HIP_ERR_HIP;
  const bool x3 = precision == F5HIP_PREC_FP16X3;
  return time_it([&] {
    static const int log2q = getenv("KB_ATTN_LOG2Q") ? 1 : 0;  // 1: treat q as carrying log2(e) -> the lazy-reference form (timing A/B)
    return launch_flash_attn(x3 ? (ctx->attn_impl == 2 ? 3 : 2) : 1, qh, x3 ? ql : nullptr, kh, x3 ? kl : nullptr, vh, x3 ? vl : nullptr, ldv, batch2, heads, n, nullptr, oh,
                             x3 ? ol : nullptr, s, 0, nullptr, 0, 1, 1, nullptr, nullptr, log2q);
  }, iters, s, avg_ms);
}

}  // extern "C"
