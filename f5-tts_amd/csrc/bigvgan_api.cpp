// bigvgan.cpp — C-ABI entry points of the BigVGAN-v2 generator (include/f5hip.h, f5hip_bigvgan_*): context, weights, and the enqueue
// logic of `vocoder(mel)` (reference src/f5_tts/infer/utils_infer.py:130-144,512-513; upstream NVIDIA/BigVGAN bigvgan.py BigVGAN.forward,
// AMPBlock1 / AMPBlock2).  No torch types, no CPU fallback: every numeric step is a kernel of bigvgan.hip or the MFMA GEMM (gemm.h).
//
// Formulation (restated and checked on the CPU in tests/bigvgan_model.py against oracle/bigvgan_oracle.py):
//   * activations are channels-last fp32 [b, L, C];
//   * Conv1d(Cin -> Cout, k, dilation d, "same") = GEMM [L, k*cpad] x [Cout, k*cpad]^T over the tap-gathered operand
//     col[l, j*cpad + c] = y[l + (j - k/2) d, c]   (cpad = Cin rounded up to 32: the operand block of the packed fp16x3 layout);
//   * ConvTranspose1d(Cin -> Cout, k, stride u, padding (k-u)/2) = the same with taps {s0, s0+1, ...} (s0 = -1, 3 taps for k = 2u) and
//     u*Cout output columns: column r*Cout + co is output phase r, so the [L, u*Cout] result IS the [L*u, Cout] upsampled tensor;
//     weight row (r, co), column (t, ci) = w[ci, co, r + pad - (s0 + t) u] where that tap exists, else 0;
//   * bias and the resblock's residual add (x = xt + x) ride in the GEMM epilogue (EpiStore);
//   * Activation1d, the mean over the parallel resblocks and conv_post (+ tanh / clamp) are the kernels of bigvgan.hip.
// First version: the operand is materialised once per conv (k-fold write amplification of an HBM-bound tensor).  DESIGN.md section 8
// lists the follow-up (tap-shifted A rows inside the GEMM's k-loop, activation fused into the operand emission).
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <string>
#include <unordered_map>
#include <vector>

#include "bigvgan_host.h"
#include "engine.h"

namespace {

thread_local std::string g_bv_create_err;

struct ConvW {        // one Conv1d / ConvTranspose1d as a GEMM
  int cin = 0, cout = 0, ntaps = 0, shift0 = 0, dstep = 1, cpad = 0, K = 0, N = 0;
  DevBuf w32, w16, wpk, bias;  // [N, K] fp32 / fp16 rows, [N, 2K] packed hi/lo rows, [N] fp32
};
struct ActW {
  DevBuf alpha, beta;  // [C]
};
struct ResW {
  int k = 0;
  std::vector<int> dil;
  std::vector<ConvW> c1, c2;  // AMPBlock2: c2 empty
  std::vector<ActW> act;      // AMPBlock1: 2 per dilation (before c1, before c2); AMPBlock2: 1 per dilation
};
struct HostTensor {
  std::string name;
  std::vector<int64_t> shape;
  std::vector<float> data;
  bool loaded = false;
  int64_t numel() const { int64_t n = 1; for (int64_t d : shape) n *= d; return n; }
};

}  // namespace

struct f5hip_bigvgan {
  f5hip_bigvgan_config cfg{};
  int device = 0;
  std::mutex mu;
  std::string err;
  std::vector<HostTensor> tensors;
  std::unordered_map<std::string, int> index;
  bool finalized = false;
  // derived
  ConvW conv_pre;
  std::vector<ConvW> ups;
  std::vector<std::vector<ResW>> res;  // [stage][kernel]
  ActW act_post;
  DevBuf post_w7, post_bias;
  int c_last = 0;
  float filt[12];
  // workspace (grow-only)
  DevBuf xa, xb, tt, yy, col, rr[4];
  // measurement: option "profile" times every launch with HIP events on the launch stream, per kernel class
  bool profile = false;
  KStat stats[4];             // BV_* classes
  std::vector<ProfRec> prof;
  int conv_impl = 2;          // 0 = tap-gathered operand + plain GEMM, 1 = implicit GEMM with tap-shifted rows (conv_gemm.h),
                              // 2 = 1 + Activation1d writes the conv's operand copy itself (the default since it was timed: 945 against
                              // 977 / 1010 ms per configs[4] step, the generator's share 99 against 164 ms; tools/r2_call20.sh)
  int stop_after_stage = -1;  // parity tap (tests): >= 0 makes forward() return the channels-last stage tensor instead of the waveform
};

namespace {

#define HIPCHK(expr)                                                                                   \
  do {                                                                                                 \
    hipError_t _e = (expr);                                                                            \
    if (_e != hipSuccess) {                                                                            \
      char _b[512];                                                                                    \
      snprintf(_b, sizeof(_b), "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
      v->err = _b;                                                                                     \
      return F5HIP_ERR_HIP;                                                                            \
    }                                                                                                  \
  } while (0)

#define FAIL(code, ...)                          \
  do {                                           \
    char _b[512];                                \
    snprintf(_b, sizeof(_b), __VA_ARGS__);       \
    v->err = _b;                                 \
    return code;                                 \
  } while (0)

#define CHK(expr)                   \
  do {                              \
    int _r = (expr);                \
    if (_r != F5HIP_OK) return _r;  \
  } while (0)

enum { BV_GEMM = 0, BV_ACT = 1, BV_OPERAND = 2, BV_OTHER = 3, BV_COUNT = 4 };
const char* const BV_NAMES[BV_COUNT] = {"conv_gemm", "activation1d", "operand", "other"};
struct BvProf {  // as Prof in api.cpp: events around one launch, resolved after the forward pass
  f5hip_bigvgan* v;
  hipStream_t s;
  ProfRec rec{};
  BvProf(f5hip_bigvgan* v_, hipStream_t st, int kclass, double flops, double bytes) : v(v_), s(st) {
    if (v->profile) {
      rec.kclass = kclass; rec.flops = flops; rec.bytes = bytes;
      (void)hipEventCreate(&rec.e0);
      (void)hipEventCreate(&rec.e1);
      (void)hipEventRecord(rec.e0, s);
    }
  }
  ~BvProf() {
    if (v->profile) {
      (void)hipEventRecord(rec.e1, s);
      v->prof.push_back(rec);
    }
  }
};
void bv_collect(f5hip_bigvgan* v, hipStream_t s) {
  if (v->prof.empty()) return;
  (void)hipStreamSynchronize(s);
  for (auto& r : v->prof) {
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, r.e0, r.e1) == hipSuccess) {
      KStat& k = v->stats[r.kclass];
      k.calls += 1; k.ms += ms; k.flops += r.flops; k.bytes += r.bytes;
    }
    (void)hipEventDestroy(r.e0);
    (void)hipEventDestroy(r.e1);
  }
  v->prof.clear();
}

int rup32(int x) { return (x + 31) / 32 * 32; }

void add_tensor(f5hip_bigvgan* v, const std::string& name, std::vector<int64_t> shape) {
  HostTensor t;
  t.name = name;
  t.shape = std::move(shape);
  v->index[name] = (int)v->tensors.size();
  v->tensors.push_back(std::move(t));
}

int stage_channels(const f5hip_bigvgan_config& c, int i) { return c.upsample_initial_channel >> (i + 1); }

void build_tensor_table(f5hip_bigvgan* v) {
  const auto& c = v->cfg;
  const int c0 = c.upsample_initial_channel;
  add_tensor(v, "conv_pre.weight", {c0, c.num_mels, 7});
  add_tensor(v, "conv_pre.bias", {c0});
  auto add_act = [&](const std::string& pfx, int ch) {
    add_tensor(v, pfx + ".act.alpha", {ch});
    if (c.activation == 1) add_tensor(v, pfx + ".act.beta", {ch});
  };
  for (int i = 0; i < c.num_upsamples; ++i) {
    const int cin = c0 >> i, ch = stage_channels(c, i);
    const std::string u = "ups." + std::to_string(i) + ".0";
    add_tensor(v, u + ".weight", {cin, ch, c.upsample_kernel_sizes[i]});  // ConvTranspose1d layout [in, out, k]
    add_tensor(v, u + ".bias", {ch});
    for (int j = 0; j < c.num_kernels; ++j) {
      const std::string r = "resblocks." + std::to_string(i * c.num_kernels + j);
      const int k = c.resblock_kernel_sizes[j], nd = c.resblock_num_dilations[j];
      for (int m = 0; m < nd; ++m) {
        if (c.resblock == 1) {
          for (const char* cv : {".convs1.", ".convs2."}) {
            add_tensor(v, r + cv + std::to_string(m) + ".weight", {ch, ch, k});
            add_tensor(v, r + cv + std::to_string(m) + ".bias", {ch});
          }
        } else {
          add_tensor(v, r + ".convs." + std::to_string(m) + ".weight", {ch, ch, k});
          add_tensor(v, r + ".convs." + std::to_string(m) + ".bias", {ch});
        }
      }
      const int nact = c.resblock == 1 ? 2 * nd : nd;
      for (int q = 0; q < nact; ++q) add_act(r + ".activations." + std::to_string(q), ch);
    }
  }
  const int cl = c0 >> c.num_upsamples;
  add_act("activation_post", cl);
  add_tensor(v, "conv_post.weight", {1, cl, 7});
  if (c.use_bias_at_final) add_tensor(v, "conv_post.bias", {1});
}

const HostTensor& T(const f5hip_bigvgan* v, const std::string& name) { return v->tensors[v->index.at(name)]; }

int upload_gemm_weight(f5hip_bigvgan* v, ConvW& cw, const std::vector<float>& mat, const std::vector<float>& bias) {
  const int64_t n = (int64_t)cw.N * cw.K;
  HIPCHK(cw.w32.ensure(n * sizeof(float)));
  HIPCHK(cw.w16.ensure(n * sizeof(f16)));
  HIPCHK(cw.wpk.ensure(2 * n * sizeof(f16)));
  HIPCHK(cw.bias.ensure(((size_t)cw.N + 4) * sizeof(float)));
  HIPCHK(hipMemcpy(cw.w32.p, mat.data(), n * sizeof(float), hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(cw.bias.p, bias.data(), (size_t)cw.N * sizeof(float), hipMemcpyHostToDevice));
  HIPCHK(launch_split_f16(cw.w32.as<float>(), n, 1.0f, cw.w16.as<f16>(), nullptr, nullptr));
  HIPCHK(launch_split_f16_packed(cw.w32.as<float>(), cw.N, cw.K, cw.wpk.as<f16>(), nullptr));
  return F5HIP_OK;
}

// Conv1d weight [Cout, Cin, k] -> [Cout, k * cpad], column j * cpad + ci
int make_conv(f5hip_bigvgan* v, ConvW& cw, const std::string& name, int dilation) {
  const HostTensor& w = T(v, name + ".weight");
  const HostTensor& b = T(v, name + ".bias");
  const int cout = (int)w.shape[0], cin = (int)w.shape[1], k = (int)w.shape[2];
  cw.cin = cin; cw.cout = cout; cw.ntaps = k; cw.shift0 = -(k / 2) * dilation; cw.dstep = dilation;
  cw.cpad = rup32(cin); cw.K = k * cw.cpad; cw.N = cout;
  std::vector<float> mat;
  bv_conv_matrix(w.data.data(), cout, cin, k, cw.cpad, mat);
  return upload_gemm_weight(v, cw, mat, b.data);
}

// ConvTranspose1d weight [Cin, Cout, k], stride u -> [u * Cout, ntaps * cpad]
int make_convt(f5hip_bigvgan* v, ConvW& cw, const std::string& name, int u) {
  const HostTensor& w = T(v, name + ".weight");
  const HostTensor& b = T(v, name + ".bias");
  const int cin = (int)w.shape[0], cout = (int)w.shape[1], k = (int)w.shape[2];
  cw.cin = cin; cw.cout = cout; cw.dstep = 1;
  bv_convt_taps(k, u, cw.shift0, cw.ntaps);
  cw.cpad = rup32(cin); cw.K = cw.ntaps * cw.cpad; cw.N = u * cout;
  std::vector<float> mat, bias((size_t)cw.N);
  bv_convt_matrix(w.data.data(), cin, cout, k, u, cw.cpad, cw.shift0, cw.ntaps, mat);
  for (int r = 0; r < u; ++r)
    for (int co = 0; co < cout; ++co) bias[(size_t)r * cout + co] = b.data[co];
  return upload_gemm_weight(v, cw, mat, bias);
}

int make_act(f5hip_bigvgan* v, ActW& a, const std::string& pfx) {
  const HostTensor& al = T(v, pfx + ".act.alpha");
  const HostTensor& be = v->cfg.activation == 1 ? T(v, pfx + ".act.beta") : al;  // Snake: beta = alpha
  HIPCHK(a.alpha.ensure(al.data.size() * sizeof(float)));
  HIPCHK(a.beta.ensure(be.data.size() * sizeof(float)));
  HIPCHK(hipMemcpy(a.alpha.p, al.data.data(), al.data.size() * sizeof(float), hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(a.beta.p, be.data.data(), be.data.size() * sizeof(float), hipMemcpyHostToDevice));
  return F5HIP_OK;
}

int finalize_impl(f5hip_bigvgan* v) {
  const auto& c = v->cfg;
  for (const auto& t : v->tensors)
    if (!t.loaded) FAIL(F5HIP_ERR_STATE, "tensor '%s' was never loaded", t.name.c_str());
  HIPCHK(init_gemm_kernels());
  HIPCHK(init_bigvgan_kernels());
  bv_kaiser_sinc_12(v->filt);
  CHK(make_conv(v, v->conv_pre, "conv_pre", 1));
  v->ups.assign(c.num_upsamples, ConvW{});
  v->res.assign(c.num_upsamples, std::vector<ResW>(c.num_kernels));
  for (int i = 0; i < c.num_upsamples; ++i) {
    CHK(make_convt(v, v->ups[i], "ups." + std::to_string(i) + ".0", c.upsample_rates[i]));
    for (int j = 0; j < c.num_kernels; ++j) {
      ResW& r = v->res[i][j];
      const std::string p = "resblocks." + std::to_string(i * c.num_kernels + j);
      const int nd = c.resblock_num_dilations[j];
      r.k = c.resblock_kernel_sizes[j];
      r.dil.assign(c.resblock_dilation_sizes[j], c.resblock_dilation_sizes[j] + nd);
      r.c1.assign(nd, ConvW{});
      r.c2.assign(c.resblock == 1 ? nd : 0, ConvW{});
      r.act.assign(c.resblock == 1 ? 2 * nd : nd, ActW{});
      for (int m = 0; m < nd; ++m) {
        if (c.resblock == 1) {
          CHK(make_conv(v, r.c1[m], p + ".convs1." + std::to_string(m), r.dil[m]));
          CHK(make_conv(v, r.c2[m], p + ".convs2." + std::to_string(m), 1));
        } else {
          CHK(make_conv(v, r.c1[m], p + ".convs." + std::to_string(m), r.dil[m]));
        }
      }
      for (size_t q = 0; q < r.act.size(); ++q) CHK(make_act(v, r.act[q], p + ".activations." + std::to_string(q)));
    }
  }
  CHK(make_act(v, v->act_post, "activation_post"));
  {
    const HostTensor& w = T(v, "conv_post.weight");  // [1, C, 7] -> [7, C]
    const int C = (int)w.shape[1];
    v->c_last = C;
    std::vector<float> w7((size_t)7 * C);
    for (int ci = 0; ci < C; ++ci)
      for (int j = 0; j < 7; ++j) w7[(size_t)j * C + ci] = w.data[(size_t)ci * 7 + j];
    HIPCHK(v->post_w7.ensure(w7.size() * sizeof(float)));
    HIPCHK(hipMemcpy(v->post_w7.p, w7.data(), w7.size() * sizeof(float), hipMemcpyHostToDevice));
    if (c.use_bias_at_final) {
      HIPCHK(v->post_bias.ensure(4 * sizeof(float)));
      HIPCHK(hipMemcpy(v->post_bias.p, T(v, "conv_post.bias").data.data(), sizeof(float), hipMemcpyHostToDevice));
    }
  }
  HIPCHK(hipDeviceSynchronize());
  for (auto& t : v->tensors) { t.data.clear(); t.data.shrink_to_fit(); }  // the device copies are the weights now
  v->finalized = true;
  return F5HIP_OK;
}

int op_of(int precision) { return precision == F5HIP_PREC_FP32 ? OP_F32 : precision == F5HIP_PREC_FP16 ? OP_F16 : OP_F16X3; }

// one conv: src [B, L, cin] (strides given) -> dst [B, L, N] (+ res)
//   conv_impl 0: tap-gathered operand [L, ntaps * cpad] (launch_im2col_taps) + the plain GEMM — k-fold write amplification, but built
//                from the kernels the DiT path has exercised; the cross-check of
//   conv_impl 1: ONE operand copy [L, cpad] + the implicit-GEMM kernel with tap-shifted rows (conv_gemm.h)
//   conv_impl 2: as 1, and the Activation1d in front of a resblock conv writes that operand copy itself (src == nullptr here)
bool can_implicit(const f5hip_bigvgan* v, const ConvW& cw, int op) { return v->conv_impl >= 1 && (op != OP_F16 || cw.cpad % 64 == 0); }

int run_conv(f5hip_bigvgan* v, const ConvW& cw, int op, const float* src, int64_t sb, int64_t sl, int64_t sc, int B, int L, float* dst,
             const float* res, hipStream_t st) {
  const int mul = op == OP_F16X3 ? 2 : 1;
  const bool implicit = can_implicit(v, cw, op);
  const int64_t ld = (int64_t)(implicit ? cw.cpad : cw.K) * mul;
  const double esz = op == OP_F16 ? 2.0 : 4.0;  // operand bytes per element (packed hi/lo = 4)
  if (!src) {  // the operand copy is already in v->col (fused emission)
    if (!implicit) FAIL(F5HIP_ERR_STATE, "internal: fused operand without the implicit-GEMM path");
  } else {
    const int taps = implicit ? 1 : cw.ntaps;
    BvProf pr(v, st, BV_OPERAND, 0, (double)B * L * ((double)cw.cin * 4.0 + (double)taps * cw.cpad * esz));
    if (implicit) HIPCHK(launch_im2col_taps(src, sb, sl, sc, B, L, cw.cin, 1, 0, 1, cw.cpad, op, v->col.p, ld, (int64_t)L * ld, st));
    else HIPCHK(launch_im2col_taps(src, sb, sl, sc, B, L, cw.cin, cw.ntaps, cw.shift0, cw.dstep, cw.cpad, op, v->col.p, ld, (int64_t)L * ld, st));
  }
  GemmCore g{};
  g.A = v->col.p;
  g.W = op == OP_F32 ? (const void*)cw.w32.p : op == OP_F16 ? (const void*)cw.w16.p : (const void*)cw.wpk.p;
  g.lda = ld; g.ldw = (int64_t)cw.K * mul; g.strideA = (int64_t)L * ld; g.strideW = 0;
  g.M = L; g.N = cw.N; g.K = cw.K; g.a_rows = L; g.w_rows = cw.N;
  EpiStore e{};
  e.alpha = 1.f; e.act = ACT_NONE; e.bias = cw.bias.as<float>(); e.out32 = dst; e.ldo = cw.N; e.ldres = cw.N; e.res = res;
  e.zdiv = 1; e.so1 = (int64_t)L * cw.N; e.so2 = 0;
  {  // algorithmic work: 2 L N (taps cin) FLOP; bytes: the operand as read + weights + result (+ residual)
    BvProf pr(v, st, BV_GEMM, 2.0 * B * L * cw.N * (double)cw.ntaps * cw.cin,
              (double)B * L * ((implicit ? cw.cpad : cw.K) * esz + cw.N * 4.0 * (res ? 2 : 1)) + (double)cw.N * cw.K * esz);
    if (implicit) HIPCHK(launch_conv_gemm(op, g, cw.ntaps, cw.shift0, cw.dstep, cw.cpad, e, B, st));
    else HIPCHK(launch_gemm_store_variant(op, g, e, B, cw.N <= 64 ? 1 : -1, st));  // narrow outputs (48 / 24 channels): the 128x64 tile
  }
  return F5HIP_OK;
}

// Activation1d followed by a conv of the same resblock: y = act(x); dst = conv(y) (+ res)
int run_act_conv(f5hip_bigvgan* v, const ActW& a, const ConvW& cw, int op, const float* x, int B, int L, int C, float* dst, const float* res,
                 hipStream_t st) {
  const int ls = v->cfg.snake_logscale;
  if (v->conv_impl == 2 && can_implicit(v, cw, op)) {
    {
      BvProf pr(v, st, BV_ACT, 0, (double)B * L * (C * 4.0 + cw.cpad * (op == OP_F16 ? 2.0 : 4.0)));
      HIPCHK(launch_aa_snake(x, nullptr, a.alpha.as<float>(), a.beta.as<float>(), v->filt, B, L, C, ls, st, v->col.p, op, cw.cpad));
    }
    return run_conv(v, cw, op, nullptr, 0, 0, 0, B, L, dst, res, st);
  }
  float* y = v->yy.as<float>();
  {
    BvProf pr(v, st, BV_ACT, 0, (double)B * L * C * 8.0);
    HIPCHK(launch_aa_snake(x, y, a.alpha.as<float>(), a.beta.as<float>(), v->filt, B, L, C, ls, st));
  }
  return run_conv(v, cw, op, y, (int64_t)L * C, C, 1, B, L, dst, res, st);
}

int forward_impl(f5hip_bigvgan* v, const float* mel, int B, int T, int channel_major, int precision, float* out, hipStream_t st) {
  const auto& c = v->cfg;
  const int op = op_of(precision);
  // workspace: the largest stage tensor and the largest operand
  int64_t max_act = (int64_t)T * c.upsample_initial_channel, max_col = (int64_t)T * v->conv_pre.K;
  {
    int64_t L = T;
    for (int i = 0; i < c.num_upsamples; ++i) {
      max_col = std::max(max_col, L * v->ups[i].K);
      max_act = std::max(max_act, L * v->ups[i].N);
      L *= c.upsample_rates[i];
      for (const ResW& r : v->res[i])
        for (const ConvW& cw : r.c1) max_col = std::max(max_col, L * cw.K);
    }
  }
  const size_t act_bytes = (size_t)B * max_act * sizeof(float);
  HIPCHK(v->xa.ensure(act_bytes));
  HIPCHK(v->xb.ensure(act_bytes));
  HIPCHK(v->tt.ensure(act_bytes));
  HIPCHK(v->yy.ensure(act_bytes));
  for (int j = 0; j < c.num_kernels; ++j) HIPCHK(v->rr[j].ensure(act_bytes));
  HIPCHK(v->col.ensure((size_t)B * max_col * (op == OP_F32 ? 4 : op == OP_F16 ? 2 : 4)));

  float* x = v->xa.as<float>();
  float* xn = v->xb.as<float>();
  float* y = v->yy.as<float>();
  float* tt = v->tt.as<float>();
  // conv_pre on the mel in the caller's layout
  if (channel_major) CHK(run_conv(v, v->conv_pre, op, mel, (int64_t)c.num_mels * T, 1, T, B, T, x, nullptr, st));
  else CHK(run_conv(v, v->conv_pre, op, mel, (int64_t)c.num_mels * T, c.num_mels, 1, B, T, x, nullptr, st));
  int L = T, C = c.upsample_initial_channel;
  if (v->stop_after_stage == 0) {
    HIPCHK(hipMemcpyAsync(out, x, (size_t)B * L * C * sizeof(float), hipMemcpyDeviceToDevice, st));
    return F5HIP_OK;
  }
  for (int i = 0; i < c.num_upsamples; ++i) {
    // x [B, L, C] -> xn [B, L, u * C'] == [B, L * u, C']
    CHK(run_conv(v, v->ups[i], op, x, (int64_t)L * C, C, 1, B, L, xn, nullptr, st));
    L *= c.upsample_rates[i];
    C = stage_channels(c, i);
    const int64_t sb = (int64_t)L * C;
    const float* rptr[4] = {nullptr, nullptr, nullptr, nullptr};
    for (int j = 0; j < c.num_kernels; ++j) {
      const ResW& r = v->res[i][j];
      float* rj = v->rr[j].as<float>();
      const float* rin = xn;  // the block's running stream: the stage input first, its own buffer afterwards
      for (size_t m = 0; m < r.c1.size(); ++m) {
        if (c.resblock == 1) {
          CHK(run_act_conv(v, r.act[2 * m], r.c1[m], op, rin, B, L, C, tt, nullptr, st));
          CHK(run_act_conv(v, r.act[2 * m + 1], r.c2[m], op, tt, B, L, C, rj, rin, st));  // x = xt + x
        } else {
          CHK(run_act_conv(v, r.act[m], r.c1[m], op, rin, B, L, C, rj, rin, st));
        }
        rin = rj;
      }
      rptr[j] = rj;
    }
    {
      BvProf pr(v, st, BV_OTHER, 0, (double)B * sb * 4.0 * (c.num_kernels + 1));
      HIPCHK(launch_mean_streams(rptr, c.num_kernels, (int64_t)B * sb, x, st));  // x = xs / num_kernels
    }
    if (v->stop_after_stage == i + 1) {
      HIPCHK(hipMemcpyAsync(out, x, (size_t)B * sb * sizeof(float), hipMemcpyDeviceToDevice, st));
      return F5HIP_OK;
    }
  }
  {
    BvProf pr(v, st, BV_ACT, 0, (double)B * L * C * 8.0);
    HIPCHK(launch_aa_snake(x, y, v->act_post.alpha.as<float>(), v->act_post.beta.as<float>(), v->filt, B, L, C, c.snake_logscale, st));
  }
  {
    BvProf pr(v, st, BV_OTHER, 2.0 * B * L * 7.0 * C, (double)B * L * (C + 1) * 4.0);
    HIPCHK(launch_conv_post(y, v->post_w7.as<float>(), c.use_bias_at_final ? v->post_bias.as<float>() : nullptr, B, L, C, c.use_tanh_at_final, out, st));
  }
  return F5HIP_OK;
}

}  // namespace

extern "C" {

const char* f5hip_bigvgan_last_error(const f5hip_bigvgan* v) { return v ? v->err.c_str() : g_bv_create_err.c_str(); }

int f5hip_bigvgan_create(const f5hip_bigvgan_config* c, int device, f5hip_bigvgan** out) {
  if (!c || !out) { g_bv_create_err = "null argument"; return F5HIP_ERR_INVALID; }
  *out = nullptr;
  const auto bad = [&](const char* m) { g_bv_create_err = m; return F5HIP_ERR_INVALID; };
  if (c->num_mels <= 0 || c->num_mels % 4) return bad("num_mels must be a positive multiple of 4");
  if (c->num_upsamples < 1 || c->num_upsamples > 8) return bad("num_upsamples must be in [1, 8]");
  if (c->num_kernels < 1 || c->num_kernels > 4) return bad("num_kernels must be in [1, 4]");
  if (c->resblock != 1 && c->resblock != 2) return bad("resblock must be 1 (AMPBlock1) or 2 (AMPBlock2)");
  if (c->activation != 0 && c->activation != 1) return bad("activation must be 0 (snake) or 1 (snakebeta)");
  if (c->upsample_initial_channel <= 0 || (c->upsample_initial_channel >> c->num_upsamples) << c->num_upsamples != c->upsample_initial_channel)
    return bad("upsample_initial_channel must be divisible by 2^num_upsamples");
  if ((c->upsample_initial_channel >> c->num_upsamples) % 4) return bad("the last stage must keep a multiple of 4 channels");
  for (int i = 0; i < c->num_upsamples; ++i) {
    const int u = c->upsample_rates[i], k = c->upsample_kernel_sizes[i];
    if (u < 1 || k < u || (k - u) % 2) return bad("upsample_kernel_sizes[i] - upsample_rates[i] must be even and >= 0 (output length = frames * rate)");
  }
  for (int j = 0; j < c->num_kernels; ++j) {
    if (c->resblock_kernel_sizes[j] < 1 || c->resblock_kernel_sizes[j] % 2 == 0) return bad("resblock kernel sizes must be odd");
    if (c->resblock_num_dilations[j] < 1 || c->resblock_num_dilations[j] > 4) return bad("1 to 4 dilations per resblock");
    for (int m = 0; m < c->resblock_num_dilations[j]; ++m)
      if (c->resblock_dilation_sizes[j][m] < 1) return bad("dilations must be >= 1");
  }
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) {
    g_bv_create_err = "no such HIP device (libf5hip has no CPU fallback)";
    return F5HIP_ERR_HIP;
  }
  if (hipSetDevice(device) != hipSuccess) { g_bv_create_err = "hipSetDevice failed"; return F5HIP_ERR_HIP; }
  f5hip_bigvgan* v = new f5hip_bigvgan();
  v->cfg = *c;
  v->device = device;
  build_tensor_table(v);
  *out = v;
  return F5HIP_OK;
}

int f5hip_bigvgan_destroy(f5hip_bigvgan* v) {
  if (!v) return F5HIP_OK;
  (void)hipSetDevice(v->device);
  (void)hipDeviceSynchronize();
  auto free_conv = [](ConvW& c) { c.w32.release(); c.w16.release(); c.wpk.release(); c.bias.release(); };
  auto free_act = [](ActW& a) { a.alpha.release(); a.beta.release(); };
  free_conv(v->conv_pre);
  for (auto& u : v->ups) free_conv(u);
  for (auto& stage : v->res)
    for (auto& r : stage) {
      for (auto& c : r.c1) free_conv(c);
      for (auto& c : r.c2) free_conv(c);
      for (auto& a : r.act) free_act(a);
    }
  free_act(v->act_post);
  DevBuf* bufs[] = {&v->post_w7, &v->post_bias, &v->xa, &v->xb, &v->tt, &v->yy, &v->col, &v->rr[0], &v->rr[1], &v->rr[2], &v->rr[3]};
  for (DevBuf* b : bufs) b->release();
  delete v;
  return F5HIP_OK;
}

int f5hip_bigvgan_num_tensors(const f5hip_bigvgan* v) { return v ? (int)v->tensors.size() : 0; }

int f5hip_bigvgan_tensor_info(const f5hip_bigvgan* v, int i, const char** name, int64_t* numel) {
  if (!v || i < 0 || i >= (int)v->tensors.size()) return F5HIP_ERR_INVALID;
  if (name) *name = v->tensors[i].name.c_str();
  if (numel) *numel = v->tensors[i].numel();
  return F5HIP_OK;
}

int f5hip_bigvgan_load_tensor(f5hip_bigvgan* v, const char* name, const float* data, int64_t numel) {
  if (!v) return F5HIP_ERR_INVALID;
  std::lock_guard<std::mutex> lk(v->mu);
  if (!name || !data) FAIL(F5HIP_ERR_INVALID, "null argument");
  auto it = v->index.find(name);
  if (it == v->index.end()) FAIL(F5HIP_ERR_INVALID, "unknown tensor '%s'", name);
  HostTensor& t = v->tensors[it->second];
  if (numel != t.numel()) FAIL(F5HIP_ERR_INVALID, "tensor '%s': expected %lld elements, got %lld", name, (long long)t.numel(), (long long)numel);
  t.data.assign(data, data + numel);
  t.loaded = true;
  v->finalized = false;
  return F5HIP_OK;
}

int f5hip_bigvgan_finalize(f5hip_bigvgan* v) {
  if (!v) return F5HIP_ERR_INVALID;
  std::lock_guard<std::mutex> lk(v->mu);
  HIPCHK(hipSetDevice(v->device));
  return finalize_impl(v);
}

int f5hip_bigvgan_set_option(f5hip_bigvgan* v, const char* key, int64_t value) {
  if (!v) return F5HIP_ERR_INVALID;
  std::lock_guard<std::mutex> lk(v->mu);
  if (!key) FAIL(F5HIP_ERR_INVALID, "null key");
  if (!strcmp(key, "stop_after_stage")) {
    if (value < -1 || value > v->cfg.num_upsamples) FAIL(F5HIP_ERR_INVALID, "stop_after_stage must be in [-1, num_upsamples]");
    v->stop_after_stage = (int)value;
    return F5HIP_OK;
  }
  if (!strcmp(key, "profile")) { v->profile = value != 0; return F5HIP_OK; }
  if (!strcmp(key, "conv_impl")) {
    if (value < 0 || value > 2) FAIL(F5HIP_ERR_INVALID, "conv_impl must be 0, 1 or 2");
    v->conv_impl = (int)value;
    return F5HIP_OK;
  }
  FAIL(F5HIP_ERR_INVALID, "unknown option '%s'", key);
}

int f5hip_bigvgan_num_kernel_stats(const f5hip_bigvgan*) { return BV_COUNT; }

int f5hip_bigvgan_kernel_stat(const f5hip_bigvgan* v, int i, const char** name, int64_t* calls, double* ms, double* flops, double* bytes) {
  if (!v || i < 0 || i >= BV_COUNT) return F5HIP_ERR_INVALID;
  if (name) *name = BV_NAMES[i];
  if (calls) *calls = v->stats[i].calls;
  if (ms) *ms = v->stats[i].ms;
  if (flops) *flops = v->stats[i].flops;
  if (bytes) *bytes = v->stats[i].bytes;
  return F5HIP_OK;
}

int f5hip_bigvgan_reset_kernel_stats(f5hip_bigvgan* v) {
  if (!v) return F5HIP_ERR_INVALID;
  std::lock_guard<std::mutex> lk(v->mu);
  for (auto& k : v->stats) k = KStat{};
  return F5HIP_OK;
}

int f5hip_bigvgan_forward(f5hip_bigvgan* v, const float* mel, int batch, int frames, int channel_major, int precision, float* out,
                          void* stream) {
  if (!v) return F5HIP_ERR_INVALID;
  std::lock_guard<std::mutex> lk(v->mu);
  if (!v->finalized) FAIL(F5HIP_ERR_STATE, "weights not finalised (f5hip_bigvgan_finalize)");
  if (!mel || !out || batch <= 0 || frames <= 0) FAIL(F5HIP_ERR_INVALID, "bad argument: mel/out null or batch/frames <= 0");
  if (precision < F5HIP_PREC_FP32 || precision > F5HIP_PREC_FP16M) FAIL(F5HIP_ERR_INVALID, "bad precision %d", precision);
  if (precision == F5HIP_PREC_FP16M) precision = F5HIP_PREC_FP16X3;  // the MX lines are the DiT / UNetT block GEMMs': the conv GEMMs run the three-term product
  HIPCHK(hipSetDevice(v->device));
  const int rc = forward_impl(v, mel, batch, frames, channel_major, precision, out, reinterpret_cast<hipStream_t>(stream));
  bv_collect(v, reinterpret_cast<hipStream_t>(stream));
  return rc;
}

}  // extern "C"
