// api.cpp — C-ABI entry points of libf5hip (include/f5hip.h): context, weight blob, and the
// enqueue logic of the sampler / mel / vocoder pipelines.  No torch types, no CPU fallback: every
// numeric step is a launch of a kernel from gemm/elementwise/convpos/attention/audio.hip.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>

#include "engine.h"

namespace {

thread_local std::string g_create_err;

#define HIPCHK(expr)                                                                                   \
  do {                                                                                                 \
    hipError_t _e = (expr);                                                                            \
    if (_e != hipSuccess) {                                                                            \
      char _b[512];                                                                                    \
      snprintf(_b, sizeof(_b), "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
      ctx->err = _b;                                                                                   \
      return F5HIP_ERR_HIP;                                                                            \
    }                                                                                                  \
  } while (0)

#define FAIL(code, ...)                          \
  do {                                           \
    char _b[512];                                \
    snprintf(_b, sizeof(_b), __VA_ARGS__);       \
    ctx->err = _b;                               \
    return code;                                 \
  } while (0)

#define CHK(expr)                   \
  do {                              \
    int _r = (expr);                \
    if (_r != F5HIP_OK) return _r;  \
  } while (0)

// typed pinned staging of the current call (engine.h HostStage): the queued copy reads it after the entry point returned
template <typename T>
T* stage(f5hip_ctx* ctx, size_t n) { return reinterpret_cast<T*>(ctx->stage.alloc(n * sizeof(T))); }
#define STAGE(T, var, n)                                                    \
  T* var = stage<T>(ctx, (n));                                              \
  if (!var) FAIL(F5HIP_ERR_HIP, "pinned staging allocation of %zu bytes failed%s%s%s%s", (size_t)(n) * sizeof(T),                               \
                 ctx->stage.last_error != hipSuccess ? ": " : "", ctx->stage.last_error != hipSuccess ? ctx->stage.last_step : "",                 \
                 ctx->stage.last_error != hipSuccess ? ": " : "", ctx->stage.last_error != hipSuccess ? hipGetErrorString(ctx->stage.last_error) : "")

// Every entry point that enqueues work calls this first: device, the staging slot of this call, and — the workspace being one per context —
// a GPU-side wait for the previous call's work when this call arrives on a different stream (no host wait in either case).
int call_begin(f5hip_ctx* ctx, hipStream_t st) {
  HIPCHK(hipSetDevice(ctx->device));
  HIPCHK(ctx->stage.begin());
  if (ctx->have_last && ctx->last_stream != st) HIPCHK(hipStreamWaitEvent(st, ctx->ev_last, 0));
  return F5HIP_OK;
}
int call_end(f5hip_ctx* ctx, hipStream_t st) {
  HIPCHK(ctx->stage.end(st));
  if (!ctx->ev_last) HIPCHK(hipEventCreateWithFlags(&ctx->ev_last, hipEventDisableTiming));
  HIPCHK(hipEventRecord(ctx->ev_last, st));
  ctx->last_stream = st;
  ctx->have_last = true;
  return F5HIP_OK;
}
// begin() / finish() of one entry point; the destructor finishes a call that leaves through an error return, so that whatever it had
// already queued on `st` is ordered before the next call's work on the shared workspace (ev_last) and its staging slot gets its event.
struct CallScope {
  f5hip_ctx* ctx;
  hipStream_t st;
  bool open = false;
  CallScope(f5hip_ctx* c, hipStream_t s) : ctx(c), st(s) {}
  CallScope(const CallScope&) = delete;
  CallScope& operator=(const CallScope&) = delete;
  int begin() {
    const int r = call_begin(ctx, st);
    open = r == F5HIP_OK || ctx->stage.in_call;
    return r;
  }
  int finish() {
    open = false;
    return call_end(ctx, st);
  }
  ~CallScope() {
    if (open) (void)call_end(ctx, st);
  }
};

int add_slot(f5hip_ctx* ctx, const std::string& name, int64_t numel, bool optional = false) {
  Slot s;
  s.name = name;
  s.numel = numel;
  s.offset = ctx->blob_elems;
  s.optional = optional;
  ctx->index[name] = (int)ctx->slots.size();
  ctx->slots.push_back(s);
  ctx->blob_elems += (numel + 3) & ~int64_t(3);  // keep every tensor 16-byte aligned
  return (int)ctx->slots.size() - 1;
}

const float* W(const f5hip_ctx* ctx, const std::string& name) {
  auto it = ctx->index.find(name);
  if (it == ctx->index.end()) return nullptr;
  return ctx->blob + ctx->slots[it->second].offset;
}

// Blob order is chosen so that fused operands are contiguous: [to_q|to_k|to_v] per block, and the
// AdaLN linears of ALL blocks back to back (one GEMM computes every block's modulation for every step).
void add_vocos_slots(f5hip_ctx* ctx);

// MMDiT (reference backbones/mmdit.py:112-134, modules.py:773-814).  Order matters in two places: to_q|to_k|to_v (and their biases) are
// contiguous so one GEMM computes the fused projection, and the AdaLN linears of all blocks are contiguous per stream so one GEMM per
// stream produces every block's modulation for a time step.
void build_slots_mmdit(f5hip_ctx* ctx) {
  const auto& c = ctx->cfg;
  const int64_t D = c.dim, mel = c.mel_dim, inner = (int64_t)c.heads * c.dim_head, F = c.ff_inner;
  const std::string p = "transformer.";
  add_slot(ctx, p + "time_embed.time_mlp.0.weight", D * 256);
  add_slot(ctx, p + "time_embed.time_mlp.0.bias", D);
  add_slot(ctx, p + "time_embed.time_mlp.2.weight", D * D);
  add_slot(ctx, p + "time_embed.time_mlp.2.bias", D);
  add_slot(ctx, p + "text_embed.text_embed.weight", (int64_t)(c.text_num_embeds + 1) * D);
  add_slot(ctx, p + "audio_embed.linear.weight", D * 2 * mel);
  add_slot(ctx, p + "audio_embed.linear.bias", D);
  const int64_t cpg = D / c.conv_pos_groups;
  for (int j = 0; j < 2; ++j) {
    const std::string b = p + "audio_embed.conv_pos_embed.conv1d." + std::to_string(2 * j) + ".";
    add_slot(ctx, b + "weight", D * cpg * c.conv_pos_kernel);
    add_slot(ctx, b + "bias", D);
  }
  add_slot(ctx, p + "rotary_embed.inv_freq", c.dim_head / 2, /*optional=*/true);
  auto blk = [&](int i) { return p + "transformer_blocks." + std::to_string(i) + "."; };
  for (int i = 0; i < c.depth; ++i) add_slot(ctx, blk(i) + "attn_norm_x.linear.weight", 6 * D * D);
  for (int i = 0; i < c.depth; ++i) add_slot(ctx, blk(i) + "attn_norm_x.linear.bias", 6 * D);
  for (int i = 0; i < c.depth; ++i) add_slot(ctx, blk(i) + "attn_norm_c.linear.weight", (i == c.depth - 1 ? 2 : 6) * D * D);
  for (int i = 0; i < c.depth; ++i) add_slot(ctx, blk(i) + "attn_norm_c.linear.bias", (i == c.depth - 1 ? 2 : 6) * D);
  for (int i = 0; i < c.depth; ++i) {
    const std::string b = blk(i);
    const bool last = i == c.depth - 1;
    for (const char* sfx : {"", "_c"}) {
      for (const char* w : {"to_q", "to_k", "to_v"}) add_slot(ctx, b + "attn." + w + sfx + ".weight", inner * D);
      for (const char* w : {"to_q", "to_k", "to_v"}) add_slot(ctx, b + "attn." + w + sfx + ".bias", inner);
    }
    add_slot(ctx, b + "attn.to_out.0.weight", D * inner);
    add_slot(ctx, b + "attn.to_out.0.bias", D);
    if (!last) {
      add_slot(ctx, b + "attn.to_out_c.weight", D * inner);
      add_slot(ctx, b + "attn.to_out_c.bias", D);
    }
    if (c.qk_norm)
      for (const char* w : {"q_norm", "k_norm", "c_q_norm", "c_k_norm"}) add_slot(ctx, b + "attn." + w + ".weight", c.dim_head);
    for (const char* ff : {"ff_x", "ff_c"}) {
      if (last && ff[3] == 'c') continue;
      add_slot(ctx, b + ff + ".ff.0.0.weight", F * D);
      add_slot(ctx, b + ff + ".ff.0.0.bias", F);
      add_slot(ctx, b + ff + ".ff.2.weight", D * F);
      add_slot(ctx, b + ff + ".ff.2.bias", D);
    }
  }
  add_slot(ctx, p + "norm_out.linear.weight", 2 * D * D);
  add_slot(ctx, p + "norm_out.linear.bias", 2 * D);
  add_slot(ctx, p + "proj_out.weight", mel * D);
  add_slot(ctx, p + "proj_out.bias", mel);
  add_slot(ctx, p + "text_embed.freqs_cis", 1024 * D, /*optional=*/true);  // non-persistent buffer (mmdit.py:40)
  ctx->dit_elems = ctx->blob_elems;
  add_vocos_slots(ctx);
}

void build_slots(f5hip_ctx* ctx) {
  const auto& c = ctx->cfg;
  if (c.backbone == 2) return build_slots_mmdit(ctx);
  const int64_t D = c.dim, T = c.text_dim, mel = c.mel_dim, inner = (int64_t)c.heads * c.dim_head, F = c.ff_inner;
  const std::string p = "transformer.";
  add_slot(ctx, p + "time_embed.time_mlp.0.weight", D * 256);
  add_slot(ctx, p + "time_embed.time_mlp.0.bias", D);
  add_slot(ctx, p + "time_embed.time_mlp.2.weight", D * D);
  add_slot(ctx, p + "time_embed.time_mlp.2.bias", D);
  add_slot(ctx, p + "text_embed.text_embed.weight", (int64_t)(c.text_num_embeds + 1) * T);
  for (int i = 0; i < c.conv_layers; ++i) {
    const std::string b = p + "text_embed.text_blocks." + std::to_string(i) + ".";
    add_slot(ctx, b + "dwconv.weight", T * 7);
    add_slot(ctx, b + "dwconv.bias", T);
    add_slot(ctx, b + "norm.weight", T);
    add_slot(ctx, b + "norm.bias", T);
    add_slot(ctx, b + "pwconv1.weight", 2 * T * T);
    add_slot(ctx, b + "pwconv1.bias", 2 * T);
    add_slot(ctx, b + "grn.gamma", 2 * T);
    add_slot(ctx, b + "grn.beta", 2 * T);
    add_slot(ctx, b + "pwconv2.weight", 2 * T * T);
    add_slot(ctx, b + "pwconv2.bias", T);
  }
  add_slot(ctx, p + "input_embed.proj.weight", D * (2 * mel + T));
  add_slot(ctx, p + "input_embed.proj.bias", D);
  const int64_t cpg = D / c.conv_pos_groups;
  for (int j = 0; j < 2; ++j) {
    const std::string b = p + "input_embed.conv_pos_embed.conv1d." + std::to_string(2 * j) + ".";
    add_slot(ctx, b + "weight", D * cpg * c.conv_pos_kernel);
    add_slot(ctx, b + "bias", D);
  }
  add_slot(ctx, p + "rotary_embed.inv_freq", c.dim_head / 2, /*optional=*/true);
  if (c.backbone == 1) {  // UNetT: reference unett.py:147-186 -> keys layers.{i}.{0: skip_proj, 1: attn_norm, 2: attn, 3: ff_norm, 4: ff}
    for (int i = 0; i < c.depth; ++i) {
      const std::string b = p + "layers." + std::to_string(i) + ".";
      if (i >= c.depth / 2 && c.skip_connect_type == 0) add_slot(ctx, b + "0.weight", D * 2 * D);
      add_slot(ctx, b + "1.g", D);
      add_slot(ctx, b + "2.to_q.weight", inner * D);
      add_slot(ctx, b + "2.to_k.weight", inner * D);
      add_slot(ctx, b + "2.to_v.weight", inner * D);
      add_slot(ctx, b + "2.to_q.bias", inner);
      add_slot(ctx, b + "2.to_k.bias", inner);
      add_slot(ctx, b + "2.to_v.bias", inner);
      add_slot(ctx, b + "2.to_out.0.weight", D * inner);
      add_slot(ctx, b + "2.to_out.0.bias", D);
      if (c.qk_norm) {
        add_slot(ctx, b + "2.q_norm.weight", c.dim_head);
        add_slot(ctx, b + "2.k_norm.weight", c.dim_head);
      }
      add_slot(ctx, b + "3.g", D);
      add_slot(ctx, b + "4.ff.0.0.weight", F * D);
      add_slot(ctx, b + "4.ff.0.0.bias", F);
      add_slot(ctx, b + "4.ff.2.weight", D * F);
      add_slot(ctx, b + "4.ff.2.bias", D);
    }
    add_slot(ctx, p + "norm_out.g", D);
    add_slot(ctx, p + "proj_out.weight", mel * D);
    add_slot(ctx, p + "proj_out.bias", mel);
  }
  for (int i = 0; i < (c.backbone == 1 ? 0 : c.depth); ++i) add_slot(ctx, p + "transformer_blocks." + std::to_string(i) + ".attn_norm.linear.weight", 6 * D * D);
  const int dit_depth = c.backbone == 1 ? 0 : c.depth;
  for (int i = 0; i < dit_depth; ++i) add_slot(ctx, p + "transformer_blocks." + std::to_string(i) + ".attn_norm.linear.bias", 6 * D);
  for (int i = 0; i < dit_depth; ++i) {
    const std::string b = p + "transformer_blocks." + std::to_string(i) + ".";
    add_slot(ctx, b + "attn.to_q.weight", inner * D);
    add_slot(ctx, b + "attn.to_k.weight", inner * D);
    add_slot(ctx, b + "attn.to_v.weight", inner * D);
    add_slot(ctx, b + "attn.to_q.bias", inner);
    add_slot(ctx, b + "attn.to_k.bias", inner);
    add_slot(ctx, b + "attn.to_v.bias", inner);
    add_slot(ctx, b + "attn.to_out.0.weight", D * inner);
    add_slot(ctx, b + "attn.to_out.0.bias", D);
    if (c.qk_norm) {
      add_slot(ctx, b + "attn.q_norm.weight", c.dim_head);
      add_slot(ctx, b + "attn.k_norm.weight", c.dim_head);
    }
    add_slot(ctx, b + "ff.ff.0.0.weight", F * D);
    add_slot(ctx, b + "ff.ff.0.0.bias", F);
    add_slot(ctx, b + "ff.ff.2.weight", D * F);
    add_slot(ctx, b + "ff.ff.2.bias", D);
  }
  if (c.backbone != 1) {
    if (c.long_skip_connection) add_slot(ctx, p + "long_skip_connection.weight", D * 2 * D);
    add_slot(ctx, p + "norm_out.linear.weight", 2 * D * D);
    add_slot(ctx, p + "norm_out.linear.bias", 2 * D);
    add_slot(ctx, p + "proj_out.weight", mel * D);
    add_slot(ctx, p + "proj_out.bias", mel);
  }
  add_slot(ctx, p + "text_embed.freqs_cis", 8192 * T, /*optional=*/true);  // non-persistent buffer (dit.py:48)
  ctx->dit_elems = ctx->blob_elems;
  add_vocos_slots(ctx);
}

void add_vocos_slots(f5hip_ctx* ctx) {
  if (ctx->has_vocos) {
    const auto& v = ctx->vcfg;
    const int64_t C = v.dim, I = v.intermediate_dim;
    add_slot(ctx, "backbone.embed.weight", C * v.input_channels * 7);
    add_slot(ctx, "backbone.embed.bias", C);
    add_slot(ctx, "backbone.norm.weight", C);
    add_slot(ctx, "backbone.norm.bias", C);
    for (int i = 0; i < v.num_layers; ++i) {
      const std::string b = "backbone.convnext." + std::to_string(i) + ".";
      add_slot(ctx, b + "dwconv.weight", C * 7);
      add_slot(ctx, b + "dwconv.bias", C);
      add_slot(ctx, b + "norm.weight", C);
      add_slot(ctx, b + "norm.bias", C);
      add_slot(ctx, b + "pwconv1.weight", I * C);
      add_slot(ctx, b + "pwconv1.bias", I);
      add_slot(ctx, b + "pwconv2.weight", C * I);
      add_slot(ctx, b + "pwconv2.bias", C);
      add_slot(ctx, b + "gamma", C);
    }
    add_slot(ctx, "backbone.final_layer_norm.weight", C);
    add_slot(ctx, "backbone.final_layer_norm.bias", C);
    add_slot(ctx, "head.out.weight", (int64_t)(v.n_fft + 2) * C);
    add_slot(ctx, "head.out.bias", v.n_fft + 2);
    add_slot(ctx, "head.istft.window", v.n_fft, /*optional=*/true);
  }
}

bool slot_loaded(const f5hip_ctx* ctx, const std::string& name) {
  auto it = ctx->index.find(name);
  return it != ctx->index.end() && ctx->slots[it->second].loaded;
}

// ---- measurement -------------------------------------------------------------------------------
struct Prof {
  f5hip_ctx* ctx;
  hipStream_t s;
  int kc;
  double flops, bytes;
  ProfRec rec{};
  bool on;
  Prof(f5hip_ctx* c, hipStream_t st, int kclass, double fl, double by) : ctx(c), s(st), kc(kclass), flops(fl), bytes(by), on(c->profile) {
    if (on) {
      rec.kclass = kc;
      rec.flops = fl;
      rec.bytes = by;
      (void)hipEventCreate(&rec.e0);
      (void)hipEventCreate(&rec.e1);
      (void)hipEventRecord(rec.e0, s);
    }
  }
  ~Prof() {
    if (on) {
      (void)hipEventRecord(rec.e1, s);
      ctx->prof.push_back(rec);
    }
  }
};

void collect_prof(f5hip_ctx* ctx, hipStream_t s) {
  if (ctx->prof.empty()) return;
  (void)hipStreamSynchronize(s);
  for (auto& r : ctx->prof) {
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, r.e0, r.e1) == hipSuccess) {
      KStat& k = ctx->stats[r.kclass];
      k.calls += 1;
      k.ms += ms;
      k.flops += r.flops;
      k.bytes += r.bytes;
    }
    (void)hipEventDestroy(r.e0);
    (void)hipEventDestroy(r.e1);
  }
  ctx->prof.clear();
}

const char* kclass_name(int k) {
  static const char* names[KC_COUNT] = {"gemm_block", "attention", "convpos", "ln_modulate", "gemm_misc", "text_embed",
                                        "elementwise", "mel", "vocos_gemm", "vocos_other", "istft"};
  return names[k];
}

// ---- GEMM helpers ------------------------------------------------------------------------------
GemmCore core(const void* A, int64_t lda, const void* Wt, int64_t ldw, int M, int N, int K) {
  GemmCore g{};
  g.A = A; g.A_lo = nullptr; g.W = Wt; g.W_lo = nullptr;
  g.lda = lda; g.ldw = ldw; g.strideA = 0; g.strideW = 0;
  g.M = M; g.N = N; g.K = K; g.a_rows = M; g.w_rows = N;
  return g;
}
// a weight operand as a block GEMM takes it: the rows in the operand mode's format and, for the conditioned half-precision copies, the
// per-output-channel factors that undo the conditioning (GemmCore::w_alpha)
struct WOp {
  const void* w;
  const float* alpha;
};
GemmCore core(const void* A, int64_t lda, const WOp& Wt, int64_t ldw, int M, int N, int K) {
  GemmCore g = core(A, lda, Wt.w, ldw, M, N, K);
  g.w_alpha = Wt.alpha;
  return g;
}
EpiStore epi_store(float* out32, int64_t ldo, const float* bias, int act = ACT_NONE) {
  EpiStore e{};
  e.alpha = 1.f; e.act = act; e.bias = bias; e.out32 = out32; e.ldo = ldo; e.ldres = ldo;
  return e;
}

double gemm_flops(int64_t M, int64_t N, int64_t K) { return 2.0 * (double)M * (double)N * (double)K; }

// One function evaluation of the ODE solve: which time-table row it uses and which state buffers it reads / updates.
struct Stage {
  int eidx;            // row of the per-evaluation time tables (embeddings, AdaLN modulations, update coefficient)
  const float* yin;    // state the backbone is evaluated on            [B*n, mel]
  const float* ybase;  // dst = ybase + coef[eidx] * velocity
  float* ydst;
  float* traj;         // optional copy of dst (trajectory slot) or null
};

// weight operand of a block GEMM in the given operand mode: fp32 blob rows, plain fp16 rows, or packed hi/lo rows
WOp wsel(const f5hip_ctx* ctx, int op, const float* w32, const f16* hi, const f16* pk, const f16* mx = nullptr) {
  if (op == OP_F32) return WOp{w32, nullptr};
  const void* w = op == OP_F16 ? (const void*)hi : op == OP_F16M ? (const void*)mx : (const void*)pk;
  const auto it = ctx->walpha.find(w);
  return WOp{w, it == ctx->walpha.end() ? nullptr : it->second};
}

// the conv-position weights as MX lines for THIS call: fp16m calls (mx_call) whose conv runs in the split mode and whose tiles were packed
inline const f16* conv_mx(const f5hip_ctx* ctx, int op, int j) {
  return ctx->mx_call && op == OP_F16X3 && ctx->conv_wmx[j].p ? ctx->conv_wmx[j].as<f16>() : nullptr;
}

// What the attention SCORES are computed from in the half-precision parity modes (fp16x3, fp16m):
//   QK_PLAIN  fp16 q, k — 1 MFMA per product (attn_impl 3: the default of rounds 2-4; always in the plain fp16 mode)
//   QK_SPLIT  hi/lo-split q and k, 3 MFMAs per product (attn_impl 4; attn_impl 2 splits P and V as well)
//   QK_MX     fp16 hi . hi + both correction products as one MX-fp6 MFMA per 32 head channels, 1.5 MFMA-equivalents (attn_impl 0 / 5: the
//             default since round 5) — where the q|k|v GEMM runs a pipelined kernel, whose epilogue packs the P words; else QK_SPLIT.
// Rounds 2-4 ran plain fp16 scores after measuring <= 2.9e-4 max-abs on every golden minted from seeded Gaussian weights.  Round 5's
// trained-like golden (heavy-tailed weights, per-channel gains: larger logits) moves by 1.1e-3 with them — over the 1e-3 tolerance — by
// 5.8e-4 with split q, k and by 3.7e-4 with everything split (profiles/r05g_attn_precision_fp16m.log): the scores feed an
// exponential, so their ABSOLUTE error counts, and it grows with the logits.
enum { QK_PLAIN = 0, QK_SPLIT = 1, QK_MX = 2 };
// attn_impl 2, 6, 7 read V as hi + lo halves (the q|k|v epilogue writes the second V^T plane for them)
inline bool attn_v_split(const f5hip_ctx* ctx) { return ctx->attn_impl == 2 || ctx->attn_impl == 6 || ctx->attn_impl == 7; }
int qk_scheme_wanted(const f5hip_ctx* ctx, int op) {
  if (op != OP_F16X3 || ctx->attn_impl == 3) return QK_PLAIN;
  return ctx->attn_impl == 2 || ctx->attn_impl == 4 ? QK_SPLIT : QK_MX;
}
// SDPA's default 1/sqrt(dim_head) on q (modules.py:511-520).  The flash kernel takes q with log2(e) folded in as well: its exponentials are
// base-2 (v_exp_f32) and its scores then need no multiply — and the lazy reference maximum becomes possible (attention_kernel.h LAZY).  The
// materialised fp32 path (exact_attn) softmaxes natural-log scores.
float attn_qscale(int dh, bool exact_attn) { return (exact_attn ? 1.0f : 1.4426950408889634f) / sqrtf((float)dh); }


// FP16M: fp16x3 everywhere except the four block GEMMs of the DiT backbone, which read MX lines when ctx->mx_call says so (f5hip_sample)
int op_of(int precision) { return precision == F5HIP_PREC_FP32 ? OP_F32 : precision == F5HIP_PREC_FP16 ? OP_F16 : OP_F16X3; }

// ---- finalize: derived layouts -------------------------------------------------------------------
int finalize_impl(f5hip_ctx* ctx) {
  const auto& c = ctx->cfg;
  const int64_t D = c.dim, T = c.text_dim, inner = (int64_t)c.heads * c.dim_head, F = c.ff_inner;
  for (auto& s : ctx->slots)
    if (!s.loaded && !s.optional) FAIL(F5HIP_ERR_STATE, "tensor '%s' was never loaded", s.name.c_str());
  hipStream_t st = nullptr;
  const std::string p = "transformer.";

  // f16 hi/lo copies of the per-step GEMM weights
  const int64_t per_block = 3 * inner * D + D * inner + F * D + D * F;
  if (D % 32 || inner % 32 || F % 32) FAIL(F5HIP_ERR_UNSUPPORTED, "dim, heads*dim_head and ff_inner must be multiples of 32 (packed fp16x3 operand rows)");
  const bool unett = c.backbone == 1;
  const bool skip_concat = unett && c.skip_connect_type == 0;
  const int64_t skip_elems = skip_concat ? (int64_t)(c.depth / 2) * D * 2 * D : (!unett && c.long_skip_connection) ? D * 2 * D : 0;
  const bool mmdit = c.backbone == 2;
  const int64_t cstream_elems = mmdit ? per_block * (c.depth - 1) + 3 * inner * D : 0;  // text stream; the last block only projects q/k/v
  // fp16m (fp16 + MX-fp6 correction lines, common.h): the four block GEMMs of the DiT / UNetT backbones when every one of them is a launch the
  // pipelined kernel takes (rows of whole 128-byte lines, at least a 3-stage ring of them) and the fused q|k|v epilogue applies
  ctx->mx_ok = ctx->mx_weights_opt && (c.backbone == 0 || c.backbone == 1) && c.dim_head == 64 && !c.qk_norm && !c.long_skip_connection && D / 32 >= 4 && inner / 32 >= 4 && F / 32 >= 4;
  const int64_t mx_elems = ctx->mx_ok ? per_block * c.depth * 2 : 0;
  HIPCHK(ctx->half_pool.ensure((size_t)((per_block * c.depth + skip_elems + cstream_elems) * 3 + mx_elems) * sizeof(f16)));  // plain hi + packed hi/lo (+ MX lines)
  f16* hp = ctx->half_pool.as<f16>();
  // weight conditioning (GemmCore::w_alpha): rows of every half-precision weight copy, scale | alpha each
  const int64_t rows_block = 3 * inner + D + F + D;
  const int64_t cond_rows = rows_block * c.depth + (skip_elems ? (skip_concat ? (int64_t)(c.depth / 2) * D : D) : 0) + (mmdit ? rows_block * c.depth : 0);
  HIPCHK(ctx->cond_pool.ensure((size_t)cond_rows * 2 * sizeof(float)));
  float* cp = ctx->cond_pool.as<float>();
  ctx->walpha.clear();
  const bool no_cond = getenv("F5HIP_NO_WEIGHT_CONDITIONING") != nullptr;  // A/B switch (tests, tools/): round 2's plain split; read at every finalize
  ctx->blocks.assign(c.depth, BlockW{});
  for (int i = 0; i < c.depth; ++i) {
    const std::string b = p + (unett ? "layers." : "transformer_blocks.") + std::to_string(i) + ".";
    const std::string ba = b + (unett ? "2." : "attn."), bf = b + (unett ? "4." : mmdit ? "ff_x." : "ff.");
    BlockW& bw = ctx->blocks[i];
    bw.wqkv = W(ctx, ba + "to_q.weight");
    bw.bqkv = W(ctx, ba + "to_q.bias");
    bw.wo = W(ctx, ba + "to_out.0.weight");
    bw.bo = W(ctx, ba + "to_out.0.bias");
    bw.w1 = W(ctx, bf + "ff.0.0.weight");
    bw.b1 = W(ctx, bf + "ff.0.0.bias");
    bw.w2 = W(ctx, bf + "ff.2.weight");
    bw.b2 = W(ctx, bf + "ff.2.bias");
    if (unett) {
      bw.g_attn = W(ctx, b + "1.g");
      bw.g_ff = W(ctx, b + "3.g");
      bw.wskip = (skip_concat && i >= c.depth / 2) ? W(ctx, b + "0.weight") : nullptr;
    }
    if (c.qk_norm) {
      bw.qn = W(ctx, ba + "q_norm.weight");
      bw.kn = W(ctx, ba + "k_norm.weight");
    }
    auto carve = [&](const float* src, int64_t rows, int64_t K, f16*& hi, f16*& pk, f16** mx = nullptr) -> hipError_t {
      hi = hp; hp += rows * K;
      pk = hp; hp += 2 * rows * K;
      if (mx) { *mx = hp; hp += 2 * rows * K; }
      if (no_cond) {
        hipError_t e = launch_split_f16(src, rows * K, 1.0f, hi, nullptr, st);
        if (e != hipSuccess) return e;
        if (mx && (e = launch_pack_mx_rows(src, K, rows, (int)K, nullptr, *mx, 1, st)) != hipSuccess) return e;
        return launch_split_f16_packed(src, rows, (int)K, pk, st);
      }
      float* scale = cp; cp += rows;
      float* alpha = cp; cp += rows;
      ctx->walpha[hi] = alpha;
      ctx->walpha[pk] = alpha;
      hipError_t e = launch_condition_weight(src, (int)rows, (int)K, scale, alpha, hi, pk, st);
      if (e != hipSuccess || !mx) return e;
      ctx->walpha[*mx] = alpha;  // the MX lines hold the same conditioned rows
      return launch_pack_mx_rows(src, K, rows, (int)K, scale, *mx, 1, st);
    };
    const bool mxw = ctx->mx_ok;
    HIPCHK(carve(bw.wqkv, 3 * inner, D, bw.wqkv_hi, bw.wqkv_pk, mxw ? &bw.wqkv_mx : nullptr));
    HIPCHK(carve(bw.wo, D, inner, bw.wo_hi, bw.wo_pk, mxw ? &bw.wo_mx : nullptr));
    HIPCHK(carve(bw.w1, F, D, bw.w1_hi, bw.w1_pk, mxw ? &bw.w1_mx : nullptr));
    HIPCHK(carve(bw.w2, D, F, bw.w2_hi, bw.w2_pk, mxw ? &bw.w2_mx : nullptr));
    if (bw.wskip) HIPCHK(carve(bw.wskip, D, 2 * D, bw.wskip_hi, bw.wskip_pk));
    if (mmdit) {  // text stream of the block (modules.py:791-814)
      const bool last = i == c.depth - 1;
      bw.wqkv_c = W(ctx, ba + "to_q_c.weight");
      bw.bqkv_c = W(ctx, ba + "to_q_c.bias");
      HIPCHK(carve(bw.wqkv_c, 3 * inner, D, bw.wqkv_c_hi, bw.wqkv_c_pk));
      if (!last) {
        bw.wo_c = W(ctx, ba + "to_out_c.weight");
        bw.bo_c = W(ctx, ba + "to_out_c.bias");
        bw.w1_c = W(ctx, b + "ff_c.ff.0.0.weight");
        bw.b1_c = W(ctx, b + "ff_c.ff.0.0.bias");
        bw.w2_c = W(ctx, b + "ff_c.ff.2.weight");
        bw.b2_c = W(ctx, b + "ff_c.ff.2.bias");
        HIPCHK(carve(bw.wo_c, D, inner, bw.wo_c_hi, bw.wo_c_pk));
        HIPCHK(carve(bw.w1_c, F, D, bw.w1_c_hi, bw.w1_c_pk));
        HIPCHK(carve(bw.w2_c, D, F, bw.w2_c_hi, bw.w2_c_pk));
      }
      if (c.qk_norm) {
        bw.qn_c = W(ctx, ba + "c_q_norm.weight");
        bw.kn_c = W(ctx, ba + "c_k_norm.weight");
      }
    }
    if (!unett && c.long_skip_connection && i == c.depth - 1) {
      ctx->wlong = W(ctx, p + "long_skip_connection.weight");
      HIPCHK(carve(ctx->wlong, D, 2 * D, ctx->wlong_hi, ctx->wlong_pk));
    }
  }
  if (unett) {
    ctx->norm_out_g = W(ctx, p + "norm_out.g");
  } else if (mmdit) {
    ctx->adaln_w = W(ctx, p + "transformer_blocks.0.attn_norm_x.linear.weight");
    ctx->adaln_b = W(ctx, p + "transformer_blocks.0.attn_norm_x.linear.bias");
    ctx->adaln_c_w = W(ctx, p + "transformer_blocks.0.attn_norm_c.linear.weight");
    ctx->adaln_c_b = W(ctx, p + "transformer_blocks.0.attn_norm_c.linear.bias");
  } else {
    ctx->adaln_w = W(ctx, p + "transformer_blocks.0.attn_norm.linear.weight");
    ctx->adaln_b = W(ctx, p + "transformer_blocks.0.attn_norm.linear.bias");
  }
  {
    const int64_t n = (int64_t)c.mel_dim * D;
    HIPCHK(ctx->wp_hi.ensure(n * sizeof(f16)));
    HIPCHK(ctx->wp_pk.ensure(2 * n * sizeof(f16)));
    HIPCHK(launch_split_f16(W(ctx, p + "proj_out.weight"), n, 1.0f, ctx->wp_hi.as<f16>(), nullptr, st));
    HIPCHK(launch_split_f16_packed(W(ctx, p + "proj_out.weight"), c.mel_dim, (int)D, ctx->wp_pk.as<f16>(), st));
  }
  // conv_pos weights -> per-tap operand tiles
  const int cpg = (int)(D / c.conv_pos_groups);
  for (int j = 0; j < 2; ++j) {
    const int64_t n = D * cpg * c.conv_pos_kernel;
    HIPCHK(ctx->conv_w32[j].ensure(n * sizeof(float)));
    HIPCHK(ctx->conv_whi[j].ensure(n * sizeof(f16)));
    HIPCHK(ctx->conv_wlo[j].ensure(n * sizeof(f16)));
    HIPCHK(launch_convpos_pack(W(ctx, p + (mmdit ? "audio_embed" : "input_embed") + ".conv_pos_embed.conv1d." + std::to_string(2 * j) + ".weight"), (int)D, cpg,
                               c.conv_pos_kernel, ctx->conv_w32[j].as<float>(), ctx->conv_whi[j].as<f16>(), ctx->conv_wlo[j].as<f16>(), st));
    if (ctx->mx_ok && cpg == 64) {  // fp16m: rows (group, tap, co) of 64 input channels -> two MX lines each (convpos_mx_kernel)
      HIPCHK(ctx->conv_wmx[j].ensure(2 * n * sizeof(f16)));
      HIPCHK(launch_pack_mx_rows(ctx->conv_w32[j].as<float>(), cpg, n / cpg, cpg, nullptr, ctx->conv_wmx[j].as<f16>(), 1, st));
    } else {
      ctx->conv_wmx[j].release();
    }
  }
  // depthwise weights [C,1,7] -> [7,C]
  const int64_t vC = ctx->has_vocos ? ctx->vcfg.dim : 0;
  HIPCHK(ctx->dwpack.ensure((size_t)(7 * (T * c.conv_layers + vC * (ctx->has_vocos ? ctx->vcfg.num_layers : 0)) + 4) * sizeof(float)));
  float* dwp = ctx->dwpack.as<float>();
  ctx->tblocks.assign(c.conv_layers, TextBlockW{});
  for (int i = 0; i < c.conv_layers; ++i) {
    const std::string b = p + "text_embed.text_blocks." + std::to_string(i) + ".";
    TextBlockW& tb = ctx->tblocks[i];
    tb.dw_b = W(ctx, b + "dwconv.bias"); tb.ln_w = W(ctx, b + "norm.weight"); tb.ln_b = W(ctx, b + "norm.bias");
    tb.pw1_w = W(ctx, b + "pwconv1.weight"); tb.pw1_b = W(ctx, b + "pwconv1.bias");
    tb.gamma = W(ctx, b + "grn.gamma"); tb.beta = W(ctx, b + "grn.beta");
    tb.pw2_w = W(ctx, b + "pwconv2.weight"); tb.pw2_b = W(ctx, b + "pwconv2.bias");
    tb.dw7 = dwp; dwp += 7 * T;
    HIPCHK(launch_dw_pack(W(ctx, b + "dwconv.weight"), (int)T, tb.dw7, st));
  }
  // absolute sinusoid position table of the text encoder (reference model/modules.py:207-218), fp32 op order as torch
  if (mmdit) {  // mmdit.py:39-40,57-60: 1024 positions of width dim; positions past the table reuse its last row (get_pos_embed_indices
                // clips, modules.py:229) — materialised here as 8192 rows so the embedding kernel indexes it directly
    const int64_t n = (int64_t)8192 * D;
    HIPCHK(ctx->freqs_cis.ensure(n * sizeof(float)));
    std::vector<float> tab(n);
    const int half = (int)D / 2;
    for (int k = 0; k < half; ++k) {
      const float f = 1.0f / powf(10000.0f, (float)(2 * k) / (float)D);
      for (int pos = 0; pos < 8192; ++pos) {
        const float a = (float)std::min(pos, 1023) * f;
        tab[(int64_t)pos * D + k] = cosf(a);
        tab[(int64_t)pos * D + half + k] = sinf(a);
      }
    }
    HIPCHK(hipMemcpy(ctx->freqs_cis.p, tab.data(), n * sizeof(float), hipMemcpyHostToDevice));
    if (slot_loaded(ctx, p + "text_embed.freqs_cis"))  // a checkpoint that does carry the buffer wins for the rows it has
      HIPCHK(hipMemcpy(ctx->freqs_cis.p, W(ctx, p + "text_embed.freqs_cis"), (size_t)1024 * D * sizeof(float), hipMemcpyDeviceToDevice));
  } else if (c.conv_layers > 0) {
    const int64_t n = (int64_t)8192 * T;
    HIPCHK(ctx->freqs_cis.ensure(n * sizeof(float)));
    if (slot_loaded(ctx, p + "text_embed.freqs_cis")) {
      HIPCHK(hipMemcpy(ctx->freqs_cis.p, W(ctx, p + "text_embed.freqs_cis"), n * sizeof(float), hipMemcpyDeviceToDevice));
    } else {
      std::vector<float> tab(n);
      const int half = (int)T / 2;
      for (int k = 0; k < half; ++k) {
        const float e = (float)(2 * k) / (float)T;
        const float f = 1.0f / powf(10000.0f, e);
        for (int pos = 0; pos < 8192; ++pos) {
          const float a = (float)pos * f;
          tab[(int64_t)pos * T + k] = cosf(a);
          tab[(int64_t)pos * T + half + k] = sinf(a);
        }
      }
      HIPCHK(hipMemcpy(ctx->freqs_cis.p, tab.data(), n * sizeof(float), hipMemcpyHostToDevice));
    }
  }
  {
    const int half = c.dim_head / 2;
    HIPCHK(ctx->inv_freq.ensure(half * sizeof(float)));
    if (slot_loaded(ctx, p + "rotary_embed.inv_freq")) {
      HIPCHK(hipMemcpy(ctx->inv_freq.p, W(ctx, p + "rotary_embed.inv_freq"), half * sizeof(float), hipMemcpyDeviceToDevice));
    } else {
      std::vector<float> f(half);
      for (int k = 0; k < half; ++k) f[k] = 1.0f / powf(10000.0f, (float)(2 * k) / (float)c.dim_head);
      HIPCHK(hipMemcpy(ctx->inv_freq.p, f.data(), half * sizeof(float), hipMemcpyHostToDevice));
    }
  }
  // audio tables: twiddles, periodic hann window, HTK mel filterbank (torchaudio melscale_fbanks, norm=None)
  {
    const int nfft = 1024, nbin = 513, nmel = c.mel_dim;
    std::vector<float> tw(nfft), win(nfft), fb((size_t)nbin * nmel);
    for (int k = 0; k < nfft / 2; ++k) {
      const double a = 2.0 * M_PI * (double)k / (double)nfft;
      tw[2 * k] = (float)cos(a);
      tw[2 * k + 1] = (float)sin(a);
    }
    for (int i = 0; i < nfft; ++i) win[i] = (float)(0.5 - 0.5 * cos(2.0 * M_PI * (double)i / (double)nfft));
    const double sr = 24000.0, fmin = 0.0, fmax = sr / 2;
    auto hz2mel = [](double f) { return 2595.0 * log10(1.0 + f / 700.0); };
    const double mmin = hz2mel(fmin), mmax = hz2mel(fmax);
    std::vector<double> fpts(nmel + 2);
    for (int i = 0; i < nmel + 2; ++i) {
      const double m = mmin + (mmax - mmin) * (double)i / (double)(nmel + 1);
      fpts[i] = 700.0 * (pow(10.0, m / 2595.0) - 1.0);
    }
    for (int k = 0; k < nbin; ++k) {
      const double fr = (sr / 2) * (double)k / (double)(nbin - 1);
      for (int m = 0; m < nmel; ++m) {
        const double down = (fr - fpts[m]) / (fpts[m + 1] - fpts[m]);
        const double up = (fpts[m + 2] - fr) / (fpts[m + 2] - fpts[m + 1]);
        const double v = std::max(0.0, std::min(down, up));
        fb[(size_t)k * nmel + m] = (float)v;
      }
    }
    // BigVGAN-type mel (modules.py:50): librosa.filters.mel defaults = Slaney mel scale (linear to 1 kHz at 200/3 Hz per mel, then
    // log with step ln(6.4)/27), triangles on the FFT bin centres, each scaled by 2 / bandwidth; float32 table, float64 arithmetic
    std::vector<float> fbs((size_t)nbin * nmel);
    {
      const double f_sp = 200.0 / 3.0, min_log_hz = 1000.0, min_log_mel = min_log_hz / f_sp, logstep = log(6.4) / 27.0;
      auto hz2mel_s = [&](double f) { return f >= min_log_hz ? min_log_mel + log(f / min_log_hz) / logstep : f / f_sp; };
      const double lo = hz2mel_s(fmin), hi = hz2mel_s(fmax);
      std::vector<double> mel_f(nmel + 2);
      for (int i = 0; i < nmel + 2; ++i) {
        // numpy.linspace: start + i * step, the last point set to stop exactly
        const double m = i == nmel + 1 ? hi : lo + (double)i * ((hi - lo) / (double)(nmel + 1));
        mel_f[i] = m >= min_log_mel ? min_log_hz * exp(logstep * (m - min_log_mel)) : f_sp * m;
      }
      const double d = (double)nfft * (1.0 / sr);
      for (int m = 0; m < nmel; ++m) {
        const double enorm = 2.0 / (mel_f[m + 2] - mel_f[m]);
        for (int k = 0; k < nbin; ++k) {
          const double fr = (double)k / d;
          const double lower = (fr - mel_f[m]) / (mel_f[m + 1] - mel_f[m]);
          const double upper = (mel_f[m + 2] - fr) / (mel_f[m + 2] - mel_f[m + 1]);
          const float w = (float)std::max(0.0, std::min(lower, upper));  // weights[] is float32 before the normalisation
          fbs[(size_t)k * nmel + m] = (float)((double)w * enorm);
        }
      }
    }
    // the support of every triangle: the mel kernel sums a channel over [first, last + 1) only (same terms, same order as the dense product)
    auto ranges = [&](const std::vector<float>& tab) {
      std::vector<int> r(2 * (size_t)nmel, 0);
      for (int m = 0; m < nmel; ++m) {
        int lo = nbin, hi = 0;
        for (int k = 0; k < nbin; ++k)
          if (tab[(size_t)k * nmel + m] != 0.0f) { lo = std::min(lo, k); hi = k + 1; }
        r[2 * m] = std::min(lo, hi);
        r[2 * m + 1] = hi;
      }
      return r;
    };
    const std::vector<int> rg = ranges(fb), rgs = ranges(fbs);
    // ... and the weights of each support as one contiguous, zero-padded run (what mel_kernel reads): [nmel][ld], ld = widest support up to 4
    auto compact = [&](const std::vector<float>& tab, const std::vector<int>& r, int& ld) {
      ld = 4;
      for (int m = 0; m < nmel; ++m) ld = std::max(ld, (r[2 * m + 1] - r[2 * m] + 3) & ~3);
      std::vector<float> w((size_t)nmel * ld, 0.0f);
      for (int m = 0; m < nmel; ++m)
        for (int k = r[2 * m]; k < r[2 * m + 1]; ++k) w[(size_t)m * ld + (k - r[2 * m])] = tab[(size_t)k * nmel + m];
      return w;
    };
    const std::vector<float> cw = compact(fb, rg, ctx->melw_ld), cws = compact(fbs, rgs, ctx->melw_slaney_ld);
    HIPCHK(ctx->melw.ensure(cw.size() * sizeof(float)));
    HIPCHK(ctx->melw_slaney.ensure(cws.size() * sizeof(float)));
    HIPCHK(hipMemcpy(ctx->melw.p, cw.data(), cw.size() * sizeof(float), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(ctx->melw_slaney.p, cws.data(), cws.size() * sizeof(float), hipMemcpyHostToDevice));
    HIPCHK(ctx->melrange.ensure(rg.size() * sizeof(int)));
    HIPCHK(ctx->melrange_slaney.ensure(rgs.size() * sizeof(int)));
    HIPCHK(hipMemcpy(ctx->melrange.p, rg.data(), rg.size() * sizeof(int), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(ctx->melrange_slaney.p, rgs.data(), rgs.size() * sizeof(int), hipMemcpyHostToDevice));
    HIPCHK(ctx->melfb_slaney.ensure(fbs.size() * sizeof(float)));
    HIPCHK(hipMemcpy(ctx->melfb_slaney.p, fbs.data(), fbs.size() * sizeof(float), hipMemcpyHostToDevice));
    HIPCHK(ctx->twiddle.ensure(tw.size() * sizeof(float)));
    HIPCHK(ctx->window.ensure(win.size() * sizeof(float)));
    HIPCHK(ctx->melfb.ensure(fb.size() * sizeof(float)));
    HIPCHK(hipMemcpy(ctx->twiddle.p, tw.data(), tw.size() * sizeof(float), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(ctx->window.p, win.data(), win.size() * sizeof(float), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(ctx->melfb.p, fb.data(), fb.size() * sizeof(float), hipMemcpyHostToDevice));
  }
  if (ctx->has_vocos) {
    const auto& v = ctx->vcfg;
    if (v.n_fft != 1024 || v.hop_length != 256) FAIL(F5HIP_ERR_UNSUPPORTED, "vocos head: only n_fft=1024 / hop=256 is built");
    const int64_t C = v.dim;
    ctx->vlayers.assign(v.num_layers, VocosLayerW{});
    for (int i = 0; i < v.num_layers; ++i) {
      const std::string b = "backbone.convnext." + std::to_string(i) + ".";
      VocosLayerW& l = ctx->vlayers[i];
      l.dw_b = W(ctx, b + "dwconv.bias"); l.ln_w = W(ctx, b + "norm.weight"); l.ln_b = W(ctx, b + "norm.bias");
      l.pw1_w = W(ctx, b + "pwconv1.weight"); l.pw1_b = W(ctx, b + "pwconv1.bias");
      l.pw2_w = W(ctx, b + "pwconv2.weight"); l.pw2_b = W(ctx, b + "pwconv2.bias");
      l.gamma = W(ctx, b + "gamma");
      l.dw7 = dwp; dwp += 7 * C;
      HIPCHK(launch_dw_pack(W(ctx, b + "dwconv.weight"), (int)C, l.dw7, st));
    }
    // head padded to a multiple of 4 output rows (1026 -> 1028)
    const int nout = v.n_fft + 2, npad = (nout + 3) & ~3;
    HIPCHK(ctx->vhead_w.ensure((size_t)npad * C * sizeof(float), nullptr, true));
    HIPCHK(ctx->vhead_b.ensure((size_t)npad * sizeof(float), nullptr, true));
    HIPCHK(hipMemcpy(ctx->vhead_w.p, W(ctx, "head.out.weight"), (size_t)nout * C * sizeof(float), hipMemcpyDeviceToDevice));
    HIPCHK(hipMemcpy(ctx->vhead_b.p, W(ctx, "head.out.bias"), (size_t)nout * sizeof(float), hipMemcpyDeviceToDevice));
  }
  HIPCHK(init_gemm_kernels());
  HIPCHK(init_convpos_kernels());
  HIPCHK(init_attention_kernels());
  HIPCHK(hipDeviceSynchronize());
  ctx->finalized = true;
  return F5HIP_OK;
}

// ---- time-grid tables: every step's time embedding and AdaLN modulation in 4 GEMMs -------------------
// te: the E evaluation times of the solve (euler: t_0..t_{steps-1}; midpoint: t_i and t_i + dt_i/2); coef: the E update coefficients
int prepare_time(f5hip_ctx* ctx, const float* te, const float* coef, int steps, float cfg_strength, hipStream_t st) {
  const auto& c = ctx->cfg;
  const int64_t D = c.dim;
  const std::string p = "transformer.";
  const float* t = te;
  HIPCHK(ctx->dt_dev.ensure(std::max(steps, 64) * sizeof(float)));
  HIPCHK(ctx->cfg_dev.ensure(16));
  {  // coef / cfg / the time grid are temporaries of the caller: staged in pinned memory, copied by the stream
    STAGE(float, hc, (size_t)steps + 1);
    memcpy(hc, coef, steps * sizeof(float));
    hc[steps] = cfg_strength;
    HIPCHK(hipMemcpyAsync(ctx->dt_dev.p, hc, steps * sizeof(float), hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(ctx->cfg_dev.p, hc + steps, sizeof(float), hipMemcpyHostToDevice, st));
  }
  // the tables below depend on the grid AND on the weights: t_host is cleared whenever a weight changes (invalidate_weight_caches)
  const bool same = (int)ctx->t_host.size() == steps && memcmp(ctx->t_host.data(), t, steps * sizeof(float)) == 0;
  if (same) return F5HIP_OK;
  bool moved = false;
  HIPCHK(ctx->t_dev.ensure(steps * sizeof(float), &moved));
  HIPCHK(ctx->tsin.ensure((size_t)steps * 256 * sizeof(float), &moved));
  HIPCHK(ctx->th1.ensure((size_t)steps * D * sizeof(float), &moved));
  HIPCHK(ctx->tsilu.ensure((size_t)steps * D * sizeof(float), &moved));
  const bool unett = c.backbone == 1;
  if (unett) {
    HIPCHK(ctx->temb.ensure((size_t)steps * D * sizeof(float), &moved));
  } else {
    HIPCHK(ctx->mods.ensure((size_t)steps * c.depth * 6 * D * sizeof(float), &moved));
    HIPCHK(ctx->fmods.ensure((size_t)steps * 2 * D * sizeof(float), &moved));
    if (c.backbone == 2) HIPCHK(ctx->cmods.ensure((size_t)steps * ((c.depth - 1) * 6 + 2) * D * sizeof(float), &moved));
  }
  if (moved) ctx->ws_epoch++;
  ctx->t_host.assign(t, t + steps);
  {
    STAGE(float, ht, (size_t)steps);
    memcpy(ht, t, steps * sizeof(float));
    HIPCHK(hipMemcpyAsync(ctx->t_dev.p, ht, steps * sizeof(float), hipMemcpyHostToDevice, st));
  }
  {
    Prof pr(ctx, st, KC_GEMM_MISC, 0, 0);
    HIPCHK(launch_time_sinus(ctx->t_dev.as<float>(), steps, 256, ctx->tsin.as<float>(), st));
    // TimestepEmbedding (reference model/modules.py:852-862) then SiLU (input of every AdaLN linear, :322,343)
    GemmCore g = core(ctx->tsin.p, 256, W(ctx, p + "time_embed.time_mlp.0.weight"), 256, steps, (int)D, 256);
    HIPCHK(launch_gemm_store(OP_F32, g, epi_store(ctx->th1.as<float>(), D, W(ctx, p + "time_embed.time_mlp.0.bias"), ACT_SILU), 1, st));
    g = core(ctx->th1.p, D, W(ctx, p + "time_embed.time_mlp.2.weight"), D, steps, (int)D, (int)D);
    if (unett) {  // the raw embedding is the time token (unett.py:272); there is no AdaLN
      HIPCHK(launch_gemm_store(OP_F32, g, epi_store(ctx->temb.as<float>(), D, W(ctx, p + "time_embed.time_mlp.2.bias")), 1, st));
      return F5HIP_OK;
    }
    HIPCHK(launch_gemm_store(OP_F32, g, epi_store(ctx->tsilu.as<float>(), D, W(ctx, p + "time_embed.time_mlp.2.bias"), ACT_SILU), 1, st));
    const int nm = c.depth * 6 * (int)D;
    g = core(ctx->tsilu.p, D, ctx->adaln_w, D, steps, nm, (int)D);
    HIPCHK(launch_gemm_store(OP_F32, g, epi_store(ctx->mods.as<float>(), nm, ctx->adaln_b), 1, st));
    g = core(ctx->tsilu.p, D, W(ctx, p + "norm_out.linear.weight"), D, steps, 2 * (int)D, (int)D);
    HIPCHK(launch_gemm_store(OP_F32, g, epi_store(ctx->fmods.as<float>(), 2 * D, W(ctx, p + "norm_out.linear.bias")), 1, st));
    if (c.backbone == 2) {  // text-stream AdaLN of every block: 6 chunks, the last block (AdaLayerNorm_Final) only (scale, shift)
      const int nc = ((c.depth - 1) * 6 + 2) * (int)D;
      g = core(ctx->tsilu.p, D, ctx->adaln_c_w, D, steps, nc, (int)D);
      HIPCHK(launch_gemm_store(OP_F32, g, epi_store(ctx->cmods.as<float>(), nc, ctx->adaln_c_b), 1, st));
    }
  }
  return F5HIP_OK;
}

// ---- workspace -----------------------------------------------------------------------------------
int ensure_workspace(f5hip_ctx* ctx, int B, int n, int nt, int op, bool exact_attn) {
  const auto& c = ctx->cfg;
  const int64_t D = c.dim, T = c.text_dim, mel = c.mel_dim, inner = (int64_t)c.heads * c.dim_head, F = c.ff_inner;
  const bool unett = c.backbone == 1, mmdit = c.backbone == 2;
  if (ctx->attn_kv_split > 1 && !exact_attn && !mmdit) {  // partial results of the key-split attention: [sequences * heads * tokens, parts, 64 + 2]
    bool moved = false;
    HIPCHK(ctx->attn_part.ensure((size_t)2 * B * c.heads * (n + 1) * ctx->attn_kv_split * 66 * sizeof(float), &moved));
    if (moved) ctx->ws_epoch++;
  }
  // tokens per sequence inside the backbone: UNetT prepends the time token; MMDiT attends over audio frames + text tokens jointly
  // (its row-wise buffers hold the audio rows of all sequences first, then the text rows)
  const int ns = n + (unett ? 1 : mmdit ? nt : 0);
  const int64_t BN = (int64_t)B * n, M = 2 * (int64_t)B * ns;
  bool moved = false;
#define ENS(buf, bytes) HIPCHK(ctx->buf.ensure((size_t)(bytes), &moved))
  ENS(tok, std::max<int64_t>(BN, (int64_t)B * nt) * 4); ENS(valid, std::max<int64_t>(BN, (int64_t)B * nt)); ENS(textkeep, M); ENS(rowvalid, M);
  ENS(condmask, BN); ENS(kvlen, 2 * B * 4);
  if (mmdit) {
    ENS(ctext0, 2 * (int64_t)B * nt * D * 4); ENS(cmask, 2 * (int64_t)B * nt); ENS(kvlen2, 2 * B * 4);
  } else {
    ENS(tx, M * T * 4); ENS(ta, M * T * 4); ENS(th, M * 2 * T * 4); ENS(tg, M * 2 * T * 4); ENS(sumsq, (size_t)2 * B * 2 * T * 4 * (1 + grn_sumsq_slices(n)));  // sums, then the per-slice partial sums
  }
  ENS(step_cond, BN * mel * 4); ENS(cconst, M * D * 4); ENS(y, BN * mel * 4);
  ENS(h, M * D * 4); ENS(c1, M * D * 4); ENS(x, M * D * 4);
  ENS(vel, M * mel * 4); ENS(dbg_vel, BN * mel * 4); ENS(rope, (int64_t)ns * c.dim_head * 4);  // MMDiT: max(n, nt) positions fit
  if (unett && c.skip_connect_type != 2) ENS(skipcat, (int64_t)(c.depth / 2) * M * 2 * D * (op == OP_F16 ? 2 : 4));
  if (!unett && c.long_skip_connection) ENS(skipcat, M * 2 * D * (op == OP_F16 ? 2 : 4));
  if (c.text_average_upsampling) ENS(avgidx, BN * 4);
  if (c.qk_norm && !exact_attn) { ENS(q32, M * inner * 4); ENS(k32, M * inner * 4); }  // raw q/k rows between the GEMM and the norm kernel
  if (op == OP_F32) {
    ENS(a32, M * D * 4); ENS(o32, M * inner * 4); ENS(f32, M * F * 4);
  } else {
    const int64_t pl = op == OP_F16X3 ? 2 : 1;  // packed hi/lo rows are twice as long
    ENS(a_hi, M * D * 2 * pl); ENS(o_hi, M * inner * 2 * pl); ENS(f_hi, M * F * 2 * pl);
  }
  if (exact_attn) {
    const int64_t np = (ns + 3) & ~3;
    ENS(q32, M * inner * 4); ENS(k32, M * inner * 4);
    if ((size_t)(2 * B * c.heads * c.dim_head * np * 4) > ctx->vt32.cap) {
      HIPCHK(ctx->vt32.ensure((size_t)(2 * B * c.heads * c.dim_head * np * 4), &moved, /*zero=*/true));
    } else if (ctx->ws_n != n || ctx->ws_nt != nt) {
      HIPCHK(hipMemset(ctx->vt32.p, 0, ctx->vt32.cap));  // padding columns must be zero for the new row stride
    }
    ENS(scores, (int64_t)2 * B * c.heads * ns * np * 4);
  } else {
    const int64_t ldv = (ns + 7) & ~7;
    ENS(q16, M * inner * 2); ENS(k16, M * inner * 2);
    // V^T slabs: the pad columns [n, ldv) are never written by the QKV epilogue and must not hold NaN/Inf bit patterns
    HIPCHK(ctx->vt16.ensure((size_t)((int64_t)2 * B * inner * ldv * 2), &moved, /*zero=*/true));
    if (op == OP_F16X3) {
      ENS(q16_lo, M * inner * 2); ENS(k16_lo, M * inner * 2);
      HIPCHK(ctx->vt16_lo.ensure((size_t)((int64_t)2 * B * inner * ldv * 2), &moved, /*zero=*/true));
    }
  }
#undef ENS
  if (moved) ctx->ws_epoch++;
  ctx->ws_B = B;
  ctx->ws_n = n;
  ctx->ws_nt = nt;
  return F5HIP_OK;
}

// ---- text embedding (once per utterance; reference dit.py:86-139, cached across steps :294-310) -----
int run_text_embed(f5hip_ctx* ctx, int B, int n, const int64_t* text, int nt, const int64_t* duration, int use_mask, hipStream_t st) {
  const auto& c = ctx->cfg;
  const int T = c.text_dim;
  const int64_t BN = (int64_t)B * n, M = 2 * BN;
  STAGE(int32_t, tok, (size_t)BN);
  STAGE(int32_t, avg, (size_t)(c.text_average_upsampling ? BN : 1));
  STAGE(uint8_t, valid, (size_t)BN);
  STAGE(uint8_t, keep, (size_t)M);
  if (c.text_average_upsampling) std::fill(avg, avg + BN, -1);
  std::vector<int32_t> vpos;
  for (int b = 0; b < B; ++b) {
    // DiT: per-sample valid length when a mask is passed (dit.py:295-298); UNetT: the padded frame count for every sample (unett.py:218-228)
    const int64_t sl = (use_mask && c.backbone != 1) ? std::min<int64_t>(duration[b], n) : n;
    for (int pos = 0; pos < n; ++pos) {
      int64_t id = pos < nt ? text[(int64_t)b * nt + pos] + 1 : 0;  // +1, 0 = filler (dit.py:87,95-96)
      const bool ok = pos < sl;
      if (!ok) id = 0;
      if (id < 0 || id > c.text_num_embeds) FAIL(F5HIP_ERR_INVALID, "text id %lld out of range at [%d,%d]", (long long)(id - 1), b, pos);
      tok[(int64_t)b * n + pos] = (int32_t)id;
      valid[(int64_t)b * n + pos] = ok ? 1 : 0;
      const uint8_t k = (c.text_mask_padding && id == 0) ? 0 : 1;
      keep[(int64_t)b * n + pos] = k;
      keep[BN + (int64_t)b * n + pos] = k;
    }
    if (c.text_average_upsampling) {
      // average_upsample_text_by_mask (dit.py:55-84): the tl valid tokens are spread over the first sl frames, token j repeated
      // sl / tl times and the last sl % tl tokens once more; frames behind stay zero
      vpos.clear();
      for (int pos = 0; pos < n; ++pos)
        if (tok[(int64_t)b * n + pos] != 0) vpos.push_back(pos);
      const int64_t tl = (int64_t)vpos.size(), al = sl;
      if (tl > 0 && al > 0) {
        const int64_t base = al / tl, rem = al % tl;
        int64_t o = 0;
        for (int64_t j = 0; j < tl && o < al; ++j)
          for (int64_t r = base + (j >= tl - rem ? 1 : 0); r > 0 && o < al; --r) avg[(int64_t)b * n + o++] = vpos[j];
      }
    }
  }
  if (c.text_average_upsampling) HIPCHK(hipMemcpyAsync(ctx->avgidx.p, avg, BN * 4, hipMemcpyHostToDevice, st));
  HIPCHK(hipMemcpyAsync(ctx->tok.p, tok, BN * 4, hipMemcpyHostToDevice, st));
  HIPCHK(hipMemcpyAsync(ctx->valid.p, valid, BN, hipMemcpyHostToDevice, st));
  HIPCHK(hipMemcpyAsync(ctx->textkeep.p, keep, M, hipMemcpyHostToDevice, st));
  // algorithmic traffic of the text encoder (per ConvNeXtV2 block read x + write out + write h + read h, h = 2x) and
  // its pointwise-GEMM FLOPs, all inside this one scope
  const double x_elems = (double)M * T;
  Prof pr(ctx, st, KC_TEXT, c.conv_layers * 2.0 * gemm_flops(M, 2 * T, T), x_elems * 4.0 + c.conv_layers * 6.0 * x_elems * 4.0);
  float* tx = ctx->tx.as<float>();
  HIPCHK(launch_text_embed(ctx->tok.as<int32_t>(), ctx->valid.as<uint8_t>(), W(ctx, "transformer.text_embed.text_embed.weight"),
                           ctx->freqs_cis.as<float>(), B, n, T, c.text_mask_padding, c.conv_layers > 0, tx, st));
  const uint8_t* keepd = c.text_mask_padding ? ctx->textkeep.as<uint8_t>() : nullptr;
  for (int i = 0; i < c.conv_layers; ++i) {
    const TextBlockW& tb = ctx->tblocks[i];
    // ConvNeXtV2Block (reference model/modules.py:270-280)
    HIPCHK(launch_dwconv7_ln(tx, 2 * B, n, T, tb.dw7, tb.dw_b, tb.ln_w, tb.ln_b, 1e-6f, ctx->ta.as<float>(), st));
    GemmCore g = core(ctx->ta.p, T, tb.pw1_w, T, (int)M, 2 * T, T);
    HIPCHK(launch_gemm_store(OP_F32, g, epi_store(ctx->th.as<float>(), 2 * T, tb.pw1_b, ACT_GELU_ERF), 1, st));
    HIPCHK(launch_grn_stats(ctx->th.as<float>(), 2 * B, n, 2 * T, ctx->sumsq.as<float>(), ctx->sumsq.as<float>() + (int64_t)2 * B * 2 * T, st));
    HIPCHK(launch_grn_apply(ctx->th.as<float>(), ctx->sumsq.as<float>(), tb.gamma, tb.beta, 2 * B, n, 2 * T, ctx->tg.as<float>(), st));
    g = core(ctx->tg.p, 2 * T, tb.pw2_w, 2 * T, (int)M, T, 2 * T);
    EpiStore e = epi_store(tx, T, tb.pw2_b);
    e.res = tx; e.ldres = T;
    e.rowmask = keepd; e.mask_mode = 2;  // masked_fill after the residual add (dit.py:127)
    HIPCHK(launch_gemm_store(OP_F32, g, e, 1, st));
  }
  if (c.text_average_upsampling) {  // dit.py:131-137, after the text encoder; the cond and uncond halves share the token mask (dit.py:104-108)
    HIPCHK(launch_gather_seq_rows(tx, ctx->avgidx.as<int32_t>(), 2 * B, B, n, T, ctx->ta.as<float>(), st));
    HIPCHK(hipMemcpyAsync(tx, ctx->ta.p, (size_t)M * T * 4, hipMemcpyDeviceToDevice, st));
  }
  return F5HIP_OK;
}


// ---- attention over [2B * H] (batch', head) slabs of ns tokens: q/k/v were written by the QKV epilogue -------------------------------
// S sequences starting at sequence s0 of the packed [cond | uncond] batch (o32/o_hi/o_lo/kvlen are already offset by the caller)
int run_attention(f5hip_ctx* ctx, int S, int s0, int n, int op, bool exact_attn, int qks, const int32_t* kvlen, float* o32, f16* o_hi, f16* o_lo, int pk,
                  int64_t ldO, hipStream_t st, const int32_t* kvlen2 = nullptr, int seg2_off = 0, const int32_t* cu_rows = nullptr) {
  const auto& c = ctx->cfg;
  const int H = c.heads, dh = c.dim_head, inner = H * dh;
  const int64_t qoff = (int64_t)s0 * H * n * dh;                      // q/k slabs [seq*H, n, dh]
  {
      Prof pr(ctx, st, KC_ATTN, 4.0 * (double)S * H * (double)n * n * dh, 0);
      if (exact_attn) {
        // materialised-score attention in fp32: S = QK^T (batched GEMM), row softmax, O = PV (batched GEMM)
        const int np = (n + 3) & ~3;
        float* sc = ctx->scores.as<float>() + (int64_t)s0 * H * n * np;
        const float* vt = ctx->vt32.as<float>() + (int64_t)s0 * inner * np;
        GemmCore g = core(ctx->q32.as<float>() + qoff, dh, ctx->k32.as<float>() + qoff, dh, n, np, dh);
        g.w_rows = n; g.strideA = (int64_t)n * dh; g.strideW = (int64_t)n * dh;
        EpiStore e = epi_store(sc, np, nullptr);
        e.zdiv = 1; e.so1 = (int64_t)n * np; e.so2 = 0;
        HIPCHK(launch_gemm_store(OP_F32, g, e, S * H, st));
        HIPCHK(launch_softmax_rows(sc, (int64_t)S * H * n, np, n, H, kvlen, n, st, kvlen2, seg2_off,
                                   ctx->attn_stats ? ctx->attn_stats_buf.as<double>() : nullptr));
        g = core(sc, np, vt, np, n, dh, np);
        g.strideA = (int64_t)n * np; g.strideW = (int64_t)dh * np;
        EpiStore e2 = epi_store(o32, inner, nullptr);
        e2.zdiv = H; e2.so1 = (int64_t)n * inner; e2.so2 = dh;
        if (op != OP_F32) {  // fp16 operand of the out-projection: same (batch', head) addressing, packed rows in fp16x3 mode
          e2.out16 = o_hi; e2.out16_lo = o_lo; e2.pk16 = pk; e2.ldo16 = ldO;
          e2.so1_16 = (int64_t)n * ldO; e2.so2_16 = dh;
        }
        HIPCHK(launch_gemm_store(OP_F32, g, e2, S * H, st));
      } else {
        // qks: what the q|k|v epilogue of this chunk left in the second planes of q and k (run_qkv) — nothing, fp16 remainders or MX P words
        // (split scores: where the q|k|v launch cannot write P words — MMDiT, qk_norm, tiny row counts — attn_impl 6 / 7 run the everything-split form)
        const bool x3 = qks == QK_SPLIT, all3 = x3 && attn_v_split(ctx), lo = qks != QK_PLAIN;
        const int ldv = (n + 7) & ~7;
        const int64_t voff = (int64_t)s0 * inner * ldv;
        // kernel form: 1 plain, 2 split q k, 3 everything split, 4 MX-corrected scores, 5 / 6 the same with V / V and P as hi + lo halves
        const bool vlo = all3 || (qks == QK_MX && (ctx->attn_impl == 6 || ctx->attn_impl == 7));
        const int form = qks == QK_MX ? (ctx->attn_impl == 6 ? 5 : ctx->attn_impl == 7 ? 6 : 4) : x3 ? (all3 ? 3 : 2) : 1;
        HIPCHK(launch_flash_attn(form, ctx->q16.as<f16>() + qoff, lo ? ctx->q16_lo.as<f16>() + qoff : nullptr,
                                 ctx->k16.as<f16>() + qoff, lo ? ctx->k16_lo.as<f16>() + qoff : nullptr, ctx->vt16.as<f16>() + voff,
                                 vlo ? ctx->vt16_lo.as<f16>() + voff : nullptr, ldv, S, H, n, kvlen, o_hi, o_lo, st, pk,
                                 kvlen2, seg2_off, 1, ctx->attn_part.p ? ctx->attn_kv_split : 1,
                                 ctx->attn_part.p ? ctx->attn_part.as<float>() + (int64_t)s0 * H * n * ctx->attn_kv_split * 66 : nullptr,
                                 ctx->attn_part.p ? ctx->attn_part.as<float>() + (int64_t)s0 * H * n * ctx->attn_kv_split * 66 + (int64_t)S * H * n * ctx->attn_kv_split * 64 : nullptr,
                                 /*log2q: flash_qscale() put log2(e) into q*/ 1, cu_rows));  // co_launches stays 1: counting the other CFG chain's launch as concurrent made B=1 8 % slower
                                                      // (the two chains are rarely in attention at the same time)
      }
    }
  return F5HIP_OK;
}

// ---- fused to_q|to_k|to_v GEMM of one block: bias + rope + 1/sqrt(dh) + head split in the epilogue (modules.py:481-509; SDPA default
// scale).  M rows = sequences [s0, s0 + M/ns) of ns tokens.  With qk_norm the epilogue leaves q and k raw (fp32, bias only) and
// qk_norm_rope_kernel applies RMSNorm(dim_head) -> rope -> scale and writes what the attention kernels read (modules.py:493-509).
// *qks: what the second planes of q and k hold afterwards (QK_*: run_attention's argument)
int run_qkv(f5hip_ctx* ctx, const BlockW& bw, const void* A, int64_t ldA, int M, int ns, int s0, int op, bool exact_attn, int* qks, int wbytes,
            hipStream_t st, const uint32_t* rowinfo = nullptr, int nslab = 0) {
  const auto& c = ctx->cfg;
  const int D = c.dim, H = c.heads, dh = c.dim_head, inner = H * dh;
  const int64_t qoff = rowinfo ? 0 : (int64_t)s0 * H * ns * dh;
  EpiQKV e{};
  e.bias = bw.bqkv; e.rope_cs = ctx->rope.as<float>(); e.nseq = ns; e.heads = H; e.dh = dh;
  e.pe_heads = c.pe_attn_head; e.qscale = attn_qscale(dh, exact_attn);
  e.rowinfo = rowinfo; e.nslab = nslab;  // packed rows: rowinfo names the sequences by their index in the whole batch, so the slabs are NOT offset by s0
  e.qk_raw = c.qk_norm ? 1 : 0;
  if (exact_attn || c.qk_norm) { e.q32 = ctx->q32.as<float>() + qoff; e.k32 = ctx->k32.as<float>() + qoff; }
  if (exact_attn) {
    e.ldvt = (ns + 3) & ~3;
    e.vt32 = ctx->vt32.as<float>() + (int64_t)s0 * inner * e.ldvt;
  } else {
    e.ldvt = (ns + 7) & ~7;
    const int64_t voff = rowinfo ? 0 : (int64_t)s0 * inner * e.ldvt;
    e.q16 = ctx->q16.as<f16>() + qoff; e.k16 = ctx->k16.as<f16>() + qoff; e.vt16 = ctx->vt16.as<f16>() + voff;
  }
  GemmCore g = (core(A, ldA, wsel(ctx, op, bw.wqkv, bw.wqkv_hi, bw.wqkv_pk, bw.wqkv_mx), ldA, M, 3 * inner, D));
  *qks = exact_attn ? QK_PLAIN : qk_scheme_wanted(ctx, op == OP_F16M ? OP_F16X3 : op);
  if (*qks == QK_MX && !gemm_qkv_takes_pp(op, g, e)) *qks = QK_SPLIT;  // the generic kernels and the qk_norm detour write fp16 remainders
  if (*qks != QK_PLAIN) {  // second planes only for what the flash kernel will read
    const int64_t voff = rowinfo ? 0 : (int64_t)s0 * inner * e.ldvt;
    e.q16_lo = ctx->q16_lo.as<f16>() + qoff; e.k16_lo = ctx->k16_lo.as<f16>() + qoff;
    e.mx_qk = *qks == QK_MX;
    if (attn_v_split(ctx)) e.vt16_lo = ctx->vt16_lo.as<f16>() + voff;
  }
  {
    Prof pr(ctx, st, KC_GEMM_BLOCK, gemm_flops(M, 3 * inner, D), (double)M * D * wbytes + 3.0 * inner * D * wbytes + (double)M * 3 * inner * wbytes);
    HIPCHK(launch_gemm_qkv(op, g, e, st));
  }
  if (c.qk_norm) {
    Prof pr(ctx, st, KC_ELEMWISE, 0, 2.0 * M * inner * (4.0 + (exact_attn ? 4.0 : 2.0 * (e.q16_lo ? 2 : 1))));
    HIPCHK(launch_qk_norm_rope(e.q32, e.k32, bw.qn, bw.kn, e.rope_cs, (int64_t)(M / ns) * H * ns, ns, H, dh, c.pe_attn_head, e.qscale, 1e-6f,
                               e.q16, e.q16_lo, e.k16, e.k16_lo, st));
  }
  return F5HIP_OK;
}

// ---- the block loop, final norm and output projection of run_step over PACKED rows (option "packed_rows") -------------------------------
// Rows [p0, p0 + Mp) of the packed order = the valid rows of sequences [s0, s0 + S); x arrives gathered in ctx->xpk.  Same kernels as the
// padded loop: only the q|k|v epilogue (row -> (sequence, token) from a table) and the attention (output rows from cu_rows, no blocks past a
// sequence's end) know about the layout; no row mask is needed any more.  The velocity is scattered back to the padded layout (zeros in
// the padding) for the CFG / Euler update.
int run_blocks_packed(f5hip_ctx* ctx, int B, int n, const Stage& sg, int op, int S, int s0, int64_t p0, int Mp, hipStream_t st, int br) {
  const int step = sg.eidx, nb = ctx->nb;
  const auto& c = ctx->cfg;
  const int D = c.dim, mel = c.mel_dim, inner = c.heads * c.dim_head, F = c.ff_inner;
  const int64_t BN = (int64_t)B * n;
  const std::string p = "transformer.";
  float* x = ctx->xpk.as<float>() + p0 * D;
  const float* mods_step = ctx->mods.as<float>() + (int64_t)step * c.depth * 6 * D;
  const int pk = op == OP_F16X3 ? 1 : 0, wbytes = 2;
  const bool mx = ctx->mx_call;                     // fp16m: the four block GEMMs read MX lines (same row strides as the hi | lo lines)
  const int opb = mx ? OP_F16M : op, pkb = mx ? 2 : pk;
  const int64_t ldA = (int64_t)D * (pk ? 2 : 1), ldO = (int64_t)inner * (pk ? 2 : 1), ldF = (int64_t)F * (pk ? 2 : 1);
  const int64_t ldAb = ldA, ldOb = ldO, ldFb = ldF;  // strides of the block operands (MX lines keep the hi | lo strides)
  f16* a_hi = ctx->a_hi.as<f16>() + p0 * ldA;
  f16* a_lo = pk ? a_hi + 32 : nullptr;
  f16* o_all = ctx->o_hi.as<f16>();  // the attention writes row cu_rows[s] + q of the WHOLE packed order
  f16* o_hi = o_all + p0 * ldOb;  // (the attention addresses the buffer by global packed row)
  f16* f_hi = ctx->f_hi.as<f16>() + p0 * ldF;
  f16* f_lo = pk ? f_hi + 32 : nullptr;
  const int32_t* kvlen = ctx->kvlen.as<int32_t>() + s0;
  const int32_t* cu = ctx->cu_rows.as<int32_t>() + s0;
  const uint32_t* rowinfo = ctx->rowinfo.as<uint32_t>() + p0;
  const double ln_bytes = (double)Mp * D * (4 + wbytes * (pk ? 2 : 1));
  for (int i = 0; i < c.depth; ++i) {
    const BlockW& bw = ctx->blocks[i];
    const float* md = mods_step + (int64_t)i * 6 * D;
    {
      Prof pr(ctx, st, KC_LNMOD, 0, ln_bytes);
      HIPCHK(launch_layernorm(x, D, Mp, D, 1e-6f, nullptr, nullptr, md + D, md, nullptr, a_hi, a_lo, D, st, pkb, ldAb));
    }
    int qks = QK_PLAIN;
    CHK(run_qkv(ctx, bw, a_hi, ldAb, Mp, n, s0, opb, false, &qks, wbytes, st, rowinfo, nb * B));
    CHK(run_attention(ctx, S, s0, n, op, false, qks, kvlen, nullptr, o_all, pk ? o_all + 32 : nullptr, pkb, ldOb, st, nullptr, 0, cu));
    {
      Prof pr(ctx, st, KC_GEMM_BLOCK, gemm_flops(Mp, D, inner), (double)Mp * inner * wbytes + (double)inner * D * wbytes + 2.0 * Mp * D * 4);
      GemmCore g = core(o_hi, ldOb, wsel(ctx, opb, bw.wo, bw.wo_hi, bw.wo_pk, bw.wo_mx), ldOb, Mp, D, inner);
      EpiStore e = epi_store(x, D, bw.bo);
      e.colscale = md + 2 * D; e.res = x; e.ldres = D;  // every row is a valid row: no mask (modules.py:554-556 zeroes the padding only)
      HIPCHK(launch_gemm_store(opb, g, e, 1, st));
    }
    {
      Prof pr(ctx, st, KC_LNMOD, 0, ln_bytes);
      HIPCHK(launch_layernorm(x, D, Mp, D, 1e-6f, nullptr, nullptr, md + 4 * D, md + 3 * D, nullptr, a_hi, a_lo, D, st, pkb, ldAb));
    }
    {
      Prof pr(ctx, st, KC_GEMM_BLOCK, gemm_flops(Mp, F, D), (double)Mp * D * wbytes + (double)F * D * wbytes + (double)Mp * F * wbytes);
      GemmCore g = core(a_hi, ldAb, wsel(ctx, opb, bw.w1, bw.w1_hi, bw.w1_pk, bw.w1_mx), ldAb, Mp, F, D);
      EpiStore e = epi_store(nullptr, F, bw.b1, ACT_GELU_TANH);
      e.out16 = f_hi; e.out16_lo = f_lo; e.pk16 = pkb; e.ldo16 = ldFb;
      HIPCHK(launch_gemm_store(opb, g, e, 1, st));
    }
    {
      Prof pr(ctx, st, KC_GEMM_BLOCK, gemm_flops(Mp, D, F), (double)Mp * F * wbytes + (double)F * D * wbytes + 2.0 * Mp * D * 4);
      GemmCore g = core(f_hi, ldFb, wsel(ctx, opb, bw.w2, bw.w2_hi, bw.w2_pk, bw.w2_mx), ldFb, Mp, D, F);
      EpiStore e = epi_store(x, D, bw.b2);
      e.colscale = md + 5 * D; e.res = x; e.ldres = D;
      HIPCHK(launch_gemm_store(opb, g, e, 1, st));
    }
  }
  {  // AdaLayerNorm_Final + proj_out on the packed rows, then back to the padded layout
    const float* fm = ctx->fmods.as<float>() + (int64_t)step * 2 * D;
    {
      Prof pr(ctx, st, KC_LNMOD, 0, ln_bytes);
      HIPCHK(launch_layernorm(x, D, Mp, D, 1e-6f, nullptr, nullptr, fm, fm + D, nullptr, a_hi, a_lo, D, st, pk, ldA));
    }
    float* vp = ctx->velpk.as<float>() + p0 * mel;
    {
      Prof pr(ctx, st, KC_GEMM_MISC, gemm_flops(Mp, mel, D), 0);
      GemmCore g = core(a_hi, ldA, wsel(ctx, op, W(ctx, p + "proj_out.weight"), ctx->wp_hi.as<f16>(), ctx->wp_pk.as<f16>()), ldA, Mp, mel, D);
      HIPCHK(launch_gemm_store(op, g, epi_store(vp, mel, W(ctx, p + "proj_out.bias")), 1, st));
    }
    Prof pr(ctx, st, KC_ELEMWISE, 0, 2.0 * Mp * mel * 4);
    float* vel = ctx->vel.as<float>();
    HIPCHK(hipMemsetAsync(vel + (int64_t)s0 * n * mel, 0, (size_t)S * n * mel * sizeof(float), st));  // the padding moves by zero
    HIPCHK(launch_scatter_rows(vp, ctx->rowmap.as<int32_t>() + p0, Mp, mel, vel, st));
  }
  if (br < 0) {
    Prof pr(ctx, st, KC_ELEMWISE, 0, 4.0 * BN * mel * 4);
    HIPCHK(launch_cfg_euler(sg.ybase, sg.ydst, ctx->vel.as<float>(), BN * mel, nb == 2, ctx->dt_dev.as<float>() + step, ctx->cfg_dev.as<float>(),
                            sg.traj, ctx->dbg_vel.as<float>(), st));
  }
  return F5HIP_OK;
}

// ---- one ODE function evaluation + Euler update ---------------------------------------------------
// br < 0: the packed batch (cond rows, then uncond rows with CFG) followed by the CFG/ODE update.  br = 0 / 1: only the cond / uncond
// branch (B sequences) and NO update — the two branches never interact inside the backbone, so enqueue_steps may run them on two
// streams and join before the update.
int run_step(f5hip_ctx* ctx, int B, int n, const Stage& sg, int op, bool exact_attn, int use_mask, hipStream_t st, int br = -1) {
  const int step = sg.eidx, nb = ctx->nb;
  const auto& c = ctx->cfg;
  const int D = c.dim, mel = c.mel_dim, inner = c.heads * c.dim_head, F = c.ff_inner;
  const int64_t BN = (int64_t)B * n;
  const int S = br < 0 ? nb * B : B, s0 = br < 0 ? 0 : br * B;  // sequences handled here, first sequence
  const int64_t r0 = (int64_t)s0 * n;                            // first row
  const int M = S * n;
  const std::string p = "transformer.";
  const uint8_t* rowvalid = use_mask ? ctx->rowvalid.as<uint8_t>() + r0 : nullptr;
  float* x = ctx->x.as<float>() + r0 * D;
  float* h = ctx->h.as<float>() + r0 * D;
  float* c1 = ctx->c1.as<float>() + r0 * D;
  const float* cconst = ctx->cconst.as<float>() + r0 * D;
  const int wbytes = op == OP_F32 ? 4 : 2;
  const int npl = op == OP_F16X3 ? 3 : 1;

  {  // InputEmbedding.proj: only the x columns are per-step (dit.py:162); cond/text part is in cconst
    Prof pr(ctx, st, KC_GEMM_MISC, gemm_flops(BN, D, mel), 0);
    GemmCore g = core(sg.yin, mel, W(ctx, p + "input_embed.proj.weight"), 2 * mel + c.text_dim, (int)BN, D, mel);
    EpiStore e = epi_store(h, D, nullptr);
    e.res = cconst; e.ldres = D;
    if (br < 0 && nb == 2) { e.out2 = h + BN * D; e.res2 = cconst + BN * D; }  // uncond rows: same x columns, uncond constant part
    HIPCHK(launch_gemm_store(OP_F32, g, e, 1, st));
  }
  {  // ConvPositionEmbedding + residual (dit.py:163, modules.py:187-201)
    const int cpg = D / c.conv_pos_groups;
    Prof pr(ctx, st, KC_CONVPOS, 2 * gemm_flops(M, D, (int64_t)cpg * c.conv_pos_kernel) * (npl == 3 ? 1 : 1), 0);
    HIPCHK(launch_convpos(op, h, ctx->conv_w32[0].as<float>(), ctx->conv_whi[0].as<f16>(), ctx->conv_wlo[0].as<f16>(),
                          W(ctx, p + "input_embed.conv_pos_embed.conv1d.0.bias"), rowvalid, nullptr, S, n, D, c.conv_pos_groups,
                          c.conv_pos_kernel, c1, st, 0, 0, conv_mx(ctx, op, 0)));
    HIPCHK(launch_convpos(op, c1, ctx->conv_w32[1].as<float>(), ctx->conv_whi[1].as<f16>(), ctx->conv_wlo[1].as<f16>(),
                          W(ctx, p + "input_embed.conv_pos_embed.conv1d.2.bias"), rowvalid, h, S, n, D, c.conv_pos_groups,
                          c.conv_pos_kernel, x, st, 0, 0, conv_mx(ctx, op, 1)));
  }
  // Packed rows (option "packed_rows"; tables built by f5hip_sample): from here to the velocity the rows are the VALID rows of this call's
  // sequences, gathered once — the block GEMMs, LayerNorms and the attention cost sum(lengths), not sequences x longest (the reference's varlen
  // attention, modules.py:522-543, extended to the row-wise layers, whose padded rows nobody reads).
  const bool packed = ctx->pk_rows > 0;
  const int64_t p0 = packed ? ctx->cu_host[s0] : 0;  // first packed row of this call's sequences
  const int Mp = packed ? (int)(ctx->cu_host[s0 + S] - p0) : M;
  if (packed) {
    Prof pr(ctx, st, KC_ELEMWISE, 0, 2.0 * Mp * D * 4);
    HIPCHK(launch_gather_rows(ctx->x.as<float>(), ctx->rowmap.as<int32_t>() + p0, Mp, D, ctx->xpk.as<float>() + p0 * D, st));
    return run_blocks_packed(ctx, B, n, sg, op, S, s0, p0, Mp, st, br);
  }
  const float* mods_step = ctx->mods.as<float>() + (int64_t)step * c.depth * 6 * D;
  const int pk = op == OP_F16X3 ? 1 : 0;      // fp16x3 operands are packed hi/lo rows: lo plane = hi + 32 halves, row stride 2K
  const bool mx = ctx->mx_call;               // fp16m: the four block GEMMs read MX lines (same row strides as the hi | lo lines)
  const int opb = mx ? OP_F16M : op, pkb = mx ? 2 : pk;
  const int64_t ldA = (int64_t)D * (pk ? 2 : 1), ldO = (int64_t)inner * (pk ? 2 : 1), ldF = (int64_t)F * (pk ? 2 : 1);
  const int64_t ldAb = ldA, ldOb = ldO, ldFb = ldF;  // strides of the block operands (MX lines keep the hi | lo strides)
  float* a32 = op == OP_F32 ? ctx->a32.as<float>() + r0 * D : nullptr;
  f16* a_hi = op != OP_F32 ? ctx->a_hi.as<f16>() + r0 * ldA : nullptr;
  f16* a_lo = pk ? a_hi + 32 : nullptr;
  const void* A = op == OP_F32 ? (const void*)a32 : (const void*)a_hi;
  float* o32 = op == OP_F32 ? ctx->o32.as<float>() + r0 * inner : nullptr;
  f16* o_hi = op != OP_F32 ? ctx->o_hi.as<f16>() + r0 * ldO : nullptr;
  f16* o_lo = pk ? o_hi + 32 : nullptr;
  float* f32 = op == OP_F32 ? ctx->f32.as<float>() + r0 * F : nullptr;
  f16* f_hi = op != OP_F32 ? ctx->f_hi.as<f16>() + r0 * ldF : nullptr;
  f16* f_lo = pk ? f_hi + 32 : nullptr;
  const int32_t* kvlen = (c.attn_mask_enabled && use_mask) ? ctx->kvlen.as<int32_t>() + s0 : nullptr;
  const double ln_bytes = (double)M * D * (4 + wbytes * (op == OP_F16X3 ? 2 : 1));
  // long_skip_connection (dit.py:228,354-365): cat(x_after_blocks, x_before_blocks) is ONE [M, 2D] operand in the mode's layout, its
  // right half written here, its left half after the last block; the Linear(2D -> D, no bias) is then a single K = 2D GEMM
  const int64_t ldC = 2 * ldA;
  char* catbuf = c.long_skip_connection ? ctx->skipcat.as<char>() + r0 * 2 * D * (op == OP_F16 ? 2 : 4) : nullptr;
  auto emit_cat = [&](int col0) -> hipError_t {  // operand copy of x (layernorm mode 2 = no normalisation) into columns [col0, col0 + D)
    if (op == OP_F32) return launch_layernorm(x, D, M, D, 0.f, nullptr, nullptr, nullptr, nullptr, reinterpret_cast<float*>(catbuf) + col0, nullptr, nullptr, 2 * D, st, 0, 0, 2);
    f16* hi = reinterpret_cast<f16*>(catbuf) + pk_off(col0, pk);
    return launch_layernorm(x, D, M, D, 0.f, nullptr, nullptr, nullptr, nullptr, nullptr, hi, pk ? hi + 32 : nullptr, 0, st, pk, ldC, 2);
  };
  if (catbuf) {
    Prof pr(ctx, st, KC_LNMOD, 0, ln_bytes);
    HIPCHK(emit_cat(D));
  }

  for (int i = 0; i < c.depth; ++i) {
    const BlockW& bw = ctx->blocks[i];
    const float* md = mods_step + (int64_t)i * 6 * D;  // shift_msa, scale_msa, gate_msa, shift_mlp, scale_mlp, gate_mlp (modules.py:323)
    {
      Prof pr(ctx, st, KC_LNMOD, 0, ln_bytes);
      HIPCHK(launch_layernorm(x, D, M, D, 1e-6f, nullptr, nullptr, md + D, md, a32, a_hi, a_lo, D, st, pkb, ldAb));
    }
    int qks = QK_PLAIN;
    CHK(run_qkv(ctx, bw, A, ldAb, M, n, s0, opb, exact_attn, &qks, wbytes, st));
    CHK(run_attention(ctx, S, s0, n, op, exact_attn, qks, kvlen, o32, o_hi, o_lo, pkb, ldOb, st));
    {  // to_out + mask + gated residual: x += gate_msa * masked(attn) (modules.py:548-556,751)
      Prof pr(ctx, st, KC_GEMM_BLOCK, gemm_flops(M, D, inner), (double)M * inner * wbytes + (double)inner * D * wbytes + 2.0 * M * D * 4);
      GemmCore g = (core(op == OP_F32 ? (const void*)o32 : (const void*)o_hi, ldOb, wsel(ctx, opb, bw.wo, bw.wo_hi, bw.wo_pk, bw.wo_mx), ldOb, M, D, inner));
      EpiStore e = epi_store(x, D, bw.bo);
      e.colscale = md + 2 * D; e.rowmask = rowvalid; e.mask_mode = 1; e.res = x; e.ldres = D;
      HIPCHK(launch_gemm_store(opb, g, e, 1, st));
    }
    {
      Prof pr(ctx, st, KC_LNMOD, 0, ln_bytes);
      HIPCHK(launch_layernorm(x, D, M, D, 1e-6f, nullptr, nullptr, md + 4 * D, md + 3 * D, a32, a_hi, a_lo, D, st, pkb, ldAb));
    }
    {  // FeedForward: Linear -> tanh-GELU (modules.py:353-364,741)
      Prof pr(ctx, st, KC_GEMM_BLOCK, gemm_flops(M, F, D), (double)M * D * wbytes + (double)F * D * wbytes + (double)M * F * wbytes);
      GemmCore g = (core(A, ldAb, wsel(ctx, opb, bw.w1, bw.w1_hi, bw.w1_pk, bw.w1_mx), ldAb, M, F, D));
      EpiStore e = epi_store(f32, F, bw.b1, ACT_GELU_TANH);
      e.out16 = f_hi; e.out16_lo = f_lo; e.pk16 = pkb; e.ldo16 = ldFb;
      HIPCHK(launch_gemm_store(opb, g, e, 1, st));
    }
    {  // x += gate_mlp * ff (modules.py:755)
      Prof pr(ctx, st, KC_GEMM_BLOCK, gemm_flops(M, D, F), (double)M * F * wbytes + (double)F * D * wbytes + 2.0 * M * D * 4);
      GemmCore g = (core(op == OP_F32 ? (const void*)f32 : (const void*)f_hi, ldFb, wsel(ctx, opb, bw.w2, bw.w2_hi, bw.w2_pk, bw.w2_mx), ldFb, M, D, F));
      EpiStore e = epi_store(x, D, bw.b2);
      e.colscale = md + 5 * D; e.res = x; e.ldres = D;
      HIPCHK(launch_gemm_store(opb, g, e, 1, st));
    }
  }
  if (catbuf) {  // x = long_skip_connection(cat(x, residual)) (dit.py:364-365)
    {
      Prof pr(ctx, st, KC_LNMOD, 0, ln_bytes);
      HIPCHK(emit_cat(0));
    }
    Prof pr(ctx, st, KC_GEMM_MISC, gemm_flops(M, D, 2 * D), 0);
    GemmCore g = core(catbuf, ldC, wsel(ctx, op, ctx->wlong, ctx->wlong_hi, ctx->wlong_pk), ldC, M, D, 2 * D);
    HIPCHK(launch_gemm_store(op, g, epi_store(x, D, nullptr), 1, st));
  }
  {  // AdaLayerNorm_Final: chunk order is (scale, shift) (modules.py:344) + proj_out (dit.py:367-368)
    const float* fm = ctx->fmods.as<float>() + (int64_t)step * 2 * D;
    {
      Prof pr(ctx, st, KC_LNMOD, 0, ln_bytes);
      HIPCHK(launch_layernorm(x, D, M, D, 1e-6f, nullptr, nullptr, fm, fm + D, a32, a_hi, a_lo, D, st, pk, ldA));
    }
    Prof pr(ctx, st, KC_GEMM_MISC, gemm_flops(M, mel, D), 0);
    GemmCore g = core(A, ldA, wsel(ctx, op, W(ctx, p + "proj_out.weight"), ctx->wp_hi.as<f16>(), ctx->wp_pk.as<f16>()), ldA, M, mel, D);
    HIPCHK(launch_gemm_store(op, g, epi_store(ctx->vel.as<float>() + r0 * mel, mel, W(ctx, p + "proj_out.bias")), 1, st));
  }
  if (br < 0) {  // CFG combine + Euler update (cfm.py:190-191, torchdiffeq euler on the given grid)
    Prof pr(ctx, st, KC_ELEMWISE, 0, 4.0 * BN * mel * 4);
    HIPCHK(launch_cfg_euler(sg.ybase, sg.ydst, ctx->vel.as<float>(), BN * mel, nb == 2, ctx->dt_dev.as<float>() + step, ctx->cfg_dev.as<float>(),
                            sg.traj, ctx->dbg_vel.as<float>(), st));
  }
  return F5HIP_OK;
}


// ---- one ODE function evaluation + Euler update, UNetT backbone (reference src/f5_tts/model/backbones/unett.py:244-307) -------------
int run_step_unett(f5hip_ctx* ctx, int B, int n, const Stage& sg, int op, bool exact_attn, int use_mask, hipStream_t st, int br = -1) {
  const int step = sg.eidx, nb = ctx->nb;
  const auto& c = ctx->cfg;
  const int D = c.dim, mel = c.mel_dim, inner = c.heads * c.dim_head, F = c.ff_inner;
  const int ns = n + 1;
  const int S = br < 0 ? nb * B : B, s0 = br < 0 ? 0 : br * B;  // sequences handled here (br: see run_step)
  const int64_t BN = (int64_t)B * n;
  const int64_t r0 = (int64_t)s0 * ns, r0n = (int64_t)s0 * n;   // first token row / first frame row
  const int M = S * ns, Mall = nb * B * ns;
  const std::string p = "transformer.";
  const uint8_t* rowvalid = use_mask ? ctx->rowvalid.as<uint8_t>() + r0 : nullptr;  // [S, ns], time token valid (unett.py:273-274)
  float* x = ctx->x.as<float>() + r0 * D;
  float* h = ctx->h.as<float>() + r0n * D;
  float* c1 = ctx->c1.as<float>() + r0n * D;
  const float* cconst = ctx->cconst.as<float>() + r0n * D;
  const int wbytes = op == OP_F32 ? 4 : 2;
  const int pk = op == OP_F16X3 ? 1 : 0;
  const bool mx = ctx->mx_call;  // fp16m: q|k|v, out, FF1, FF2 read MX lines (same row strides); the skip projection and proj_out stay fp16x3
  const int opb = mx ? OP_F16M : op, pkb = mx ? 2 : pk;
  const int64_t pl = pk ? 2 : 1;
  const int64_t ldA = D * pl, ldO = inner * pl, ldF = F * pl, ldC = 2 * D * pl;  // operand row strides (elements)
  const int64_t ldAb = ldA, ldOb = ldO, ldFb = ldF;  // strides of the block operands (MX lines keep the hi | lo strides)
  float* a32 = op == OP_F32 ? ctx->a32.as<float>() + r0 * D : nullptr;
  f16* a_hi = op != OP_F32 ? ctx->a_hi.as<f16>() + r0 * ldA : nullptr;
  f16* a_lo = pk ? a_hi + 32 : nullptr;
  const void* A = op == OP_F32 ? (const void*)a32 : (const void*)a_hi;
  float* o32 = op == OP_F32 ? ctx->o32.as<float>() + r0 * inner : nullptr;
  f16* o_hi = op != OP_F32 ? ctx->o_hi.as<f16>() + r0 * ldO : nullptr;
  f16* o_lo = pk ? o_hi + 32 : nullptr;
  float* f32 = op == OP_F32 ? ctx->f32.as<float>() + r0 * F : nullptr;
  f16* f_hi = op != OP_F32 ? ctx->f_hi.as<f16>() + r0 * ldF : nullptr;
  f16* f_lo = pk ? f_hi + 32 : nullptr;
  const int32_t* kvlen = (c.attn_mask_enabled && use_mask) ? ctx->kvlen.as<int32_t>() + s0 : nullptr;
  const double ln_bytes = (double)M * D * (4 + wbytes * pl);
  // concat buffer of skip level l: [M, 2D] in the operand layout; columns [0, D) = current x, [D, 2D) = the saved skip
  const int64_t cat_elem_bytes = op == OP_F16 ? 2 : 4;  // fp32: 4 B; fp16: 2 B; packed hi/lo: 2 x 2 B per logical element
  auto cat_ptr = [&](int level) { return ctx->skipcat.as<char>() + ((int64_t)level * Mall + r0) * 2 * D * cat_elem_bytes; };
  // write an operand copy of x (mode 2 = no normalisation) into columns [col0, col0 + D) of a concat buffer
  auto emit_cat = [&](int level, int col0) -> hipError_t {
    char* base = cat_ptr(level);
    if (op == OP_F32) return launch_layernorm(x, D, M, D, 0.f, nullptr, nullptr, nullptr, nullptr, reinterpret_cast<float*>(base) + col0, nullptr, nullptr, 2 * D, st, 0, 0, 2);
    f16* hi = reinterpret_cast<f16*>(base) + pk_off(col0, pk);
    return launch_layernorm(x, D, M, D, 0.f, nullptr, nullptr, nullptr, nullptr, nullptr, hi, pk ? hi + 32 : nullptr, 0, st, pk, ldC, 2);
  };

  {  // InputEmbedding.proj (unett.py:100): per-step x columns + the step-invariant cond/text part
    Prof pr(ctx, st, KC_GEMM_MISC, gemm_flops(BN, D, mel), 0);
    GemmCore g = core(sg.yin, mel, W(ctx, p + "input_embed.proj.weight"), 2 * mel + c.text_dim, (int)BN, D, mel);
    EpiStore e = epi_store(h, D, nullptr);
    e.res = cconst; e.ldres = D;
    if (br < 0 && nb == 2) { e.out2 = h + BN * D; e.res2 = cconst + BN * D; }
    HIPCHK(launch_gemm_store(OP_F32, g, e, 1, st));
  }
  {  // ConvPositionEmbedding WITHOUT a mask (unett.py:101) + residual, written behind the time token of each sequence
    const int cpg = D / c.conv_pos_groups;
    Prof pr(ctx, st, KC_CONVPOS, 2 * gemm_flops((int64_t)S * n, D, (int64_t)cpg * c.conv_pos_kernel), 0);
    HIPCHK(launch_convpos(op, h, ctx->conv_w32[0].as<float>(), ctx->conv_whi[0].as<f16>(), ctx->conv_wlo[0].as<f16>(),
                          W(ctx, p + "input_embed.conv_pos_embed.conv1d.0.bias"), nullptr, nullptr, S, n, D, c.conv_pos_groups, c.conv_pos_kernel,
                          c1, st, 0, 0, conv_mx(ctx, op, 0)));
    HIPCHK(launch_convpos(op, c1, ctx->conv_w32[1].as<float>(), ctx->conv_whi[1].as<f16>(), ctx->conv_wlo[1].as<f16>(),
                          W(ctx, p + "input_embed.conv_pos_embed.conv1d.2.bias"), nullptr, h, S, n, D, c.conv_pos_groups, c.conv_pos_kernel, x, st,
                          ns, 1, conv_mx(ctx, op, 1)));
    HIPCHK(launch_set_token_rows(x, ctx->temb.as<float>() + (int64_t)step * D, S, ns, D, st));  // unett.py:272
  }

  for (int i = 0; i < c.depth; ++i) {
    const BlockW& bw = ctx->blocks[i];
    if (c.skip_connect_type == 2) {  // "none": no skip connections at all (unett.py:289-295)
    } else if (c.skip_connect_type == 1) {  // "add": x = x + skips.pop() (unett.py:294-295); the level's slab holds an fp32 [M, D] copy
      float* slab = reinterpret_cast<float*>(cat_ptr(i < c.depth / 2 ? i : c.depth - 1 - i));
      Prof pr(ctx, st, KC_ELEMWISE, 0, (double)M * D * 4 * (i < c.depth / 2 ? 2 : 3));
      if (i < c.depth / 2) HIPCHK(hipMemcpyAsync(slab, x, (size_t)M * D * 4, hipMemcpyDeviceToDevice, st));
      else HIPCHK(launch_add_inplace(x, slab, (int64_t)M * D, st));
    } else if (i < c.depth / 2) {  // skips.append(x) (unett.py:286-287): kept as the right half of that level's concat operand
      Prof pr(ctx, st, KC_LNMOD, 0, ln_bytes);
      HIPCHK(emit_cat(i, D));
    } else {  // x = skip_proj(cat(x, skips.pop())) (unett.py:289-293)
      const int level = c.depth - 1 - i;
      {
        Prof pr(ctx, st, KC_LNMOD, 0, ln_bytes);
        HIPCHK(emit_cat(level, 0));
      }
      Prof pr(ctx, st, KC_GEMM_BLOCK, gemm_flops(M, D, 2 * D), 0);
      GemmCore g = core(cat_ptr(level), ldC, wsel(ctx, op, bw.wskip, bw.wskip_hi, bw.wskip_pk), ldC, M, D, 2 * D);
      HIPCHK(launch_gemm_store(op, g, epi_store(x, D, nullptr), 1, st));
    }
    {  // attn_norm: x_transformers RMSNorm
      Prof pr(ctx, st, KC_LNMOD, 0, ln_bytes);
      HIPCHK(launch_layernorm(x, D, M, D, 0.f, bw.g_attn, nullptr, nullptr, nullptr, a32, a_hi, a_lo, D, st, pkb, ldAb, 1));
    }
    int qks = QK_PLAIN;
    CHK(run_qkv(ctx, bw, A, ldAb, M, ns, s0, opb, exact_attn, &qks, wbytes, st));
    CHK(run_attention(ctx, S, s0, ns, op, exact_attn, qks, kvlen, o32, o_hi, o_lo, pkb, ldOb, st));
    {  // x = attn(...) + x, padded rows of the attention output zero-filled (modules.py:548-556; unett.py:300)
      Prof pr(ctx, st, KC_GEMM_BLOCK, gemm_flops(M, D, inner), 0);
      GemmCore g = core(op == OP_F32 ? (const void*)o32 : (const void*)o_hi, ldOb, wsel(ctx, opb, bw.wo, bw.wo_hi, bw.wo_pk, bw.wo_mx), ldOb, M, D, inner);
      EpiStore e = epi_store(x, D, bw.bo);
      e.rowmask = rowvalid; e.mask_mode = 1; e.res = x; e.ldres = D;
      HIPCHK(launch_gemm_store(opb, g, e, 1, st));
    }
    {
      Prof pr(ctx, st, KC_LNMOD, 0, ln_bytes);
      HIPCHK(launch_layernorm(x, D, M, D, 0.f, bw.g_ff, nullptr, nullptr, nullptr, a32, a_hi, a_lo, D, st, pkb, ldAb, 1));
    }
    {
      Prof pr(ctx, st, KC_GEMM_BLOCK, gemm_flops(M, F, D), 0);
      GemmCore g = core(A, ldAb, wsel(ctx, opb, bw.w1, bw.w1_hi, bw.w1_pk, bw.w1_mx), ldAb, M, F, D);
      EpiStore e = epi_store(f32, F, bw.b1, ACT_GELU_TANH);
      e.out16 = f_hi; e.out16_lo = f_lo; e.pk16 = pkb; e.ldo16 = ldFb;
      HIPCHK(launch_gemm_store(opb, g, e, 1, st));
    }
    {  // x = ff(...) + x (unett.py:301)
      Prof pr(ctx, st, KC_GEMM_BLOCK, gemm_flops(M, D, F), 0);
      GemmCore g = core(op == OP_F32 ? (const void*)f32 : (const void*)f_hi, ldFb, wsel(ctx, opb, bw.w2, bw.w2_hi, bw.w2_pk, bw.w2_mx), ldFb, M, D, F);
      EpiStore e = epi_store(x, D, bw.b2);
      e.res = x; e.ldres = D;
      HIPCHK(launch_gemm_store(opb, g, e, 1, st));
    }
  }
  {  // norm_out(x)[:, 1:, :] -> proj_out (unett.py:305-307): one GEMM per sequence over rows 1..n of the normalised operand
    {
      Prof pr(ctx, st, KC_LNMOD, 0, ln_bytes);
      HIPCHK(launch_layernorm(x, D, M, D, 0.f, ctx->norm_out_g, nullptr, nullptr, nullptr, a32, a_hi, a_lo, D, st, pk, ldA, 1));
    }
    Prof pr(ctx, st, KC_GEMM_MISC, gemm_flops((int64_t)S * n, mel, D), 0);
    const char* Arow1 = reinterpret_cast<const char*>(A) + ldA * (op == OP_F32 ? 4 : 2);  // skip the time token of sequence 0
    GemmCore g = core(Arow1, ldA, wsel(ctx, op, W(ctx, p + "proj_out.weight"), ctx->wp_hi.as<f16>(), ctx->wp_pk.as<f16>()), ldA, n, mel, D);
    g.strideA = (int64_t)ns * ldA;
    EpiStore e = epi_store(ctx->vel.as<float>() + r0n * mel, mel, W(ctx, p + "proj_out.bias"));
    e.zdiv = 1; e.so1 = (int64_t)n * mel; e.so2 = 0;
    HIPCHK(launch_gemm_store(op, g, e, S, st));
  }
  if (br < 0) {
    Prof pr(ctx, st, KC_ELEMWISE, 0, 4.0 * BN * mel * 4);
    HIPCHK(launch_cfg_euler(sg.ybase, sg.ydst, ctx->vel.as<float>(), BN * mel, nb == 2, ctx->dt_dev.as<float>() + step, ctx->cfg_dev.as<float>(),
                            sg.traj, ctx->dbg_vel.as<float>(), st));
  }
  return F5HIP_OK;
}

// ---- MMDiT (reference backbones/mmdit.py, MMDiTBlock modules.py:763-845, JointAttnProcessor modules.py:563-705) -----------------------
// Row layout of every row-wise buffer (x, a_*, f_*): the audio rows of all S sequences first ([S*n, .]), then the text rows
// ([S*nt, .]).  Attention slabs are joint per (sequence, head): tokens [0, n) = audio frames, [n, n + nt) = text tokens.

// text stream input, once per utterance (mmdit.py:43-66): embedding + absolute sinusoid positions, padding rows zeroed
int run_text_embed_mmdit(f5hip_ctx* ctx, int B, int nt, const int64_t* text, hipStream_t st) {
  const auto& c = ctx->cfg;
  const int64_t BT = (int64_t)B * nt;
  if (nt > 8192) FAIL(F5HIP_ERR_INVALID, "nt=%d exceeds the 8192-row text position table", nt);
  STAGE(int32_t, tok, (size_t)BT);
  STAGE(uint8_t, valid, (size_t)BT);
  STAGE(uint8_t, cm, (size_t)2 * BT);
  memset(valid, 1, BT);
  for (int64_t i = 0; i < BT; ++i) {
    const int64_t id = text[i] + 1;  // 0 = filler / batch padding (mmdit.py:44)
    if (id < 0 || id > c.text_num_embeds) FAIL(F5HIP_ERR_INVALID, "text id %lld out of range at %lld", (long long)(id - 1), (long long)i);
    tok[i] = (int32_t)id;
    cm[i] = cm[BT + i] = id != 0;  // c_mask (mmdit.py:232)
  }
  HIPCHK(hipMemcpyAsync(ctx->tok.p, tok, BT * 4, hipMemcpyHostToDevice, st));
  HIPCHK(hipMemcpyAsync(ctx->valid.p, valid, BT, hipMemcpyHostToDevice, st));
  HIPCHK(hipMemcpyAsync(ctx->cmask.p, cm, 2 * BT, hipMemcpyHostToDevice, st));
  Prof pr(ctx, st, KC_TEXT, 0, 2.0 * BT * c.dim * 4.0 * 2);
  HIPCHK(launch_text_embed(ctx->tok.as<int32_t>(), ctx->valid.as<uint8_t>(), W(ctx, "transformer.text_embed.text_embed.weight"),
                           ctx->freqs_cis.as<float>(), B, nt, c.dim, c.text_mask_padding, 1, ctx->ctext0.as<float>(), st));
  return F5HIP_OK;
}

int run_step_mmdit(f5hip_ctx* ctx, int B, int n, int nt, const Stage& sg, int op, bool exact_attn, int use_mask, hipStream_t st) {
  const int step = sg.eidx, nb = ctx->nb;
  const auto& c = ctx->cfg;
  const int D = c.dim, mel = c.mel_dim, inner = c.heads * c.dim_head, F = c.ff_inner, H = c.heads, dh = c.dim_head;
  const int S = nb * B, ns = n + nt;
  const int64_t BN = (int64_t)B * n;
  const int Mx = S * n, Mc = S * nt;
  const std::string p = "transformer.";
  const uint8_t* rowvalid = use_mask ? ctx->rowvalid.as<uint8_t>() : nullptr;  // audio mask [S, n] (modules.py:699-700)
  const uint8_t* cmask = ctx->cmask.as<uint8_t>();                              // text mask [S, nt], always applied (modules.py:701-702)
  float* x = ctx->x.as<float>();                 // audio stream state [S*n, D]
  float* cs = x + (int64_t)Mx * D;               // text stream state  [S*nt, D]
  float* h = ctx->h.as<float>();
  float* c1 = ctx->c1.as<float>();
  const float* cconst = ctx->cconst.as<float>();
  const int wbytes = op == OP_F32 ? 4 : 2;
  const int pk = op == OP_F16X3 ? 1 : 0;
  const int64_t pl = pk ? 2 : 1, ldA = D * pl, ldO = inner * pl, ldF = F * pl;
  const int esz = op == OP_F32 ? 4 : 2;          // bytes per stored operand element
  // operand buffers: audio rows then text rows
  char* a_base = op == OP_F32 ? ctx->a32.as<char>() : ctx->a_hi.as<char>();
  char* o_base = op == OP_F32 ? ctx->o32.as<char>() : ctx->o_hi.as<char>();
  char* f_base = op == OP_F32 ? ctx->f32.as<char>() : ctx->f_hi.as<char>();
  auto a_rows = [&](int64_t row) { return a_base + row * ldA * esz; };
  auto f_rows = [&](int64_t row) { return f_base + row * ldF * esz; };
  // LayerNorm (no affine, eps 1e-6) with AdaLN modulation into the operand buffer
  auto ln_mod = [&](const float* src, int M, int64_t row0, const float* scale, const float* shift) -> hipError_t {
    Prof pr(ctx, st, KC_LNMOD, 0, (double)M * D * (4 + wbytes * pl));
    char* dst = a_rows(row0);
    if (op == OP_F32) return launch_layernorm(src, D, M, D, 1e-6f, nullptr, nullptr, scale, shift, reinterpret_cast<float*>(dst), nullptr, nullptr, D, st);
    f16* hi = reinterpret_cast<f16*>(dst);
    return launch_layernorm(src, D, M, D, 1e-6f, nullptr, nullptr, scale, shift, nullptr, hi, pk ? hi + 32 : nullptr, D, st, pk, ldA);
  };
  // joint attention mask (modules.py:643-657) only when the config enables it and a mask exists
  const bool amask = c.attn_mask_enabled && use_mask;
  const int32_t* kvlen = amask ? ctx->kvlen.as<int32_t>() : nullptr;
  const int32_t* kvlen2 = amask ? ctx->kvlen2.as<int32_t>() : nullptr;

  {  // AudioEmbedding (mmdit.py:79-85): linear over cat(x, cond) — the cond part is step-invariant (cconst) — then conv_pos + residual, no mask
    Prof pr(ctx, st, KC_GEMM_MISC, gemm_flops(BN, D, mel), 0);
    GemmCore g = core(sg.yin, mel, W(ctx, p + "audio_embed.linear.weight"), 2 * mel, (int)BN, D, mel);
    EpiStore e = epi_store(h, D, nullptr);
    e.res = cconst; e.ldres = D;
    if (nb == 2) { e.out2 = h + BN * D; e.res2 = cconst + BN * D; }
    HIPCHK(launch_gemm_store(OP_F32, g, e, 1, st));
  }
  {
    const int cpg = D / c.conv_pos_groups;
    Prof pr(ctx, st, KC_CONVPOS, 2 * gemm_flops(Mx, D, (int64_t)cpg * c.conv_pos_kernel), 0);
    HIPCHK(launch_convpos(op, h, ctx->conv_w32[0].as<float>(), ctx->conv_whi[0].as<f16>(), ctx->conv_wlo[0].as<f16>(),
                          W(ctx, p + "audio_embed.conv_pos_embed.conv1d.0.bias"), nullptr, nullptr, S, n, D, c.conv_pos_groups, c.conv_pos_kernel,
                          c1, st));
    HIPCHK(launch_convpos(op, c1, ctx->conv_w32[1].as<float>(), ctx->conv_whi[1].as<f16>(), ctx->conv_wlo[1].as<f16>(),
                          W(ctx, p + "audio_embed.conv_pos_embed.conv1d.2.bias"), nullptr, h, S, n, D, c.conv_pos_groups, c.conv_pos_kernel, x, st));
  }
  {  // the text stream starts every evaluation from the cached embedding (mmdit.py:190-206)
    Prof pr(ctx, st, KC_ELEMWISE, 0, 2.0 * Mc * D * 4);
    HIPCHK(hipMemcpyAsync(cs, ctx->ctext0.p, (size_t)Mc * D * 4, hipMemcpyDeviceToDevice, st));
  }
  const float* mods_step = ctx->mods.as<float>() + (int64_t)step * c.depth * 6 * D;
  const float* cmods_step = ctx->cmods.as<float>() + (int64_t)step * ((c.depth - 1) * 6 + 2) * D;

  // the joint slabs are written by two launches (and the qk_norm detour): MX P words are not offered here, the precise scores are the split's
  const int qks = exact_attn ? QK_PLAIN : (qk_scheme_wanted(ctx, op) == QK_PLAIN ? QK_PLAIN : QK_SPLIT);
  // fused q|k|v projection of one stream into the joint slabs
  auto qkv = [&](const BlockW& bw, bool text) -> int {
    const int M = text ? Mc : Mx, nseq = text ? nt : n;
    EpiQKV e{};
    e.bias = text ? bw.bqkv_c : bw.bqkv; e.rope_cs = ctx->rope.as<float>(); e.nseq = nseq; e.heads = H; e.dh = dh;
    e.pe_heads = -1; e.qscale = attn_qscale(dh, exact_attn);  // rope on every head of both streams, each from position 0 (modules.py:626-636)
    e.slab_n = ns; e.pos_off = text ? n : 0;
    e.qk_raw = c.qk_norm ? 1 : 0;
    if (exact_attn || c.qk_norm) { e.q32 = ctx->q32.as<float>(); e.k32 = ctx->k32.as<float>(); }
    if (exact_attn) {
      e.ldvt = (ns + 3) & ~3;
      e.vt32 = ctx->vt32.as<float>();
    } else {
      e.ldvt = (ns + 7) & ~7;
      e.q16 = ctx->q16.as<f16>(); e.k16 = ctx->k16.as<f16>(); e.vt16 = ctx->vt16.as<f16>();
      if (qks != QK_PLAIN) {
        e.q16_lo = ctx->q16_lo.as<f16>(); e.k16_lo = ctx->k16_lo.as<f16>();
        if (attn_v_split(ctx)) e.vt16_lo = ctx->vt16_lo.as<f16>();
      }
    }
    Prof pr(ctx, st, KC_GEMM_BLOCK, gemm_flops(M, 3 * inner, D), (double)M * D * wbytes + 3.0 * inner * D * wbytes + (double)M * 3 * inner * wbytes);
    GemmCore g = core(a_rows(text ? Mx : 0), ldA, text ? wsel(ctx, op, bw.wqkv_c, bw.wqkv_c_hi, bw.wqkv_c_pk) : wsel(ctx, op, bw.wqkv, bw.wqkv_hi, bw.wqkv_pk),
                      ldA, M, 3 * inner, D);
    HIPCHK(launch_gemm_qkv(op, g, e, st));
    return F5HIP_OK;
  };
  // to_out / to_out_c over the stream's rows of the joint attention output, one GEMM batch per sequence:
  //   state += gate * masked(o . W^T + b)
  auto out_proj = [&](bool text, const WOp& Wsel, const float* bias, const float* gate, float* state) -> int {
    const int M = text ? nt : n;
    Prof pr(ctx, st, KC_GEMM_BLOCK, gemm_flops((int64_t)S * M, D, inner), 0);
    GemmCore g = core(o_base + (int64_t)(text ? n : 0) * ldO * esz, ldO, Wsel, ldO, M, D, inner);
    g.strideA = (int64_t)ns * ldO;
    EpiStore e = epi_store(state, D, bias);
    e.colscale = gate; e.res = state; e.ldres = D;
    e.rowmask = text ? cmask : rowvalid; e.mask_mode = 1; e.smask = M;
    e.zdiv = 1; e.so1 = (int64_t)M * D; e.so2 = 0;
    HIPCHK(launch_gemm_store(op, g, e, S, st));
    return F5HIP_OK;
  };
  // FeedForward of one stream: state += gate * (gelu_tanh(a . W1^T + b1) . W2^T + b2)
  auto feed_forward = [&](int M, int64_t row0, const WOp& W1, const float* b1, const WOp& W2, const float* b2, const float* gate, float* state) -> int {
    {
      Prof pr(ctx, st, KC_GEMM_BLOCK, gemm_flops(M, F, D), 0);
      GemmCore g = core(a_rows(row0), ldA, W1, ldA, M, F, D);
      EpiStore e = epi_store(op == OP_F32 ? reinterpret_cast<float*>(f_rows(row0)) : nullptr, F, b1, ACT_GELU_TANH);
      if (op != OP_F32) { f16* fh = reinterpret_cast<f16*>(f_rows(row0)); e.out16 = fh; e.out16_lo = pk ? fh + 32 : nullptr; e.pk16 = pk; e.ldo16 = ldF; }
      HIPCHK(launch_gemm_store(op, g, e, 1, st));
    }
    Prof pr(ctx, st, KC_GEMM_BLOCK, gemm_flops(M, D, F), 0);
    GemmCore g = core(f_rows(row0), ldF, W2, ldF, M, D, F);
    EpiStore e = epi_store(state, D, b2);
    e.colscale = gate; e.res = state; e.ldres = D;
    HIPCHK(launch_gemm_store(op, g, e, 1, st));
    return F5HIP_OK;
  };

  for (int i = 0; i < c.depth; ++i) {
    const BlockW& bw = ctx->blocks[i];
    const bool last = i == c.depth - 1;  // context_pre_only (mmdit.py:118)
    const float* mx = mods_step + (int64_t)i * 6 * D;   // shift_msa, scale_msa, gate_msa, shift_mlp, scale_mlp, gate_mlp (modules.py:323)
    const float* mc = cmods_step + (int64_t)i * 6 * D;  // same for the text stream; the last block holds (scale, shift) only (modules.py:344)
    HIPCHK(ln_mod(x, Mx, 0, mx + D, mx));
    HIPCHK(last ? ln_mod(cs, Mc, Mx, mc, mc + D) : ln_mod(cs, Mc, Mx, mc + D, mc));
    CHK(qkv(bw, false));
    CHK(qkv(bw, true));
    if (c.qk_norm) {
      Prof pr(ctx, st, KC_ELEMWISE, 0, 0);
      HIPCHK(launch_qk_norm_rope(ctx->q32.as<float>(), ctx->k32.as<float>(), bw.qn, bw.kn, ctx->rope.as<float>(), (int64_t)S * H * ns, ns, H, dh, -1,
                                 attn_qscale(dh, exact_attn), 1e-6f, exact_attn ? nullptr : ctx->q16.as<f16>(),
                                 (!exact_attn && qks != QK_PLAIN) ? ctx->q16_lo.as<f16>() : nullptr,
                                 exact_attn ? nullptr : ctx->k16.as<f16>(),
                                 (!exact_attn && qks != QK_PLAIN) ? ctx->k16_lo.as<f16>() : nullptr, st, n, bw.qn_c, bw.kn_c));
    }
    CHK(run_attention(ctx, S, 0, ns, op, exact_attn, qks, kvlen, op == OP_F32 ? reinterpret_cast<float*>(o_base) : nullptr,
                      op != OP_F32 ? reinterpret_cast<f16*>(o_base) : nullptr, pk ? reinterpret_cast<f16*>(o_base) + 32 : nullptr, pk, ldO, st, kvlen2, n));
    if (!last) {  // text stream (modules.py:829-837)
      CHK(out_proj(true, wsel(ctx, op, bw.wo_c, bw.wo_c_hi, bw.wo_c_pk), bw.bo_c, mc + 2 * D, cs));
      HIPCHK(ln_mod(cs, Mc, Mx, mc + 4 * D, mc + 3 * D));
      CHK(feed_forward(Mc, Mx, wsel(ctx, op, bw.w1_c, bw.w1_c_hi, bw.w1_c_pk), bw.b1_c, wsel(ctx, op, bw.w2_c, bw.w2_c_hi, bw.w2_c_pk), bw.b2_c, mc + 5 * D, cs));
    }
    // audio stream (modules.py:839-843)
    CHK(out_proj(false, wsel(ctx, op, bw.wo, bw.wo_hi, bw.wo_pk), bw.bo, mx + 2 * D, x));
    HIPCHK(ln_mod(x, Mx, 0, mx + 4 * D, mx + 3 * D));
    CHK(feed_forward(Mx, 0, wsel(ctx, op, bw.w1, bw.w1_hi, bw.w1_pk), bw.b1, wsel(ctx, op, bw.w2, bw.w2_hi, bw.w2_pk), bw.b2, mx + 5 * D, x));
  }
  {  // norm_out + proj_out (mmdit.py:259-260)
    const float* fm = ctx->fmods.as<float>() + (int64_t)step * 2 * D;
    HIPCHK(ln_mod(x, Mx, 0, fm, fm + D));
    Prof pr(ctx, st, KC_GEMM_MISC, gemm_flops(Mx, mel, D), 0);
    GemmCore g = core(a_rows(0), ldA, wsel(ctx, op, W(ctx, p + "proj_out.weight"), ctx->wp_hi.as<f16>(), ctx->wp_pk.as<f16>()), ldA, Mx, mel, D);
    HIPCHK(launch_gemm_store(op, g, epi_store(ctx->vel.as<float>(), mel, W(ctx, p + "proj_out.bias")), 1, st));
  }
  {
    Prof pr(ctx, st, KC_ELEMWISE, 0, 4.0 * BN * mel * 4);
    HIPCHK(launch_cfg_euler(sg.ybase, sg.ydst, ctx->vel.as<float>(), BN * mel, nb == 2, ctx->dt_dev.as<float>() + step, ctx->cfg_dev.as<float>(),
                            sg.traj, ctx->dbg_vel.as<float>(), st));
  }
  return F5HIP_OK;
}

// The whole solve on the given evaluation tables: euler = one evaluation per step; midpoint (torchdiffeq fixed-grid "midpoint":
// y_mid = y + f(t, y) dt/2; y += dt f(t + dt/2, y_mid)) = two, through the scratch state ymid.
int enqueue_steps(f5hip_ctx* ctx, int B, int n, int nt, int steps, int method, int op, bool exact_attn, int use_mask, float* traj, hipStream_t st) {
  const bool unett = ctx->cfg.backbone == 1, mmdit = ctx->cfg.backbone == 2;
  const int64_t slab = (int64_t)B * n * ctx->cfg.mel_dim;
  float* y = ctx->y.as<float>();
  float* ymid = ctx->ymid.as<float>();
  // Small batches are latency-bound per kernel (20-90 us launches, 1-2 waves of workgroups): the cond and uncond branches of the CFG
  // batch are independent until the combine, so they run as two concurrent kernel chains (fork / join per evaluation; inside a
  // graph capture the side stream becomes a parallel branch of the graph).
  // Two chains against one, measured with the round-2 tiles at N = 1406 (same box, tools/r2_call23.sh / r2_call24.sh; ms per step):
  //   B = 1: 85.1 / 84.4 (the one-round tiles of a 2812-row launch win)   B = 2: 138 / 156   B = 3: 206 / 251   B = 4: 258 / 284
  //   B = 6: 376 / 432   B = 8: 254 / 269 (NFE 8)   B = 12: 390 / 406   B = 16: 522 / 497   B = 24: 779 / 752   B = 32: 960 / 978
  // Round 3, with the two-workgroups-per-CU tiles at 4k .. 40k rows (NFE 8, same box, profiles/r03g_chains.log; one chain / two chains):
  //   B = 2: 71.1 / 70.8   B = 3: 108.6 / 96.4   B = 4: 130.4 / 125.5   B = 6: 194.9 / 182.4   B = 8: 249.6 / 243.8   B = 12: 370.4 / 369.3
  //   B = 16: 488.7 / 489.8   B = 24: 740.5 / 729.8 — two chains are never worse any more, so: two chains from B = 2 on.
  const int64_t rows1 = ctx->pk_rows > 0 ? ctx->pk_rows / ctx->nb : (int64_t)B * n;  // rows of one chain (packed rows: the valid ones)
  const bool auto_split = rows1 >= 2048;
  const bool split = !mmdit && ctx->nb == 2 && !ctx->profile && (ctx->branch_streams == 1 || (ctx->branch_streams < 0 && auto_split));
  if (split && !ctx->side_stream) {
    HIPCHK(hipStreamCreateWithFlags(&ctx->side_stream, hipStreamNonBlocking));
    HIPCHK(hipEventCreateWithFlags(&ctx->ev_fork, hipEventDisableTiming));
    HIPCHK(hipEventCreateWithFlags(&ctx->ev_join, hipEventDisableTiming));
  }
  auto eval = [&](const Stage& sg) -> int {
    auto one = [&](hipStream_t s_, int br) {
      return unett ? run_step_unett(ctx, B, n, sg, op, exact_attn, use_mask, s_, br) : run_step(ctx, B, n, sg, op, exact_attn, use_mask, s_, br);
    };
    if (mmdit) return run_step_mmdit(ctx, B, n, nt, sg, op, exact_attn, use_mask, st);
    if (!split) return one(st, -1);
    HIPCHK(hipEventRecord(ctx->ev_fork, st));
    HIPCHK(hipStreamWaitEvent(ctx->side_stream, ctx->ev_fork, 0));
    CHK(one(st, 0));
    CHK(one(ctx->side_stream, 1));
    HIPCHK(hipEventRecord(ctx->ev_join, ctx->side_stream));
    HIPCHK(hipStreamWaitEvent(st, ctx->ev_join, 0));
    HIPCHK(launch_cfg_euler(sg.ybase, sg.ydst, ctx->vel.as<float>(), slab, 1, ctx->dt_dev.as<float>() + sg.eidx, ctx->cfg_dev.as<float>(), sg.traj,
                            ctx->dbg_vel.as<float>(), st));
    return F5HIP_OK;
  };
  for (int s = 0; s < steps; ++s) {
    float* tr = traj ? traj + (int64_t)(s + 1) * slab : nullptr;
    if (method == 0) {
      CHK(eval(Stage{s, y, y, y, tr}));
    } else {
      CHK(eval(Stage{2 * s, y, y, ymid, nullptr}));
      CHK(eval(Stage{2 * s + 1, ymid, y, y, tr}));
    }
  }
  return F5HIP_OK;
}

}  // namespace

// =================================================================================================
extern "C" {

int f5hip_abi_version(void) { return F5HIP_ABI_VERSION; }

const char* f5hip_last_error(const f5hip_ctx* ctx) { return ctx ? ctx->err.c_str() : g_create_err.c_str(); }

int f5hip_create(const f5hip_dit_config* dc, const f5hip_vocos_config* vc, int device, f5hip_ctx** out) {
  if (!dc || !out) { g_create_err = "null argument"; return F5HIP_ERR_INVALID; }
  const auto bad = [&](const char* m) { g_create_err = m; return F5HIP_ERR_INVALID; };
  if (dc->dim <= 0 || dc->depth <= 0 || dc->heads <= 0) return bad("dim, depth and heads must be > 0");  // heads*dim_head may differ from dim (modules.py:397-400)
  if (dc->dim_head != 64) return bad("dim_head must be 64 (attention kernels are built for dh=64)");
  if (dc->dim % 4 || dc->text_dim % 4 || dc->mel_dim % 4 || dc->ff_inner % 8 || dc->dim % 8) return bad("dim/text_dim/mel_dim/ff_inner alignment");
  if (dc->conv_pos_groups <= 0 || dc->dim % dc->conv_pos_groups) return bad("conv_pos_groups must divide dim");
  const int cpg = dc->dim / dc->conv_pos_groups;
  if (cpg != 16 && cpg != 32 && cpg != 48 && cpg != 64) return bad("dim/conv_pos_groups must be 16, 32, 48 or 64");
  if (!(dc->conv_pos_kernel & 1)) return bad("conv_pos_kernel must be odd");
  if (dc->mel_dim > 256) return bad("mel_dim must be <= 256");
  if (dc->backbone < 0 || dc->backbone > 2) return bad("backbone must be 0 (DiT), 1 (UNetT) or 2 (MMDiT)");
  if (dc->backbone == 2 && (dc->text_dim != dc->dim || dc->conv_layers != 0 || dc->depth < 1 || dc->pe_attn_head >= 0 || dc->long_skip_connection ||
                            dc->text_average_upsampling))
    return bad("MMDiT: text_dim == dim, no text conv blocks, rope on all heads, no DiT-only switches (mmdit.py:94-134)");
  if (dc->backbone == 1 && (dc->conv_layers != 0 || (dc->depth & 1))) return bad("UNetT: conv_layers must be 0 and depth even (unett.py:130)");
  if (dc->qk_norm != 0 && dc->qk_norm != 1) return bad("Unimplemented qk_norm (modules.py:409): 0 = None, 1 = rms_norm");
  if (dc->qk_norm && (dc->dim_head % 4 || dc->dim_head > 256 || (dc->dim_head & (dc->dim_head - 1)))) return bad("qk_norm: dim_head must be a power of two <= 256");
  if (dc->text_average_upsampling && !dc->text_mask_padding) return bad("text_embedding_average_upsampling requires text_mask_padding to be True (dit.py:43)");
  if (dc->backbone == 1 && (dc->long_skip_connection || dc->text_average_upsampling)) return bad("long_skip_connection / average upsampling are DiT options (dit.py:181-189)");
  if (dc->skip_connect_type < 0 || dc->skip_connect_type > 2) return bad("skip_connect_type: 0 = concat, 1 = add, 2 = none (unett.py:127)");
  if (dc->backbone != 1 && dc->skip_connect_type != 0) return bad("skip_connect_type is a UNetT option (unett.py:127)");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) {
    g_create_err = "no such HIP device (libf5hip has no CPU fallback)";
    return F5HIP_ERR_HIP;
  }
  if (hipSetDevice(device) != hipSuccess) { g_create_err = "hipSetDevice failed"; return F5HIP_ERR_HIP; }
  f5hip_ctx* ctx = new f5hip_ctx();
  ctx->cfg = *dc;
  ctx->device = device;
  if (vc) { ctx->vcfg = *vc; ctx->has_vocos = true; }
  build_slots(ctx);
  if (hipMalloc(reinterpret_cast<void**>(&ctx->blob), ctx->blob_elems * sizeof(float)) != hipSuccess) {
    g_create_err = "hipMalloc of the weight blob failed";
    delete ctx;
    return F5HIP_ERR_HIP;
  }
  (void)hipMemset(ctx->blob, 0, ctx->blob_elems * sizeof(float));
  *out = ctx;
  return F5HIP_OK;
}

int f5hip_destroy(f5hip_ctx* ctx) {
  if (!ctx) return F5HIP_OK;
  (void)hipSetDevice(ctx->device);
  (void)hipDeviceSynchronize();
  if (ctx->graph_exec) (void)hipGraphExecDestroy(ctx->graph_exec);
  if (ctx->cap_stream) (void)hipStreamDestroy(ctx->cap_stream);
  if (ctx->side_stream) { (void)hipStreamDestroy(ctx->side_stream); (void)hipEventDestroy(ctx->ev_fork); (void)hipEventDestroy(ctx->ev_join); }
  DevBuf* bufs[] = {&ctx->half_pool, &ctx->cond_pool, &ctx->conv_w32[0], &ctx->conv_w32[1], &ctx->conv_whi[0], &ctx->conv_whi[1], &ctx->conv_wlo[0],
                    &ctx->conv_wlo[1], &ctx->conv_wmx[0], &ctx->conv_wmx[1], &ctx->wp_hi, &ctx->wp_pk, &ctx->dwpack, &ctx->freqs_cis, &ctx->inv_freq, &ctx->vhead_w, &ctx->vhead_b,
                    &ctx->twiddle, &ctx->window, &ctx->melfb, &ctx->melfb_slaney, &ctx->melrange, &ctx->melrange_slaney, &ctx->melw, &ctx->melw_slaney, &ctx->t_dev, &ctx->dt_dev, &ctx->cfg_dev, &ctx->tsin, &ctx->th1, &ctx->tsilu,
                    &ctx->mods, &ctx->fmods, &ctx->temb, &ctx->skipcat, &ctx->ymid, &ctx->traj_buf, &ctx->tok, &ctx->valid, &ctx->textkeep, &ctx->rowvalid, &ctx->condmask, &ctx->kvlen, &ctx->tx,
                    &ctx->ta, &ctx->th, &ctx->tg, &ctx->sumsq, &ctx->step_cond, &ctx->cconst, &ctx->y, &ctx->h, &ctx->c1, &ctx->x, &ctx->a32,
                    &ctx->a_hi, &ctx->o32, &ctx->o_hi, &ctx->f32, &ctx->f_hi, &ctx->q32, &ctx->k32,
                    &ctx->vt32, &ctx->scores, &ctx->q16, &ctx->k16, &ctx->vt16, &ctx->q16_lo, &ctx->k16_lo, &ctx->vt16_lo, &ctx->vel, &ctx->rope, &ctx->dbg_vel, &ctx->vcol, &ctx->vx,
                    &ctx->va, &ctx->vh, &ctx->vlogits, &ctx->attn_part, &ctx->attn_stats_buf};
  for (DevBuf* b : bufs) b->release();
  ctx->stage.release();
  if (ctx->ev_last) (void)hipEventDestroy(ctx->ev_last);
  if (ctx->blob) (void)hipFree(ctx->blob);
  delete ctx;
  return F5HIP_OK;
}

int f5hip_num_tensors(const f5hip_ctx* ctx) { return ctx ? (int)ctx->slots.size() : 0; }

int f5hip_tensor_info(const f5hip_ctx* ctx, int i, const char** name, int64_t* numel, int64_t* off) {
  if (!ctx || i < 0 || i >= (int)ctx->slots.size()) return F5HIP_ERR_INVALID;
  if (name) *name = ctx->slots[i].name.c_str();
  if (numel) *numel = ctx->slots[i].numel;
  if (off) *off = ctx->slots[i].offset;
  return F5HIP_OK;
}

// Everything derived from the weights that is cached across calls: the per-step time-embedding / AdaLN tables are keyed on the time grid
// (prepare_time) and must not survive a weight change, and a captured graph holds no weight-derived pointer that moves, but is dropped with
// them for good measure.
static void invalidate_weight_caches(f5hip_ctx* ctx) {
  ctx->finalized = false;
  ctx->t_host.clear();
  ctx->ws_epoch++;
}

int f5hip_load_tensor(f5hip_ctx* ctx, const char* name, const float* data, int64_t numel) {
  if (!ctx || !name || !data) return F5HIP_ERR_INVALID;
  std::lock_guard<std::mutex> lk(ctx->mu);
  auto it = ctx->index.find(name);
  if (it == ctx->index.end()) FAIL(F5HIP_ERR_INVALID, "unexpected tensor '%s'", name);
  Slot& s = ctx->slots[it->second];
  if (s.numel != numel) FAIL(F5HIP_ERR_INVALID, "tensor '%s': expected %lld elements, got %lld", name, (long long)s.numel, (long long)numel);
  HIPCHK(hipSetDevice(ctx->device));
  HIPCHK(hipMemcpy(ctx->blob + s.offset, data, numel * sizeof(float), hipMemcpyHostToDevice));
  s.loaded = true;
  invalidate_weight_caches(ctx);
  return F5HIP_OK;
}

int f5hip_weight_blob(f5hip_ctx* ctx, void** p, int64_t* bytes) {
  if (!ctx || !p || !bytes) return F5HIP_ERR_INVALID;
  *p = ctx->blob;
  *bytes = ctx->blob_elems * (int64_t)sizeof(float);
  return F5HIP_OK;
}

int f5hip_mark_all_loaded(f5hip_ctx* ctx) {
  if (!ctx) return F5HIP_ERR_INVALID;
  std::lock_guard<std::mutex> lk(ctx->mu);
  for (auto& s : ctx->slots)
    if (!s.optional) s.loaded = true;
  invalidate_weight_caches(ctx);
  return F5HIP_OK;
}

int f5hip_loaded_mask(f5hip_ctx* ctx, uint8_t* mask, int n) {
  if (!ctx || !mask) return F5HIP_ERR_INVALID;
  std::lock_guard<std::mutex> lk(ctx->mu);
  if (n != (int)ctx->slots.size()) FAIL(F5HIP_ERR_INVALID, "loaded mask: %d entries for %d tensors", n, (int)ctx->slots.size());
  for (int i = 0; i < n; ++i) mask[i] = ctx->slots[i].loaded ? 1 : 0;
  return F5HIP_OK;
}

int f5hip_set_loaded_mask(f5hip_ctx* ctx, const uint8_t* mask, int n) {
  if (!ctx || !mask) return F5HIP_ERR_INVALID;
  std::lock_guard<std::mutex> lk(ctx->mu);
  if (n != (int)ctx->slots.size()) FAIL(F5HIP_ERR_INVALID, "loaded mask: %d entries for %d tensors", n, (int)ctx->slots.size());
  for (int i = 0; i < n; ++i) ctx->slots[i].loaded = mask[i] != 0;
  invalidate_weight_caches(ctx);
  return F5HIP_OK;
}

int f5hip_finalize_weights(f5hip_ctx* ctx) {
  if (!ctx) return F5HIP_ERR_INVALID;
  std::lock_guard<std::mutex> lk(ctx->mu);
  HIPCHK(hipSetDevice(ctx->device));
  ctx->t_host.clear();  // the blob may have been written directly (f5hip_weight_blob: the RCCL receive path)
  ctx->ws_epoch++;
  return finalize_impl(ctx);
}

int f5hip_set_option(f5hip_ctx* ctx, const char* key, int64_t value) {
  if (!ctx || !key) return F5HIP_ERR_INVALID;
  std::lock_guard<std::mutex> lk(ctx->mu);
  const std::string k = key;
  if (k == "use_graph") ctx->use_graph = value != 0;
  else if (k == "profile") ctx->profile = value != 0;
  else if (k == "attn_impl") { ctx->attn_impl = (int)value; ctx->ws_epoch++; }  // invalidates a captured graph
  else if (k == "branch_streams") { ctx->branch_streams = (int)value; ctx->ws_epoch++; }
  else if (k == "packed_rows") { ctx->packed_opt = value ? 1 : 0; ctx->ws_epoch++; }
  else if (k == "mx_weights") {  // read by the NEXT f5hip_finalize_weights (the copies are carved from the pool it sizes)
    ctx->mx_weights_opt = value ? 1 : 0;
  }
  else if (k == "attn_stats") {  // 0 off (default) / 1: the materialised-score attention (precision FP32, or attn_impl 1) accumulates how sharp
    // its softmax rows are — f5hip_attention_stats reads the figures; the flash kernels do not (they never see a whole row's sum at a point
    // where it is cheap to publish), so the option only has an effect on that path
    if (value) {
      HIPCHK(hipSetDevice(ctx->device));
      HIPCHK(ctx->attn_stats_buf.ensure(4 * sizeof(double), nullptr, true));  // (zeroed when first allocated)
    }
    ctx->attn_stats = value ? 1 : 0;
    ctx->ws_epoch++;  // (a captured graph holds the launch's pointer argument)
  }
  else if (k == "attn_kv_split") {  // 1 = off (default); 2..8 = flash attention with every query block cut into that many key ranges
    if (value < 1 || value > 8) FAIL(F5HIP_ERR_INVALID, "attn_kv_split must be in [1, 8]");
    ctx->attn_kv_split = (int)value;
    ctx->ws_epoch++;
  }
  else FAIL(F5HIP_ERR_INVALID, "unknown option '%s'", key);
  return F5HIP_OK;
}

int f5hip_attention_stats(f5hip_ctx* ctx, double* out4, int reset) {
  if (!ctx || !out4) return F5HIP_ERR_INVALID;
  std::lock_guard<std::mutex> lk(ctx->mu);
  if (!ctx->attn_stats_buf.p) FAIL(F5HIP_ERR_INVALID, "attention statistics were never switched on (option attn_stats)");
  HIPCHK(hipSetDevice(ctx->device));
  HIPCHK(hipDeviceSynchronize());  // a diagnostic call: whatever stream the samples ran on has finished
  HIPCHK(hipMemcpy(out4, ctx->attn_stats_buf.p, 4 * sizeof(double), hipMemcpyDeviceToHost));
  if (reset) HIPCHK(hipMemset(ctx->attn_stats_buf.p, 0, 4 * sizeof(double)));
  return F5HIP_OK;
}

int f5hip_num_kernel_stats(const f5hip_ctx*) { return KC_COUNT; }
int f5hip_kernel_stat(const f5hip_ctx* ctx, int i, const char** name, int64_t* calls, double* ms, double* flops, double* bytes) {
  if (!ctx || i < 0 || i >= KC_COUNT) return F5HIP_ERR_INVALID;
  if (name) *name = kclass_name(i);
  if (calls) *calls = ctx->stats[i].calls;
  if (ms) *ms = ctx->stats[i].ms;
  if (flops) *flops = ctx->stats[i].flops;
  if (bytes) *bytes = ctx->stats[i].bytes;
  return F5HIP_OK;
}
int f5hip_reset_kernel_stats(f5hip_ctx* ctx) {
  if (!ctx) return F5HIP_ERR_INVALID;
  std::lock_guard<std::mutex> lk(ctx->mu);
  for (auto& s : ctx->stats) s = KStat{};
  return F5HIP_OK;
}

// ---- mel -----------------------------------------------------------------------------------------
int f5hip_mel(f5hip_ctx* ctx, const float* wav, int batch, int64_t nsamp, float* out, int frame_major, int mel_type, void* stream) {
  if (!ctx || !wav || !out || batch <= 0) return F5HIP_ERR_INVALID;
  std::lock_guard<std::mutex> lk(ctx->mu);
  if (!ctx->finalized) FAIL(F5HIP_ERR_STATE, "weights not finalised");
  if (mel_type != 0 && mel_type != 1) FAIL(F5HIP_ERR_INVALID, "mel_type %d: only vocos (0) and bigvgan (1) exist (modules.py:127)", mel_type);
  // vocos type: centred frames, n_fft/2 reflect padding; bigvgan type: (n_fft - hop)/2 padding, frames that fit (modules.py:59-72)
  const int pad = mel_type == 1 ? (1024 - 256) / 2 : 512;
  if (nsamp < pad + 1) FAIL(F5HIP_ERR_INVALID, "reflect padding needs more than %d samples (got %lld)", pad, (long long)nsamp);
  const int frames = mel_type == 1 ? (int)((nsamp + 2 * pad - 1024) / 256) + 1 : 1 + (int)(nsamp / 256);
  if (frames <= 0) FAIL(F5HIP_ERR_INVALID, "wave of %lld samples is shorter than one frame", (long long)nsamp);
  hipStream_t st = (hipStream_t)stream;
  CallScope scope(ctx, st);
  CHK(scope.begin());
  {
    Prof pr(ctx, st, KC_MEL, 0, (double)batch * (nsamp * 4.0 + (double)frames * ctx->cfg.mel_dim * 4.0));
    HIPCHK(launch_mel(wav, batch, nsamp, frames, ctx->twiddle.as<float>(), ctx->window.as<float>(),
                      mel_type == 1 ? ctx->melw_slaney.as<float>() : ctx->melw.as<float>(),
                      mel_type == 1 ? ctx->melrange_slaney.as<int>() : ctx->melrange.as<int>(), mel_type == 1 ? ctx->melw_slaney_ld : ctx->melw_ld,
                      ctx->cfg.mel_dim, frame_major, pad,
                      mel_type == 1 ? 1e-9f : 0.f, out, st));
  }
  CHK(scope.finish());
  collect_prof(ctx, st);
  return F5HIP_OK;
}

// ---- sampler -------------------------------------------------------------------------------------
int f5hip_sample(f5hip_ctx* ctx, int B, int n, const float* cond, const uint8_t* cond_mask, const int64_t* text, int nt,
                 const int64_t* duration, int use_mask, const float* y0, const float* t, int steps, int ode_method, float cfg_strength,
                 int precision, float* out, float* trajectory, void* stream) {
  if (!ctx) return F5HIP_ERR_INVALID;
  std::lock_guard<std::mutex> lk(ctx->mu);
  if (!ctx->finalized) FAIL(F5HIP_ERR_STATE, "weights not finalised");
  if (!cond || !cond_mask || !text || !duration || !y0 || !t || !out) FAIL(F5HIP_ERR_INVALID, "null argument");
  if (B <= 0 || n <= 0 || steps <= 0 || nt <= 0) FAIL(F5HIP_ERR_INVALID, "batch, n, nt and steps must be positive");
  if (n > 8192 && ctx->cfg.conv_layers > 0) FAIL(F5HIP_ERR_INVALID, "n=%d exceeds the 8192-frame text position table (dit.py:47)", n);
  if (ode_method != 0 && ode_method != 1) FAIL(F5HIP_ERR_UNSUPPORTED, "ode_method %d: only euler (0) and midpoint (1) are built", ode_method);
  if (precision < F5HIP_PREC_FP32 || precision > F5HIP_PREC_FP16M) FAIL(F5HIP_ERR_INVALID, "bad precision %d", precision);
  for (int b = 0; b < B; ++b)
    if (duration[b] <= 0 || duration[b] > n) FAIL(F5HIP_ERR_INVALID, "duration[%d]=%lld outside (0, n=%d]", b, (long long)duration[b], n);
  hipStream_t st = (hipStream_t)stream;
  CallScope scope(ctx, st);
  CHK(scope.begin());
  const auto& c = ctx->cfg;
  const int op = op_of(precision);
  // attn_impl: 0 auto (fp32 -> materialised fp32 scores; fp16 -> flash attention, plain fp16 operands; fp16x3 / fp16m -> flash attention with
  // MX-corrected scores, qk_scheme_wanted above), 1 force materialised, 2 flash with every operand split, 3 flash with plain fp16 operands
  // in every mode (the default of rounds 2-4), 4 flash with split q, k and plain P, V, 5 = 0 for the half-precision modes, 6 / 7 = 0 with V / V
  // and P as hi + lo halves (round 6: the margin against sharper attention than any golden's)
  const bool exact_attn = ctx->attn_impl == 1 || (ctx->attn_impl == 0 && (precision == F5HIP_PREC_FP32 || !flash_attn_available()));
  if (!exact_attn && precision == F5HIP_PREC_FP32) FAIL(F5HIP_ERR_INVALID, "flash attention needs an fp16 precision mode");
  const int mel = c.mel_dim, D = c.dim;
  const int64_t BN = (int64_t)B * n;
  // fp16m: MX lines for the block GEMMs where they are built (finalize: mx_ok) and the call is one the pipelined kernel and the flash
  // epilogue take; anything else runs the call in fp16x3 — never less accurate, so the mode needs no error path
  // (the fused q|k|v launch of the pipelined kernel needs >= 8 tokens per sequence; tuning knobs that force a tile without an MX
  // instantiation or the general q|k|v index path take the launches away from it: gemm_mx_tiles_usable)
  ctx->mx_call = precision == F5HIP_PREC_FP16M && ctx->mx_ok && !exact_attn && ctx->attn_kv_split <= 1 && n >= 8 && gemm_mx_tiles_usable() &&
                 (2 * (BN + B) + 512) * 4 * std::max<int64_t>(std::max<int64_t>(D, c.ff_inner), (int64_t)c.heads * c.dim_head) < (int64_t)0x7ff00000;

  // cfg_strength < 1e-5: the reference evaluates only the conditional branch (cfm.py:166-177); otherwise cond + uncond rows are packed
  const int nb = cfg_strength < 1e-5f ? 1 : 2;
  if (ctx->nb != nb) { ctx->nb = nb; ctx->ws_epoch++; }
  CHK(ensure_workspace(ctx, B, n, nt, op, exact_attn));
  HIPCHK(ctx->ymid.ensure((size_t)BN * mel * sizeof(float)));
  {  // evaluation times and update coefficients of the chosen fixed-grid solver (torchdiffeq euler / midpoint on the supplied grid)
    const int E = ode_method == 0 ? steps : 2 * steps;
    std::vector<float> te(E), coef(E);
    for (int i = 0; i < steps; ++i) {
      const float dt = t[i + 1] - t[i];
      if (ode_method == 0) { te[i] = t[i]; coef[i] = dt; }
      else { const float half = 0.5f * dt; te[2 * i] = t[i]; coef[2 * i] = half; te[2 * i + 1] = t[i] + half; coef[2 * i + 1] = dt; }
    }
    CHK(prepare_time(ctx, te.data(), coef.data(), E, cfg_strength, st));
  }

  // masks (cfm.py:128-158)
  {
    // rows of the backbone's token sequences: UNetT prepends the (always valid) time token (unett.py:272-274)
    const int tok0 = c.backbone == 1 ? 1 : 0, ns = n + tok0;
    const int64_t BNs = (int64_t)B * ns;
    STAGE(uint8_t, rv, (size_t)2 * BNs);
    STAGE(int32_t, kv, (size_t)2 * B);
    STAGE(int32_t, kv2, (size_t)2 * B);
    STAGE(uint8_t, cmk, (size_t)BN);
    for (int b = 0; b < B; ++b) {
      for (int r = 0; r < ns; ++r) rv[(int64_t)b * ns + r] = rv[BNs + (int64_t)b * ns + r] = (r < tok0 || r - tok0 < duration[b]) ? 1 : 0;
      kv[b] = kv[B + b] = (int32_t)duration[b] + tok0;
    }
    HIPCHK(hipMemcpyAsync(ctx->rowvalid.p, rv, 2 * BNs, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(ctx->kvlen.p, kv, 2 * B * 4, hipMemcpyHostToDevice, st));
    std::fill(kv2, kv2 + 2 * B, 0);
    if (c.backbone == 2) {  // MMDiT joint key mask = cat(audio mask, c_mask) (modules.py:643-648): the valid text tokens as a second run
      for (int b = 0; b < B; ++b) {
        int len = 0;
        while (len < nt && text[(int64_t)b * nt + len] != -1) ++len;
        if (c.attn_mask_enabled && use_mask)
          for (int j = len; j < nt; ++j)
            if (text[(int64_t)b * nt + j] != -1)
              FAIL(F5HIP_ERR_UNSUPPORTED, "MMDiT with attn_mask_enabled: padding (-1) inside the text of sample %d (only trailing padding is built)", b);
        kv2[b] = kv2[B + b] = len;
      }
      HIPCHK(hipMemcpyAsync(ctx->kvlen2.p, kv2, 2 * B * 4, hipMemcpyHostToDevice, st));
    }
    memcpy(cmk, cond_mask, BN);  // cond_mask is a HOST array of the caller (include/f5hip.h): staged like the rest
    HIPCHK(hipMemcpyAsync(ctx->condmask.p, cmk, BN, hipMemcpyHostToDevice, st));
  }
  {  // packed rows: only where dropping the padded rows cannot change a valid row — padded keys must already be masked out of the attention
     // (attn_mask_enabled; with it off the reference lets the padding attend, modules.py:511-520) — and only for what is built (DiT, flash)
    int64_t valid = 0;
    for (int b = 0; b < B; ++b) valid += duration[b];
    const bool packed = ctx->packed_opt && use_mask && c.attn_mask_enabled && c.backbone == 0 && !exact_attn && !c.long_skip_connection && !c.qk_norm &&
                        ctx->attn_kv_split <= 1 && valid < BN && n < 65536 && 2 * B < 65536;
    ctx->pk_rows = packed ? nb * valid : 0;
    if (packed) {
      const int64_t Mp = ctx->pk_rows;
      STAGE(int32_t, rm, (size_t)Mp);
      STAGE(uint32_t, ri, (size_t)Mp);
      STAGE(int32_t, cu, (size_t)nb * B + 1);
      ctx->cu_host.assign((size_t)nb * B + 1, 0);
      int64_t r = 0;
      for (int s = 0; s < nb * B; ++s) {
        cu[s] = (int32_t)r;
        const int64_t len = duration[s % B];
        for (int64_t t2 = 0; t2 < len; ++t2, ++r) { rm[r] = (int32_t)((int64_t)s * n + t2); ri[r] = ((uint32_t)s << 16) | (uint32_t)t2; }
      }
      cu[nb * B] = (int32_t)r;
      std::copy(cu, cu + nb * B + 1, ctx->cu_host.begin());
      bool moved = false, mv = false;
      HIPCHK(ctx->rowmap.ensure((size_t)2 * BN * 4, &mv)); moved |= mv;
      HIPCHK(ctx->rowinfo.ensure((size_t)2 * BN * 4, &mv)); moved |= mv;
      HIPCHK(ctx->cu_rows.ensure((size_t)(2 * B + 1) * 4, &mv)); moved |= mv;
      HIPCHK(ctx->xpk.ensure((size_t)2 * BN * D * 4, &mv)); moved |= mv;
      HIPCHK(ctx->velpk.ensure((size_t)2 * BN * mel * 4, &mv)); moved |= mv;
      if (moved) ctx->ws_epoch++;
      HIPCHK(hipMemcpyAsync(ctx->rowmap.p, rm, (size_t)Mp * 4, hipMemcpyHostToDevice, st));
      HIPCHK(hipMemcpyAsync(ctx->rowinfo.p, ri, (size_t)Mp * 4, hipMemcpyHostToDevice, st));
      HIPCHK(hipMemcpyAsync(ctx->cu_rows.p, cu, (size_t)(nb * B + 1) * 4, hipMemcpyHostToDevice, st));
      // The packed q|k|v epilogue writes tokens below duration[b] only, and the flash kernel multiplies the V^T columns of masked keys by
      // P = 0: whatever an earlier call (another n, the padded layout) left in [duration[b], ldv) must be finite — cleared once per call
      // (2 B inner ldv halves: microseconds)
      const int64_t ldv = (n + 7) & ~7;
      HIPCHK(hipMemsetAsync(ctx->vt16.p, 0, (size_t)((int64_t)2 * B * c.heads * c.dim_head * ldv * 2), st));
      if (ctx->vt16_lo.p && attn_v_split(ctx)) HIPCHK(hipMemsetAsync(ctx->vt16_lo.p, 0, (size_t)((int64_t)2 * B * c.heads * c.dim_head * ldv * 2), st));
    }
  }
  {
    Prof pr(ctx, st, KC_ELEMWISE, 0, 0);
    HIPCHK(launch_mask_select(cond, ctx->condmask.as<uint8_t>(), BN, mel, ctx->step_cond.as<float>(), st));  // cfm.py:151-153
    HIPCHK(hipMemcpyAsync(ctx->y.p, y0, BN * mel * sizeof(float), hipMemcpyDeviceToDevice, st));
    if (trajectory) HIPCHK(hipMemcpyAsync(trajectory, y0, BN * mel * sizeof(float), hipMemcpyDeviceToDevice, st));
    HIPCHK(launch_rope_table(ctx->inv_freq.as<float>(), c.backbone == 2 ? std::max(n, nt) : n + (c.backbone == 1 ? 1 : 0), c.dim_head / 2,
                             ctx->rope.as<float>(), st));
  }
  if (c.backbone == 2) {
    CHK(run_text_embed_mmdit(ctx, B, nt, text, st));
    // step-invariant part of AudioEmbedding.linear (mmdit.py:79-83): W[:, mel:] . cond + b for the cond rows, b alone for the uncond rows
    Prof pr(ctx, st, KC_GEMM_MISC, 0, 0);
    const float* Wp = W(ctx, "transformer.audio_embed.linear.weight");
    const float* bp = W(ctx, "transformer.audio_embed.linear.bias");
    float* cc = ctx->cconst.as<float>();
    GemmCore g = core(ctx->step_cond.p, mel, Wp + mel, 2 * mel, (int)BN, D, mel);
    HIPCHK(launch_gemm_store(OP_F32, g, epi_store(cc, D, bp), 1, st));
    if (nb == 2) {  // alpha = 0: the epilogue just broadcasts the bias
      EpiStore e0 = epi_store(cc + BN * D, D, bp);
      e0.alpha = 0.f;
      HIPCHK(launch_gemm_store(OP_F32, g, e0, 1, st));
    }
  } else {
  CHK(run_text_embed(ctx, B, n, text, nt, duration, use_mask, st));
  {  // step-invariant part of InputEmbedding.proj: W_c.cond + W_t.text + b  (cond branch), W_t.text_uncond + b (uncond, cond=0)
    Prof pr(ctx, st, KC_GEMM_MISC, 0, 0);
    const float* Wp = W(ctx, "transformer.input_embed.proj.weight");
    const float* bp = W(ctx, "transformer.input_embed.proj.bias");
    const int ldw = 2 * mel + c.text_dim;
    float* cc = ctx->cconst.as<float>();
    GemmCore g = core(ctx->step_cond.p, mel, Wp + mel, ldw, (int)BN, D, mel);
    HIPCHK(launch_gemm_store(OP_F32, g, epi_store(cc, D, bp), 1, st));
    g = core(ctx->tx.p, c.text_dim, Wp + 2 * mel, ldw, (int)BN, D, c.text_dim);
    EpiStore e = epi_store(cc, D, nullptr);
    e.res = cc; e.ldres = D;
    HIPCHK(launch_gemm_store(OP_F32, g, e, 1, st));
    g = core(ctx->tx.as<float>() + BN * c.text_dim, c.text_dim, Wp + 2 * mel, ldw, (int)BN, D, c.text_dim);
    HIPCHK(launch_gemm_store(OP_F32, g, epi_store(cc + BN * D, D, bp), 1, st));
  }
  }

  // ---- the NFE loop: eager, or one hipGraph replay ------------------------------------------------
  bool done = false;
  if (ctx->use_graph && !ctx->profile) {
    auto& k = ctx->graph_key;
    // The captured loop writes the trajectory into a context-owned buffer (copied out below), so the graph does not depend on any
    // caller pointer and is replayed for every call of the same shape.
    float* tbuf = nullptr;
    if (trajectory) {
      bool moved = false;
      HIPCHK(ctx->traj_buf.ensure((size_t)(steps + 1) * BN * mel * sizeof(float), &moved));
      if (moved) ctx->ws_epoch++;
      tbuf = ctx->traj_buf.as<float>();
    }
    const bool hit = ctx->graph_exec && k.B == B && k.n == n && k.nt == (c.backbone == 2 ? nt : 0) && k.steps == steps && k.prec == precision && k.use_mask == use_mask &&
                     k.method == ode_method && k.traj == tbuf && k.ws_epoch == ctx->ws_epoch && k.pk_rows == ctx->pk_rows &&
                     k.pk_cond == (ctx->pk_rows ? ctx->cu_host[B] : 0);
    if (!hit) {
      if (ctx->graph_exec) { (void)hipGraphExecDestroy(ctx->graph_exec); ctx->graph_exec = nullptr; }
      if (!ctx->cap_stream) HIPCHK(hipStreamCreateWithFlags(&ctx->cap_stream, hipStreamNonBlocking));
      hipGraph_t graph = nullptr;
      HIPCHK(hipStreamBeginCapture(ctx->cap_stream, hipStreamCaptureModeThreadLocal));
      int r = enqueue_steps(ctx, B, n, nt, steps, ode_method, op, exact_attn, use_mask, tbuf, ctx->cap_stream);
      hipError_t ce = hipStreamEndCapture(ctx->cap_stream, &graph);
      if (r != F5HIP_OK) { if (graph) (void)hipGraphDestroy(graph); return r; }
      HIPCHK(ce);
      HIPCHK(hipGraphInstantiate(&ctx->graph_exec, graph, nullptr, nullptr, 0));
      (void)hipGraphDestroy(graph);
      k.B = B; k.n = n; k.nt = c.backbone == 2 ? nt : 0; k.steps = steps; k.prec = precision; k.use_mask = use_mask; k.method = ode_method; k.traj = tbuf; k.ws_epoch = ctx->ws_epoch;
      k.pk_rows = ctx->pk_rows; k.pk_cond = ctx->pk_rows ? ctx->cu_host[B] : 0;
    }
    HIPCHK(hipGraphLaunch(ctx->graph_exec, st));
    if (trajectory)  // states 1..steps (state 0 = y0 was copied above)
      HIPCHK(hipMemcpyAsync(trajectory + BN * mel, tbuf + BN * mel, (size_t)steps * BN * mel * sizeof(float), hipMemcpyDeviceToDevice, st));
    done = true;
  }
  if (!done) CHK(enqueue_steps(ctx, B, n, nt, steps, ode_method, op, exact_attn, use_mask, trajectory, st));

  {  // out = where(cond_mask, cond, y_final) (cfm.py:221-223)
    Prof pr(ctx, st, KC_ELEMWISE, 0, 0);
    HIPCHK(launch_where_rows(ctx->condmask.as<uint8_t>(), cond, ctx->y.as<float>(), BN, mel, out, st));
  }
  ctx->last_B = B;
  ctx->last_n = n;
  CHK(scope.finish());
  collect_prof(ctx, st);
  return F5HIP_OK;
}

int f5hip_debug_tensor(f5hip_ctx* ctx, int which, float* dst, int64_t numel, void* stream) {
  if (!ctx || !dst) return F5HIP_ERR_INVALID;
  std::lock_guard<std::mutex> lk(ctx->mu);
  const auto& c = ctx->cfg;
  const int64_t BN = (int64_t)ctx->last_B * ctx->last_n;
  if (BN == 0) FAIL(F5HIP_ERR_STATE, "no sample call yet");
  const float* src = nullptr;
  int64_t want = 0;
  if (c.backbone == 2 && which < 2) FAIL(F5HIP_ERR_UNSUPPORTED, "MMDiT keeps the text stream at its own length: no [batch, n, text_dim] text tap");
  switch (which) {
    case 0: src = ctx->tx.as<float>(); want = BN * c.text_dim; break;
    case 1: src = ctx->tx.as<float>() + BN * c.text_dim; want = BN * c.text_dim; break;
    case 2: src = ctx->dbg_vel.as<float>(); want = BN * c.mel_dim; break;
    case 3: src = ctx->h.as<float>(); want = 2 * BN * c.dim; break;
    default: FAIL(F5HIP_ERR_INVALID, "unknown debug tensor %d", which);
  }
  if (numel != want) FAIL(F5HIP_ERR_INVALID, "debug tensor %d has %lld elements, caller asked for %lld", which, (long long)want, (long long)numel);
  HIPCHK(hipMemcpyAsync(dst, src, want * sizeof(float), hipMemcpyDeviceToDevice, (hipStream_t)stream));
  return F5HIP_OK;
}

// ---- vocoder -------------------------------------------------------------------------------------
}  // extern "C"
namespace {
// head.out Linear -> exp / clip(1e2) / cos / sin -> inverse STFT (one kernel: transform + overlap-add); workspace vlogits sized by the caller
int vocos_head(f5hip_ctx* ctx, const float* hidden, int B, int T, float* out, hipStream_t st) {
  const auto& v = ctx->vcfg;
  const int C = v.dim, nout = v.n_fft + 2, npad = (nout + 3) & ~3;
  const int64_t R = (int64_t)B * T;
  {
    Prof pr(ctx, st, KC_VOCOS_GEMM, gemm_flops(R, nout, C), 0);
    GemmCore g = core(hidden, C, ctx->vhead_w.p, C, (int)R, npad, C);
    HIPCHK(launch_gemm_store(OP_F32, g, epi_store(ctx->vlogits.as<float>(), npad, ctx->vhead_b.as<float>()), 1, st));
  }
  {
    Prof pr(ctx, st, KC_ISTFT, 0, (double)R * npad * 4 + (double)B * 256.0 * (T - 1) * 4);  // logits once, samples once
    HIPCHK(launch_istft(ctx->vlogits.as<float>(), npad, B, T, ctx->twiddle.as<float>(), ctx->window.as<float>(), out, st));
  }
  return F5HIP_OK;
}
}  // namespace
extern "C" {
int f5hip_vocos_decode(f5hip_ctx* ctx, const float* melp, int B, int T, int channel_major, float* out, void* stream) {
  if (!ctx || !melp || !out) return F5HIP_ERR_INVALID;
  std::lock_guard<std::mutex> lk(ctx->mu);
  if (!ctx->has_vocos) FAIL(F5HIP_ERR_STATE, "context was created without a vocoder config");
  if (!ctx->finalized) FAIL(F5HIP_ERR_STATE, "weights not finalised");
  if (B <= 0 || T < 2) FAIL(F5HIP_ERR_INVALID, "vocos decode needs batch > 0 and frames >= 2");
  hipStream_t st = (hipStream_t)stream;
  CallScope scope(ctx, st);
  CHK(scope.begin());
  const auto& v = ctx->vcfg;
  const int C = v.dim, I = v.intermediate_dim, Cin = v.input_channels;
  const int64_t R = (int64_t)B * T;
  const int kcol = Cin * 7, ldc = (kcol + 3) & ~3;
  const int nout = v.n_fft + 2, npad = (nout + 3) & ~3;
  HIPCHK(ctx->vcol.ensure((size_t)R * ldc * 4));
  HIPCHK(ctx->vx.ensure((size_t)R * C * 4));
  HIPCHK(ctx->va.ensure((size_t)R * C * 4));
  HIPCHK(ctx->vh.ensure((size_t)R * I * 4));
  HIPCHK(ctx->vlogits.ensure((size_t)R * npad * 4));
  float* vx = ctx->vx.as<float>();
  {  // embed Conv1d(100 -> C, k=7, pad 3) as im2col + GEMM, then LayerNorm
    Prof pr(ctx, st, KC_VOCOS_OTHER, 0, 0);
    HIPCHK(launch_im2col7(melp, B, T, Cin, channel_major, ctx->vcol.as<float>(), ldc, st));
  }
  {
    Prof pr(ctx, st, KC_VOCOS_GEMM, gemm_flops(R, C, kcol), 0);
    GemmCore g = core(ctx->vcol.p, ldc, W(ctx, "backbone.embed.weight"), kcol, (int)R, C, kcol);
    HIPCHK(launch_gemm_store(OP_F32, g, epi_store(vx, C, W(ctx, "backbone.embed.bias")), 1, st));
  }
  {
    Prof pr(ctx, st, KC_VOCOS_OTHER, 0, 0);
    HIPCHK(launch_layernorm(vx, C, (int)R, C, 1e-6f, W(ctx, "backbone.norm.weight"), W(ctx, "backbone.norm.bias"), nullptr, nullptr, vx, nullptr,
                            nullptr, C, st));
  }
  for (int i = 0; i < v.num_layers; ++i) {
    const VocosLayerW& l = ctx->vlayers[i];
    {
      Prof pr(ctx, st, KC_VOCOS_OTHER, 0, 2.0 * R * C * 4);
      HIPCHK(launch_dwconv7_ln(vx, B, T, C, l.dw7, l.dw_b, l.ln_w, l.ln_b, 1e-6f, ctx->va.as<float>(), st));
    }
    Prof pr(ctx, st, KC_VOCOS_GEMM, 2 * gemm_flops(R, I, C), 0);
    GemmCore g = core(ctx->va.p, C, l.pw1_w, C, (int)R, I, C);
    HIPCHK(launch_gemm_store(OP_F32, g, epi_store(ctx->vh.as<float>(), I, l.pw1_b, ACT_GELU_ERF), 1, st));
    g = core(ctx->vh.p, I, l.pw2_w, I, (int)R, C, I);
    EpiStore e = epi_store(vx, C, l.pw2_b);
    e.colscale = l.gamma; e.res = vx; e.ldres = C;
    HIPCHK(launch_gemm_store(OP_F32, g, e, 1, st));
  }
  {
    Prof pr(ctx, st, KC_VOCOS_OTHER, 0, 0);
    HIPCHK(launch_layernorm(vx, C, (int)R, C, 1e-6f, W(ctx, "backbone.final_layer_norm.weight"), W(ctx, "backbone.final_layer_norm.bias"), nullptr,
                            nullptr, ctx->va.as<float>(), nullptr, nullptr, C, st));
  }
  CHK(vocos_head(ctx, ctx->va.as<float>(), B, T, out, st));
  CHK(scope.finish());
  collect_prof(ctx, st);
  return F5HIP_OK;
}

// The ISTFTHead alone (vocos heads.py; the reference's runnable copy: runtime/triton_trtllm/scripts/export_vocoder_to_onnx.py:43-59):
// hidden [B*T, dim] fp32 on the device -> waveform [B, 256 (T - 1)].
int f5hip_vocos_head(f5hip_ctx* ctx, const float* hidden, int B, int T, float* out, void* stream) {
  if (!ctx || !hidden || !out) return F5HIP_ERR_INVALID;
  std::lock_guard<std::mutex> lk(ctx->mu);
  if (!ctx->has_vocos) FAIL(F5HIP_ERR_STATE, "context was created without a vocoder config");
  if (!ctx->finalized) FAIL(F5HIP_ERR_STATE, "weights not finalised");
  if (B <= 0 || T < 2) FAIL(F5HIP_ERR_INVALID, "vocos head needs batch > 0 and frames >= 2");
  hipStream_t st = (hipStream_t)stream;
  CallScope scope(ctx, st);
  CHK(scope.begin());
  const int npad = (ctx->vcfg.n_fft + 2 + 3) & ~3;
  HIPCHK(ctx->vlogits.ensure((size_t)B * T * npad * 4));
  CHK(vocos_head(ctx, hidden, B, T, out, st));
  CHK(scope.finish());
  collect_prof(ctx, st);
  return F5HIP_OK;
}

int f5hip_istft(f5hip_ctx* ctx, const float* logits, int64_t ld, int B, int T, float* out, void* stream) {
  if (!ctx || !logits || !out) return F5HIP_ERR_INVALID;
  std::lock_guard<std::mutex> lk(ctx->mu);
  if (!ctx->finalized) FAIL(F5HIP_ERR_STATE, "weights not finalised (the twiddle / window tables are built there)");
  if (B <= 0 || T < 2 || ld < 1026 || (ld & 3)) FAIL(F5HIP_ERR_INVALID, "istft needs batch > 0, frames >= 2, ld >= 1026 and ld %% 4 == 0");
  hipStream_t st = (hipStream_t)stream;
  CallScope scope(ctx, st);
  CHK(scope.begin());
  {
    Prof pr(ctx, st, KC_ISTFT, 0, (double)B * T * ld * 4 + (double)B * 256.0 * (T - 1) * 4);
    HIPCHK(launch_istft(logits, ld, B, T, ctx->twiddle.as<float>(), ctx->window.as<float>(), out, st));
  }
  CHK(scope.finish());
  collect_prof(ctx, st);
  return F5HIP_OK;
}

}  // extern "C"
