// engine.h — host-side state of libf5hip: context, packed weight blob, derived layouts, workspace.
#pragma once
#ifndef F5_HIPEMU  // under the host shim (tests/hipemu) common.h brings the runtime stand-ins
#include <hip/hip_runtime.h>
#endif

#include <cstdint>
#include <mutex>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/f5hip.h"
#include "kernels.h"

struct Slot {
  std::string name;
  int64_t numel = 0;
  int64_t offset = 0;  // in floats, into the blob
  bool loaded = false;
  bool optional = false;
};

struct DevBuf {
  void* p = nullptr;
  size_t cap = 0;
  // grow-only; returns true when the pointer changed (invalidates captured graphs)
  hipError_t ensure(size_t bytes, bool* moved = nullptr, bool zero = false) {
    if (bytes <= cap) return hipSuccess;
    if (p) {
      hipError_t e = hipFree(p);
      if (e != hipSuccess) return e;
      p = nullptr;
      cap = 0;
    }
    size_t want = (bytes + 255) & ~size_t(255);
    hipError_t e = hipMalloc(&p, want);
    if (e != hipSuccess) return e;
    cap = want;
    if (moved) *moved = true;
    if (zero) return hipMemset(p, 0, want);
    return hipSuccess;
  }
  template <typename T>
  T* as() const { return reinterpret_cast<T*>(p); }
  void release() {
    if (p) (void)hipFree(p);
    p = nullptr;
    cap = 0;
  }
};

// Pinned, context-owned staging for small host inputs (masks, token ids, time tables) that a QUEUED copy reads after the entry point has
// returned: the boundary enqueues and returns (include/f5hip.h), so nothing on the call path may wait for the stream.  A ring of slots, one per
// call in flight; a slot is reused only after the event recorded behind its call's last copy has completed (blocks the host only when
// RING calls are outstanding).
struct HostStage {
  static constexpr int RING = 4;
  struct Chunk { char* p = nullptr; size_t cap = 0, used = 0; };
  struct SlotS {
    std::vector<Chunk> chunks;
    hipEvent_t done = nullptr;
    bool recorded = false;  // an event is pending behind this slot's copies
    bool open = false;      // begin() without end(): an error path left copies queued without an event
  } slot[RING];
  int cur = -1;
  bool in_call = false;  // between begin() and end()
  bool taken = false;    // this call has claimed a ring slot (its first alloc())
  // A call claims a ring slot only when it stages something (its first alloc()): entry points that stage nothing (f5hip_mel, f5hip_istft)
  // neither advance the ring nor make the host wait once RING calls are outstanding.
  hipError_t begin() {
    in_call = true;
    taken = false;
    return hipSuccess;
  }
  hipError_t last_error = hipSuccess;  // why the last alloc() returned null (the caller formats it) ...
  const char* last_step = "";          // ... and which step that was
  hipError_t claim() {
    const int next = (cur + 1) % RING;
    SlotS& s = slot[next];
    hipError_t e = hipSuccess;
    if (s.open) e = hipDeviceSynchronize();
    else if (s.recorded) e = hipEventSynchronize(s.done);
    if (e != hipSuccess) { last_error = e; last_step = "waiting for the staging slot's previous call"; return e; }  // the ring does not advance: the next alloc() retries this slot
    last_error = hipSuccess;  // (a later hipHostMalloc failure in alloc() must not be reported with a stale claim() cause)
    last_step = "";
    cur = next;
    s.recorded = false;
    s.open = true;
    for (auto& c : s.chunks) c.used = 0;
    taken = true;
    return hipSuccess;
  }
  void* alloc(size_t bytes) {
    if (!taken && claim() != hipSuccess) return nullptr;
    SlotS& s = slot[cur];
    bytes = (bytes + 63) & ~size_t(63);
    for (auto& c : s.chunks)
      if (c.cap - c.used >= bytes) { void* r = c.p + c.used; c.used += bytes; return r; }
    Chunk c;
    c.cap = bytes > (size_t(1) << 20) ? bytes : (size_t(1) << 20);
    if (const hipError_t e = hipHostMalloc(reinterpret_cast<void**>(&c.p), c.cap, 0); e != hipSuccess) { last_error = e; last_step = "hipHostMalloc"; return nullptr; }
    c.used = bytes;
    s.chunks.push_back(c);
    return c.p;
  }
  // Always reached once per begin() — the entry points hold a CallScope (api.cpp) whose destructor calls it on every return path — so a
  // failed call's queued copies get their event too and the slot is reclaimed by an event wait, not by a device-wide synchronise.
  hipError_t end(hipStream_t st) {
    in_call = false;
    if (!taken) return hipSuccess;
    SlotS& s = slot[cur];
    if (!s.done) { hipError_t e = hipEventCreateWithFlags(&s.done, hipEventDisableTiming); if (e != hipSuccess) return e; }
    hipError_t e = hipEventRecord(s.done, st);
    if (e != hipSuccess) return e;  // (the slot stays `open`: the next claim of it synchronises the device)
    s.recorded = true;
    s.open = false;
    return hipSuccess;
  }
  void release() {
    for (auto& s : slot) {
      for (auto& c : s.chunks) (void)hipHostFree(c.p);
      s.chunks.clear();
      if (s.done) (void)hipEventDestroy(s.done);
      s.done = nullptr;
    }
  }
};

struct BlockW {  // per DiT block
  const float *wqkv, *bqkv, *wo, *bo, *w1, *b1, *w2, *b2;  // fp32 (blob)
  f16 *wqkv_hi, *wo_hi, *w1_hi, *w2_hi;  // plain fp16 rows [N, K]             (precision FP16)
  f16 *wqkv_pk, *wo_pk, *w1_pk, *w2_pk;  // packed hi/lo rows [N, 2K] (gemm.h) (precision FP16X3)
  f16 *wqkv_mx = nullptr, *wo_mx = nullptr, *w1_mx = nullptr, *w2_mx = nullptr;  // MX lines [N, 2K] (common.h) (precision FP16M; DiT / UNetT backbone, ctx->mx_ok)
  // UNetT only: RMSNorm gains and the later-half skip projection Linear(2D -> D, no bias)
  const float *g_attn = nullptr, *g_ff = nullptr, *wskip = nullptr;
  f16 *wskip_hi = nullptr, *wskip_pk = nullptr;
  const float *qn = nullptr, *kn = nullptr;  // qk_norm == rms_norm: RMSNorm gains over dim_head (modules.py:402-409)
  // MMDiT only: the text ("context") stream of the block (modules.py:791-814); absent (null) in the last, context_pre_only block
  const float *wqkv_c = nullptr, *bqkv_c = nullptr, *wo_c = nullptr, *bo_c = nullptr, *w1_c = nullptr, *b1_c = nullptr, *w2_c = nullptr,
              *b2_c = nullptr, *qn_c = nullptr, *kn_c = nullptr;
  f16 *wqkv_c_hi = nullptr, *wo_c_hi = nullptr, *w1_c_hi = nullptr, *w2_c_hi = nullptr;
  f16 *wqkv_c_pk = nullptr, *wo_c_pk = nullptr, *w1_c_pk = nullptr, *w2_c_pk = nullptr;
};
struct TextBlockW {
  const float *dw_b, *ln_w, *ln_b, *pw1_w, *pw1_b, *gamma, *beta, *pw2_w, *pw2_b;
  float* dw7;  // [7, T]
};
struct VocosLayerW {
  const float *dw_b, *ln_w, *ln_b, *pw1_w, *pw1_b, *pw2_w, *pw2_b, *gamma;
  float* dw7;
};

enum KClass {
  KC_GEMM_BLOCK = 0,  // QKV / out / FF1 / FF2 of the DiT blocks (the dominant kernel)
  KC_ATTN,
  KC_CONVPOS,
  KC_LNMOD,
  KC_GEMM_MISC,
  KC_TEXT,
  KC_ELEMWISE,
  KC_MEL,
  KC_VOCOS_GEMM,
  KC_VOCOS_OTHER,
  KC_ISTFT,
  KC_COUNT
};

struct KStat {
  int64_t calls = 0;
  double ms = 0, flops = 0, bytes = 0;
};

struct ProfRec {
  int kclass;
  hipEvent_t e0, e1;
  double flops, bytes;
};

struct f5hip_ctx {
  f5hip_dit_config cfg{};
  f5hip_vocos_config vcfg{};
  bool has_vocos = false;
  int device = 0;
  std::mutex mu;
  std::string err;

  // weights
  std::vector<Slot> slots;
  std::unordered_map<std::string, int> index;
  float* blob = nullptr;
  int64_t blob_elems = 0;
  int64_t dit_elems = 0;
  bool finalized = false;

  // derived layouts
  DevBuf half_pool;  // all f16 hi/lo copies
  DevBuf cond_pool;  // per-output-channel conditioning of those copies: scale[rows] | alpha[rows] per weight matrix (GemmCore::w_alpha)
  std::unordered_map<const void*, const float*> walpha;  // half-precision weight copy (plain or packed) -> its alpha vector
  std::vector<BlockW> blocks;
  std::vector<TextBlockW> tblocks;
  std::vector<VocosLayerW> vlayers;
  DevBuf conv_w32[2], conv_whi[2], conv_wlo[2];
  DevBuf conv_wmx[2];  // fp16m: the per-tap tiles as MX lines [G][K][co][2 x 128 B] (64 channels per group only; else empty)
  DevBuf wp_hi, wp_pk;                 // proj_out f16: plain rows, packed hi/lo rows
  DevBuf dwpack;                       // [7,C] depthwise weights (text + vocos)
  DevBuf freqs_cis;                    // [8192, text_dim]
  DevBuf inv_freq;                     // [dh/2]
  DevBuf vhead_w, vhead_b;             // padded vocos head [1028, C], [1028]
  DevBuf melrange, melrange_slaney;              // per mel channel: [first, one past last) bin of its filterbank triangle
  DevBuf melw, melw_slaney;                      // ... and the weights of those bins as contiguous zero-padded runs [nmel][melw_ld]
  int melw_ld = 0, melw_slaney_ld = 0;
  DevBuf twiddle, window, melfb, melfb_slaney;  // audio tables (HTK filterbank of the Vocos-type mel, slaney one of the BigVGAN type)
  const float *adaln_w = nullptr, *adaln_b = nullptr;  // [depth*6D, D], [depth*6D]

  // time-grid dependent tables (cached on the last grid)
  std::vector<float> t_host;
  DevBuf t_dev, dt_dev, cfg_dev, tsin, th1, tsilu, mods, fmods;
  DevBuf temb;                         // UNetT: raw time embedding per step [steps, D] (the time token)
  const float* norm_out_g = nullptr;   // UNetT final RMSNorm gain
  DevBuf skipcat;                      // UNetT: depth/2 concat buffers [2B*(n+1), 2D] in the mode's operand layout (skip type "add": the
                                       // same slabs hold fp32 [rows, D] copies); DiT long_skip_connection: one such buffer
  const float* wlong = nullptr;        // DiT long_skip_connection.weight [D, 2D] (dit.py:228) and its fp16 operand copies
  f16 *wlong_hi = nullptr, *wlong_pk = nullptr;
  // MMDiT: AdaLN tables of the text stream, its per-utterance embedding [2B*nt, D], its row mask [2B*nt] and valid-token counts [2B]
  const float *adaln_c_w = nullptr, *adaln_c_b = nullptr;  // [(depth-1)*6D + 2D, D], [(depth-1)*6D + 2D]
  DevBuf cmods, ctext0, cmask, kvlen2;
  int ws_nt = 0;
  DevBuf avgidx;                       // text average upsampling: source token position per frame [B, n] int32, -1 = zero row

  // workspace (grow-only)
  int ws_B = 0, ws_n = 0;
  DevBuf tok, valid, textkeep, rowvalid, condmask, kvlen;
  // option "packed_rows" (DiT backbone, flash attention, attn_mask_enabled): the block loop of a ragged batch runs over the VALID rows only.
  // Row r of the packed order (sequence by sequence: cond 0 .. B-1, then uncond) is padded row rowmap[r] = token rowinfo[r] & 0xffff of
  // sequence rowinfo[r] >> 16; sequence s starts at packed row cu_rows[s].  pk_rows = 0: the padded layout is in use.
  int packed_opt = 0;
  int mx_weights_opt = 1;  // option "mx_weights": build the MX-line copies of the block weights at finalize (2 more halves per element)
  int64_t pk_rows = 0;
  std::vector<int32_t> cu_host;
  DevBuf rowmap, rowinfo, cu_rows, xpk, velpk;
  DevBuf tx, ta, th, tg, sumsq;
  DevBuf step_cond, cconst, y, h, c1, x;
  DevBuf a32, a_hi, o32, o_hi, f32, f_hi;  // *_hi: plain fp16 rows, or packed hi/lo rows (twice the size) in fp16x3 mode
  DevBuf q32, k32, vt32, scores, q16, k16, vt16, q16_lo, k16_lo, vt16_lo;
  DevBuf traj_buf;                     // trajectory slots written by the captured graph (copied to the caller's buffer)
  DevBuf vel, rope, dbg_vel, ymid;     // ymid: scratch ODE state of the midpoint solver
  int nb = 2;                          // packed branches per utterance: 2 = cond + uncond (CFG), 1 = cond only (cfg_strength < 1e-5)
  // vocos workspace
  DevBuf vcol, vx, va, vh, vlogits;

  // options / measurement
  bool use_graph = false;
  bool profile = false;
  int attn_impl = 0;  // 0 auto (fp32: materialised scores; fp16: flash attention, plain fp16 operands; fp16x3 / fp16m: flash attention, scores
                      // from fp16 hi . hi + MX-fp6 corrections), 1 materialised fp32, 2 flash with all operands split, 3 flash with plain
                      // fp16 operands in every mode, 4 flash with split q, k + plain P, V, 5 = 0 (api.cpp qk_scheme_wanted)
  KStat stats[KC_COUNT];
  std::vector<ProfRec> prof;

  // last sample (debug taps)
  int last_B = 0, last_n = 0;

  // graph cache
  hipGraphExec_t graph_exec = nullptr;
  struct GraphKey {
    int B = 0, n = 0, nt = 0, steps = 0, prec = -1, use_mask = 0, method = 0;
    int64_t pk_rows = 0, pk_cond = 0;  // packed rows in total / of the cond half (launch sizes of the captured loop)
    float* traj = nullptr;
    uint64_t ws_epoch = 0;
  } graph_key;
  uint64_t ws_epoch = 0;
  hipStream_t cap_stream = nullptr;
  bool mx_ok = false;          // finalize: the MX-line weight copies exist (DiT backbone whose block GEMMs the pipelined kernel takes)
  bool mx_call = false;        // f5hip_sample: this call's block GEMMs read MX lines (precision FP16M and every condition holds)
  int attn_kv_split = 1;       // option "attn_kv_split": key ranges per query block in the flash kernel (1 = off); attn_part = its scratch
  DevBuf attn_part;
  int attn_stats = 0;          // option "attn_stats": the materialised-score attention accumulates the rows' largest probabilities in
  DevBuf attn_stats_buf;       // attn_stats_buf (4 doubles, f5hip_attention_stats)
  // host inputs of queued copies (the entry points never wait for the stream) and the ordering of calls that arrive on different streams:
  // the workspace is one per context, so a call on a new stream first waits (on the GPU, not the host) for the previous call's work
  HostStage stage;
  hipStream_t last_stream = nullptr;
  hipEvent_t ev_last = nullptr;
  bool have_last = false;
  // cond / uncond branches on two streams (small batches): -1 auto, 0 off, 1 on
  int branch_streams = -1;
  hipStream_t side_stream = nullptr;
  hipEvent_t ev_fork = nullptr, ev_join = nullptr;
};
