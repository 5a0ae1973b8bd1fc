/*
 * f5hip.h — C ABI of libf5hip.so: the MI355X (gfx950) engine for the F5-TTS inference hot path.
 *
 * The reference (SWivid/F5-TTS) has no FFI on this path: its seam is the duck-typed Python objects
 * returned by load_model()/load_vocoder() (reference src/f5_tts/infer/utils_infer.py:238-276, :106-145)
 * and consumed by infer_batch_process() (:440-593).  The closest thing to an operator ABI in the tree
 * is the TensorRT-LLM engine call (reference
 * src/f5_tts/runtime/triton_trtllm/model_repo_f5_tts/f5_tts/1/f5_tts_trtllm.py:268-277,338-364:
 * caller-allocated device buffers, a raw stream handle, boolean status).  This header follows that
 * precedent: plain pointers and sizes, caller-owned I/O device buffers, library-owned weights and
 * workspace, int status codes (0 = ok) + f5hip_last_error().  No torch types cross this boundary.
 *
 * Each entry point names the reference interface it replaces.  The Python binding that a reference
 * maintainer would add is f5-tts_amd/binding.py (ctypes) — shown in INTEGRATION.md.
 *
 * All "device" pointers are HIP device pointers valid on the context's device; all launches go to
 * the `stream` argument (a hipStream_t passed as void*; NULL = the legacy default stream).  Calls
 * return after enqueueing unless stated otherwise; outputs are complete when the stream reaches the
 * end of the enqueued work.  Host arrays passed to a call (text ids, masks, time grids) are copied into
 * pinned, context-owned staging before the call returns, so the caller may free or reuse them at once;
 * no entry point on the sample / mel / vocoder paths waits for the stream (exceptions: the first call at
 * a new shape grows the workspace with hipMalloc, which synchronises the device; "profile" mode reads
 * its events back).  A context owns ONE workspace: its mutex covers the host-side enqueue only, so
 * concurrent f5hip_sample() calls from a thread pool (reference utils_infer.py:540-543) overlap their
 * host work with the GPU work already queued; on the GPU the calls of one context run in submission
 * order — a call that arrives on a different stream than its predecessor first waits (hipStreamWaitEvent,
 * GPU side) for the predecessor's work.
 */
#ifndef F5HIP_H
#define F5HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define F5HIP_ABI_VERSION 10 /* v10: option "attn_stats" + f5hip_attention_stats; v9: option "mx_weights"; precision value 4 (a
                                microbenchmark-only operand form of v8) removed */

/* status codes */
enum {
  F5HIP_OK = 0,
  F5HIP_ERR_INVALID = 1,   /* bad argument / shape (mirrors the reference's asserts, cfm.py:109,124) */
  F5HIP_ERR_HIP = 2,       /* a HIP runtime call failed */
  F5HIP_ERR_STATE = 3,     /* call order (weights not finalised, ...) */
  F5HIP_ERR_UNSUPPORTED = 4
};

/* GEMM operand precision of the per-step DiT path (everything else is always fp32):
 *   FP32    exact fp32 MFMA (v_mfma_f32_32x32x2_f32)                     — parity mode
 *   FP16X3  fp16 hi/lo split operands, 3 MFMAs per product, fp32 accum    — ~fp32 accuracy
 *   FP16    fp16 operands, fp32 accumulate (what the reference runs on GPU: utils_infer.py:191-199)
 *   FP16M   FP16X3 with the two correction terms of every product of the DiT block GEMMs (q|k|v, out, FF1, FF2) taken as ONE MX-fp6
 *           matrix instruction per 32 k instead of two fp16 ones (1.5 MFMA-equivalents per product instead of 3; same accuracy class:
 *           DESIGN.md section 2).  Backbones / shapes / options the MX path is not built for run as FP16X3 (never less accurate).
 *           (Value 4 named a 96-byte-row operand form of FP16M in ABI v8's microbenchmarks — an experiment that lost 13-20 % per GEMM,
 *           now tools/experiments/removed_r05.patch; the value is rejected.) */
enum { F5HIP_PREC_FP32 = 0, F5HIP_PREC_FP16X3 = 1, F5HIP_PREC_FP16 = 2, F5HIP_PREC_FP16M = 3 };

/* Architecture of the backbone: the keyword arguments of reference src/f5_tts/model/backbones/dit.py:171-192 (DiT) /
 * unett.py:109-128 (UNetT) that change inference arithmetic. */
typedef struct f5hip_dit_config {
  int32_t dim, depth, heads, dim_head, ff_inner;
  int32_t mel_dim, text_num_embeds, text_dim, conv_layers;
  int32_t text_mask_padding;   /* bool */
  int32_t pe_attn_head;        /* -1 = rope on all heads (None), k>0 = first k heads */
  int32_t attn_mask_enabled;   /* bool: key-padding mask inside attention */
  int32_t conv_pos_kernel, conv_pos_groups;
  int32_t backbone;            /* 0 = DiT (F5-TTS, dit.py), 1 = UNetT (E2-TTS, reference src/f5_tts/model/backbones/unett.py:108-307:
                                  time embedding as a prepended token, x_transformers RMSNorm pre-norm, concat skip connections),
                                  2 = MMDiT (reference src/f5_tts/model/backbones/mmdit.py:91-262: the text is a second token stream of its
                                  own length nt with its own weights; every block attends jointly over [n audio frames | nt text tokens];
                                  requires text_dim == dim, conv_layers == 0, pe_attn_head == -1) */
  /* optional constructor switches no shipped yaml enables (dit.py:181-189, unett.py:120-127); all 0 for the released models.
   * qk_norm applies to both backbones (UNetT keys: layers.{i}.2.q_norm.weight). */
  int32_t qk_norm;               /* 0 = None, 1 = "rms_norm": RMSNorm(dim_head, eps 1e-6) on q and k before rope (modules.py:402-409,493-496);
                                    adds tensors ...attn.q_norm.weight / k_norm.weight [dim_head] per block */
  int32_t long_skip_connection;  /* bool: x = Linear(2*dim -> dim, no bias)(cat(x_after_blocks, x_before_blocks)) (dit.py:228,354-365);
                                    adds transformer.long_skip_connection.weight [dim, 2*dim] */
  int32_t text_average_upsampling; /* bool: text_embedding_average_upsampling (dit.py:55-84,131-137); requires text_mask_padding */
  int32_t skip_connect_type;     /* UNetT only (unett.py:127,289-295): 0 = "concat" (skip_proj Linear(2*dim -> dim)), 1 = "add", 2 = "none" */
} f5hip_dit_config;

/* Vocos (charactr/vocos-mel-24khz config.yaml; loader reference utils_infer.py:106-129). */
typedef struct f5hip_vocos_config {
  int32_t input_channels, dim, intermediate_dim, num_layers, n_fft, hop_length;
} f5hip_vocos_config;

typedef struct f5hip_ctx f5hip_ctx;

int f5hip_abi_version(void);

/* ---- lifetime -------------------------------------------------------------------------------- */
/* replaces: CFM(transformer=DiT(**cfg)).to(device) in load_model (utils_infer.py:258-271).
 * vocos_cfg may be NULL (no vocoder).  device = HIP device ordinal. */
int f5hip_create(const f5hip_dit_config* dit_cfg, const f5hip_vocos_config* vocos_cfg, int device, f5hip_ctx** out);
int f5hip_destroy(f5hip_ctx* ctx);
/* last error text of this context (or of the failed f5hip_create when ctx == NULL); never NULL */
const char* f5hip_last_error(const f5hip_ctx* ctx);

/* ---- weights --------------------------------------------------------------------------------- */
/* replaces: model.load_state_dict(...) in load_checkpoint (utils_infer.py:190-232) and
 * vocoder.load_state_dict (utils_infer.py:127).  `name` is the reference state-dict key after the
 * EMA prefix strip ("transformer.transformer_blocks.0.attn.to_q.weight", "backbone.embed.weight", ...).
 * `data` is fp32, contiguous, HOST memory, `numel` elements; it is copied synchronously into the
 * context's packed device blob.  Unknown names return F5HIP_ERR_INVALID (the adapter drops the keys
 * the reference drops: "initted", "step", legacy mel_stft buffers, rotary inv_freq is accepted and ignored). */
int f5hip_load_tensor(f5hip_ctx* ctx, const char* name, const float* data, int64_t numel);
/* number of named tensors the context expects / i-th name and element count (for loaders and tests) */
int f5hip_num_tensors(const f5hip_ctx* ctx);
int f5hip_tensor_info(const f5hip_ctx* ctx, int index, const char** name, int64_t* numel, int64_t* blob_offset);
/* The packed fp32 weight blob on the device (DiT then Vocos), for the one-time RCCL broadcast from
 * rank 0 — the reference instead lets every rank read the checkpoint
 * (src/f5_tts/eval/eval_infer_batch.py:140-172). */
int f5hip_weight_blob(f5hip_ctx* ctx, void** device_ptr, int64_t* bytes);
/* Mark every tensor as loaded (used after the blob was filled by a broadcast instead of load_tensor). */
int f5hip_mark_all_loaded(f5hip_ctx* ctx);
/* Which tensors have been loaded, in f5hip_tensor_info order (n = f5hip_num_tensors): the sender of the blob reads its mask, the
 * receivers set it, so a checkpoint's OPTIONAL buffers (rotary inv_freq, text freqs_cis, the iSTFT window — recomputed when absent)
 * are honoured on every rank exactly as on the rank that read the file.  Setting a mask un-finalises the context. */
int f5hip_loaded_mask(f5hip_ctx* ctx, uint8_t* mask, int n);
int f5hip_set_loaded_mask(f5hip_ctx* ctx, const uint8_t* mask, int n);
/* Build derived device layouts (fp16 hi/lo operand copies, fused QKV, per-tap conv weights, FFT/mel
 * tables).  Must be called after all tensors are loaded and before any compute call.  Synchronous. */
int f5hip_finalize_weights(f5hip_ctx* ctx);

/* ---- mel front-end ---------------------------------------------------------------------------- */
/* replaces: MelSpec.forward (reference src/f5_tts/model/modules.py:138-151) for both mel_spec_type values:
 *   mel_type 0 "vocos"   get_vocos_mel_spectrogram (modules.py:80-109): reflect-pad n_fft/2, STFT(1024, hop 256, periodic hann), |.|,
 *                        HTK mel filterbank [513->100] (torchaudio, norm=None), log(clamp 1e-5); frames = 1 + n_samples/256
 *   mel_type 1 "bigvgan" get_bigvgan_mel_spectrogram (modules.py:35-77): reflect-pad (n_fft-hop)/2 = 384, STFT without centring,
 *                        sqrt(re^2+im^2+1e-9), slaney-scale slaney-normalised filterbank (librosa.filters.mel), log(clamp 1e-5);
 *                        frames = n_samples/256 (n_samples >= 640)
 * wav: device fp32 [batch, n_samples]; out: device fp32;
 * frame_major != 0 -> out[batch, frames, 100] (what CFM.sample consumes after its permute, cfm.py:107-108),
 * else out[batch, 100, frames] (what MelSpec.forward returns). */
int f5hip_mel(f5hip_ctx* ctx, const float* wav, int batch, int64_t n_samples, float* out, int frame_major, int mel_type, void* stream);

/* ---- sampler ----------------------------------------------------------------------------------- */
/* replaces: CFM.sample from "duration" onwards (reference src/f5_tts/model/cfm.py:128-223) including the
 * odeint euler loop (cfm.py:218), DiT.forward(cfg_infer=True, cache=True) (dit.py:319-370), the CFG combine
 * (cfm.py:190-191) and the prompt restore (cfm.py:221-223).  The host adapter keeps the reference's cheap
 * host logic (tokenise, duration clamp, seeded torch-CPU noise, time grid) and passes its results in.
 *
 *   batch, n        number of utterances, padded frame count (= duration.amax())
 *   cond            device fp32 [batch, n, mel]: prompt mel padded with zeros to n frames (cfm.py:145)
 *   cond_mask       host uint8 [batch, n]: lens_to_mask(lens) & edit_mask, padded False (cfm.py:128-130,149)
 *   text            host int64 [batch, nt]: token ids, -1 = batch padding (utils.py:99-106)
 *   duration        host int64 [batch]: per-utterance frame count; rows >= duration[b] are padding.
 *                   use_mask != 0 <=> the reference passes mask = lens_to_mask(duration) (batch > 1, cfm.py:155-158)
 *   y0              device fp32 [batch, n, mel]: initial noise (cfm.py:196-201), zero on padding rows
 *   t               host fp32 [steps + 1]: the time grid (cfm.py:211-216)
 *   ode_method      0 = euler (reference default, utils_infer.py:60), 1 = midpoint — torchdiffeq fixed-grid solvers on exactly the
 *                   supplied grid (odeint_kwargs["method"], cfm.py:218); NFE = steps (euler) / 2*steps (midpoint)
 *   cfg_strength    classifier-free guidance strength; < 1e-5 evaluates the conditional branch only (cfm.py:166-177)
 *   precision       F5HIP_PREC_*
 *   out             device fp32 [batch, n, mel]: where(cond_mask, cond, y_final)
 *   trajectory      device fp32 [steps + 1, batch, n, mel] or NULL: every ODE state (cfm.py:218 return value)
 */
int f5hip_sample(f5hip_ctx* ctx, int batch, int n, const float* cond, const uint8_t* cond_mask, const int64_t* text,
                 int nt, const int64_t* duration, int use_mask, const float* y0, const float* t, int steps, int ode_method,
                 float cfg_strength, int precision, float* out, float* trajectory, void* stream);

/* Debug/parity taps (tests only): copies of internal tensors of the LAST f5hip_sample call.
 *   which = 0: text_embed cond [batch, n, text_dim]   1: text_embed uncond
 *           2: velocity of the last step [batch, n, mel] (after CFG)
 *           3: packed hidden state after input embedding of the last step [2*batch, n, dim]
 * dst: device fp32 buffer of `numel` elements. */
int f5hip_debug_tensor(f5hip_ctx* ctx, int which, float* dst, int64_t numel, void* stream);

/* ---- vocoder ------------------------------------------------------------------------------------ */
/* replaces: vocoder.decode(mel) (reference utils_infer.py:510-511; `vocos` package: backbone + ISTFTHead,
 * head restated in-repo at runtime/triton_trtllm/scripts/export_vocoder_to_onnx.py:45-59).
 * mel: device fp32; channel_major != 0 -> [batch, 100, frames] (the reference's call layout), else [batch, frames, 100].
 * out: device fp32 [batch, 256 * (frames - 1)] (torch.istft(center=True) length). */
int f5hip_vocos_decode(f5hip_ctx* ctx, const float* mel, int batch, int frames, int channel_major, float* out, void* stream);

/* The head's inverse STFT alone — replaces: ISTFTHead's `self.istft(S)` with S = exp(log-magnitude, clipped at 1e2) * (cos + i sin)(phase),
 * torch.istft(n_fft 1024, hop 256, hann, center=True) semantics (restated in-repo at export_vocoder_to_onnx.py:45-59; the reference's own
 * runnable conv-iSTFT, runtime/triton_trtllm/scripts/conv_stft.py:193-234, is what tests/golden/istft_conv_reference.npz holds).
 * logits: device fp32 [batch * frames, ld], columns [0, 513) log-magnitude, [513, 1026) phase (ld >= 1026, ld % 4 == 0).
 * out: device fp32 [batch, 256 * (frames - 1)].  A component entry point for parity tests of the kernels f5hip_vocos_decode ends with. */
int f5hip_istft(f5hip_ctx* ctx, const float* logits, int64_t ld, int batch, int frames, float* out, void* stream);
/* The Vocos ISTFTHead alone: hidden [B*T, dim] (the backbone's output after final_layer_norm, fp32, device memory) -> head.out Linear ->
 * exp -> clip(1e2) -> (cos, sin) -> inverse STFT, waveform [B, 256 (T - 1)].  Replaces vocos `ISTFTHead.forward`; the reference's own
 * runnable copy is runtime/triton_trtllm/scripts/export_vocoder_to_onnx.py:43-59 (+ scripts/conv_stft.py:193-234), which is what
 * tests/golden/vocos_head_ref.npz was minted from. */
int f5hip_vocos_head(f5hip_ctx* ctx, const float* hidden, int B, int T, float* out, void* stream);

/* ---- BigVGAN generator (mel_spec_type "bigvgan") -------------------------------------------------- */
/* replaces: the object load_vocoder(vocoder_name="bigvgan") returns — bigvgan.BigVGAN.from_pretrained(...), .remove_weight_norm(),
 * .eval() (reference utils_infer.py:130-144) — and its call `vocoder(mel)` (utils_infer.py:512-513).  The generator's source is an
 * un-vendored submodule of the reference (.gitmodules:1-3, NVIDIA/BigVGAN, src/third_party/BigVGAN is empty): field and tensor
 * names below are upstream's (bigvgan.py, config.json of nvidia/bigvgan_v2_24khz_100band_256x); parity is pinned only against
 * this repo's CPU restatement of the published algorithm (oracle/bigvgan_oracle.py), see DESIGN.md.
 * A separate context type: the reference's vocoder is a separate object too. */
typedef struct f5hip_bigvgan_config {
  int32_t num_mels, num_upsamples, upsample_initial_channel;
  int32_t upsample_rates[8], upsample_kernel_sizes[8];
  int32_t resblock;                 /* 1 = AMPBlock1 (convs1 + convs2), 2 = AMPBlock2 (convs) */
  int32_t num_kernels;              /* parallel resblocks per stage (<= 4) */
  int32_t resblock_kernel_sizes[4];
  int32_t resblock_num_dilations[4];
  int32_t resblock_dilation_sizes[4][4];
  int32_t activation;               /* 0 = "snake", 1 = "snakebeta" */
  int32_t snake_logscale, use_tanh_at_final, use_bias_at_final;   /* bools */
} f5hip_bigvgan_config;

typedef struct f5hip_bigvgan f5hip_bigvgan;

int f5hip_bigvgan_create(const f5hip_bigvgan_config* cfg, int device, f5hip_bigvgan** out);
int f5hip_bigvgan_destroy(f5hip_bigvgan* v);
/* last error text of this context (or of the failed f5hip_bigvgan_create when v == NULL); never NULL */
const char* f5hip_bigvgan_last_error(const f5hip_bigvgan* v);
/* replaces: generator.load_state_dict(...) inside from_pretrained + remove_weight_norm().  `name` is the generator state-dict key
 * AFTER weight-norm removal ("conv_pre.weight", "ups.0.0.weight", "resblocks.0.convs1.0.bias", "resblocks.0.activations.0.act.alpha",
 * "activation_post.act.beta", "conv_post.weight", ...); the host binding folds weight_g / weight_v.  `data`: fp32 HOST memory. */
int f5hip_bigvgan_num_tensors(const f5hip_bigvgan* v);
int f5hip_bigvgan_tensor_info(const f5hip_bigvgan* v, int index, const char** name, int64_t* numel);
int f5hip_bigvgan_load_tensor(f5hip_bigvgan* v, const char* name, const float* data, int64_t numel);
/* after every tensor is loaded: builds the GEMM weight matrices (a Conv1d / ConvTranspose1d is one GEMM over a tap-gathered operand)
 * in all three operand layouts and uploads them.  Synchronous. */
int f5hip_bigvgan_finalize(f5hip_bigvgan* v);
/* replaces: vocoder(mel) (utils_infer.py:512-513, BigVGAN.forward).
 * mel: device fp32; channel_major != 0 -> [batch, num_mels, frames] (the reference's call layout), else [batch, frames, num_mels].
 * precision: F5HIP_PREC_* operand mode of the conv GEMMs (activations, resampling filters and conv_post are always fp32).
 * out: device fp32 [batch, frames * prod(upsample_rates)] (the reference's [batch, 1, T*hop] without the singleton axis). */
int f5hip_bigvgan_forward(f5hip_bigvgan* v, const float* mel, int batch, int frames, int channel_major, int precision, float* out,
                          void* stream);
/* key/value options: "profile" (0/1, see f5hip_bigvgan_kernel_stat), "conv_impl" (0 = every conv as a tap-gathered operand + the plain MFMA GEMM; 1 = one operand copy + the
 * implicit-GEMM kernel with tap-shifted rows, csrc/conv_gemm.h; 2 = 1 with the operand copy written by the Activation1d kernel itself); "stop_after_stage" (parity tap for the tests; -1 = off): k >= 0 makes f5hip_bigvgan_forward write the
 * channels-last fp32 tensor [batch, L_k, C_k] after conv_pre (k = 0) / after upsampling stage k (k >= 1: L_k = frames * prod(rates[:k]),
 * C_k = upsample_initial_channel >> k) into `out` instead of the waveform. */
int f5hip_bigvgan_set_option(f5hip_bigvgan* v, const char* key, int64_t value);
/* Per-kernel-class statistics accumulated while option "profile" is 1 (HIP events on the launch stream around every launch; the call then
 * synchronises the stream): classes "conv_gemm" (algorithmic FLOPs 2 L N taps Cin), "activation1d", "operand" (tap-gathered / operand
 * emission), "other" (mean over resblocks, conv_post) with their algorithmic bytes.  Same shape as f5hip_kernel_stat. */
int f5hip_bigvgan_num_kernel_stats(const f5hip_bigvgan* v);
int f5hip_bigvgan_kernel_stat(const f5hip_bigvgan* v, int index, const char** name, int64_t* calls, double* total_ms, double* flops, double* bytes);
int f5hip_bigvgan_reset_kernel_stats(f5hip_bigvgan* v);

/* ---- engine options / measurement ---------------------------------------------------------------- */
/* key/value options: "use_graph" (0/1: replay the NFE loop as a hipGraph), "profile" (0/1: time each kernel
 * class with hipEvents on the launch stream; forces eager launches),
 * "attn_impl" (what F.scaled_dot_product_attention, model/modules.py:511-520, is computed from — 0 (default): FP32 -> materialised fp32
 * scores; FP16 -> flash attention on plain fp16 q, k, P, V; FP16X3 / FP16M -> flash attention whose SCORES are fp16 hi . hi + both correction
 * products as one MX-fp6 matrix instruction per 32 head channels, P and V plain fp16 (DESIGN.md sections 2, 4.3: plain fp16 scores moved a
 * golden with trained-like weight statistics by 1.1e-3); 1: materialised fp32 scores in every mode; 2: flash, every operand hi/lo split;
 * 3: flash on plain fp16 operands in every mode (the default of ABI v5-v8 builds); 4: flash, q and k hi/lo split (3 MFMAs per score
 * product), P and V plain; 5 = 0 for the half-precision modes; 6: the default's MX-corrected scores with V read as fp16 hi + lo halves
 * (O = V_hi P + V_lo P), 7: P split as well — the margin against attention sharper than the trained-like goldens': DESIGN.md section 2,
 * "sharpness sweep"),
 * "branch_streams" (-1 auto / 0 / 1: run the cond and uncond branches of the CFG batch as two concurrent kernel chains),
 * "packed_rows" (0 off (default) / 1: a ragged batch — use_mask with durations below n — of a DiT with attn_mask_enabled runs its block
 * loop over the VALID rows only: the reference's varlen attention path, model/modules.py:522-543, extended to the row-wise layers.  Rows past a
 * sequence's duration then keep the prompt / zero state instead of the values the padded layout would compute for them; the reference's
 * callers slice every utterance to its own length, utils_infer.py:507, eval_infer_batch.py:199.  Ignored where it cannot apply: key mask
 * off, qk_norm, long skip, UNetT / MMDiT, the materialised fp32 attention, "attn_kv_split" > 1, n >= 65536 or 2 * batch >= 65536 (the
 * row table packs (sequence, token) into 16 + 16 bits)),
 * "attn_kv_split" (1 off (default) / 2..8: flash attention with every query block cut into that many key ranges + a merge kernel —
 * shorter workgroups for small batches, csrc/attention_kernel.h),
 * "mx_weights" (1 (default) / 0; read by the next f5hip_finalize_weights: whether the MX-line copies of the block weights that
 * F5HIP_PREC_FP16M multiplies are built — 2 more halves per block-weight element on top of the fp32 blob and the plain / hi|lo half copies,
 * +1.3 GB for F5-TTS Base; with 0 an FP16M call runs as FP16X3, for users of the other precisions who want the memory back),
 * "attn_stats" (0 (default) / 1: the materialised-score attention — precision FP32, or "attn_impl" 1 — accumulates how sharp its softmax
 * rows are; read with f5hip_attention_stats). */
int f5hip_set_option(f5hip_ctx* ctx, const char* key, int64_t value);
/* How sharp is this checkpoint's attention?  No counterpart in the reference (F.scaled_dot_product_attention, model/modules.py:511-520,
 * is exact in fp32 whatever the logits); here the half-precision attention forms hold the 1e-3 tolerance up to a sharpness that DESIGN.md
 * section 2 ("sharpness sweep") measures, and this call measures a checkpoint against it: with option "attn_stats" = 1, every softmax row
 * of the materialised-score path contributes its LARGEST probability.  out4 = { largest over all rows, sum over rows, rows, rows whose
 * largest probability exceeds 1/2 } since the option was set or the figures were last reset (reset != 0 zeroes them after reading).
 * Synchronises the device.  INTEGRATION.md, "Which attention form does my checkpoint need?", has the thresholds. */
int f5hip_attention_stats(f5hip_ctx* ctx, double* out4, int reset);
/* Per-kernel-class statistics accumulated while "profile" is on: calls, total milliseconds, algorithmic
 * FLOPs and algorithmic bytes.  index in [0, f5hip_num_kernel_stats). */
int f5hip_num_kernel_stats(const f5hip_ctx* ctx);
int f5hip_kernel_stat(const f5hip_ctx* ctx, int index, const char** name, int64_t* calls, double* total_ms,
                      double* flops, double* bytes);
int f5hip_reset_kernel_stats(f5hip_ctx* ctx);

/* (The kernel microbenchmarks and fault reproducers — f5hip_bench_* — are not part of the product: include/f5hip_bench.h,
 * libf5hip_bench.so.) */

#ifdef __cplusplus
}
#endif
#endif /* F5HIP_H */
