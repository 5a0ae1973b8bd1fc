// kernels.h — host-callable launchers of every HIP kernel in libf5hip (internal; the public ABI is include/f5hip.h)
#pragma once
#include <stdint.h>

#include "gemm.h"  // common.h pulls in the HIP runtime (or the host shim under F5_HIPEMU)

// GEMM operand kind.  OP_F16M: fp16 + MX-fp6 correction lines (common.h "fp16m").  It exists for the pipelined block-GEMM kernels ONLY: a
// launch that those kernels do not take FAILS (hipErrorInvalidValue) instead of falling back to the generic kernel, which has no MX k-loop.
// Callers therefore pre-validate: the engine chooses the mode per call (api.cpp `mx_call`: model shape, at least 8 tokens per sequence,
// gemm_mx_tiles_usable()) and runs the call in OP_F16X3 when any precondition fails.
enum { OP_F32 = 0, OP_F16 = 1, OP_F16X3 = 2, OP_F16M = 3 };

// ---- gemm.hip ---------------------------------------------------------------------------------
// batch = gridDim.z.  Tile is chosen from (M, N): 128x128, or 64x128 when the grid would underfill 256 CUs.
hipError_t launch_gemm_store(int op, const GemmCore& g, const EpiStore& e, int batch, hipStream_t s);
hipError_t launch_gemm_qkv(int op, const GemmCore& g, const EpiQKV& e, hipStream_t s);
bool gemm_qkv_takes_pp(int op, const GemmCore& g, const EpiQKV& e);  // would this launch run a pipelined kernel (the only writers of EpiQKV::mx_qk)?
hipError_t launch_gemm_qkv_variant(int op, const GemmCore& g, const EpiQKV& e, int variant, hipStream_t s);  // microbenchmarks / tests: < 0 = heuristic
// explicit tile variant (microbenchmarks): 0 = 64x128, 1 = 128x64, 2 = 128x128 (rows x channels), -1 = heuristic
hipError_t launch_gemm_store_variant(int op, const GemmCore& g, const EpiStore& e, int batch, int variant, hipStream_t s);
// do the tuning knobs of this process leave every OP_F16M launch on a tile instantiated for MX lines? (gemm.hip)
bool gemm_mx_tiles_usable();
// one-time: raise the dynamic-LDS limit of every instantiation (must not happen inside a stream capture)
hipError_t init_gemm_kernels();
// race_probe.hip: the reproducer of round 2's co-residency fault in the fused q|k|v epilogue (microbenchmarks / tests only)
hipError_t launch_pp_qkv_probe(const GemmCore& g, const EpiQKV& e, int variant, int expt, int abl, int lds_pad, uint32_t* dbg, hipStream_t s);
hipError_t launch_noise(const void* src, uint32_t bytes, int wgs, int rounds, int kind, int lds_bytes, uint32_t* sink, hipStream_t s);
hipError_t init_convpos_kernels();
hipError_t init_bigvgan_kernels();

// ---- elementwise.hip --------------------------------------------------------------------------
// LayerNorm over the last dim D (D % 4 == 0, D <= 2048), eps inside sqrt.  Either affine (weight/bias) or
// AdaLN modulation: y = ln(x) * (1 + scale) + shift.  Writes fp32 and/or f16 hi(/lo) planes.
hipError_t launch_layernorm(const float* x, int64_t ldx, int M, int D, float eps, const float* weight, const float* bias,
                            const float* scale, const float* shift, float* out32, f16* out16, f16* out16_lo, int64_t ldo,
                            hipStream_t s, int pk16 = 0, int64_t ldo16 = 0, int mode = 0);
// mode: 0 = LayerNorm, 1 = x_transformers RMSNorm (F.normalize(x) * sqrt(D) * weight; bias/scale/shift unused),
//       2 = no normalisation (operand conversion / copy only)
// rows [seq, 0, :] of x [S, nseq, D] <- t[D]   (UNetT: the time embedding is a token, reference backbones/unett.py:272)
hipError_t launch_set_token_rows(float* x, const float* t, int S, int nseq, int D, hipStream_t s);
// text token embedding + absolute sinusoid position + masks (reference model/backbones/dit.py:86-127)
//   tok [B, n] int32 (0 = filler), valid [B, n] u8 (pos < seq_len[b]); out [2B, n, T]: rows [0,B) cond, [B,2B) uncond (ids zeroed)
hipError_t launch_text_embed(const int32_t* tok, const uint8_t* valid, const float* table, const float* freqs_cis,
                             int B, int n, int T, int mask_padding, int has_pos, float* out, hipStream_t s);
// depthwise conv k=7 (zero pad per sequence) + bias + LayerNorm(affine): x [S, n, C] -> out [S, n, C]
//   w7 [7, C] (tap-major), C % 256 == 0 or C <= 2048 with C % 4 == 0
hipError_t launch_dwconv7_ln(const float* x, int S, int n, int C, const float* w7, const float* cbias, const float* ln_w,
                             const float* ln_b, float eps, float* out, hipStream_t s);
// GRN (reference model/modules.py:242-245): Gx[s, c] = sqrt(sum_n h[s,n,c]^2), nx = Gx / (mean_c(Gx) + 1e-6) (launch_grn_stats); then
//   out = gamma * (h * nx) + beta + h (launch_grn_apply)
int grn_sumsq_slices(int n);  // sequence slices of the two-pass reduction; `part` holds S * slices * C floats
hipError_t launch_grn_stats(const float* h, int S, int n, int C, float* nx, float* part, hipStream_t s);
hipError_t launch_grn_apply(const float* h, const float* nx, const float* gamma, const float* beta, int S, int n, int C,
                            float* out, hipStream_t s);
// zero rows where mask[row] != 0 (masked_fill), x [rows, C]
hipError_t launch_zero_rows(float* x, const uint8_t* mask, int64_t rows, int C, hipStream_t s);
// step_cond = where(cond_mask, cond, 0)
hipError_t launch_mask_select(const float* a, const uint8_t* mask, int64_t rows, int C, float* out, hipStream_t s);
// out = where(mask, a, b)
hipError_t launch_where_rows(const uint8_t* mask, const float* a, const float* b, int64_t rows, int C, float* out, hipStream_t s);
// CFG combine + one explicit ODE stage (reference cfm.py:190-191 + torchdiffeq fixed-grid euler / midpoint stages):
//   g = has_uncond ? vc + (vc - vu) * cfg : vc;   dst = base + coef * g;   optional copies of dst (trajectory) and g (debug tap)
hipError_t launch_cfg_euler(const float* base, float* dst, const float* v, int64_t half_elems, int has_uncond, const float* coef_ptr,
                            const float* cfg_ptr, float* traj_next, float* vel_dbg, hipStream_t s);
// optional variants: q/k RMSNorm(dh) + rope + q scale over raw fp32 q/k rows [BH, n, dh] (in place when q16 == null, else to the fp16
// (hi/lo) planes); gather of text rows per sequence (idx [B, n], -1 = zero row; cond and uncond halves share it); x += y
hipError_t launch_qk_norm_rope(float* q32, float* k32, const float* wq, const float* wk, const float* rope_cs, int64_t rows, int nseq,
                               int heads, int dh, int pe_heads, float qscale, float eps, f16* q16, f16* q16_lo, f16* k16, f16* k16_lo,
                               hipStream_t s, int n1 = 0, const float* wq2 = nullptr, const float* wk2 = nullptr);
// n1 > 0 (MMDiT joint slabs): tokens >= n1 of every slab belong to the text stream — gains wq2 / wk2, rope position = token - n1
hipError_t launch_gather_seq_rows(const float* src, const int32_t* idx, int S, int B, int n, int C, float* out, hipStream_t s);
hipError_t launch_add_inplace(float* x, const float* y, int64_t n, hipStream_t s);
// sinusoidal time embedding (reference model/modules.py:157-169): t [S] -> out [S, 256]
hipError_t launch_time_sinus(const float* t, int S, int dim, float* out, hipStream_t s);
// rope table: out [n, dh/2, 2] = (cos, sin)(pos * inv_freq[i])
hipError_t launch_rope_table(const float* inv_freq, int n, int half, float* out, hipStream_t s);
// fp32 -> f16 hi/lo planes (weights, one-time), optional power-of-two prescale
hipError_t launch_split_f16(const float* src, int64_t n, float prescale, f16* hi, f16* lo, hipStream_t s);
// [rows, K] fp32 -> packed fp16x3 operand [rows, 2K]: k-blocks of 32 as [32 hi | 32 lo] (gemm.h), K % 32 == 0
hipError_t launch_split_f16_packed(const float* src, int64_t rows, int K, f16* dst, hipStream_t s);
// [rows, K] fp32 (row stride ld floats) x rowscale[r] (or null) -> MX operand rows of OP_F16M (common.h): weight != 0 packs the W side
hipError_t launch_pack_mx_rows(const float* src, int64_t ld, int64_t rows, int K, const float* rowscale, f16* dst, int weight, hipStream_t s);
// W [rows, K] fp32 -> per-row power-of-two scale (largest entry to [2^12, 2^13)) and its inverse, the plain fp16 copy hi [rows, K] and the
// packed hi | lo copy pk [rows, 2K] of the scaled rows (GemmCore::w_alpha takes `alpha`)
// dst[r, :] = src[rowmap[r], :] (gather) / dst[rowmap[r], :] = src[r, :] (scatter) over `rows` rows of C floats (C % 4 == 0)
hipError_t launch_gather_rows(const float* src, const int32_t* rowmap, int64_t rows, int C, float* dst, hipStream_t s);
hipError_t launch_scatter_rows(const float* src, const int32_t* rowmap, int64_t rows, int C, float* dst, hipStream_t s);
hipError_t launch_condition_weight(const float* src, int rows, int K, float* scale, float* alpha, f16* hi, f16* pk, hipStream_t s);
// conv_pos weights [D, cpg, K] -> per-tap operand layout [G][K][cpg(co)][cpg(ci)] (fp32 + f16 hi/lo)
hipError_t launch_convpos_pack(const float* w, int D, int cpg, int K, float* w32, f16* whi, f16* wlo, hipStream_t s);
// [C, 1, 7] depthwise weights -> [7, C]
hipError_t launch_dw_pack(const float* w, int C, float* w7, hipStream_t s);
// row softmax for the exact (materialised-score) attention: S [rows, ld], cols >= kvlen(row) get 0
hipError_t launch_softmax_rows(float* S, int64_t rows, int ld, int nseq, int heads, const int32_t* kvlen_per_batch, int kv_default,
                               hipStream_t s, const int32_t* kvlen2 = nullptr, int seg2_off = 0,  // second key run: see launch_flash_attn
                               double* stats = nullptr);  // {max, sum, rows, rows > 1/2} of the rows' largest probabilities, accumulated (or null)
// im2col for the Vocos embed conv (k=7, pad 3): mel [B, T, Cin] (frame-major) -> col [B*T, 7*Cin] with k index = ci*7 + tap
hipError_t launch_im2col7(const float* mel, int B, int T, int Cin, int channel_major, float* col, int64_t ldc, hipStream_t s);

// ---- convpos.hip ------------------------------------------------------------------------------
// grouped Conv1d(k, groups) as implicit GEMM on MFMA + bias + row mask + Mish (+ residual)
//   x [S, n, D] fp32; w per-tap layout; out [S, n, D] fp32; rowvalid [S*n] u8 or null (reference
//   model/modules.py:187-201: input and conv output zero-filled outside the mask)
hipError_t launch_convpos(int op, const float* x, const float* w32, const f16* whi, const f16* wlo, const float* bias,
                          const uint8_t* rowvalid, const float* residual, int S, int n, int D, int groups, int K, float* out,
                          hipStream_t s, int out_n = 0, int out_off = 0,  // output row of (seq, m) = seq * out_n + m + out_off (out_n 0 = n)
                          const f16* wmx = nullptr);  // MX lines of the per-tap tiles (64 channels per group, op OP_F16X3): the 1.5-MFMA form

// ---- attention.hip ----------------------------------------------------------------------------
// flash-style non-causal attention, fp32 softmax + accumulate.  nsplit 1: plain fp16 operands; 3: every operand hi/lo split; 2: q, k split;
// 4: scores = fp16 hi . hi + MX-fp6 corrections (q_lo / k_lo hold the P words PpEpiQKV::mx_qk writes), P and V plain fp16.
//   q,k [BH, n, 64] f16 (q pre-scaled); vt [BH, 64, ldv] f16 (V transposed, ldv % 8 == 0); o16(/lo) [B', n, H*64];
//   kvlen per batch' or null.  *_lo planes are required for nsplit == 3.  o_packed: o16 is a packed fp16x3 operand
//   (row stride 2*H*64, o16_lo == o16 + 32).
bool flash_attn_available();
hipError_t init_attention_kernels();
hipError_t launch_flash_attn(int nsplit, const f16* q, const f16* q_lo, const f16* k, const f16* k_lo, const f16* vt, const f16* vt_lo,
                             int ldv, int Bp, int heads, int n, const int32_t* kvlen, f16* o16, f16* o16_lo, hipStream_t s,
                             int o_packed = 0, const int32_t* kvlen2 = nullptr, int seg2_off = 0, int co_launches = 1, int kv_split = 1,
                             float* part_o = nullptr, float* part_ml = nullptr, int log2q = 0, const int32_t* cu_rows = nullptr);
// cu_rows (with kvlen): packed output rows — row cu_rows[b'] + q for query q < kvlen[b'] of sequence b' (FlashArgs::cu_rows)
// log2q = 1: q carries log2(e) on top of 1/sqrt(dh) (scores are base-2 logarithms; enables the lazy reference maximum, attention_kernel.h)
// kv_split > 1: every query block is cut into kv_split workgroups over contiguous key ranges (small batches: more, shorter workgroups);
// part_o [Bp*heads*n, kv_split, 64] / part_ml [Bp*heads*n, kv_split, 2] fp32 scratch for the unnormalised partial results
// co_launches: how many identical launches run concurrently on other streams (the cond / uncond chains): enters the block-size choice
// kvlen2 / seg2_off: a second run of valid keys [seg2_off, seg2_off + kvlen2[b']) behind [0, kvlen[b']) (MMDiT joint attention mask)

// ---- audio.hip --------------------------------------------------------------------------------
struct AudioTables {
  const float* twiddle;   // [512, 2] cos/sin(2*pi*k/1024)
  const float* window;    // [1024] periodic hann
  const float* melfb;     // [513, 100] HTK triangles
  const float* env_inv;   // unused (envelope computed per call)
};
// melrange [nmel][2]: first bin and one past the last bin with a non-zero filterbank weight, per mel channel; melw [nmel][melw_ld]: the
// weights of those bins, first bin first, zero-padded (melw_ld % 4 == 0)
hipError_t launch_mel(const float* wav, int B, int64_t nsamp, int frames, const float* twiddle, const float* window,
                      const float* melw, const int* melrange, int melw_ld, int nmel, int frame_major, int pad, float mag_eps, float* out, hipStream_t s);
// head logits [B*T, ld] (log-mag | phase) -> inverse real transform, window, overlap-add, envelope normalisation, centre trim ->
// wav [B, 256*(T-1)]: one kernel, no intermediate in HBM
hipError_t launch_istft(const float* logits, int64_t ld, int B, int T, const float* twiddle, const float* window, float* wav, hipStream_t s);

// ---- bigvgan.hip ------------------------------------------------------------------------------
// BigVGAN generator path, channels-last fp32 activations [B, L, C] (see bigvgan.hip for the formulas each kernel evaluates).
// Activation1d (x2 upsample -> Snake/SnakeBeta -> x2 downsample) with the host-computed 12-tap kaiser-sinc filter `filt12` (HOST pointer)
// oper != null: instead of y, write the GEMM operand copy [B, L, cpad] in layout `op` (channels [C, cpad) zero) for launch_conv_gemm
hipError_t launch_aa_snake(const float* x, float* y, const float* alpha, const float* beta, const float* filt12, int B, int L, int C,
                           int logscale, hipStream_t s, void* oper = nullptr, int op = 0, int cpad = 0);
// tap-gathered GEMM operand: out[(b, l), j * cpad + c] = src[b, l + shift0 + j * dstep, c] or 0, in operand layout `op` (OP_*);
// sb / sl / sc = element strides of src (batch, time step, channel); ldo / ob = row / batch stride of out in elements of its type
hipError_t launch_im2col_taps(const float* src, int64_t sb, int64_t sl, int64_t sc, int B, int L, int C, int ntaps, int shift0, int dstep,
                              int cpad, int op, void* out, int64_t ldo, int64_t ob, hipStream_t s);
// out = (r[0] + ... + r[nk-1]) / nk over n floats (n % 4 == 0, nk <= 4)
hipError_t launch_mean_streams(const float* const* r, int nk, int64_t n, float* out, hipStream_t s);
// Conv1d(C -> 1, k 7, pad 3) + tanh or clamp(-1, 1): y [B, L, C], w7 [7, C] tap-major, bias [1] or null -> out [B, L]
hipError_t launch_conv_post(const float* y, const float* w7, const float* bias, int B, int L, int C, int use_tanh, float* out, hipStream_t s);
// Conv1d / ConvTranspose1d as an implicit GEMM over ONE operand copy Y [batch, L, cpad] (conv_gemm.h): g.A = Y (row stride g.lda,
// batch stride g.strideA), g.W = [N, ntaps * cpad] as for launch_im2col_taps + launch_gemm_store, g.K = ntaps * cpad, g.M = g.a_rows = L.
// Needs cpad % 32 == 0 (fp32 / fp16x3) or cpad % 64 == 0 (fp16): a tap segment is a whole number of 128-byte k-tiles.
hipError_t launch_conv_gemm(int op, const GemmCore& g, int ntaps, int shift0, int dstep, int cpad, const EpiStore& e, int batch, hipStream_t s);
