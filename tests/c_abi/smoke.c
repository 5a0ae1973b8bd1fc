/* The C ABI used from plain C: no Python, no torch — only include/f5hip.h and the HIP runtime for device buffers.
 * Builds a tiny DiT + Vocos context, fills every named tensor with a deterministic pattern, runs mel -> sample -> vocos_decode and
 * checks the properties that do not need an oracle: finite outputs, prompt frames restored bit for bit (cfm.py:221-223), the
 * trajectory's first state equals y0, determinism across two calls.  Exit code 0 = ok.  (tests/test_abi.py builds it; the GPU
 * suite runs it.) */
#include <hip/hip_runtime_api.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "f5hip.h"

#define CHECK(x) do { int _s = (x); if (_s != 0) { fprintf(stderr, "%s failed: %d (%s)\n", #x, _s, f5hip_last_error(ctx)); return 1; } } while (0)
#define HIP(x) do { hipError_t _e = (x); if (_e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(_e)); return 1; } } while (0)

static uint32_t lcg(uint32_t* s) { *s = *s * 1664525u + 1013904223u; return *s; }
static float unif(uint32_t* s) { return (float)(lcg(s) >> 8) * (1.0f / 8388608.0f) - 1.0f; } /* [-1, 1) */

int main(void) {
  f5hip_ctx* ctx = NULL;
  f5hip_dit_config dc;
  f5hip_vocos_config vc = {100, 128, 384, 2, 1024, 256};
  memset(&dc, 0, sizeof dc);
  dc.dim = 256; dc.depth = 2; dc.heads = 4; dc.dim_head = 64; dc.ff_inner = 512; dc.mel_dim = 100; dc.text_num_embeds = 255;
  dc.text_dim = 128; dc.conv_layers = 2; dc.text_mask_padding = 1; dc.pe_attn_head = -1; dc.conv_pos_kernel = 31; dc.conv_pos_groups = 16;
  if (f5hip_abi_version() != F5HIP_ABI_VERSION) { fprintf(stderr, "ABI version mismatch\n"); return 1; }
  {
    int st = f5hip_create(&dc, &vc, 0, &ctx);
    if (st != 0) { fprintf(stderr, "f5hip_create: %d (%s)\n", st, f5hip_last_error(NULL)); return st == F5HIP_ERR_HIP ? 77 : 1; } /* 77 = no GPU */
  }
  /* weights: every tensor the context names, small values (LayerNorm-like gains near 1 are not needed for a smoke run) */
  const int nt_ = f5hip_num_tensors(ctx);
  uint32_t seed = 12345u;
  for (int i = 0; i < nt_; ++i) {
    const char* name; int64_t numel, off;
    CHECK(f5hip_tensor_info(ctx, i, &name, &numel, &off));
    float* buf = (float*)malloc((size_t)numel * sizeof(float));
    const float scale = strstr(name, "norm.weight") || strstr(name, "gamma") ? 0.0f : 0.05f;
    for (int64_t k = 0; k < numel; ++k) buf[k] = scale * unif(&seed) + (scale == 0.0f ? 1.0f : 0.0f);
    CHECK(f5hip_load_tensor(ctx, name, buf, numel));
    free(buf);
  }
  CHECK(f5hip_finalize_weights(ctx));

  /* prompt: 40 hops of a two-tone wave -> 41 mel frames; generate up to 96 frames */
  enum { NW = 256 * 40, LENS = 41, N = 96, MEL = 100, NTXT = 20, STEPS = 4 };
  float* wav_h = (float*)malloc(NW * sizeof(float));
  for (int i = 0; i < NW; ++i) wav_h[i] = 0.1f * sinf(0.05f * (float)i) + 0.05f * sinf(0.31f * (float)i);
  float *wav_d, *mel_d, *cond_d, *y0_d, *out_d, *out2_d, *traj_d, *wave_d;
  HIP(hipMalloc((void**)&wav_d, NW * sizeof(float)));
  HIP(hipMalloc((void**)&mel_d, (size_t)LENS * MEL * sizeof(float)));
  HIP(hipMalloc((void**)&cond_d, (size_t)N * MEL * sizeof(float)));
  HIP(hipMalloc((void**)&y0_d, (size_t)N * MEL * sizeof(float)));
  HIP(hipMalloc((void**)&out_d, (size_t)N * MEL * sizeof(float)));
  HIP(hipMalloc((void**)&out2_d, (size_t)N * MEL * sizeof(float)));
  HIP(hipMalloc((void**)&traj_d, (size_t)(STEPS + 1) * N * MEL * sizeof(float)));
  HIP(hipMalloc((void**)&wave_d, (size_t)256 * (N - LENS - 1) * sizeof(float)));
  HIP(hipMemcpy(wav_d, wav_h, NW * sizeof(float), hipMemcpyHostToDevice));
  CHECK(f5hip_mel(ctx, wav_d, 1, NW, mel_d, /*frame_major=*/1, /*mel_type=*/0, NULL));
  HIP(hipMemset(cond_d, 0, (size_t)N * MEL * sizeof(float)));
  HIP(hipMemcpy(cond_d, mel_d, (size_t)LENS * MEL * sizeof(float), hipMemcpyDeviceToDevice));

  uint8_t cond_mask[N];
  int64_t text[NTXT], duration[1] = {N};
  float t[STEPS + 1];
  float* y0_h = (float*)malloc((size_t)N * MEL * sizeof(float));
  for (int i = 0; i < N; ++i) cond_mask[i] = i < LENS;
  for (int i = 0; i < NTXT; ++i) text[i] = 1 + (int64_t)(lcg(&seed) % 254u);
  for (int i = 0; i <= STEPS; ++i) t[i] = (float)i / STEPS;
  for (int i = 0; i < N * MEL; ++i) y0_h[i] = unif(&seed);
  HIP(hipMemcpy(y0_d, y0_h, (size_t)N * MEL * sizeof(float), hipMemcpyHostToDevice));

  for (int prec = F5HIP_PREC_FP32; prec <= F5HIP_PREC_FP16; ++prec) {
    CHECK(f5hip_sample(ctx, 1, N, cond_d, cond_mask, text, NTXT, duration, 0, y0_d, t, STEPS, 0, 2.0f, prec, out_d, traj_d, NULL));
    CHECK(f5hip_sample(ctx, 1, N, cond_d, cond_mask, text, NTXT, duration, 0, y0_d, t, STEPS, 0, 2.0f, prec, out2_d, NULL, NULL));
    HIP(hipDeviceSynchronize());
    float* out_h = (float*)malloc((size_t)N * MEL * sizeof(float));
    float* out2_h = (float*)malloc((size_t)N * MEL * sizeof(float));
    float* cond_h = (float*)malloc((size_t)N * MEL * sizeof(float));
    float* tr_h = (float*)malloc((size_t)N * MEL * sizeof(float));
    HIP(hipMemcpy(out_h, out_d, (size_t)N * MEL * sizeof(float), hipMemcpyDeviceToHost));
    HIP(hipMemcpy(out2_h, out2_d, (size_t)N * MEL * sizeof(float), hipMemcpyDeviceToHost));
    HIP(hipMemcpy(cond_h, cond_d, (size_t)N * MEL * sizeof(float), hipMemcpyDeviceToHost));
    HIP(hipMemcpy(tr_h, traj_d, (size_t)N * MEL * sizeof(float), hipMemcpyDeviceToHost));
    double moved = 0.0;
    for (int i = 0; i < N * MEL; ++i) {
      if (!isfinite(out_h[i])) { fprintf(stderr, "precision %d: non-finite output at %d\n", prec, i); return 1; }
      if (out_h[i] != out2_h[i]) { fprintf(stderr, "precision %d: two identical calls differ at %d\n", prec, i); return 1; }
      if (tr_h[i] != y0_h[i]) { fprintf(stderr, "trajectory[0] != y0 at %d\n", i); return 1; }
      if (i < LENS * MEL && out_h[i] != cond_h[i]) { fprintf(stderr, "prompt frame not restored at %d\n", i); return 1; }
      if (i >= LENS * MEL) moved += fabs((double)out_h[i] - (double)y0_h[i]);
    }
    if (moved == 0.0) { fprintf(stderr, "precision %d: the generated frames never left the noise\n", prec); return 1; }
    free(out_h); free(out2_h); free(cond_h); free(tr_h);
  }
  /* vocode the generated part (frame-major slice of `out`) */
  CHECK(f5hip_vocos_decode(ctx, out_d + (size_t)LENS * MEL, 1, N - LENS, /*channel_major=*/0, wave_d, NULL));
  HIP(hipDeviceSynchronize());
  {
    const int nwav = 256 * (N - LENS - 1);
    float* w = (float*)malloc((size_t)nwav * sizeof(float));
    HIP(hipMemcpy(w, wave_d, (size_t)nwav * sizeof(float), hipMemcpyDeviceToHost));
    double e = 0.0;
    for (int i = 0; i < nwav; ++i) { if (!isfinite(w[i])) { fprintf(stderr, "non-finite wave sample\n"); return 1; } e += (double)w[i] * w[i]; }
    if (e == 0.0) { fprintf(stderr, "silent wave\n"); return 1; }
    free(w);
  }
  /* how sharp are this model's softmax rows (ABI v10)?  one fp32 call with the option on: rows were counted, the figures are probabilities */
  {
    double st[4];
    if (f5hip_attention_stats(ctx, st, 0) == 0) { fprintf(stderr, "attention statistics readable before the option was ever set\n"); return 1; }
    CHECK(f5hip_set_option(ctx, "attn_stats", 1));
    CHECK(f5hip_sample(ctx, 1, N, cond_d, cond_mask, text, NTXT, duration, 0, y0_d, t, STEPS, 0, 2.0f, F5HIP_PREC_FP32, out2_d, NULL, NULL));
    CHECK(f5hip_attention_stats(ctx, st, 1));
    CHECK(f5hip_set_option(ctx, "attn_stats", 0));
    if (!(st[2] > 0.0 && st[1] / st[2] >= 1.0 / N && st[1] / st[2] <= st[0] && st[0] <= 1.0 + 1e-6 && st[3] <= st[2])) {
      fprintf(stderr, "attention statistics out of range: max %g sum %g rows %g above-half %g\n", st[0], st[1], st[2], st[3]);
      return 1;
    }
    CHECK(f5hip_attention_stats(ctx, st, 0));
    if (st[2] != 0.0) { fprintf(stderr, "attention statistics not reset\n"); return 1; }
  }
  /* argument errors come back as status codes with a message, never as crashes */
  if (f5hip_sample(ctx, 1, N, cond_d, cond_mask, text, NTXT, duration, 0, y0_d, t, STEPS, 7, 2.0f, 0, out_d, NULL, NULL) == 0 ||
      strlen(f5hip_last_error(ctx)) == 0) { fprintf(stderr, "bad ode_method accepted\n"); return 1; }
  CHECK(f5hip_destroy(ctx));
  printf("c_abi smoke ok\n");
  return 0;
}
