"""TEST INFRASTRUCTURE — mint golden fixtures by running the REFERENCE ITSELF on CPU.

The reference ships no golden vectors, so the pin for this build is: outputs of the
reference's own ``CFM.sample`` / ``DiT`` code (imported verbatim from ``/root/reference/src`` through
``oracle/ref_shims.py``) on seeded synthetic weights and inputs, stored under ``tests/golden/``.
Weights/inputs are NOT stored: they are regenerated bit-identically from seeds by
``f5-tts_amd/synth.py`` (same torch build on every box).

Run in the build container only (``/root/reference`` does not exist on the GPU box):

    python oracle/make_golden.py [--full]

Also cross-checks, and records in ``tests/golden/pins.json``:
  * the standalone restatement ``oracle/f5_oracle.py`` against the live reference (max-abs),
  * torch.stft / torch.istft against the reference's own runnable conv-STFT
    (``src/f5_tts/runtime/triton_trtllm/scripts/conv_stft.py``) — the only in-repo pin for the
    STFT/iSTFT stage of the Vocos mel front-end / vocoder head.
"""
from __future__ import annotations

import argparse
import json
import math
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import f5_tts_amd  # noqa: E402,F401
from f5_tts_amd import config, synth  # noqa: E402
from oracle import f5_oracle as O  # noqa: E402
from oracle import ref_shims  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")
MEL_KW = dict(n_fft=1024, hop_length=256, win_length=1024, n_mel_channels=100, target_sample_rate=24000, mel_spec_type="vocos")

# name -> (preset, weight seed, wave samples, wave seed, batch, nt, text seed, duration, lens, sample kwargs)
CASES = {
    "tiny_v1_nfe16": dict(preset="tiny", wseed=1, nw=256 * 60, wavseed=3, batch=1, nt=40, tseed=2, duration=200, lens=None,
                          kw=dict(steps=16, cfg_strength=2.0, sway_sampling_coef=-1.0, seed=7)),
    "tiny_v0_nfe16": dict(preset="tiny_v0", wseed=2, nw=256 * 60, wavseed=3, batch=1, nt=40, tseed=2, duration=200, lens=None,
                          kw=dict(steps=16, cfg_strength=2.0, sway_sampling_coef=-1.0, seed=7)),
    "tiny_v1_nfe32_nosway": dict(preset="tiny", wseed=1, nw=256 * 50 + 77, wavseed=4, batch=1, nt=25, tseed=5, duration=150, lens=None,
                                 kw=dict(steps=32, cfg_strength=1.5, sway_sampling_coef=None, seed=11)),
    "tiny_v1_ragged_b2": dict(preset="tiny", wseed=1, nw=256 * 60, wavseed=3, batch=2, nt=40, tseed=2, duration=[200, 170], lens=[61, 50],
                              pad_from=30, kw=dict(steps=8, cfg_strength=2.0, sway_sampling_coef=-1.0, seed=7)),
    # E2-TTS flat U-Net transformer (reference backbones/unett.py): single utterance and a ragged batch
    "tiny_unett_nfe8": dict(preset="tiny_unett", wseed=3, nw=256 * 50, wavseed=6, batch=1, nt=30, tseed=4, duration=140, lens=None,
                            kw=dict(steps=8, cfg_strength=2.0, sway_sampling_coef=-1.0, seed=5)),
    "tiny_unett_ragged_b2": dict(preset="tiny_unett", wseed=3, nw=256 * 50, wavseed=6, batch=2, nt=30, tseed=4, duration=[140, 111],
                                 lens=[51, 40], pad_from=22, kw=dict(steps=6, cfg_strength=2.0, sway_sampling_coef=-1.0, seed=5)),
    # solver / guidance variants of the sampler itself (cfm.py:166-177 single branch; odeint_kwargs method="midpoint")
    "tiny_v1_midpoint": dict(preset="tiny", wseed=1, nw=256 * 40, wavseed=7, batch=1, nt=24, tseed=8, duration=120, lens=None, method="midpoint",
                             kw=dict(steps=6, cfg_strength=2.0, sway_sampling_coef=-1.0, seed=9)),
    "tiny_v1_nocfg_b2": dict(preset="tiny", wseed=1, nw=256 * 40, wavseed=7, batch=2, nt=24, tseed=8, duration=[120, 97], lens=[41, 33],
                             pad_from=18, kw=dict(steps=6, cfg_strength=0.0, sway_sampling_coef=None, seed=9)),
    # optional DiT constructor switches (dit.py:181-189), one at a time and all together on a ragged batch with the key-padding mask
    "tiny_qknorm": dict(preset="tiny_qknorm", wseed=4, nw=256 * 30, wavseed=9, batch=1, nt=20, tseed=6, duration=100, lens=None,
                        kw=dict(steps=4, cfg_strength=2.0, sway_sampling_coef=-1.0, seed=3)),
    "tiny_longskip": dict(preset="tiny_longskip", wseed=4, nw=256 * 30, wavseed=9, batch=1, nt=20, tseed=6, duration=100, lens=None,
                          kw=dict(steps=4, cfg_strength=2.0, sway_sampling_coef=-1.0, seed=3)),
    "tiny_avgup": dict(preset="tiny_avgup", wseed=4, nw=256 * 30, wavseed=9, batch=1, nt=20, tseed=6, duration=100, lens=None,
                       kw=dict(steps=4, cfg_strength=2.0, sway_sampling_coef=-1.0, seed=3)),
    "tiny_flags_ragged_b2": dict(preset="tiny_flags", wseed=4, nw=256 * 60, wavseed=3, batch=2, nt=40, tseed=2, duration=[200, 170], lens=[61, 50],
                                 pad_from=30, kw=dict(steps=8, cfg_strength=2.0, sway_sampling_coef=-1.0, seed=7)),
    # UNetT constructor switches (unett.py:120-127): skip_connect_type "add" / "none", qk_norm, key-padding mask
    "tiny_unett_add_ragged_b2": dict(preset="tiny_unett_add", wseed=5, nw=256 * 50, wavseed=6, batch=2, nt=30, tseed=4, duration=[140, 111],
                                     lens=[51, 40], pad_from=22, kw=dict(steps=4, cfg_strength=2.0, sway_sampling_coef=-1.0, seed=5)),
    "tiny_unett_noskip": dict(preset="tiny_unett_noskip", wseed=5, nw=256 * 30, wavseed=6, batch=1, nt=20, tseed=4, duration=100, lens=None,
                              kw=dict(steps=4, cfg_strength=2.0, sway_sampling_coef=-1.0, seed=5)),
    # MMDiT backbone (reference backbones/mmdit.py): two token streams with joint attention
    "tiny_mmdit_nfe6": dict(preset="tiny_mmdit", wseed=6, nw=256 * 40, wavseed=7, batch=1, nt=24, tseed=8, duration=120, lens=None,
                            kw=dict(steps=6, cfg_strength=2.0, sway_sampling_coef=-1.0, seed=9)),
    "tiny_mmdit_ragged_b2": dict(preset="tiny_mmdit", wseed=6, nw=256 * 40, wavseed=7, batch=2, nt=24, tseed=8, duration=[120, 97], lens=[41, 33],
                                 pad_from=18, kw=dict(steps=4, cfg_strength=2.0, sway_sampling_coef=-1.0, seed=9)),
    "tiny_mmdit_mask_ragged_b2": dict(preset="tiny_mmdit_mask", wseed=6, nw=256 * 40, wavseed=7, batch=2, nt=24, tseed=8, duration=[120, 97],
                                      lens=[41, 33], pad_from=18, kw=dict(steps=4, cfg_strength=2.0, sway_sampling_coef=-1.0, seed=9)),
    "tiny_mmdit_nopad_nocfg_b2": dict(preset="tiny_mmdit_nopad", wseed=6, nw=256 * 40, wavseed=7, batch=2, nt=24, tseed=8, duration=[120, 97],
                                      lens=[41, 33], pad_from=18, kw=dict(steps=4, cfg_strength=0.0, sway_sampling_coef=None, seed=9)),
    # 48 channels per conv-position group (dim 768 / 16 groups in the Small models)
    "tiny48_ragged_b2": dict(preset="tiny48", wseed=7, nw=256 * 60, wavseed=3, batch=2, nt=40, tseed=2, duration=[200, 170], lens=[61, 50],
                             pad_from=30, kw=dict(steps=4, cfg_strength=2.0, sway_sampling_coef=-1.0, seed=7)),
    # 64 channels per conv-position group (dim 1024 / 16 groups: the Base models' width at depth 2): the MX form of the conv-position kernel (fp16m)
    "tiny1024_ragged_b2": dict(preset="tiny1024", wseed=7, nw=256 * 60, wavseed=3, batch=2, nt=40, tseed=2, duration=[200, 170], lens=[61, 50],
                               pad_from=30, kw=dict(steps=4, cfg_strength=2.0, sway_sampling_coef=-1.0, seed=7)),
    # attention inner width != model width (heads * dim_head = 512, dim = 256)
    "tiny_inner512": dict(preset="tiny_inner512", wseed=8, nw=256 * 30, wavseed=9, batch=1, nt=20, tseed=6, duration=100, lens=None,
                          kw=dict(steps=4, cfg_strength=2.0, sway_sampling_coef=-1.0, seed=3)),
    # corners of CFM.sample itself: duplicate_test / t_inter (cfm.py:141-143,205-209) and no_ref_audio (cfm.py:146-147)
    "tiny_v1_duptest": dict(preset="tiny", wseed=1, nw=256 * 40, wavseed=7, batch=1, nt=24, tseed=8, duration=120, lens=None,
                            kw=dict(steps=8, cfg_strength=2.0, sway_sampling_coef=-1.0, seed=9, duplicate_test=True, t_inter=0.25)),
    "tiny_v1_noref": dict(preset="tiny", wseed=1, nw=256 * 40, wavseed=7, batch=1, nt=24, tseed=8, duration=120, lens=None,
                          kw=dict(steps=6, cfg_strength=2.0, sway_sampling_coef=-1.0, seed=9, no_ref_audio=True)),
    # dynamic-range stress (synth.stress_dit_state_dict): per-tensor weight scales over three decades, outlier AdaLN channels, a clipped prompt
    "tiny_v1_stress": dict(preset="tiny", wseed=1, stress=True, loud=True, nw=256 * 60, wavseed=3, batch=1, nt=40, tseed=2, duration=200, lens=None,
                           kw=dict(steps=16, cfg_strength=2.0, sway_sampling_coef=-1.0, seed=7)),
    # a ragged batch of three with the key-padding mask (attn_mask_enabled): the case the packed-row path must reproduce on its valid rows
    "tiny_mask_ragged_b3": dict(preset="tiny_mask", wseed=9, nw=256 * 60, wavseed=3, batch=3, nt=40, tseed=2, duration=[200, 163, 97], lens=[61, 50, 33],
                                pad_from=30, kw=dict(steps=6, cfg_strength=2.0, sway_sampling_coef=-1.0, seed=7)),
    # trained-checkpoint weight statistics (synth.trained_like_dit_state_dict): Student-t entries, log-normal row / column gains, norm gains far from 1
    "tiny_v1_trained_like": dict(preset="tiny", wseed=1, trained=True, nw=256 * 60, wavseed=3, batch=1, nt=40, tseed=2, duration=200, lens=None,
                                 kw=dict(steps=16, cfg_strength=2.0, sway_sampling_coef=-1.0, seed=7)),
    "tiny_unett_trained_like": dict(preset="tiny_unett", wseed=2, trained=True, nw=256 * 50, wavseed=6, batch=1, nt=30, tseed=4, duration=140, lens=None,
                                    kw=dict(steps=8, cfg_strength=2.0, sway_sampling_coef=-1.0, seed=5)),
    # trained-like statistics on the backbones whose q | k | v projection DETOURS the pipelined epilogue: MMDiT's joint
    # slabs, the qk-norm pass, and both under the key mask
    "tiny_mmdit_trained_like": dict(preset="tiny_mmdit", wseed=3, trained=True, nw=256 * 40, wavseed=7, batch=1, nt=24, tseed=8, duration=120, lens=None,
                                    kw=dict(steps=6, cfg_strength=2.0, sway_sampling_coef=-1.0, seed=9)),
    "tiny_qknorm_trained_like": dict(preset="tiny_qknorm", wseed=3, trained=True, nw=256 * 30, wavseed=9, batch=1, nt=20, tseed=6, duration=100, lens=None,
                                     kw=dict(steps=4, cfg_strength=2.0, sway_sampling_coef=-1.0, seed=3)),
    "tiny_mmdit_mask_trained_like_b2": dict(preset="tiny_mmdit_mask", wseed=3, trained=True, nw=256 * 40, wavseed=7, batch=2, nt=24, tseed=8,
                                            duration=[120, 97], lens=[41, 33], pad_from=18, kw=dict(steps=4, cfg_strength=2.0, sway_sampling_coef=-1.0, seed=9)),
    "tiny_v1_b3_fixed": dict(preset="tiny", wseed=1, nw=256 * 30, wavseed=9, batch=3, nt=20, tseed=6, duration=96, lens=None,
                             kw=dict(steps=6, cfg_strength=2.0, sway_sampling_coef=-1.0, seed=0)),
}
# The tiny points of the SHARPNESS SWEEP (to_q, to_k x s: logits x s^2): measurements of how the half-precision modes' error moves with the
# logits (tools/sharpness_sweep.py), asserted by tests/test_gpu_parity.py::test_sharpness_sweep_* with bounds of
# their own — not members of the parity matrix above, whose tolerance they leave at logits x 16
SWEEP_CASES = {
    "tiny_v1_trained_like_sharp2": dict(preset="tiny", wseed=1, trained=True, sharp=2.0, nw=256 * 60, wavseed=3, batch=1, nt=40, tseed=2, duration=200, lens=None,
                                        kw=dict(steps=16, cfg_strength=2.0, sway_sampling_coef=-1.0, seed=7)),
    "tiny_v1_trained_like_sharp4": dict(preset="tiny", wseed=1, trained=True, sharp=4.0, nw=256 * 60, wavseed=3, batch=1, nt=40, tseed=2, duration=200, lens=None,
                                        kw=dict(steps=16, cfg_strength=2.0, sway_sampling_coef=-1.0, seed=7)),
}
FULL_CASES = {
    # the Small models at full width/depth on a shorter utterance (2.1 s prompt, 500 frames) to keep the fixtures small
    "small_v1": dict(preset="F5TTS_v1_Small", wseed=0, nw=256 * 200, wavseed=0, batch=1, nt=80, tseed=0, duration=500, lens=None,
                     kw=dict(steps=8, cfg_strength=2.0, sway_sampling_coef=-1.0, seed=0)),
    "small_e2": dict(preset="E2TTS_Small", wseed=0, nw=256 * 200, wavseed=0, batch=1, nt=80, tseed=0, duration=500, lens=None,
                     kw=dict(steps=4, cfg_strength=2.0, sway_sampling_coef=-1.0, seed=0)),
    # the Small model with the key-padding mask on a ragged batch of four (lengths 500 / 431 / 333 / 250): packed rows at a real width
    "small_mask_ragged_b4": dict(preset="F5TTS_v1_Small_mask", wseed=0, nw=256 * 200, wavseed=0, batch=4, nt=80, tseed=0, duration=[500, 431, 333, 250],
                                 lens=[201, 160, 130, 101], pad_from=50, kw=dict(steps=4, cfg_strength=2.0, sway_sampling_coef=-1.0, seed=0)),
    # the same kind of batch (three sequences, 460 / 380 / 250 frames, key-padding mask) on the Small model with trained-checkpoint weight
    # STATISTICS (round 5): the attention-score arithmetic under large logits AND masked tails / packed rows
    "small_mask_ragged_b3_trained_like": dict(preset="F5TTS_v1_Small_mask", wseed=2, trained=True, nw=256 * 190, wavseed=4, batch=3, nt=70, tseed=6,
                                              duration=[460, 380, 250], lens=[181, 150, 101], pad_from=50,
                                              kw=dict(steps=8, cfg_strength=2.0, sway_sampling_coef=-1.0, seed=0)),
    # BASELINE.json configs[4] backbone at full size: E2-TTS Base (UNetT, depth 24, ff_mult 4), same prompt/duration as config 1
    "e2_base_cfg5": dict(preset="E2TTS_Base", wseed=0, nw=120000, wavseed=0, batch=1, nt=220, tseed=0, duration=1406, lens=None,
                         kw=dict(steps=8, cfg_strength=2.0, sway_sampling_coef=-1.0, seed=0)),
    # BASELINE.json configs[2] / [3] shape at a size the CPU finishes in minutes: F5-TTS Base, a BATCH of 4 distinct fixed-length prompts
    # (rows of the packed cond | uncond schedule: 8 x 1406), NFE 32, sway, CFG 2 — the batched path of the engine at the full model size
    "base_v1_cfg3_b4": dict(preset="F5TTS_v1_Base", wseed=0, nw=120000, wavseed=10, batch=4, nt=220, tseed=3, duration=1406, lens=None,
                            kw=dict(steps=32, cfg_strength=2.0, sway_sampling_coef=-1.0, seed=0)),
    # the configs[1] case with the dynamic-range stress applied to the weights and a loud, clipped prompt (VERDICT r02 "weak" 1): does the
    # fp16 hi/lo split survive weights and activations whose scales span three decades?
    "base_v1_stress": dict(preset="F5TTS_v1_Base", wseed=0, stress=True, loud=True, nw=120000, wavseed=0, batch=1, nt=220, tseed=0, duration=1406, lens=None,
                           kw=dict(steps=16, cfg_strength=2.0, sway_sampling_coef=-1.0, seed=0)),
    # the configs[1] case on weights with trained-checkpoint statistics (no real checkpoint is reachable; every other golden
    # uses Gaussian matrices): heavy-tailed entries, row / column gains over a factor of ~5, LayerNorm / GRN parameters far from their initial values
    "base_v1_trained_like": dict(preset="F5TTS_v1_Base", wseed=0, trained=True, nw=120000, wavseed=0, batch=1, nt=220, tseed=0, duration=1406, lens=None,
                                 kw=dict(steps=16, cfg_strength=2.0, sway_sampling_coef=-1.0, seed=0)),
    # SHARPNESS SWEEP: the two trained-like cases with every to_q / to_k (weight and bias) x 2 and x 4, i.e. every attention
    # logit x 4 and x 16 — does the error of the half-precision modes plateau (fp16 P.V is bounded by V's own rounding) or grow with the logits?
    "base_v1_trained_like_sharp1p4": dict(preset="F5TTS_v1_Base", wseed=0, trained=True, sharp=math.sqrt(2.0), nw=120000, wavseed=0, batch=1, nt=220, tseed=0,
                                          duration=1406, lens=None, kw=dict(steps=16, cfg_strength=2.0, sway_sampling_coef=-1.0, seed=0)),
    "small_mask_ragged_b3_trained_like_sharp1p4": dict(preset="F5TTS_v1_Small_mask", wseed=2, trained=True, sharp=math.sqrt(2.0), nw=256 * 190, wavseed=4, batch=3,
                                                       nt=70, tseed=6, duration=[460, 380, 250], lens=[181, 150, 101], pad_from=50,
                                                       kw=dict(steps=8, cfg_strength=2.0, sway_sampling_coef=-1.0, seed=0)),
    # logits x 1.41 (to_q, to_k x 2^0.25): the point between "inside" (x 1) and "outside" (x 2) of the default attention's tolerance — the
    # calibration of f5hip_attention_stats' thresholds ("Which attention form does my checkpoint need?")
    "base_v1_trained_like_sharp1p2": dict(preset="F5TTS_v1_Base", wseed=0, trained=True, sharp=2.0 ** 0.25, nw=120000, wavseed=0, batch=1, nt=220, tseed=0,
                                          duration=1406, lens=None, kw=dict(steps=16, cfg_strength=2.0, sway_sampling_coef=-1.0, seed=0)),
    "small_mask_ragged_b3_trained_like_sharp1p2": dict(preset="F5TTS_v1_Small_mask", wseed=2, trained=True, sharp=2.0 ** 0.25, nw=256 * 190, wavseed=4, batch=3,
                                                       nt=70, tseed=6, duration=[460, 380, 250], lens=[181, 150, 101], pad_from=50,
                                                       kw=dict(steps=8, cfg_strength=2.0, sway_sampling_coef=-1.0, seed=0)),
    "base_v1_trained_like_sharp1p7": dict(preset="F5TTS_v1_Base", wseed=0, trained=True, sharp=2.0 ** 0.75, nw=120000, wavseed=0, batch=1, nt=220, tseed=0,
                                          duration=1406, lens=None, kw=dict(steps=16, cfg_strength=2.0, sway_sampling_coef=-1.0, seed=0)),
    "small_mask_ragged_b3_trained_like_sharp1p7": dict(preset="F5TTS_v1_Small_mask", wseed=2, trained=True, sharp=2.0 ** 0.75, nw=256 * 190, wavseed=4, batch=3,
                                                       nt=70, tseed=6, duration=[460, 380, 250], lens=[181, 150, 101], pad_from=50,
                                                       kw=dict(steps=8, cfg_strength=2.0, sway_sampling_coef=-1.0, seed=0)),
    "base_v1_trained_like_sharp2": dict(preset="F5TTS_v1_Base", wseed=0, trained=True, sharp=2.0, nw=120000, wavseed=0, batch=1, nt=220, tseed=0,
                                        duration=1406, lens=None, kw=dict(steps=16, cfg_strength=2.0, sway_sampling_coef=-1.0, seed=0)),
    # logits x 16: record_only — the sampler is CHAOTIC there (near one-hot softmax rows flip between keys): the fp32 restatement differs from the fp32
    # reference by 16 (pins.json), so no fixture is stored and no arithmetic, the reference's own on another thread count included, reproduces it
    "base_v1_trained_like_sharp4": dict(preset="F5TTS_v1_Base", wseed=0, trained=True, sharp=4.0, record_only=True, nw=120000, wavseed=0, batch=1, nt=220, tseed=0,
                                        duration=1406, lens=None, kw=dict(steps=16, cfg_strength=2.0, sway_sampling_coef=-1.0, seed=0)),
    "small_mask_ragged_b3_trained_like_sharp2": dict(preset="F5TTS_v1_Small_mask", wseed=2, trained=True, sharp=2.0, nw=256 * 190, wavseed=4, batch=3, nt=70,
                                                     tseed=6, duration=[460, 380, 250], lens=[181, 150, 101], pad_from=50,
                                                     kw=dict(steps=8, cfg_strength=2.0, sway_sampling_coef=-1.0, seed=0)),
    "small_mask_ragged_b3_trained_like_sharp4": dict(preset="F5TTS_v1_Small_mask", wseed=2, trained=True, sharp=4.0, record_only=True, nw=256 * 190, wavseed=4, batch=3, nt=70,
                                                     tseed=6, duration=[460, 380, 250], lens=[181, 150, 101], pad_from=50,
                                                     kw=dict(steps=8, cfg_strength=2.0, sway_sampling_coef=-1.0, seed=0)),
    # BASELINE.json configs[4] at its own NFE and as a BATCH (VERDICT r03 "missing" 4): E2-TTS Base, two distinct fixed-length prompts, NFE 16 —
    # rows of the B = 8 schedule bench.py --model E2TTS_Base --batch 8 times (the GPU test repeats them to 8: fixed-length batches have no
    # cross-row coupling, as with base_v1_cfg3_b4 for configs[2])
    "e2_base_cfg5_b2": dict(preset="E2TTS_Base", wseed=0, nw=120000, wavseed=20, batch=2, nt=220, tseed=5, duration=1406, lens=None,
                            kw=dict(steps=16, cfg_strength=2.0, sway_sampling_coef=-1.0, seed=0)),
    # the v0 yaml at full size (src/f5_tts/configs/F5TTS_Base.yaml:20-46: pe_attn_head 1, text_mask_padding False) — the configs[1] case on it
    "base_v0_cfg1": dict(preset="F5TTS_Base", wseed=0, nw=120000, wavseed=0, batch=1, nt=220, tseed=0, duration=1406, lens=None,
                         kw=dict(steps=16, cfg_strength=2.0, sway_sampling_coef=-1.0, seed=0)),
    # BASELINE.json configs[0]/[1]: F5-TTS Base, 5 s ref + 10 s gen, NFE 16, sway, CFG 2
    "base_v1_cfg1": dict(preset="F5TTS_v1_Base", wseed=0, nw=120000, wavseed=0, batch=1, nt=220, tseed=0, duration=1406, lens=None,
                         kw=dict(steps=16, cfg_strength=2.0, sway_sampling_coef=-1.0, seed=0)),
}


def case_inputs(c):
    cfg = config.PRESETS[c["preset"]]
    wav = (synth.synth_loud_wave if c.get("loud") else synth.synth_wave)(c["nw"], seed=c["wavseed"], batch=c["batch"])
    text = synth.synth_text_ids(c["batch"], c["nt"], cfg.text_num_embeds, seed=c["tseed"])
    if "pad_from" in c:
        text[1, c["pad_from"]:] = -1
    duration = c["duration"] if isinstance(c["duration"], int) else torch.tensor(c["duration"])
    lens = torch.tensor(c["lens"]) if c["lens"] is not None else None
    return cfg, wav, text, duration, lens


def case_weights(c):
    """The seeded state dict of a case (with the dynamic-range stress when the case asks for it)."""
    cfg = config.PRESETS[c["preset"]]
    sd = synth.synth_dit_state_dict(cfg, seed=c["wseed"])
    if c.get("trained"):
        sd = synth.trained_like_dit_state_dict(sd, cfg, seed=c["wseed"])
        return synth.sharpen_attention_state_dict(sd, c["sharp"]) if c.get("sharp") else sd
    return synth.stress_dit_state_dict(sd, cfg, seed=c["wseed"]) if c.get("stress") else sd


def build_reference(cfg, sd, method="euler"):
    return ref_shims.build_reference_cfm(cfg, sd, method)


def run_case(name, c, pins):
    cfg, wav, text, duration, lens = case_inputs(c)
    sd = case_weights(c)
    method = c.get("method", "euler")
    model = build_reference(cfg, sd, method)
    t0 = time.time()
    with torch.no_grad():
        out, traj = model.sample(wav, text, duration, lens=lens, **c["kw"])
    t_ref = time.time() - t0
    out_o, traj_o = O.cfm_sample(sd, cfg, wav, text, duration, lens=lens, method=method, **c["kw"])
    d = (out - out_o).abs().max().item()
    dt = (traj - traj_o).abs().max().item()
    print(f"{name}: reference {t_ref:.1f}s  out {tuple(out.shape)}  oracle-vs-reference out {d:.2e} traj {dt:.2e}")
    steps = traj.shape[0] - 1  # duplicate_test shortens the solve (cfm.py:209)
    if not c.get("record_only"):
        np.savez_compressed(os.path.join(GOLD, name + ".npz"), out=out.numpy(), traj_1=traj[1].numpy(),
                            traj_mid=traj[steps // 2].numpy(), traj_last=traj[-1].numpy())
    pins[name] = dict(case={k: v for k, v in c.items()}, oracle_vs_reference_out=d, oracle_vs_reference_traj=dt,
                      reference_seconds=t_ref, out_absmax=out.abs().max().item())


def pin_stft(pins):
    m = ref_shims.reference_conv_stft()
    stft = m.STFT(fft_len=1024, win_hop=256, win_len=1024)
    wav = synth.synth_wave(256 * 40, seed=21)
    re, im = stft.transform(wav, return_type="realimag")
    spec = torch.stft(wav, 1024, hop_length=256, win_length=1024, window=torch.hann_window(1024), center=True, pad_mode="reflect",
                      return_complex=True)
    d_fwd = max((re - spec.real).abs().max().item(), (im - spec.imag).abs().max().item())
    T = spec.shape[-1]
    inv_ref = stft.inverse(spec.real.contiguous(), spec.imag.contiguous(), "realimag")  # 256*T samples
    inv_t = torch.istft(spec, 1024, hop_length=256, win_length=1024, window=torch.hann_window(1024), center=True)
    d_inv = (inv_ref[:, : 256 * (T - 1)] - inv_t).abs().max().item()
    d_man = (O.istft_manual(spec) - inv_t).abs().max().item()
    print(f"conv_stft pin: transform vs torch.stft {d_fwd:.2e}; inverse vs torch.istft {d_inv:.2e}; istft_manual vs torch.istft {d_man:.2e}")
    # a golden for the iSTFT stage produced by the REFERENCE's conv-iSTFT
    g = torch.Generator().manual_seed(5)
    sp = torch.complex(torch.randn(1, 513, 24, generator=g), torch.randn(1, 513, 24, generator=g))
    sp[:, 0].imag.zero_()
    sp[:, -1].imag.zero_()
    ref_wav = stft.inverse(sp.real.contiguous(), sp.imag.contiguous(), "realimag")[:, : 256 * 23]
    np.savez_compressed(os.path.join(GOLD, "istft_conv_reference.npz"), spec_re=sp.real.numpy(), spec_im=sp.imag.numpy(),
                        wav=ref_wav.numpy())
    pins["conv_stft"] = dict(transform_vs_torch_stft=d_fwd, inverse_vs_torch_istft=d_inv, istft_manual_vs_torch_istft=d_man)


def pin_mel(pins):
    """MelSpec.forward of the reference (through the torchaudio shim) vs the oracle; store a golden."""
    ref_shims.install()
    from f5_tts.model.modules import MelSpec

    wav = synth.synth_wave(256 * 37 + 100, seed=13, batch=2)
    ref = MelSpec(**MEL_KW)(wav)
    d = (ref - O.vocos_mel(wav)).abs().max().item()
    print(f"mel: reference MelSpec (torchaudio shim) vs oracle {d:.2e}")
    np.savez_compressed(os.path.join(GOLD, "mel_b2.npz"), mel=ref.numpy())
    pins["mel_b2"] = dict(nw=256 * 37 + 100, wavseed=13, batch=2, oracle_vs_reference=d)


def pin_mel_bigvgan(pins):
    """get_bigvgan_mel_spectrogram of the reference (model/modules.py:35-77; librosa's filterbank through the shim) vs the oracle."""
    ref_shims.install()
    from f5_tts.model.modules import MelSpec

    wav = synth.synth_wave(256 * 37 + 100, seed=14, batch=2)
    kw = dict(MEL_KW, mel_spec_type="bigvgan")
    ref = MelSpec(**kw)(wav)
    d = (ref - O.bigvgan_mel(wav)).abs().max().item()
    print(f"bigvgan mel: reference MelSpec (librosa filterbank restated) vs oracle {d:.2e}")
    np.savez_compressed(os.path.join(GOLD, "mel_bigvgan_b2.npz"), mel=ref.numpy())
    pins["mel_bigvgan_b2"] = dict(nw=256 * 37 + 100, wavseed=14, batch=2, oracle_vs_reference=d,
                                  note="librosa absent: slaney filterbank restated (oracle/f5_oracle.py::slaney_mel_basis), table unpinned")


def golden_vocos(pins):
    """Vocos has no source in the reference tree: the golden comes from the ORACLE restatement (parity unpinned)."""
    vcfg = config.VOCOS_TINY
    vsd = synth.synth_vocos_state_dict(vcfg, seed=1)
    mel = O.vocos_mel(synth.synth_wave(256 * 80, seed=5))
    wav = O.vocos_decode(vsd, mel, vcfg.num_layers)
    np.savez_compressed(os.path.join(GOLD, "vocos_tiny.npz"), wav=wav.numpy())
    pins["vocos_tiny"] = dict(source="oracle restatement (vocos package absent) — parity unpinned by the reference", wseed=1, wavseed=5,
                              nw=256 * 80, absmax=wav.abs().max().item())


def golden_vocos_head(pins):
    """The ISTFT head of Vocos as the REFERENCE restates it for its ONNX export (``runtime/triton_trtllm/scripts/export_vocoder_to_onnx.py:43-59``:
    ``out`` Linear -> exp -> clip(1e2) -> cos / sin -> ``conv_stft.STFT.inverse``): the class is lifted out of the script with ``ast`` (the
    script itself imports ``vocos`` / ``huggingface_hub``, absent here), given the seeded ``head.out`` layer of ``synth_vocos_state_dict`` and run on a
    seeded hidden state.  Pins the head as a unit at the full Vocos width; the backbone stays unpinned."""
    import ast
    import torch.nn as nn

    path = os.path.join(ref_shims.REFERENCE_SRC, "f5_tts", "runtime", "triton_trtllm", "scripts", "export_vocoder_to_onnx.py")
    cls = next(n for n in ast.parse(open(path, encoding="utf-8").read()).body if isinstance(n, ast.ClassDef) and n.name == "ISTFTHead")
    env = {"torch": torch, "nn": nn, "STFT": ref_shims.reference_conv_stft().STFT}
    exec(compile(ast.Module(body=[cls], type_ignores=[]), path, "exec"), env)
    for name, vcfg, vseed, T, hseed in (("vocos_head_ref", config.VOCOS_MEL_24K, 2, 96, 31), ("vocos_head_ref_tiny", config.VOCOS_TINY, 1, 40, 32)):
        vsd = synth.synth_vocos_state_dict(vcfg, seed=vseed)
        head = env["ISTFTHead"](vcfg.n_fft, vcfg.hop_length)
        head.out = nn.Linear(vcfg.dim, vcfg.n_fft + 2)
        with torch.no_grad():
            head.out.weight.copy_(vsd["head.out.weight"])
            head.out.bias.copy_(vsd["head.out.bias"])
        g = torch.Generator().manual_seed(hseed)
        hidden = torch.randn(2, T, vcfg.dim, generator=g)
        hidden[1] *= 6.0  # second row: log-magnitudes large enough for the 1e2 clip to act on many bins
        with torch.no_grad():
            # one row per call: conv_stft.STFT.inverse divides by the window envelope through `th.where(coff > 1e-8)` on a [1, 1, L]
            # tensor (conv_stft.py:221-233), which normalises batch row 0 ONLY — a batch of 2 comes back with row 1 un-normalised (0.08
            # off torch.istft).  vocos itself calls torch.istft, so the single-row behaviour is the head's meaning.
            # conv-iSTFT emits 256 T samples; the first 256 (T - 1) are torch.istft(center=True)'s
            wav = torch.cat([head(hidden[r : r + 1])[:, : 256 * (T - 1)] for r in range(hidden.shape[0])])
            logits = head.out(hidden)
        clipped = (logits[..., : vcfg.n_fft // 2 + 1] > math.log(1e2)).float().mean().item()
        d = (O.vocos_head(vsd, hidden) - wav).abs().max().item() if hasattr(O, "vocos_head") else float("nan")
        print(f"{name}: reference ISTFTHead wav {tuple(wav.shape)} absmax {wav.abs().max().item():.3f}, clipped bins {clipped:.3f}, oracle-vs-reference {d:.2e}")
        np.savez_compressed(os.path.join(GOLD, name + ".npz"), wav=wav.numpy())
        pins[name] = dict(source="reference ISTFTHead (export_vocoder_to_onnx.py:43-59) + conv_stft.STFT.inverse, lifted with ast", vocos_seed=vseed,
                          frames=T, hidden_seed=hseed, batch=2, row1_scale=6.0, clipped_bin_fraction=clipped, oracle_vs_reference=d,
                          absmax=wav.abs().max().item())


REAL_EXAMPLE = dict(wav="infer/examples/basic/basic_ref_en.wav", toml="infer/examples/basic/basic.toml", vocab="../../data/Emilia_ZH_EN_pinyin/vocab.txt",
                    preset="F5TTS_v1_Base", wseed=0, kw=dict(steps=16, cfg_strength=2.0, sway_sampling_coef=-1.0, seed=0))


def real_example_inputs():
    """The reference's own example (``infer/examples/basic/basic.toml:4-6`` + ``basic_ref_en.wav``) taken through the reference's own glue up to
    the sampler call: ``infer_process`` chunking (``utils_infer.py:400-402`` with ``chunk_text`` lifted out of the module), the RMS rule and the
    duration heuristic of ``infer_batch_process`` / ``_infer_basic`` (``utils_infer.py:455-493``), ``convert_char_to_pinyin`` (``model/utils.py:148-185``,
    the reference's function with ``rjieba.cut`` stubbed to return the whole string — exact for ASCII text).  The transcript gets
    the ". " ending that ``preprocess_ref_audio_text`` gives it (``utils_infer.py:367-372``).  Returns (audio [1, nw] float32 after the RMS rule, the
    int16 PCM it came from, token lists per chunk, durations per chunk, vocab map, chunks)."""
    import ast
    import re
    import wave

    ref_shims.install()
    from f5_tts.model.utils import convert_char_to_pinyin

    base = os.path.join(ref_shims.REFERENCE_SRC, "f5_tts")
    toml = open(os.path.join(base, REAL_EXAMPLE["toml"]), encoding="utf-8").read()
    ref_text = re.search(r'^ref_text = "(.*)"$', toml, re.M).group(1)
    gen_text = re.search(r'^gen_text = "(.*)"$', toml, re.M).group(1)
    if not ref_text.endswith(". "):
        ref_text = ref_text + " " if ref_text.endswith(".") else ref_text + ". "
    with wave.open(os.path.join(base, REAL_EXAMPLE["wav"])) as w:
        assert w.getframerate() == 24000 and w.getnchannels() == 1 and w.getsampwidth() == 2
        pcm = np.frombuffer(w.readframes(w.getnframes()), dtype="<i2").copy()
    audio = torch.from_numpy(pcm.astype(np.float32) / 32768.0)[None]  # torchaudio.load's normalisation of 16-bit PCM
    sr = 24000
    src = open(os.path.join(base, "infer", "utils_infer.py"), encoding="utf-8").read()
    fn = next(n for n in ast.parse(src).body if isinstance(n, ast.FunctionDef) and n.name == "chunk_text")
    env = {"re": re}
    exec(compile(ast.Module(body=[fn], type_ignores=[]), "utils_infer.py", "exec"), env)
    max_chars = int(len(ref_text.encode("utf-8")) / (audio.shape[-1] / sr) * (22 - audio.shape[-1] / sr) * 1.0)
    chunks = env["chunk_text"](gen_text, max_chars=max_chars)
    rms = torch.sqrt(torch.mean(torch.square(audio)))
    if rms < 0.1:
        audio = audio * 0.1 / rms
    ref_audio_len = audio.shape[-1] // 256
    texts, durations = [], []
    for g in chunks:
        speed = 0.3 if len(g.encode("utf-8")) < 10 else 1.0
        texts.append(convert_char_to_pinyin([ref_text + g])[0])
        durations.append(ref_audio_len + int(ref_audio_len / len(ref_text.encode("utf-8")) * len(g.encode("utf-8")) / speed))
    with open(os.path.join(base, REAL_EXAMPLE["vocab"]), encoding="utf-8") as f:
        vocab = {line[:-1]: i for i, line in enumerate(f)}
    return audio, pcm, float(rms), texts, durations, vocab, chunks, ref_text


def golden_real_example(pins):
    """north_star's "identical (ref_audio, ref_text, gen_text, seed)": the reference's CFM.sample at the full F5-TTS v1 Base size on ITS OWN
    example prompt and text (weights seeded: no checkpoint is reachable), one sample call per text chunk exactly as `infer_batch_process`
    issues them.  The fixture carries the inputs too (PCM, token ids, durations): the wav lives in /root/reference, absent on the GPU box."""
    audio, pcm, rms, texts, durations, vocab, chunks, ref_text = real_example_inputs()
    cfg = config.PRESETS[REAL_EXAMPLE["preset"]]
    assert len(vocab) == cfg.text_num_embeds
    sd = synth.synth_dit_state_dict(cfg, seed=REAL_EXAMPLE["wseed"])
    CFM, DiT, _ = ref_shims.reference_classes()
    model = CFM(transformer=DiT(**cfg.arch_kwargs()), mel_spec_kwargs=ref_shims.MEL_KW, odeint_kwargs=dict(method="euler"), vocab_char_map=vocab)
    model.load_state_dict(sd, strict=True)
    model.eval()
    from f5_tts.model.utils import list_str_to_idx

    save = dict(pcm=pcm, rms=np.float32(rms), durations=np.asarray(durations, dtype=np.int64))
    t_ref = 0.0
    for i, (text, dur) in enumerate(zip(texts, durations)):
        ids = list_str_to_idx([text], vocab)  # what CFM.sample derives from the string list (cfm.py:108-113)
        t0 = time.time()
        with torch.no_grad():
            out, traj = model.sample(cond=audio, text=[text], duration=dur, **REAL_EXAMPLE["kw"])
        t_ref += time.time() - t0
        out_o, _ = O.cfm_sample(sd, cfg, audio, ids, dur, **REAL_EXAMPLE["kw"])
        d = (out - out_o).abs().max().item()
        print(f"real_example chunk {i}: {len(text)} tokens, duration {dur} frames, reference {time.time() - t0:.1f}s, out absmax {out.abs().max().item():.3f}, "
              f"oracle-vs-reference {d:.2e}: {chunks[i]!r}")
        save[f"ids_{i}"] = ids[0].numpy().astype(np.int64)
        save[f"out_{i}"] = out.numpy()
        save[f"traj1_{i}"] = traj[1].numpy()
        pins[f"real_example_chunk{i}"] = dict(tokens=len(text), duration=dur, oracle_vs_reference_out=d, out_absmax=out.abs().max().item())
    np.savez_compressed(os.path.join(GOLD, "real_example.npz"), **save)
    pins["real_example"] = dict(case=REAL_EXAMPLE, ref_text=ref_text, chunks=chunks, prompt_rms=rms, prompt_samples=int(pcm.shape[0]), reference_seconds=t_ref,
                                source="reference CFM.sample (vocab_char_map given) on infer/examples/basic: basic_ref_en.wav + basic.toml ref_text / gen_text")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--full", action="store_true", help="also mint the full-size goldens (~1 min of CPU each)")
    ap.add_argument("--only", default="", help="comma-separated case names (full-size ones included) to mint, leaving the others alone")
    args = ap.parse_args()
    os.makedirs(GOLD, exist_ok=True)
    pins_path = os.path.join(GOLD, "pins.json")
    pins = json.load(open(pins_path)) if os.path.exists(pins_path) else {}
    torch.manual_seed(0)
    pin_stft(pins)
    pin_mel(pins)
    pin_mel_bigvgan(pins)
    golden_vocos(pins)
    golden_vocos_head(pins)
    only = set(filter(None, args.only.split(",")))
    for name, c in {**CASES, **SWEEP_CASES}.items():
        if not only or name in only:
            run_case(name, c, pins)
    for name, c in FULL_CASES.items():
        if (args.full and not only) or name in only:
            run_case(name, c, pins)
    if (args.full and not only) or "real_example" in only:
        golden_real_example(pins)
    pins["_meta"] = dict(torch=torch.__version__, reference="/root/reference (SWivid/F5-TTS v1.1.20)", generated_by="oracle/make_golden.py")
    json.dump(pins, open(pins_path, "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
