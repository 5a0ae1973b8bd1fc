"""GPU parity tests (run on an MI355X: `pytest -m gpu`).  Every compute call goes through the C ABI
(libf5hip.so); the oracle and the committed goldens (minted from the reference itself) are the checkers.

Tolerance: BASELINE.json north_star — <= 1e-3 max-abs on the generated mel vs the reference CPU path.
fp32 / fp16x3 must meet 1e-3 (held to 1e-4 / 3e-4 here: fp16x3 keeps P and V of the attention in plain fp16); plain fp16 (what the
reference itself runs on a GPU, utils_infer.py:191-199) is measured and bounded at 2e-2 — it is NOT the parity mode."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from f5_tts_amd import config, synth  # noqa: E402
from oracle import f5_oracle as O  # noqa: E402
from oracle import make_golden as MG  # noqa: E402

pytestmark = pytest.mark.gpu
GOLD = os.path.join(ROOT, "tests", "golden")
MEL_TOL = 1e-3  # north_star tolerance on the generated mel
TIGHT = 1e-4    # what the fp32 mode is actually held to
X3TOL = 3e-4    # fp16x3: split GEMM operands and attention scores, plain fp16 P.V (measured 1.1e-4 on the full-size model)
MXTOL = 3e-4    # fp16m: the fp16x3 path with the block GEMMs' two correction terms taken as one MX-fp6 product (common.h): the tiny goldens
FULL_TOL = 5e-4  # what the half-precision parity modes are held to on the FULL-SIZE goldens (real example, stress, configs[1..4] shapes):
                 # half the north_star tolerance — measured 2.4e-4 .. 3.6e-4


def gold(name):
    return np.load(os.path.join(GOLD, name + ".npz"))


@pytest.fixture(scope="module")
def engines():
    from f5_tts_amd.engine import F5HipEngine

    cache = {}

    def get(preset, wseed, vocos=False, stress=False, trained=False, sharp=None):
        key = (preset, wseed, vocos, stress, trained, sharp)
        if key not in cache:
            cfg = config.PRESETS[preset]
            sd = synth.synth_dit_state_dict(cfg, seed=wseed)
            if stress:
                sd = synth.stress_dit_state_dict(sd, cfg, seed=wseed)
            if trained:
                sd = synth.trained_like_dit_state_dict(sd, cfg, seed=wseed)
            if sharp:
                sd = synth.sharpen_attention_state_dict(sd, sharp)
            vcfg = config.VOCOS_TINY if vocos else None
            eng = F5HipEngine(cfg, vcfg, device=0)
            if vocos:
                sd = {**sd, **synth.synth_vocos_state_dict(vcfg, seed=1)}
            eng.load_state_dict(sd)
            cache[key] = eng
        return cache[key]

    yield get
    for e in cache.values():
        e.close()


def maxerr(a, b):
    return float((a.detach().cpu().float() - torch.as_tensor(b).float()).abs().max())


@pytest.mark.parametrize("name", sorted(MG.CASES))
@pytest.mark.parametrize("prec,tol", [("fp32", TIGHT), ("fp16x3", X3TOL), ("fp16m", MXTOL), ("fp16", 2e-2)])
def test_sample_matches_reference_golden(engines, name, prec, tol):
    from f5_tts_amd.engine import F5HipCFM

    c = MG.CASES[name]
    cfg, wav, text, duration, lens = MG.case_inputs(c)
    model = F5HipCFM(engines(c["preset"], c["wseed"], stress=c.get("stress", False), trained=c.get("trained", False), sharp=c.get("sharp")), precision=prec,
                     ode_method=c.get("method", "euler"))
    out, traj = model.sample(wav.cuda(), text, duration, lens=lens, **c["kw"])
    g = gold(name)
    steps = c["kw"]["steps"]
    if c["kw"].get("duplicate_test"):  # the solve starts at t_inter with proportionally fewer steps (cfm.py:205-209)
        steps = int(steps * (1 - c["kw"]["t_inter"]))
    assert tuple(out.shape) == g["out"].shape and traj.shape[0] == steps + 1
    assert tol <= MEL_TOL or prec == "fp16"
    assert maxerr(out, g["out"]) < tol
    assert maxerr(traj[1], g["traj_1"]) < tol
    assert maxerr(traj[steps // 2], g["traj_mid"]) < tol
    assert maxerr(traj[-1], g["traj_last"]) < tol


def test_mel_matches_reference_golden(engines):
    eng = engines("tiny", 1)
    wav = synth.synth_wave(256 * 37 + 100, seed=13, batch=2)
    m = eng.mel(wav.cuda(), frame_major=False)
    assert maxerr(m, gold("mel_b2")["mel"]) < 1e-4
    mf = eng.mel(wav.cuda(), frame_major=True)
    assert torch.equal(mf.permute(0, 2, 1), m)


def test_bigvgan_mel_matches_reference_golden(engines):
    """mel_spec_type="bigvgan" (reference modules.py:35-77): no centring, 384-sample reflect padding, sqrt(.+1e-9), slaney filterbank."""
    from f5_tts_amd.engine import F5HipCFM

    eng = engines("tiny", 1)
    wav = synth.synth_wave(256 * 37 + 100, seed=14, batch=2)
    g = gold("mel_bigvgan_b2")["mel"]
    m = eng.mel(wav.cuda(), mel_spec_type="bigvgan")
    assert tuple(m.shape) == g.shape == (2, 100, 37)
    assert maxerr(m, g) < 1e-4
    assert torch.equal(eng.mel(wav.cuda(), frame_major=True, mel_spec_type="bigvgan").permute(0, 2, 1), m)
    assert maxerr(F5HipCFM(eng, mel_spec_type="bigvgan").mel_spec(wav.cuda()), g) < 1e-4  # the model.mel_spec callable callers use
    for nw in (640, 1024, 256 * 20 + 255):
        w = synth.synth_wave(nw, seed=nw)
        ref = O.bigvgan_mel(w)
        out = eng.mel(w.cuda(), mel_spec_type="bigvgan")
        assert out.shape == ref.shape == (1, 100, nw // 256) and maxerr(out, ref) < 1e-4
    with pytest.raises(ValueError):
        eng.mel(torch.zeros(1, 300).cuda(), mel_spec_type="bigvgan")  # shorter than the reflect padding
    with pytest.raises(AssertionError):
        eng.mel(wav.cuda(), mel_spec_type="hifigan")  # modules.py:127


def test_bigvgan_type_sampler_and_glue(engines):
    """BASELINE configs[4] pairs E2-TTS with the BigVGAN-type mel: CFM.sample on a raw wave then uses get_bigvgan_mel_spectrogram
    (cfm.py:106-109 -> modules.py:138-151).  The generator itself is the caller's module (utils_infer.py:512-513)."""
    from f5_tts_amd import infer as I
    from f5_tts_amd.engine import F5HipCFM

    cfg = config.UNETT_TINY
    eng = engines("tiny_unett", 3)
    sd = synth.synth_dit_state_dict(cfg, seed=3)
    wav = synth.synth_wave(256 * 50, seed=6)
    text = synth.synth_text_ids(1, 30, cfg.text_num_embeds, seed=4)
    kw = dict(steps=6, cfg_strength=2.0, sway_sampling_coef=-1.0, seed=5)
    model = F5HipCFM(eng, mel_spec_type="bigvgan")
    out, _ = model.sample(wav.cuda(), text, 140, **kw)
    ref, _ = O.cfm_sample(sd, cfg, wav, text, 140, mel_spec_type="bigvgan", **kw)
    assert out.shape == ref.shape == (1, 140, 100) and maxerr(out, ref) < TIGHT
    seen = {}

    def generator(mel):  # stands in for the BigVGAN module: [b, 100, T] -> [b, 1, 256 T]
        seen["shape"] = tuple(mel.shape)
        return mel.mean(dim=1, keepdim=True).repeat_interleave(256, dim=-1)

    model.vocab_char_map = {c: i for i, c in enumerate(VOCAB)}
    wave, sr, spec = next(iter(I.infer_batch_process((wav, 24000), "some call me nature.", ["others call me mother nature."], model, generator,
                                                     mel_spec_type="bigvgan", nfe_step=4, seed=1)))
    assert sr == 24000 and seen["shape"][1] == 100 and wave.shape[0] == 256 * seen["shape"][2]


@pytest.mark.parametrize("nw", [513, 1024, 256 * 20, 256 * 20 + 255, 24000 * 3 + 17])
def test_mel_edge_lengths(engines, nw):
    eng = engines("tiny", 1)
    wav = synth.synth_wave(nw, seed=nw)
    m = eng.mel(wav.cuda())
    ref = O.vocos_mel(wav)
    assert m.shape == ref.shape == (1, 100, 1 + nw // 256)
    assert maxerr(m, ref) < 1e-4


def test_mel_too_short_raises(engines):
    with pytest.raises(ValueError):
        engines("tiny", 1).mel(torch.zeros(1, 100).cuda())


def test_text_embedding_and_velocity_taps(engines):
    from f5_tts_amd.engine import F5HipCFM

    c = MG.CASES["tiny_v1_nfe16"]
    cfg, wav, text, duration, lens = MG.case_inputs(c)
    eng = engines(c["preset"], c["wseed"])
    sd = synth.synth_dit_state_dict(cfg, seed=c["wseed"])
    out, traj = F5HipCFM(eng).sample(wav.cuda(), text, duration, **c["kw"])
    ref, rtraj, aux = O.cfm_sample(sd, cfg, wav, text, duration, return_steps=True, **c["kw"])
    n = ref.shape[1]
    assert maxerr(eng.debug_tensor(0, (1, n, cfg.text_dim)), aux["text_cond"]) < 1e-4
    assert maxerr(eng.debug_tensor(1, (1, n, cfg.text_dim)), aux["text_uncond"]) < 1e-4
    assert maxerr(eng.debug_tensor(2, (1, n, cfg.mel_dim)), aux["velocity"][-1]) < 1e-3
    # prompt frames are restored to the input mel (cfm.py:221-223)
    assert torch.equal(out[:, :61].cpu(), O.vocos_mel(wav).permute(0, 2, 1)[:, :61]) or maxerr(out[:, :61], ref[:, :61]) < 1e-4


def test_text_longer_than_frames_and_unknown_ids(engines):
    """text is curtailed to the frame count (dit.py:95) and id 0 rows are filler."""
    from f5_tts_amd.engine import F5HipCFM

    cfg = config.DIT_TINY
    eng = engines("tiny", 1)
    sd = synth.synth_dit_state_dict(cfg, seed=1)
    wav = synth.synth_wave(256 * 20, seed=1)
    text = synth.synth_text_ids(1, 90, cfg.text_num_embeds, seed=3)
    text[0, 5:9] = 0
    kw = dict(steps=4, cfg_strength=2.0, sway_sampling_coef=-1.0, seed=1)
    out, _ = F5HipCFM(eng).sample(wav.cuda(), text, 60, **kw)  # duration is raised to text_len + 1 = 91 (cfm.py:135-137)
    ref, _ = O.cfm_sample(sd, cfg, wav, text, 60, **kw)
    assert out.shape == ref.shape == (1, 91, 100)
    assert maxerr(out, ref) < TIGHT


def test_edit_mask_and_no_ref_audio(engines):
    from f5_tts_amd.engine import F5HipCFM

    cfg = config.DIT_TINY
    eng = engines("tiny", 1)
    sd = synth.synth_dit_state_dict(cfg, seed=1)
    wav = synth.synth_wave(256 * 40, seed=2)
    text = synth.synth_text_ids(1, 30, cfg.text_num_embeds, seed=3)
    edit = torch.ones(1, 41, dtype=torch.bool)
    edit[0, 10:20] = False
    kw = dict(steps=4, cfg_strength=2.0, sway_sampling_coef=-1.0, seed=1)
    out, _ = F5HipCFM(eng).sample(wav.cuda(), text, 100, edit_mask=edit, **kw)
    ref, _ = O.cfm_sample(sd, cfg, wav, text, 100, edit_mask=edit, **kw)
    assert maxerr(out, ref) < TIGHT


def test_vocos_decode_matches_oracle_golden(engines):
    from f5_tts_amd.engine import F5HipVocos

    eng = engines("tiny", 1, vocos=True)
    mel = O.vocos_mel(synth.synth_wave(256 * 80, seed=5))
    w = F5HipVocos(eng).decode(mel.cuda())
    g = gold("vocos_tiny")["wav"]
    assert w.shape == g.shape
    assert maxerr(w, g) < 1e-4 * max(1.0, float(np.abs(g).max()))
    # frame-major input layout gives the same samples
    w2 = eng.vocos_decode(mel.permute(0, 2, 1).contiguous().cuda(), channel_major=False)
    assert torch.equal(w, w2)


def test_vocos_batch_and_min_frames(engines):
    from f5_tts_amd.engine import F5HipVocos

    eng = engines("tiny", 1, vocos=True)
    vsd = synth.synth_vocos_state_dict(config.VOCOS_TINY, seed=1)
    mel = O.vocos_mel(synth.synth_wave(256 * 9, seed=8, batch=3))
    w = F5HipVocos(eng).decode(mel.cuda())
    ref = O.vocos_decode(vsd, mel, config.VOCOS_TINY.num_layers)
    assert maxerr(w, ref) < 1e-4 * max(1.0, float(ref.abs().max()))
    with pytest.raises(ValueError):
        eng.vocos_decode(mel[:, :, :1].contiguous().cuda())


def test_graph_replay_equals_eager(engines):
    from f5_tts_amd.engine import F5HipCFM

    c = MG.CASES["tiny_v1_nfe16"]
    cfg, wav, text, duration, lens = MG.case_inputs(c)
    eng = engines(c["preset"], c["wseed"])
    model = F5HipCFM(eng, precision="fp16x3")
    eager, _ = model.sample(wav.cuda(), text, duration, **c["kw"])
    eng.set_option("use_graph", 1)
    try:
        g1, _ = model.sample(wav.cuda(), text, duration, **c["kw"])
        g2, _ = model.sample(wav.cuda(), text, duration, **c["kw"])  # replay of the cached graph
    finally:
        eng.set_option("use_graph", 0)
    assert torch.equal(eager, g1) and torch.equal(g1, g2)


def test_attention_stats_follow_the_sharpness(engines):
    """Option "attn_stats" / f5hip_attention_stats (include/f5hip.h): the materialised-score attention publishes every softmax row's largest
    probability — the figure INTEGRATION.md uses to say which half-precision attention form a checkpoint needs.  Row count = cond + uncond
    x heads x frames x blocks x steps; 1 / n <= mean <= max <= 1; weights with sharper attention (q, k projections x sqrt 2: logits x 2) read
    sharper; the flash forms do not contribute; reading with reset zeroes."""
    from f5_tts_amd.engine import F5HipCFM

    cfg = config.DIT_TINY
    wav = synth.synth_wave(256 * 45, seed=21)
    text = synth.synth_text_ids(1, 50, cfg.text_num_embeds, seed=22)
    n, steps = 120, 2
    kw = dict(steps=steps, cfg_strength=2.0, sway_sampling_coef=-1.0, seed=5)
    got = {}
    for sharp in (None, 2.0 ** 0.5):
        eng = engines("tiny", 1, trained=True, sharp=sharp)
        try:
            eng.set_option("attn_stats", 1)
            eng.attention_stats(reset=True)
            F5HipCFM(eng, precision="fp32").sample(wav.cuda(), text, n, **kw)
            st = eng.attention_stats()
            F5HipCFM(eng, precision="fp16x3").sample(wav.cuda(), text, n, **kw)  # flash attention: nothing is published
            assert eng.attention_stats()["rows"] == 0
        finally:
            eng.set_option("attn_stats", 0)
        assert st["rows"] == 2 * cfg.heads * n * cfg.depth * steps, st
        assert 1.0 / n <= st["mean_max_prob"] <= st["max_prob"] <= 1.0 + 1e-6, st
        assert 0.0 <= st["frac_rows_above_half"] <= 1.0
        got[sharp] = st
    print(got)
    assert got[2.0 ** 0.5]["mean_max_prob"] > 1.05 * got[None]["mean_max_prob"], got


def test_flash_attention_equals_materialised_attention(engines):
    """The flash kernel (attention.hip) against the materialised-score fp32 attention (three GEMM/softmax launches)
    inside the same fp16x3 pipeline, at a frame count that is neither a multiple of the 64-key tile nor of the
    128-row query block."""
    from f5_tts_amd.engine import F5HipCFM

    cfg = config.DIT_TINY
    eng = engines("tiny", 1)
    wav = synth.synth_wave(256 * 45, seed=21)
    text = synth.synth_text_ids(1, 50, cfg.text_num_embeds, seed=22)
    kw = dict(steps=4, cfg_strength=2.0, sway_sampling_coef=-1.0, seed=5)
    model = F5HipCFM(eng, precision="fp16x3")
    try:
        eng.set_option("attn_impl", 1)
        exact, _ = model.sample(wav.cuda(), text, 333, **kw)
        eng.set_option("attn_impl", 2)  # every attention operand hi/lo split
        flash, _ = model.sample(wav.cuda(), text, 333, **kw)
        eng.set_option("attn_impl", 4)  # split q/k, plain fp16 P/V (the default of round 1)
        mixed, _ = model.sample(wav.cuda(), text, 333, **kw)
        eng.set_option("attn_impl", 0)  # default since round 5: scores from fp16 hi.hi + MX-fp6 corrections, plain fp16 P, V
        default, _ = model.sample(wav.cuda(), text, 333, **kw)
        eng.set_option("attn_impl", 3)  # plain fp16 q, k, P, V (the default of rounds 2-4)
        plain, _ = model.sample(wav.cuda(), text, 333, **kw)
    finally:
        eng.set_option("attn_impl", 0)
    assert maxerr(flash, exact.cpu()) < 5e-5
    assert maxerr(mixed, exact.cpu()) < 2e-4
    assert maxerr(default, exact.cpu()) < 2e-4
    assert maxerr(plain, exact.cpu()) < 3e-4


@pytest.mark.parametrize("prec,tol", [("fp32", TIGHT), ("fp16x3", X3TOL)])
def test_key_padding_mask_ragged_batch(prec, tol):
    """attn_mask_enabled=True (reference modules.py:513-516): keys beyond each utterance's duration are masked —
    the kvlen path of both attention implementations, on a ragged batch."""
    from dataclasses import replace

    from f5_tts_amd.engine import F5HipCFM, F5HipEngine

    cfg = replace(config.DIT_TINY, attn_mask_enabled=True)
    sd = synth.synth_dit_state_dict(cfg, seed=4)
    eng = F5HipEngine(cfg, None, device=0)
    eng.load_state_dict(sd)
    try:
        wav = synth.synth_wave(256 * 40, seed=31, batch=3)
        text = synth.synth_text_ids(3, 30, cfg.text_num_embeds, seed=32)
        text[1, 20:] = -1
        duration = torch.tensor([150, 97, 131])
        lens = torch.tensor([41, 30, 35])
        kw = dict(steps=4, cfg_strength=2.0, sway_sampling_coef=-1.0, seed=3)
        out, _ = F5HipCFM(eng, precision=prec).sample(wav.cuda(), text, duration, lens=lens, **kw)
        ref, _ = O.cfm_sample(sd, cfg, wav, text, duration, lens=lens, **kw)
        for b in range(3):  # rows beyond an utterance's duration are padding in both
            d = int(duration[b])
            assert maxerr(out[b, :d], ref[b, :d]) < tol
    finally:
        eng.close()


def test_determinism_and_batch_consistency(engines):
    """Same inputs twice -> bit-identical; a fixed-length batch of identical utterances gives identical rows."""
    from f5_tts_amd.engine import F5HipCFM

    cfg = config.DIT_TINY
    eng = engines("tiny", 1)
    wav = synth.synth_wave(256 * 30, seed=9).repeat(3, 1)
    text = synth.synth_text_ids(1, 20, cfg.text_num_embeds, seed=6).repeat(3, 1)
    kw = dict(steps=4, cfg_strength=2.0, sway_sampling_coef=-1.0, seed=0)
    a, _ = F5HipCFM(eng).sample(wav.cuda(), text, 96, **kw)
    b, _ = F5HipCFM(eng).sample(wav.cuda(), text, 96, **kw)
    assert torch.equal(a, b)
    assert torch.equal(a[0], a[1]) and torch.equal(a[1], a[2])


def test_full_size_base_config1_golden():
    """BASELINE.json configs[0]/[1]: F5-TTS Base, 5 s ref + 10 s gen, NFE 16 — against the golden minted by
    running the reference's own CFM.sample on CPU (53 s there)."""
    from f5_tts_amd.engine import F5HipCFM, F5HipEngine

    c = MG.FULL_CASES["base_v1_cfg1"]
    cfg, wav, text, duration, lens = MG.case_inputs(c)
    eng = F5HipEngine(cfg, None, device=0)
    eng.load_state_dict(synth.synth_dit_state_dict(cfg, seed=c["wseed"]))
    g = gold("base_v1_cfg1")
    try:
        for prec, tol in (("fp16m", FULL_TOL), ("fp16x3", FULL_TOL), ("fp32", TIGHT)):
            out, traj = F5HipCFM(eng, precision=prec).sample(wav.cuda(), text, duration, **c["kw"])
            e = maxerr(out[:, 468:], g["out"][:, 468:])  # the generated frames (utils_infer.py:507-509 slice)
            print(f"full-size {prec}: generated-mel max-abs {e:.2e}")
            assert e < tol
            assert maxerr(traj[1], g["traj_1"]) < tol
    finally:
        eng.close()


def test_reference_example_prompt_and_text_golden():
    """north_star "identical (ref_audio, ref_text, gen_text, seed)": the reference's OWN example — infer/examples/basic/basic_ref_en.wav with
    basic.toml's transcript and text — taken through its own glue (chunking, RMS rule, duration heuristic, convert_char_to_pinyin with the
    ASCII branch, list_str_to_idx over its Emilia vocab) and its CFM.sample at the full F5-TTS v1 Base size, one call per text chunk
    (oracle/make_golden.py::golden_real_example; 148 s + 106 s of reference CPU time).  A real recording instead of 0.1 N(0,1) noise, real
    token statistics, and sequence lengths the synthetic cases do not have (1829 and 743 frames)."""
    from f5_tts_amd.engine import F5HipCFM, F5HipEngine

    g = gold("real_example")
    kw = MG.REAL_EXAMPLE["kw"]
    cfg = config.PRESETS[MG.REAL_EXAMPLE["preset"]]
    audio = torch.from_numpy(g["pcm"].astype(np.float32) / 32768.0)[None]
    rms = torch.sqrt(torch.mean(torch.square(audio)))
    assert abs(float(rms) - float(g["rms"])) < 1e-6
    if rms < 0.1:
        audio = audio * 0.1 / rms  # utils_infer.py:455-457
    eng = F5HipEngine(cfg, None, device=0)
    eng.load_state_dict(synth.synth_dit_state_dict(cfg, seed=MG.REAL_EXAMPLE["wseed"]))
    try:
        prompt_frames = audio.shape[-1] // 256
        for i, dur in enumerate(g["durations"].tolist()):
            ids = torch.from_numpy(g[f"ids_{i}"])[None]
            for prec, tol in (("fp16m", FULL_TOL), ("fp16x3", FULL_TOL), ("fp32", TIGHT)):
                out, traj = F5HipCFM(eng, precision=prec).sample(audio.cuda(), ids, dur, **kw)
                assert tuple(out.shape) == g[f"out_{i}"].shape == (1, dur, 100)
                e = maxerr(out[:, prompt_frames:], g[f"out_{i}"][:, prompt_frames:])
                print(f"reference example chunk {i} ({dur} frames) {prec}: generated-mel max-abs {e:.2e}")
                assert e < tol
                assert maxerr(traj[1], g[f"traj1_{i}"]) < tol
    finally:
        eng.close()


def test_dynamic_range_stress_golden_full_size():
    """VERDICT r02 "weak" 1: every other golden uses N(0, 1/sqrt(in)) matrices and a 0.1 N(0,1) prompt.  This one is the configs[1] case with
    per-tensor weight scales spread over three decades (to_v / to_out, to_q / to_k, FF1 / FF2 scaled by s and 1/s, s log-uniform in
    [1e-3, 1]), 2 % outlier AdaLN channels (x 4-12) and a prompt that clips (synth.stress_dit_state_dict, synth.synth_loud_wave), minted by
    the reference's own CFM.sample.  The fp16 hi/lo split must hold the north_star tolerance on it."""
    from f5_tts_amd.engine import F5HipCFM, F5HipEngine

    c = MG.FULL_CASES["base_v1_stress"]
    cfg, wav, text, duration, lens = MG.case_inputs(c)
    eng = F5HipEngine(cfg, None, device=0)
    eng.load_state_dict(MG.case_weights(c))
    g = gold("base_v1_stress")
    try:
        for prec, tol in (("fp16m", FULL_TOL), ("fp16x3", FULL_TOL), ("fp32", TIGHT)):
            out, traj = F5HipCFM(eng, precision=prec).sample(wav.cuda(), text, duration, **c["kw"])
            e = maxerr(out[:, 468:], g["out"][:, 468:])
            print(f"dynamic-range stress, full size, {prec}: generated-mel max-abs {e:.2e} (|mel| max {np.abs(g['out']).max():.2f})")
            assert e < tol
            assert maxerr(traj[1], g["traj_1"]) < tol
    finally:
        eng.close()


def test_trained_like_weights_golden_full_size():
    """no trained checkpoint is reachable from the build container, and every other golden multiplies Gaussian matrices.
    This one is the configs[1] case on weights with a checkpoint's STATISTICS (synth.trained_like_dit_state_dict: Student-t entries — a few at
    6-10 sigma —, per-row and per-column log-normal gains, LayerNorm / GRN / bias parameters far from their initial values), minted by the
    reference's own CFM.sample: the margin of the fp16 + MX-fp6 operand scheme (block scales per 16 values, weight rows conditioned by
    powers of two) where a block holds an outlier.
    It is also the golden that ended the plain-fp16 attention scores of rounds 2-4: 1.14e-3 with them (`attn_impl` 3, asserted below to stay
    the worse choice), 4.9e-4 / 3.4e-4 (fp16m / fp16x3) with the MX-corrected scores that are the default since round 5
    (profiles/r05j_attn_precision_*.log).  Tolerance of this golden: 7e-4 — 0.7 of north_star's 1e-3; its largest logits are ~2x those of
    the Gaussian goldens and the error of the scores grows with them."""
    from f5_tts_amd.engine import F5HipCFM, F5HipEngine

    TRAINED_TOL = 7e-4
    c = MG.FULL_CASES["base_v1_trained_like"]
    cfg, wav, text, duration, lens = MG.case_inputs(c)
    eng = F5HipEngine(cfg, None, device=0)
    eng.load_state_dict(MG.case_weights(c))
    g = gold("base_v1_trained_like")
    try:
        errs = {}
        for prec, tol in (("fp16m", TRAINED_TOL), ("fp16x3", FULL_TOL), ("fp32", TIGHT)):
            out, traj = F5HipCFM(eng, precision=prec).sample(wav.cuda(), text, duration, **c["kw"])
            e = errs[prec] = maxerr(out[:, 468:], g["out"][:, 468:])
            print(f"trained-like weight statistics, full size, {prec}: generated-mel max-abs {e:.2e} (|mel| max {np.abs(g['out']).max():.2f})")
            assert e < tol
            assert maxerr(traj[1], g["traj_1"]) < tol
        eng.set_option("attn_impl", 3)  # plain fp16 q, k: what rounds 2-4 shipped
        out, _ = F5HipCFM(eng, precision="fp16m").sample(wav.cuda(), text, duration, **c["kw"])
        e3 = maxerr(out[:, 468:], g["out"][:, 468:])
        print(f"trained-like weight statistics, full size, fp16m with plain fp16 scores: {e3:.2e}")
        assert e3 > 1.5 * errs["fp16m"]
    finally:
        eng.close()


def test_full_size_e2_unett_golden():
    """BASELINE.json configs[4] backbone: E2-TTS Base (UNetT, 333 M parameters) at full size against the golden minted by running the
    reference's own UNetT through CFM.sample on CPU (54 s there)."""
    from f5_tts_amd.engine import F5HipCFM, F5HipEngine

    c = MG.FULL_CASES["e2_base_cfg5"]
    cfg, wav, text, duration, lens = MG.case_inputs(c)
    eng = F5HipEngine(cfg, None, device=0)
    eng.load_state_dict(synth.synth_dit_state_dict(cfg, seed=c["wseed"]))
    g = gold("e2_base_cfg5")
    try:
        for prec, tol in (("fp16m", FULL_TOL), ("fp16x3", FULL_TOL), ("fp32", TIGHT)):
            out, traj = F5HipCFM(eng, precision=prec).sample(wav.cuda(), text, duration, **c["kw"])
            e = maxerr(out[:, 468:], g["out"][:, 468:])
            print(f"full-size E2 {prec}: generated-mel max-abs {e:.2e}")
            assert e < tol
            assert maxerr(traj[1], g["traj_1"]) < tol
    finally:
        eng.close()


def test_full_size_v0_yaml_golden():
    """The v0 configuration at full size (src/f5_tts/configs/F5TTS_Base.yaml:20-46: pe_attn_head 1 — rope on the first head only —,
    text_mask_padding False) on the configs[1] inputs, against the golden minted by the reference's own DiT (VERDICT r03 "missing" 4)."""
    from f5_tts_amd.engine import F5HipCFM, F5HipEngine

    c = MG.FULL_CASES["base_v0_cfg1"]
    cfg, wav, text, duration, lens = MG.case_inputs(c)
    assert cfg.pe_attn_head == 1 and not cfg.text_mask_padding
    eng = F5HipEngine(cfg, None, device=0)
    eng.load_state_dict(synth.synth_dit_state_dict(cfg, seed=c["wseed"]))
    g = gold("base_v0_cfg1")
    try:
        for prec, tol in (("fp16m", FULL_TOL), ("fp16x3", FULL_TOL), ("fp32", TIGHT)):
            out, traj = F5HipCFM(eng, precision=prec).sample(wav.cuda(), text, duration, **c["kw"])
            e = maxerr(out[:, 468:], g["out"][:, 468:])
            print(f"full-size v0 yaml {prec}: generated-mel max-abs {e:.2e}")
            assert e < tol and maxerr(traj[1], g["traj_1"]) < tol
    finally:
        eng.close()


def test_configs4_shaped_batch_golden():
    """BASELINE.json configs[4] in its own shape (E2-TTS Base, batch 8, NFE 16): two distinct utterances against the golden minted by the
    reference's own UNetT through CFM.sample (e2_base_cfg5_b2), then the same two as rows 0..1 and 6..7 of the B = 8 batch that
    bench.py --model E2TTS_Base --batch 8 times (fixed-length batches have no cross-row coupling)."""
    from f5_tts_amd.engine import F5HipCFM, F5HipEngine

    c = MG.FULL_CASES["e2_base_cfg5_b2"]
    cfg, wav, text, duration, lens = MG.case_inputs(c)
    eng = F5HipEngine(cfg, None, device=0)
    eng.load_state_dict(synth.synth_dit_state_dict(cfg, seed=c["wseed"]))
    g = gold("e2_base_cfg5_b2")
    try:
        for prec in ("fp16m", "fp16x3"):
            model = F5HipCFM(eng, precision=prec)
            out, traj = model.sample(wav.cuda(), text, duration, **c["kw"])
            e = maxerr(out[:, 468:], g["out"][:, 468:])
            print(f"configs[4]-shaped E2 B=2 NFE=16 {prec}: generated-mel max-abs {e:.2e}")
            assert e < FULL_TOL and maxerr(traj[1], g["traj_1"]) < FULL_TOL
            out8, _ = model.sample(wav.repeat(4, 1).cuda(), text.repeat(4, 1), duration, **c["kw"])
            e8 = max(maxerr(out8[:2, 468:], g["out"][:, 468:]), maxerr(out8[6:, 468:], g["out"][:, 468:]))
            print(f"the same utterances as rows 0..1 and 6..7 of B=8 ({prec}): {e8:.2e}")
            assert e8 < FULL_TOL
    finally:
        eng.close()


@pytest.mark.parametrize("name", ["small_v1", "small_e2"])
def test_small_models_golden(name):
    """The shipped Small configs (dim 768 = 12 heads, 48 channels per conv-position group; configs/F5TTS_v1_Small.yaml,
    E2TTS_Small.yaml) at full width and depth against goldens minted from the reference's own DiT / UNetT."""
    from f5_tts_amd.engine import F5HipCFM, F5HipEngine

    c = MG.FULL_CASES[name]
    cfg, wav, text, duration, lens = MG.case_inputs(c)
    eng = F5HipEngine(cfg, None, device=0)
    eng.load_state_dict(synth.synth_dit_state_dict(cfg, seed=c["wseed"]))
    g = gold(name)
    try:
        for prec, tol in (("fp32", TIGHT), ("fp16x3", 3.5e-4), ("fp16m", 4e-4)):  # (small_v1: 2.8 .. 3.1e-4 in either mode, moving with every change of a summation order)
            out, traj = F5HipCFM(eng, precision=prec).sample(wav.cuda(), text, duration, **c["kw"])
            e = maxerr(out, g["out"])
            print(f"{name} {prec}: max-abs {e:.2e}")
            assert e < tol and maxerr(traj[1], g["traj_1"]) < tol
    finally:
        eng.close()


# written without GPU minutes: executed on the CPU shim (tests/test_parity_on_shim.py) and under a real gloo broadcast (tests/test_dist_cpu.py);


def test_weight_blob_receiver_equals_the_rank_that_loaded(engines):
    """The N>1 path of bench.py / dist.broadcast_engine_weights from the receiver's side, in one process: a context that never saw a
    state dict gets the packed blob and the sender's loaded mask, finalises, and must produce the sender's bits — with a checkpoint
    that carries an OPTIONAL buffer (rotary inv_freq, deliberately not the recomputed values) so the mask matters."""
    from f5_tts_amd import binding
    from f5_tts_amd.engine import F5HipCFM, F5HipEngine

    cfg, vcfg = config.DIT_TINY, config.VOCOS_TINY
    sd = {**synth.synth_dit_state_dict(cfg, seed=5), **synth.synth_vocos_state_dict(vcfg, seed=5)}
    half = cfg.dim_head // 2
    sd["transformer.rotary_embed.inv_freq"] = 1.0 / (9000.0 ** (torch.arange(half).float() / half))
    a, b, c = (F5HipEngine(cfg, vcfg, device=0) for _ in range(3))
    try:
        a.load_state_dict(sd, finalize=False)  # what rank 0 does in bench.py
        blob = a.weight_blob()
        assert blob.data_ptr() == a.weight_blob().data_ptr() and blob.dtype == torch.float32  # an alias of the context's memory, not a copy
        names = [n for n, _, _ in a.tensor_table()]
        mask = a.loaded_mask()
        assert mask.numel() == len(names) and mask[names.index("transformer.rotary_embed.inv_freq")] == 1
        assert int(mask.sum()) == len(sd) and int(b.loaded_mask().sum()) == 0
        with pytest.raises(binding.F5HipError):
            b.finalize()  # nothing received yet: fails loudly instead of running on an empty blob
        for recv in (b, c):
            recv.weight_blob().copy_(blob)  # the broadcast
        b.set_loaded_mask(mask)
        c.mark_all_loaded()  # the pre-mask protocol: optional buffers are recomputed on the receiver
        for e in (a, b, c):
            e.finalize()
        wav = synth.synth_wave(256 * 30, seed=2, batch=2).cuda()
        text = synth.synth_text_ids(2, 25, cfg.text_num_embeds, seed=4)
        kw = dict(steps=4, cfg_strength=2.0, sway_sampling_coef=-1.0, seed=3)
        outs = [F5HipCFM(e, precision="fp16x3").sample(wav, text, 90, **kw)[0] for e in (a, b, c)]
        assert torch.equal(outs[0], outs[1])
        assert not torch.equal(outs[0], outs[2])  # c rotated with the recomputed base-10000 table: the mask is what carries the buffer
        gen = outs[0][:, 30:, :].contiguous()
        assert torch.equal(a.vocos_decode(gen, channel_major=False), b.vocos_decode(gen, channel_major=False))
        with pytest.raises(ValueError):
            b.set_loaded_mask(mask[:-1])
    finally:
        for e in (a, b, c):
            e.close()


def test_invalid_arguments_raise(engines):
    from f5_tts_amd.engine import F5HipCFM

    eng = engines("tiny", 1)
    model = F5HipCFM(eng)
    wav = synth.synth_wave(256 * 10, seed=1)
    text = synth.synth_text_ids(1, 5, config.DIT_TINY.text_num_embeds, seed=1)
    with pytest.raises(ValueError):
        F5HipCFM(eng, ode_method="rk4")  # only the fixed-grid euler / midpoint solvers exist
    bad = text.clone()
    bad[0, 0] = 10_000
    with pytest.raises(ValueError):
        model.sample(wav.cuda(), bad, 40, steps=4, cfg_strength=2.0)


@pytest.mark.gpu
def test_fp16m_runs_as_fp16x3_where_the_mx_tiles_do_not_apply(engines):
    """fp16m has no generic-kernel fallback inside a launch, so the engine must decide per CALL (csrc/api.cpp mx_call): sequences shorter
    than 8 tokens — which the fused q|k|v launch of the pipelined kernel does not take (duration 6 / 7 raised 'launch_gemm_qkv
    failed') — run in fp16x3, bit for bit, instead of failing.  (The tuning knobs that take launches away from the MX tiles are read once
    per process: tests/test_pp_gemm_shim.py runs them in child processes.)"""
    from f5_tts_amd.engine import F5HipCFM

    eng = engines("tiny", 1)
    for dur in (6, 7):
        wav = synth.synth_wave(256 * 3, seed=1)
        text = synth.synth_text_ids(1, 3, config.DIT_TINY.text_num_embeds, seed=1)
        kw = dict(steps=4, cfg_strength=2.0, sway_sampling_coef=-1.0, seed=3)
        a, _ = F5HipCFM(eng, precision="fp16m").sample(wav.cuda(), text, dur, **kw)
        b, _ = F5HipCFM(eng, precision="fp16x3").sample(wav.cuda(), text, dur, **kw)
        assert torch.equal(a.cpu(), b.cpu()) and bool(torch.isfinite(a).all())


# ---- L4 glue (f5-tts_amd/infer.py): checkpoint formats and infer_process -------------------------------------------------------
VOCAB = [" "] + [chr(c) for c in range(33, 127)]  # id 0 = space (unknown), printable ASCII after it


def _write_vocab(tmp_path):
    p = tmp_path / "vocab.txt"
    p.write_text("".join(v + "\n" for v in VOCAB), encoding="utf-8")
    return str(p)


def test_checkpoint_formats_load_like_the_reference(tmp_path):
    """EMA .pt ({'ema_model_state_dict': {'ema_model.<key>', 'initted', 'step'}}), plain .pt and EMA .safetensors checkpoints
    (reference load_checkpoint, utils_infer.py:201-227) all give the engine the same weights as loading the state dict directly."""
    from dataclasses import replace

    from safetensors.torch import save_file

    from f5_tts_amd import infer as I

    cfg = replace(config.DIT_TINY, text_num_embeds=len(VOCAB))
    sd = synth.synth_dit_state_dict(cfg, seed=5)
    vocab = _write_vocab(tmp_path)
    ema = {"ema_model." + k: v for k, v in sd.items()}
    ema.update({"initted": torch.tensor(True), "step": torch.tensor(123),
                "ema_model.mel_spec.mel_stft.mel_scale.fb": torch.zeros(3), "ema_model.mel_spec.mel_stft.spectrogram.window": torch.zeros(3)})
    torch.save({"ema_model_state_dict": ema}, tmp_path / "ema.pt")
    torch.save({"model_state_dict": dict(sd)}, tmp_path / "plain.pt")
    save_file({k: v.contiguous() for k, v in ema.items() if k not in ("initted", "step")}, str(tmp_path / "ema.safetensors"))

    wav = synth.synth_wave(256 * 30, seed=2)
    kw = dict(steps=4, cfg_strength=2.0, sway_sampling_coef=-1.0, seed=3)
    text = ["hello world. this is a test"]
    outs = []
    for path, use_ema in ((None, True), ("ema.pt", True), ("plain.pt", False), ("ema.safetensors", True)):
        model = I.load_model(cfg, str(tmp_path / path) if path else None, vocab_file=vocab, use_ema=use_ema, device=0, precision="fp32",
                             vocos_cfg=None, state_dict=sd if path is None else None)
        out, _ = model.sample(wav.cuda(), text, 80, **kw)
        outs.append(out.cpu())
        model.engine.close()
    for o in outs[1:]:
        assert torch.equal(o, outs[0])
    ids = torch.tensor([[VOCAB.index(ch) for ch in text[0]]])
    ref, _ = O.cfm_sample(sd, cfg, wav, ids, 80, **kw)
    assert maxerr(outs[0], ref) < TIGHT


def test_infer_process_matches_oracle_glue(tmp_path):
    """infer_process (chunking, per-chunk duration heuristic, 468-vs-469 slice, vocoder, RMS restore, cross-fade) against the oracle
    restatement of _infer_basic (utils_infer.py:477-520) per chunk."""
    from dataclasses import replace

    from f5_tts_amd import infer as I

    cfg = replace(config.DIT_TINY, text_num_embeds=len(VOCAB))
    vcfg = config.VOCOS_TINY
    sd, vsd = synth.synth_dit_state_dict(cfg, seed=6), synth.synth_vocos_state_dict(vcfg, seed=2)
    model = I.load_model(cfg, None, vocab_file=_write_vocab(tmp_path), device=0, precision="fp32", vocos_cfg=vcfg, state_dict=sd)
    voc = I.load_vocoder("vocos", engine=model.engine, state_dict=vsd)
    try:
        audio = 0.5 * synth.synth_wave(24000 * 2, seed=4)  # rms 0.05 < target 0.1: exercises the normalise / restore branch
        ref_text = "some call me nature."
        gen_text = ("others call me mother nature. i have been here for ages, and i will be here after. respect me; or not. "
                    "i have fed species greater than you, and i have starved species greater than you. my oceans, my soil, "
                    "my flowing streams, my forests: they all can take you, or leave you. ok.")
        kw = dict(nfe_step=4, cfg_strength=2.0, sway_sampling_coef=-1.0, seed=11)
        wave_out, sr, spec = I.infer_process((audio, 24000), ref_text, gen_text, model, voc, show_info=lambda *_: None, **kw)
        # oracle side: same chunking + heuristic, CPU numerics
        rms = float(torch.sqrt(torch.mean(audio ** 2)))
        a = audio * 0.1 / rms
        rt = ref_text + " "
        seconds = audio.shape[-1] / 24000
        chunks = I.chunk_text(gen_text, max_chars=int(len(ref_text.encode()) / seconds * (22 - seconds) * 1.0))
        assert len(chunks) >= 2
        waves, specs = [], []
        for g in chunks:
            chars = I.convert_char_to_pinyin([rt + g])[0]
            ids = torch.tensor([[VOCAB.index(ch) if ch in VOCAB else 0 for ch in chars]])
            ref_len = a.shape[-1] // 256
            speed = 0.3 if len(g.encode()) < 10 else 1.0
            duration = ref_len + int(ref_len / len(rt.encode()) * len(g.encode()) / speed)
            w, out = O.infer_basic(sd, cfg, vsd, vcfg.num_layers, a, ids, duration, steps=4, cfg_strength=2.0, sway_sampling_coef=-1.0, seed=11)
            waves.append((w * rms / 0.1).squeeze().numpy())
            specs.append(out[0, ref_len:].T.numpy())
        ref_wave = I.cross_fade_concat(waves, 0.15)
        assert sr == 24000 and wave_out.shape == ref_wave.shape and spec.shape == np.concatenate(specs, axis=1).shape
        assert np.abs(spec - np.concatenate(specs, axis=1)).max() < TIGHT
        assert np.abs(wave_out - ref_wave).max() < 1e-4 * max(1.0, np.abs(ref_wave).max())
    finally:
        model.engine.close()


# ---- size-independent properties at BASELINE.json's full sizes, long sequences, degenerate text --------------------------------
def test_speech_edit_matches_oracle(engines):
    """speech_edit.py as a function: mel of the original, zero frames + edit_mask for the edited spans, sample, vocode — against the
    oracle run on the same condition."""
    from f5_tts_amd import infer as I
    from f5_tts_amd.engine import F5HipCFM, F5HipVocos

    cfg, vcfg = config.DIT_TINY, config.VOCOS_TINY
    eng = engines("tiny", 1, vocos=True)
    sd, vsd = synth.synth_dit_state_dict(cfg, seed=1), synth.synth_vocos_state_dict(vcfg, seed=1)
    model = F5HipCFM(eng, vocab_char_map={c: i for i, c in enumerate(VOCAB)})
    wav = synth.synth_wave(24000 * 2, seed=9)
    parts, fix = [[0.4, 0.8], [1.2, 1.5]], [0.5, 0.2]
    text = "some call me nature, others do not."
    wave_out, mel_out = I.speech_edit(model, F5HipVocos(eng), wav, 24000, text, parts, fix, nfe_step=6, seed=3)
    rms = torch.sqrt(torch.mean(torch.square(wav)))
    wav_n = wav * 0.1 / rms if rms < 0.1 else wav  # the script's RMS normalisation (speech_edit.py:141-143)
    cond, mask = I.build_edit_condition(O.vocos_mel(wav_n).permute(0, 2, 1), parts, fix)
    ids = torch.tensor([[model.vocab_char_map.get(c, 0) for c in I.convert_char_to_pinyin([text])[0]]])
    ref, _ = O.cfm_sample(sd, cfg, cond, ids, cond.shape[1], steps=6, cfg_strength=2.0, sway_sampling_coef=-1.0, seed=3, edit_mask=mask)
    n = cond.shape[1]  # CFM.sample raises the duration to lens + 1 (cfm.py:135-137), exactly as it does for the reference script
    assert ref.shape == (1, n + 1, 100) and mel_out.shape == (1, 100, n + 1) and maxerr(mel_out.permute(0, 2, 1), ref) < TIGHT
    assert maxerr(mel_out.permute(0, 2, 1)[:, :n][mask], cond[mask]) < 1e-4  # kept frames are the original mel
    wref = O.vocos_decode(vsd, ref.permute(0, 2, 1), vcfg.num_layers)
    if rms < 0.1:
        wref = wref * rms / 0.1
    assert wave_out.shape == wref.shape and maxerr(wave_out, wref) < 1e-3 * max(1.0, float(wref.abs().max()))


def test_f5tts_api_class(tmp_path):
    """The reference's one-class API (api.py:23-149) over the engine: preprocess the prompt file, chunk, sample, vocode, cross-fade,
    export; the same seed gives the same wave."""
    import wave as wavemod

    from f5_tts_amd import api as A

    cfg = config.DIT_TINY
    vocab = _write_vocab(tmp_path)
    from dataclasses import replace

    cfg = replace(cfg, text_num_embeds=len(VOCAB))
    sd = synth.synth_dit_state_dict(cfg, seed=5)
    vsd = synth.synth_vocos_state_dict(config.VOCOS_TINY, seed=1)
    A.PRESETS["tiny_vocab"] = cfg
    try:
        tts = A.F5TTS(model="tiny_vocab", vocab_file=vocab, device=0, precision="fp32", vocos_cfg=config.VOCOS_TINY, state_dict=sd,
                      vocoder_state_dict=vsd)
    finally:
        A.PRESETS.pop("tiny_vocab", None)
    ref = tmp_path / "ref.wav"
    pcm = (synth.synth_wave(24000 * 2, seed=4)[0].numpy() * 32767).astype("<i2")
    with wavemod.open(str(ref), "wb") as w:
        w.setnchannels(1); w.setsampwidth(2); w.setframerate(24000); w.writeframes(pcm.tobytes())
    kw = dict(ref_file=str(ref), ref_text="some call me nature", gen_text="others call me mother nature.", nfe_step=4, show_info=lambda m: None)
    out_file = tmp_path / "gen.wav"
    w1, sr, spec = tts.infer(seed=11, file_wave=str(out_file), **kw)
    w2, _, _ = tts.infer(seed=11, **kw)
    assert sr == 24000 and w1.ndim == 1 and np.isfinite(w1).all() and spec.shape[0] == 100
    assert np.array_equal(w1, w2) and tts.seed == 11
    with wavemod.open(str(out_file), "rb") as w:
        assert w.getframerate() == 24000 and w.getnframes() == w1.shape[0]
    tts.ema_model.engine.close()


def test_full_size_batch_rows_equal_single_utterance():
    """F5-TTS Base at full size: a fixed-length batch of identical utterances (packed cond+uncond launches, M = 2*B*N rows) must give
    every row the result of the B=1 call — same arithmetic, different schedules (since round 2: B = 1 is one packed chain of 2 N rows,
    B = 4 two concurrent CFG chains of 4 N rows each) and tile heuristics."""
    from f5_tts_amd.engine import F5HipCFM, F5HipEngine

    c = MG.FULL_CASES["base_v1_cfg1"]
    cfg, wav, text, duration, lens = MG.case_inputs(c)
    eng = F5HipEngine(cfg, None, device=0)
    eng.load_state_dict(synth.synth_dit_state_dict(cfg, seed=c["wseed"]))
    try:
        model = F5HipCFM(eng, precision="fp16x3")
        kw = dict(c["kw"], steps=4)
        one, _ = model.sample(wav.cuda(), text, duration, **kw)
        many, _ = model.sample(wav.repeat(4, 1).cuda(), text.repeat(4, 1), duration, **kw)
        assert many.shape[0] == 4
        # not bit-equal by construction: the B = 1 launches run other tiles (the k-split one sums in another order, the epilogues contract
        # their FMAs differently), and 22 blocks x 4 steps amplify those last-bit differences to ~2e-4 (measured 2.0e-4 / 2.2e-4 with / without
        # the k-split tile).  The race this test once found showed as 3e-2 .. 6e-2; identical rows must still be identical bit for bit.
        for b in range(4):
            assert maxerr(many[b], one[0].cpu()) < 5e-4
        assert torch.equal(many[0], many[3]) and torch.equal(many[1], many[2])
        again, _ = model.sample(wav.cuda(), text, duration, **kw)
        assert torch.equal(again, one)
    finally:
        eng.close()


def test_long_sequence_matches_oracle(engines):
    """4000 frames (42 s): more than 31 query blocks / 62 key tiles per head, offsets well past 2^16 rows."""
    from f5_tts_amd.engine import F5HipCFM

    cfg = config.DIT_TINY
    eng = engines("tiny", 1)
    sd = synth.synth_dit_state_dict(cfg, seed=1)
    wav = synth.synth_wave(256 * 300, seed=12)
    text = synth.synth_text_ids(1, 400, cfg.text_num_embeds, seed=13)
    kw = dict(steps=2, cfg_strength=2.0, sway_sampling_coef=-1.0, seed=2)
    out, _ = F5HipCFM(eng, precision="fp16x3").sample(wav.cuda(), text, 4000, **kw)
    ref, _ = O.cfm_sample(sd, cfg, wav, text, 4000, **kw)
    assert out.shape == ref.shape == (1, 4000, 100)
    assert maxerr(out, ref) < X3TOL


def test_all_padding_text_and_single_frame_prompt(engines):
    """text made only of batch padding (-1): every position is the filler token; prompt of the minimum length the mel front-end takes."""
    from f5_tts_amd.engine import F5HipCFM

    cfg = config.DIT_TINY
    eng = engines("tiny", 1)
    sd = synth.synth_dit_state_dict(cfg, seed=1)
    wav = synth.synth_wave(513, seed=14)  # 3 mel frames
    text = torch.full((1, 6), -1, dtype=torch.long)
    kw = dict(steps=3, cfg_strength=2.0, sway_sampling_coef=-1.0, seed=4)
    out, _ = F5HipCFM(eng).sample(wav.cuda(), text, 50, **kw)
    ref, _ = O.cfm_sample(sd, cfg, wav, text, 50, **kw)
    assert out.shape == ref.shape == (1, 50, 100)
    assert maxerr(out, ref) < TIGHT


# ---- round 2: the parity holes VERDICT r01 listed -------------------------------------------------------------------------------------
def test_vocos_full_size_on_the_benchmarked_mel():
    """F5HipVocos at VOCOS_MEL_24K (dim 512 x 8 layers, 1026-wide head: the vocoder bench.py times) on the 938 generated frames of the
    reference-minted golden base_v1_cfg1, against the oracle.  The oracle's backbone is a restatement (the `vocos` package is absent:
    parity unpinned for the backbone); its iSTFT is pinned by the reference's conv-iSTFT (tests/test_oracle.py) and the HIP iSTFT
    directly by the next test."""
    from f5_tts_amd.engine import F5HipEngine, F5HipVocos

    vcfg = config.VOCOS_MEL_24K
    vsd = synth.synth_vocos_state_dict(vcfg, seed=0)
    eng = F5HipEngine(config.DIT_TINY, vcfg, device=0)
    eng.load_state_dict({**synth.synth_dit_state_dict(config.DIT_TINY, seed=1), **vsd})
    try:
        mel = torch.from_numpy(gold("base_v1_cfg1")["out"][:, 468:, :]).permute(0, 2, 1).contiguous()  # [1, 100, 938] as utils_infer.py:507-511
        assert mel.shape == (1, 100, 938)
        w = F5HipVocos(eng).decode(mel.cuda())
        ref = O.vocos_decode(vsd, mel, vcfg.num_layers)
        assert w.shape == ref.shape == (1, 256 * 937)
        e, scale = maxerr(w, ref), float(ref.abs().max())
        print(f"full-size vocos: wave max-abs {e:.2e} of max |wave| {scale:.2e}")
        assert e < 1e-4 * max(1.0, scale)
        w2 = eng.vocos_decode(mel.permute(0, 2, 1).contiguous().cuda(), channel_major=False)  # the layout bench.py hands over
        assert torch.equal(w, w2)
    finally:
        eng.close()


def test_hip_istft_against_the_reference_conv_istft_fixture(engines):
    """istft_fused_kernel (the head of f5hip_vocos_decode) against the wave the reference's own runnable conv-iSTFT
    (runtime/triton_trtllm/scripts/conv_stft.py:193-234) produced for the complex spectrogram in tests/golden/istft_conv_reference.npz."""
    g = gold("istft_conv_reference")
    spec = torch.complex(torch.from_numpy(g["spec_re"]), torch.from_numpy(g["spec_im"]))  # [1, 513, T]
    b, nbin, T = spec.shape
    assert nbin == 513
    logits = torch.zeros(b, T, 1028)
    logits[:, :, :513] = spec.abs().clamp_min(1e-30).log().permute(0, 2, 1)
    logits[:, :, 513:1026] = spec.angle().permute(0, 2, 1)
    eng = engines("tiny", 1, vocos=True)
    w = eng.istft(logits.cuda())
    ref = g["wav"][:, :256 * (T - 1)]
    assert w.shape == ref.shape
    assert maxerr(w, ref) < 5e-5 * max(1.0, float(np.abs(ref).max()))


@pytest.mark.parametrize("name,vname,vseed,frames,hseed", [("vocos_head_ref", "VOCOS_MEL_24K", 2, 96, 31), ("vocos_head_ref_tiny", "VOCOS_TINY", 1, 40, 32)])
def test_vocos_head_against_the_reference_istft_head(name, vname, vseed, frames, hseed):
    """f5hip_vocos_head (head.out GEMM -> exp -> clip(1e2) -> cos / sin -> HIP inverse STFT) against the wave the REFERENCE's own ISTFTHead
    produced (runtime/triton_trtllm/scripts/export_vocoder_to_onnx.py:43-59 + scripts/conv_stft.py, lifted and run by
    oracle/make_golden.py::golden_vocos_head) for the seeded `head.out` layer and hidden state: the Vocos head pinned as a unit at the full
    width; row 1 is scaled so that 3 % of its magnitude bins hit the clip."""
    from f5_tts_amd.engine import F5HipEngine

    vcfg = getattr(config, vname)
    vsd = synth.synth_vocos_state_dict(vcfg, seed=vseed)
    hidden = torch.randn(2, frames, vcfg.dim, generator=torch.Generator().manual_seed(hseed))
    hidden[1] *= 6.0
    ref = gold(name)["wav"]
    eng = F5HipEngine(config.DIT_TINY, vcfg, device=0)
    eng.load_state_dict({**synth.synth_dit_state_dict(config.DIT_TINY, seed=1), **vsd})
    try:
        w = eng.vocos_head(hidden.cuda())
        assert w.shape == ref.shape == (2, 256 * (frames - 1))
        e, scale = maxerr(w, ref), float(np.abs(ref).max())
        print(f"{name}: head wave max-abs {e:.2e} of max |wave| {scale:.2e}")
        assert e < 5e-5 * max(1.0, scale)
    finally:
        eng.close()


@pytest.mark.parametrize("branch_streams", [0, 1])
def test_packed_rows_equal_the_padded_layout_on_the_valid_rows(engines, branch_streams, precisions=("fp16x3", "fp16m", "fp16")):
    """Option "packed_rows" (the reference's varlen path, modules.py:522-543, extended to the row-wise layers): a ragged batch with the
    key-padding mask runs its block loop over the valid rows only.  The valid rows must come out as in the padded layout (same kernels, same
    per-row arithmetic up to the tile choice) and inside the golden's tolerance; the padding is never read by the reference's callers (utils_eval / utils_infer
    slice every utterance to its own length)."""
    from f5_tts_amd.engine import F5HipCFM

    c = MG.CASES["tiny_mask_ragged_b3"]
    cfg, wav, text, duration, lens = MG.case_inputs(c)
    eng = engines(c["preset"], c["wseed"])
    g = gold("tiny_mask_ragged_b3")["out"]
    outs = {}
    try:
        eng.set_option("branch_streams", branch_streams)
        for packed in (0, 1):
            eng.set_option("packed_rows", packed)
            for prec in precisions:
                out, _ = F5HipCFM(eng, precision=prec).sample(wav.cuda(), text, duration, lens=lens, **c["kw"])
                outs[(packed, prec)] = out.cpu()
    finally:
        eng.set_option("packed_rows", 0)
        eng.set_option("branch_streams", -1)
    # the row counts differ, so the launch heuristic may pick other tiles (k-split tiles sum in another order): equal up to that rounding
    for prec, tol, same in (("fp16x3", X3TOL, 1e-4), ("fp16m", MXTOL, 1e-4), ("fp16", 2e-2, 1e-2)):
        if prec not in precisions:
            continue
        for b, d in enumerate(duration.tolist()):
            a_, p_ = outs[(0, prec)][b, :d], outs[(1, prec)][b, :d]
            assert maxerr(a_, p_) < same, f"{prec} row {b}: packed differs from padded by {maxerr(a_, p_):.2e}"
            assert maxerr(p_, g[b, :d]) < tol


def test_packed_rows_small_model_golden():
    """The Small model (dim 768, 18 blocks) with the key-padding mask on a ragged batch of four (500 / 431 / 333 / 250 frames), packed rows
    against the reference-minted golden on every valid row; 24 % of the padded layout's rows are padding."""
    from f5_tts_amd.engine import F5HipCFM, F5HipEngine

    c = MG.FULL_CASES["small_mask_ragged_b4"]
    cfg, wav, text, duration, lens = MG.case_inputs(c)
    eng = F5HipEngine(cfg, None, device=0)
    eng.load_state_dict(MG.case_weights(c))
    g = gold("small_mask_ragged_b4")["out"]
    try:
        eng.set_option("packed_rows", 1)
        for prec in ("fp16x3", "fp16m"):
            out, _ = F5HipCFM(eng, precision=prec).sample(wav.cuda(), text, duration, lens=lens, **c["kw"])
            e = max(maxerr(out[b, :d], g[b, :d]) for b, d in enumerate(duration.tolist()))
            print(f"small model, ragged batch of 4, packed rows, {prec}: max-abs over the valid rows {e:.2e}")
            # rounds 2-4 (plain fp16 attention scores): 4.4e-4 .. 5.2e-4, bound 6.5e-4; with the MX-corrected scores of round 5 it measures
            # 4.0e-4 and is held to FULL_TOL like every other full-size golden
            assert e < FULL_TOL
    finally:
        eng.close()


def test_ragged_masked_batch_on_trained_like_weights():
    """Round 5: the Small model with the key-padding mask on a ragged batch of three (460 / 380 / 250 frames) whose weights carry a trained
    checkpoint's statistics (synth.trained_like_dit_state_dict) - larger logits than any Gaussian golden, masked key tails, and the packed-row
    layout on top.  Golden minted by the reference's own CFM.sample.  Asserted in both half-precision modes, padded and packed; the plain-fp16
    attention scores of rounds 2-4 are measured next to the default for the record."""
    from f5_tts_amd.engine import F5HipCFM, F5HipEngine

    c = MG.FULL_CASES["small_mask_ragged_b3_trained_like"]
    cfg, wav, text, duration, lens = MG.case_inputs(c)
    eng = F5HipEngine(cfg, None, device=0)
    eng.load_state_dict(MG.case_weights(c))
    g = gold("small_mask_ragged_b3_trained_like")["out"]

    def err(out):  # generated frames of every row
        return max(maxerr(out[b, int(lens[b]):d], g[b, int(lens[b]):d]) for b, d in enumerate(duration.tolist()))

    try:
        for packed in (0, 1):
            eng.set_option("packed_rows", packed)
            for prec in ("fp16m", "fp16x3"):
                out, _ = F5HipCFM(eng, precision=prec).sample(wav.cuda(), text, duration, lens=lens, **c["kw"])
                e = err(out)
                print(f"trained-like ragged batch of 3, {'packed' if packed else 'padded'} rows, {prec}: {e:.2e}")
                assert e < 7e-4
        eng.set_option("packed_rows", 0)
        eng.set_option("attn_impl", 3)
        out, _ = F5HipCFM(eng, precision="fp16m").sample(wav.cuda(), text, duration, lens=lens, **c["kw"])
        print(f"  with plain fp16 scores (attn_impl 3), padded, fp16m: {err(out):.2e}")
        eng.set_option("attn_impl", 0)
        out, _ = F5HipCFM(eng, precision="fp32").sample(wav.cuda(), text, duration, lens=lens, **c["kw"])
        e32 = err(out)
        print(f"  fp32: {e32:.2e}")
        assert e32 < TIGHT
    finally:
        eng.close()


@pytest.mark.parametrize("family,tols", [
    # golden -> bounds of (fp16m default, fp16m attn_impl 6, fp16m attn_impl 7, fp16x3 attn_impl 2, fp32); measured (profiles/r06g_sharpness_sweep.md)
    ("base_v1_trained_like", {"": (7e-4, 6e-4, 6e-4, 1.5e-4, 1e-4), "_sharp1p2": (1e-3, 8e-4, 5.5e-4, 1.5e-4, 1e-4), "_sharp1p4": (2.2e-3, 1.6e-3, 7e-4, 2e-4, 1.5e-4), "_sharp1p7": (6e-3, 4e-3, 1.5e-3, 4e-4, 3.5e-4)}),
    ("small_mask_ragged_b3_trained_like", {"": (7e-4, 6e-4, 5e-4, 1e-4, 1e-4), "_sharp1p2": (1e-3, 7e-4, 5e-4, 1e-4, 1e-4), "_sharp1p4": (3e-3, 1.6e-3, 6e-4, 1e-4, 1e-4), "_sharp1p7": (5e-3, 4e-3, 1.3e-3, 1.5e-4, 1.5e-4)}),
])
def test_sharpness_sweep_full_size(family, tols):
    """VERDICT r05 item 3, measured instead of argued: the trained-like goldens with every attention logit x 1.41, x 2 and x 2.8 (to_q, to_k
    x 2^0.25 — inside the 1e-3 tolerance in the default mode: 7.0e-4 / 7.9e-4, the point f5hip_attention_stats' thresholds are calibrated on —, x sqrt 2,
    x 2^0.75; minted by the reference's own CFM.sample; at x 4 the reference's fp32 result is itself only reproducible to 1.3e-3 and at x 16
    not at all, tests/golden/pins.json).  What the sweep showed: the error of the DEFAULT half-precision attention —
    MX-corrected scores, plain fp16 P and V — GROWS with the logits (4.9e-4 -> 1.4e-3 -> 4e-3): it leaves the 1e-3 tolerance between x 1.4
    and x 2.  The carrier is fp16 V under near one-hot rows; attn_impl 6 reads V as hi + lo halves, 7 splits P as well (<= 1e-3 up to x 2.8 in
    fp16m, whose MX GEMM scheme then carries the rest), fp16x3 with everything split stays within 4x of the fp32 kernels at every point.
    Asserted: each mode stays inside the bound it was measured at (x ~1.4), and the ORDER — more split operands never hurt by more than noise."""
    from f5_tts_amd.engine import F5HipCFM, F5HipEngine

    eng = None
    try:
        for suffix, bounds in tols.items():
            name = family + suffix
            c = MG.FULL_CASES[name]
            cfg, wav, text, duration, lens = MG.case_inputs(c)
            if eng is not None:
                eng.close()
            eng = F5HipEngine(cfg, None, device=0)
            eng.load_state_dict(MG.case_weights(c))
            g = gold(name)["out"]
            durs = duration.tolist() if torch.is_tensor(duration) else [int(duration)] * g.shape[0]
            errs = []
            for (prec, impl), bound in zip((("fp16m", 0), ("fp16m", 6), ("fp16m", 7), ("fp16x3", 2), ("fp32", 0)), bounds):
                eng.set_option("attn_impl", impl)
                out, _ = F5HipCFM(eng, precision=prec).sample(wav.cuda(), text, duration, lens=lens, **c["kw"])
                e = max(maxerr(out[b, :d], g[b, :d]) for b, d in enumerate(durs))
                errs.append(e)
                assert e < bound, (name, prec, impl, e, bound)
            print(f"{name}: fp16m default {errs[0]:.2e}, V hi+lo {errs[1]:.2e}, V and P hi+lo {errs[2]:.2e}; fp16x3 all split {errs[3]:.2e}; fp32 {errs[4]:.2e}")
            assert errs[1] < 1.3 * errs[0] and errs[2] < 1.3 * errs[1] and errs[3] < errs[2]
    finally:
        if eng is not None:
            eng.close()


@pytest.mark.parametrize("name", sorted(MG.SWEEP_CASES))
def test_sharpness_sweep_tiny(engines, name):
    """The tiny sweep points (logits x 4 and x 16; the tiny model is not chaotic there: fp32 floor 5e-6): the default half-precision modes inside
    the bounds they were measured at, the split forms ordered, fp32 at its own level."""
    from f5_tts_amd.engine import F5HipCFM

    c = MG.SWEEP_CASES[name]
    cfg, wav, text, duration, lens = MG.case_inputs(c)
    eng = engines(c["preset"], c["wseed"], trained=True, sharp=c["sharp"])
    g = gold(name)["out"]
    sharp16 = c["sharp"] > 3
    try:
        errs = {}
        for prec, impl, bound in (("fp32", 0, TIGHT), ("fp16x3", 0, 6e-4 if sharp16 else 3e-4), ("fp16m", 0, 6e-4 if sharp16 else 3e-4),
                                  ("fp16m", 6, 4e-4 if sharp16 else 2e-4), ("fp16m", 7, 4e-4 if sharp16 else 2e-4), ("fp16x3", 2, 5e-5)):
            eng.set_option("attn_impl", impl)
            out, _ = F5HipCFM(eng, precision=prec).sample(wav.cuda(), text, duration, lens=lens, **c["kw"])
            errs[(prec, impl)] = e = maxerr(out, g)
            assert e < bound, (name, prec, impl, e, bound)
        print(name, {f"{p}:{i}": f"{e:.2e}" for (p, i), e in errs.items()})
    finally:
        eng.set_option("attn_impl", 0)


def test_configs2_shaped_batch_golden():
    """BASELINE.json configs[2] / [3] shape (a batch of fixed-length prompts through the packed cond | uncond schedule, NFE 32) at the full
    model size: 4 distinct utterances against the golden minted by the reference's own CFM.sample (oracle/make_golden.py base_v1_cfg3_b4),
    then the same 4 utterances inside a batch of 32 (rows 0..3 of the B = 32 packed schedule that bench.py --batch 32 --nfe 32 times;
    fixed-length batches have no cross-row coupling, so those rows must reproduce the golden as well)."""
    from f5_tts_amd.engine import F5HipCFM, F5HipEngine

    c = MG.FULL_CASES["base_v1_cfg3_b4"]
    cfg, wav, text, duration, lens = MG.case_inputs(c)
    eng = F5HipEngine(cfg, None, device=0)
    eng.load_state_dict(synth.synth_dit_state_dict(cfg, seed=c["wseed"]))
    g = gold("base_v1_cfg3_b4")
    try:
        for prec in ("fp16m", "fp16x3"):
            model = F5HipCFM(eng, precision=prec)
            out, traj = model.sample(wav.cuda(), text, duration, **c["kw"])
            e = maxerr(out[:, 468:], g["out"][:, 468:])
            print(f"configs[2]-shaped B=4 NFE=32 {prec}: generated-mel max-abs {e:.2e}")
            assert e < FULL_TOL and maxerr(traj[1], g["traj_1"]) < FULL_TOL
            wav32, text32 = wav.repeat(8, 1), text.repeat(8, 1)
            out32, _ = model.sample(wav32.cuda(), text32, duration, **c["kw"])
            e32 = maxerr(out32[:4, 468:], g["out"][:, 468:])
            print(f"the same utterances as rows 0..3 of B=32 ({prec}): {e32:.2e}")
            assert e32 < FULL_TOL
            assert maxerr(out32[28:, 468:], g["out"][:, 468:]) < FULL_TOL  # and as the last four rows (other tiles of the 256-row GEMM tiling)
    finally:
        eng.close()


def test_weights_reloaded_into_a_live_context():
    """the per-step time-embedding / AdaLN tables are cached on the time grid; reloading weights into a context that
    has already sampled must not reuse them.  Seed-2 weights loaded over seed-1 weights == a fresh seed-2 engine, bit for bit."""
    from f5_tts_amd.engine import F5HipCFM, F5HipEngine

    cfg = config.DIT_TINY
    wav = synth.synth_wave(256 * 40, seed=3)
    text = synth.synth_text_ids(1, 30, cfg.text_num_embeds, seed=2)
    kw = dict(steps=4, cfg_strength=2.0, sway_sampling_coef=-1.0, seed=7)
    live, fresh = F5HipEngine(cfg, None, device=0), F5HipEngine(cfg, None, device=0)
    try:
        for use_graph in (0, 1):
            live.set_option("use_graph", use_graph)
            live.load_state_dict(synth.synth_dit_state_dict(cfg, seed=1))
            a1, _ = F5HipCFM(live).sample(wav.cuda(), text, 120, **kw)
            live.load_state_dict(synth.synth_dit_state_dict(cfg, seed=2))  # same grid, same shapes: only the weights changed
            a2, _ = F5HipCFM(live).sample(wav.cuda(), text, 120, **kw)
            fresh.load_state_dict(synth.synth_dit_state_dict(cfg, seed=2))
            b2, _ = F5HipCFM(fresh).sample(wav.cuda(), text, 120, **kw)
            assert not torch.equal(a1, a2)
            assert torch.equal(a2, b2), float((a2 - b2).abs().max())
        ref, _ = O.cfm_sample(synth.synth_dit_state_dict(cfg, seed=2), cfg, wav, text, 120, **kw)
        assert maxerr(a2, ref) < TIGHT
    finally:
        live.close()
        fresh.close()


def test_sample_enqueues_and_returns(engines):
    """include/f5hip.h: the entry points enqueue and return.  With the NFE loop replayed as a graph the host is back long before the GPU
    has finished, and the stream is still busy at that moment."""
    import time

    from f5_tts_amd.engine import F5HipCFM

    cfg = config.DIT_TINY
    eng = engines("tiny", 1)
    eng.set_option("use_graph", 1)
    try:
        B = 8
        wav = synth.synth_wave(256 * 100, seed=4, batch=B).cuda()
        text = synth.synth_text_ids(B, 60, cfg.text_num_embeds, seed=5)
        kw = dict(steps=64, cfg_strength=2.0, sway_sampling_coef=-1.0, seed=3)
        model = F5HipCFM(eng, precision="fp16x3")
        model.sample(wav, text, 600, **kw)  # workspace + graph capture
        torch.cuda.synchronize()
        st = torch.cuda.current_stream()
        t0 = time.perf_counter()
        out, _ = model.sample(wav, text, 600, **kw)
        t_host = time.perf_counter() - t0
        busy = not st.query()
        torch.cuda.synchronize()
        t_all = time.perf_counter() - t0
        print(f"host returned after {1e3 * t_host:.2f} ms, GPU finished after {1e3 * t_all:.2f} ms")
        assert busy and t_host < 0.5 * t_all
        assert bool(torch.isfinite(out).all())
    finally:
        eng.set_option("use_graph", 0)


def test_four_threads_share_one_context(engines):
    """The reference calls sample() from a thread pool (utils_infer.py:540-543).  Four threads on one context: every result bit-identical
    to the same call made alone (the calls of a context are ordered on the GPU, their host work overlaps)."""
    from concurrent.futures import ThreadPoolExecutor

    from f5_tts_amd.engine import F5HipCFM

    cfg = config.DIT_TINY
    eng = engines("tiny", 1)
    model = F5HipCFM(eng, precision="fp16x3")
    jobs = []
    for i in range(8):
        wav = synth.synth_wave(256 * (30 + 3 * i), seed=20 + i)
        text = synth.synth_text_ids(1, 20 + i, cfg.text_num_embeds, seed=30 + i)
        jobs.append((wav, text, 100 + 7 * i, dict(steps=4 + (i % 3), cfg_strength=2.0, sway_sampling_coef=-1.0, seed=40 + i)))

    def one(j):
        wav, text, dur, kw = j
        with torch.cuda.device(0):
            out, _ = model.sample(wav.cuda(), text, dur, **kw)
            return out.cpu()

    alone = [one(j) for j in jobs]
    for _ in range(3):
        with ThreadPoolExecutor(max_workers=4) as pool:
            together = list(pool.map(one, jobs))
        for a, b in zip(alone, together):
            assert torch.equal(a, b)


# ---- the pipelined block GEMM (csrc/gemm_pp.h) on the GPU ---------------------------------------------------------------------------------
@pytest.mark.parametrize("variant", [50, 51, 55, 56, 59, 68])  # the tiles pick_pp_variant() can choose
@pytest.mark.parametrize("seqs", [2, 8])
def test_pp_qkv_epilogue_equals_generic_kernel_on_the_gpu(engines, variant, seqs):
    """The fused q|k|v projection through every production tile of the pipelined kernel against the generic kernel of gemm.h (GPU-verified
    against the reference goldens since round 1): every q / k value to fp32 rounding, V^T byte for byte, three launches each (a race would
    show as run-to-run differences; round 2 found one in the 4-wave two-per-CU tiles this way — they are not used by the engine)."""
    import ctypes as C

    from f5_tts_amd import binding

    eng = engines("tiny", 1)
    ms, diff = C.c_double(), C.c_int64()
    for _ in range(3):
        st = eng.bench_lib.f5hip_bench_qkv(eng._ctx, binding.PRECISIONS["fp16x3"], variant, seqs, 1406, 1024, 2, 1, C.byref(ms), C.byref(diff))
        assert st == 0 and diff.value == 0, (variant, seqs, st, diff.value)


@pytest.mark.parametrize("prec,variant", [("fp16m", -1), ("fp16m", 50), ("fp16m", 59), ("fp16m", 61), ("fp16m", 68), ("fp16m", 80), ("fp16x3", -1), ("fp16x3", 57)])
@pytest.mark.parametrize("seqs,nseq", [(2, 150), (2, 1406), (8, 1024)])
def test_qkv_epilogue_score_corrections_on_the_gpu(engines, capfd, prec, variant, seqs, nseq):
    """The MX-fp6 correction words the q|k|v epilogue leaves for the attention scores (round 5's default), decoded on the host against the
    generic kernel's hi + lo values with the format's bounds, and the flash kernel's NSPLIT = 2 form on them against the split-q,k form at
    logits of tens.  This is the test that separates the GPU from the host shim: with the conversion's scale register allocated inside its
    destination every word was out of bounds here and nowhere else."""
    import ctypes as C

    from f5_tts_amd import binding

    eng = engines("tiny", 1)
    ms, diff = C.c_double(), C.c_int64()
    st = eng.bench_lib.f5hip_bench_qkv(eng._ctx, binding.PRECISIONS[prec], variant, seqs, nseq, 1024, 1, 2, C.byref(ms), C.byref(diff))
    err = capfd.readouterr().err
    if prec == "fp16x3" and variant < 0 and seqs * nseq < 512:  # the heuristic sends a handful of row tiles to the generic kernel, which
        assert st == 2, (st, err[-500:])                        # writes no correction words: the launch must refuse (the engine asks
        return                                                  # gemm_qkv_takes_pp first and falls back to split q, k)
    assert st == 0 and diff.value == 0, (st, diff.value, err[-1500:])
    assert "attention on the MX planes" in err


@pytest.mark.parametrize("variant", [55, 59, 65, 66, 67, 69, 70])
@pytest.mark.parametrize("epi,M,N,K", [(1, 2812, 2048, 1024), (2, 2812, 1024, 2048), (2, 1406, 1024, 1024)])
def test_pp_store_epilogues_equal_generic_kernel_on_the_gpu(engines, capfd, variant, epi, M, N, K):
    """FF1 (GELU -> packed operand rows) and out-proj / FF2 (gated residual update) of the DiT Base shapes through the production tiles,
    among them the k-split 96x128 (two groups of waves on alternate k-tiles, partial sums exchanged through LDS): every value against the
    generic kernel's to fp32 rounding (the k-split sums in another order, so not bytes), three repetitions that must agree with each other."""
    import ctypes as C
    import re

    from f5_tts_amd import binding

    eng = engines("tiny", 1)
    os.environ["KB_CHECK"] = "1"
    try:
        ms = C.c_double()
        st = eng.bench_lib.f5hip_bench_gemm(eng._ctx, binding.PRECISIONS["fp16x3"], variant, epi, M, N, K, 1, C.byref(ms))
    finally:
        os.environ.pop("KB_CHECK", None)
    err = capfd.readouterr().err
    assert st == 0, err
    lines = re.findall(r"KB_CHECK variant \d+ epi \d+ rep \d+: (\d+) of \d+ bytes differ.*max \|diff\| (\S+) of max \|value\| (\S+)", err)
    assert len(lines) == 3 and len(set(lines)) == 1, err  # the same differing bytes / distance every time
    d, v = float(lines[0][1]), float(lines[0][2])
    assert v > 0.5 and d <= 4e-6 * v, lines


def test_full_size_runs_are_bit_reproducible():
    """F5-TTS Base, B = 1 (one packed chain, the one-round k-split tiles, graph replay) and B = 4 (two concurrent CFG chains, the 256x128
    tiles): five calls, one result, bit for bit."""
    from f5_tts_amd.engine import F5HipCFM, F5HipEngine

    c = MG.FULL_CASES["base_v1_cfg1"]
    cfg, wav, text, duration, lens = MG.case_inputs(c)
    eng = F5HipEngine(cfg, None, device=0)
    eng.load_state_dict(synth.synth_dit_state_dict(cfg, seed=c["wseed"]))
    try:
        model = F5HipCFM(eng, precision="fp16x3")
        for use_graph in (0, 1):
            eng.set_option("use_graph", use_graph)
            kw = dict(c["kw"], steps=4)
            first, _ = model.sample(wav.cuda(), text, duration, **kw)
            for _ in range(4):
                again, _ = model.sample(wav.cuda(), text, duration, **kw)
                assert torch.equal(first, again)
        eng.set_option("use_graph", 0)
        kw = dict(c["kw"], steps=2)
        w4, t4 = wav.repeat(4, 1).cuda(), text.repeat(4, 1)
        first, _ = model.sample(w4, t4, duration, **kw)
        for _ in range(3):
            again, _ = model.sample(w4, t4, duration, **kw)
            assert torch.equal(first, again)
    finally:
        eng.close()


def test_bucketed_eval_batches_on_the_gpu(engines, tmp_path):
    """eval/utils_eval.py:72-205 + eval_infer_batch.py:178-214 through the HIP engine: length-bucketed ragged batches formed with the
    engine's own mel front-end, one ragged sample() per batch, every utterance vocoded once; the loop equals the manual per-batch
    computation bit for bit, and the oracle's sampler agrees on a batch (ragged lens / durations, list-of-str text)."""
    import wave as wave_mod

    from f5_tts_amd import eval_batching as EB
    from f5_tts_amd.engine import F5HipCFM, F5HipVocos

    cfg = config.DIT_TINY
    eng = engines("tiny", 1, vocos=True)
    sd = synth.synth_dit_state_dict(cfg, seed=1)
    vocab = {chr(c): 1 + (c % (cfg.text_num_embeds - 2)) for c in range(32, 127)}
    model, voc = F5HipCFM(eng, precision="fp16x3", vocab_char_map=vocab), F5HipVocos(eng)
    meta = []
    for i, (secs, words) in enumerate([(0.45, 3), (0.5, 4), (0.47, 3), (0.9, 6), (0.52, 4), (0.95, 7)]):
        x = (0.02 if i == 1 else 0.2) * np.random.default_rng(i).standard_normal(int(secs * 24000))
        p = tmp_path / f"p{i}.wav"
        with wave_mod.open(str(p), "wb") as w:
            w.setnchannels(1); w.setsampwidth(2); w.setframerate(24000); w.writeframes((x.clip(-1, 1) * 32767).astype("<i2").tobytes())
        meta.append((f"u{i}", "ab cd.", str(p), " ".join(["w"] * words), ""))
    batches = EB.get_inference_prompt(meta, lambda wv: model.mel_spec(wv.cuda()).cpu(), tokenizer="char", infer_batch_size=150, min_secs=0,
                                      max_secs=3, num_buckets=3)
    assert sorted(u for b in batches for u in b[0]) == sorted(m[0] for m in meta) and any(len(b[0]) > 1 for b in batches)
    run = dict(nfe_step=4, cfg_strength=2.0, sway_sampling_coef=-1.0, seed=3)
    got = dict(EB.run_prompt_batches(model, voc, batches, **run))
    assert set(got) == {m[0] for m in meta}
    for utts, rms, mels, lens, totals, texts in batches:
        out, _ = model.sample(cond=mels, text=texts, duration=torch.tensor(totals), lens=torch.tensor(lens), steps=4, cfg_strength=2.0,
                              sway_sampling_coef=-1.0, seed=3)
        from f5_tts_amd.engine import list_str_to_idx

        ref, _ = O.cfm_sample(sd, cfg, mels, list_str_to_idx(texts, vocab), torch.tensor(totals), lens=torch.tensor(lens), steps=4,
                              cfg_strength=2.0, sway_sampling_coef=-1.0, seed=3)
        assert maxerr(out, ref) < X3TOL
        for i, u in enumerate(utts):
            w = voc.decode(out[i, lens[i]:totals[i]].unsqueeze(0).permute(0, 2, 1)).cpu()
            if rms[i] < 0.1:
                w = w * rms[i] / 0.1
            assert torch.equal(w, got[u]) and w.shape[-1] == 256 * (totals[i] - lens[i] - 1)
