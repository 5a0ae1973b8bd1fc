"""Host-side glue (f5-tts_amd/infer.py) against the reference's own functions where they can be lifted out of its modules
(pure-Python helpers: extracted from the reference source with ast when /root/reference exists) and by properties otherwise."""
import ast
import math
import os
import re

import numpy as np
import pytest
import torch

import f5_tts_amd  # noqa: F401
from f5_tts_amd import infer as I

REF = "/root/reference/src/f5_tts"
TEXTS = [
    "Some call me nature, others call me mother nature.",
    "I don't really care what you call me. I've been a silent spectator, watching species evolve, empires rise and fall. "
    "But always remember, I am mighty and enduring. Respect me and I'll nurture you; ignore me and you shall face the consequences.",
    "Short.",
    "no punctuation at all just words going on and on and on " * 6,
    "A;B:C,D.E!F? G",
    "",
]


def lift(path, name, env):
    """exec one top-level function of a reference module without importing the module (its imports are not installable here)"""
    src = open(path, encoding="utf-8").read()
    fn = next(n for n in ast.parse(src).body if isinstance(n, ast.FunctionDef) and n.name == name)
    code = compile(ast.Module(body=[fn], type_ignores=[]), path, "exec")
    exec(code, env)
    return env[name]


needs_ref = pytest.mark.skipif(not os.path.isdir(REF), reason="reference tree not present (GPU box)")


@needs_ref
@pytest.mark.parametrize("max_chars", [20, 60, 135, 400])
def test_chunk_text_matches_reference(max_chars):
    ref = lift(os.path.join(REF, "infer", "utils_infer.py"), "chunk_text", {"re": re})
    for t in TEXTS:
        assert I.chunk_text(t, max_chars) == ref(t, max_chars)


@needs_ref
def test_convert_char_to_pinyin_single_byte_rule_matches_reference():
    import types

    rj = types.SimpleNamespace(cut=lambda s: I._ASCII_TOKENS.findall(s))  # stand-in segmenter: the rule under test is what follows it
    ref = lift(os.path.join(REF, "model", "utils.py"), "convert_char_to_pinyin",
               {"rjieba": rj, "lazy_pinyin": None, "Style": None})
    for t in TEXTS + ["hello,world (x-ray) it's \"quoted\": yes;no"]:
        assert I.convert_char_to_pinyin([t]) == ref([t])
    with pytest.raises(ValueError):
        I.convert_char_to_pinyin(["你好"])  # needs rjieba/pypinyin


def test_cross_fade_concat_properties():
    a, b, c = np.ones(5000, np.float32), 2 * np.ones(4000, np.float32), 3 * np.ones(100, np.float32)
    out = I.cross_fade_concat([a, b, c], 0.15)
    k1, k2 = int(0.15 * 24000), 100
    assert len(out) == 5000 + 4000 + 100 - k1 - k2
    assert out[0] == 1 and out[-1] == 3
    mid = out[5000 - k1:5000]
    assert np.all(np.diff(mid) >= -1e-6) and abs(mid[0] - 1) < 1e-6 and abs(mid[-1] - 2) < 1e-6
    assert np.array_equal(I.cross_fade_concat([a, b], 0.0), np.concatenate([a, b]))


def test_resample_preserves_a_tone():
    sr0, sr1, f = 16000, 24000, 440.0
    t0 = torch.arange(16000) / sr0
    x = torch.sin(2 * math.pi * f * t0)[None]
    y = I.resample(x, sr0, sr1)
    assert y.shape == (1, 24000)
    t1 = torch.arange(24000) / sr1
    ref = torch.sin(2 * math.pi * f * t1)
    assert (y[0, 200:-200] - ref[200:-200]).abs().max() < 2e-3
    assert torch.equal(I.resample(x, sr0, sr0), x)


def test_load_wav_roundtrip(tmp_path):
    import wave

    pcm = (np.random.RandomState(0).randn(2400) * 3000).astype("<i2")
    p = tmp_path / "a.wav"
    with wave.open(str(p), "wb") as w:
        w.setnchannels(1); w.setsampwidth(2); w.setframerate(24000); w.writeframes(pcm.tobytes())
    audio, sr = I.load_wav(str(p))
    assert sr == 24000 and audio.shape == (1, 2400)
    assert np.array_equal((audio[0].numpy() * 32768).astype(np.int16), pcm)


def test_get_tokenizer(tmp_path):
    p = tmp_path / "vocab.txt"
    p.write_text(" \na\nb\nzh1\n", encoding="utf-8")
    vocab, n = I.get_tokenizer(str(p))
    assert n == 4 and vocab[" "] == 0 and vocab["zh1"] == 3
    assert I.get_tokenizer("", "byte") == (None, 256)


def test_build_edit_condition_matches_the_reference_script():
    """Frame bookkeeping of speech_edit.py:154-200 (kept frames, zero frames of the wanted length, mask), against a literal re-run of
    the script's loop."""
    import torch

    from f5_tts_amd import infer as I

    mel = torch.arange(1 * 300 * 4, dtype=torch.float32).reshape(1, 300, 4)
    parts, fix = [[0.5, 1.0], [2.0, 2.6]], [0.3, 0.9]
    cond, mask = I.build_edit_condition(mel, parts, fix)
    # literal restatement
    off, ref_c, ref_m = 0, [], []
    for (s0, e0), d in zip(parts, fix):
        sf, ef, df = round(s0 * 24000 / 256), round(e0 * 24000 / 256), round(d * 24000 / 256)
        ref_c += [mel[:, off:sf], torch.zeros(1, df, 4)]
        ref_m += [torch.ones(1, sf - off, dtype=torch.bool), torch.zeros(1, df, dtype=torch.bool)]
        off = ef
    ref_c.append(mel[:, off:])
    ref_c = torch.cat(ref_c, 1)
    ref_m = torch.cat(ref_m + [torch.ones(1, ref_c.shape[1] - sum(x.shape[1] for x in ref_m), dtype=torch.bool)], 1)
    assert torch.equal(cond, ref_c) and torch.equal(mask, ref_m)
    assert cond.shape[1] == 300 - (round(1.0 * 93.75) - round(0.5 * 93.75)) - (round(2.6 * 93.75) - round(2.0 * 93.75)) + round(0.3 * 93.75) + round(0.9 * 93.75)
    cond2, mask2 = I.build_edit_condition(mel, parts)  # spans keep their own length
    assert cond2.shape[1] == 300 and int((~mask2).sum()) == (round(1.0 * 93.75) - round(0.5 * 93.75)) + (round(2.6 * 93.75) - round(2.0 * 93.75))


def test_load_model_arch_dict_needs_its_backbone_class(monkeypatch):
    """the reference's load_model(model_cls, model_cfg, ...) (utils_infer.py:238) names the backbone class; an arch dict
    alone must not silently become a DiT.  Checked on the config that reaches the engine constructor (no GPU needed)."""
    from f5_tts_amd import infer as I

    seen = {}

    class FakeEngine:
        def __init__(self, cfg, vocos_cfg, device=0):
            seen["cfg"], seen["vocos"] = cfg, vocos_cfg

        def finalize(self):
            seen["finalized"] = True

    monkeypatch.setattr(I, "F5HipEngine", FakeEngine)
    monkeypatch.setattr(I, "F5HipCFM", lambda engine, **kw: engine)
    e2_arch = dict(dim=1024, depth=24, heads=16, ff_mult=4, text_mask_padding=False, pe_attn_head=1)  # configs/E2TTS_Base.yaml:25-31
    with pytest.raises(ValueError, match="model_cls"):
        I.load_model(e2_arch, None)
    I.load_model(e2_arch, None, model_cls="UNetT")
    assert seen["cfg"].backbone == "UNetT" and seen["cfg"].text_dim == 100 and seen["cfg"].conv_layers == 0 and seen["cfg"].depth == 24

    class DiT:  # a class object works like the reference's first argument
        pass

    I.load_model(dict(dim=1024, depth=22, heads=16, ff_mult=2, text_dim=512, conv_layers=4), None, model_cls=DiT)
    assert seen["cfg"].backbone == "DiT" and seen["cfg"].text_dim == 512 and seen["vocos"] is not None
    # mel_spec_type="bigvgan": no Vocos slots to wait for — the context is finalised by load_model itself
    seen.clear()
    I.load_model("E2TTS_Base", None, mel_spec_type="bigvgan")
    assert seen["vocos"] is None and seen.get("finalized")
