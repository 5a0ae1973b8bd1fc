"""Length-bucketed batch formation (f5-tts_amd/eval_batching.py) against the reference's own ``get_inference_prompt`` /
``padded_mel_batch`` (src/f5_tts/eval/utils_eval.py:56-205), lifted out of the module with ast (its imports — torchaudio, the ECAPA
model — are not installable here) and run on the same synthetic .wav files; then the batch loop on the host build of the engine."""
import math
import os
import random
import types
import wave

import numpy as np
import pytest
import torch
import torch.nn.functional as F

import f5_tts_amd  # noqa: F401
from f5_tts_amd import eval_batching as EB
from f5_tts_amd import infer as I
from tests.test_hipemu import CLANG, emu_engine, engine_emu_lib  # noqa: F401  (fixtures)
from tests.test_infer_host import REF, lift, needs_ref

HOP = 256


def write_wav(path, n, sr, seed, amp):
    rng = np.random.default_rng(seed)
    x = (amp * rng.standard_normal(n)).clip(-1, 1)
    with wave.open(str(path), "wb") as w:
        w.setnchannels(1); w.setsampwidth(2); w.setframerate(sr); w.writeframes((x * 32767).astype("<i2").tobytes())


def fake_mel(wav):  # [1, n] -> [1, 100, n // HOP + 1]: deterministic, length as the real front-end's
    t = wav.shape[-1] // HOP + 1
    base = F.pad(wav[0], (0, t * HOP - wav.shape[-1]))[::HOP][:t]
    return (base[None, None, :] * torch.arange(1, 101, dtype=torch.float32)[None, :, None]).contiguous()


@pytest.fixture(scope="module")
def corpus(tmp_path_factory):
    d = tmp_path_factory.mktemp("corpus")
    rng = random.Random(5)
    meta = []
    for i in range(37):
        sr = 24000 if i % 3 else 16000
        secs = rng.uniform(1.5, 7.0)
        pw, gw = d / f"p{i}.wav", d / f"g{i}.wav"
        write_wav(pw, int(secs * sr), sr, i, 0.02 if i % 5 == 0 else 0.2)  # some prompts below the target RMS
        write_wav(gw, int(rng.uniform(2.0, 9.0) * sr), sr, 100 + i, 0.1)
        ptxt = "prompt text number %d%s" % (i, "." if i % 2 else " end")
        gtxt = " ".join(["word"] * rng.randint(3, 15))
        meta.append((f"utt{i}", ptxt, str(pw), gtxt, str(gw)))
    return meta


def ref_get_inference_prompt():
    path = os.path.join(REF, "eval", "utils_eval.py")
    ta = types.SimpleNamespace(load=I.load_wav, transforms=types.SimpleNamespace(Resample=lambda a, b: (lambda x: I.resample(x, a, b))))
    env = {"math": math, "random": random, "torch": torch, "F": F, "torchaudio": ta, "tqdm": lambda it, **kw: it,
           "MelSpec": lambda **kw: fake_mel, "convert_char_to_pinyin": I.convert_char_to_pinyin}
    lift(path, "padded_mel_batch", env)
    return lift(path, "get_inference_prompt", env)


@needs_ref
@pytest.mark.parametrize("kw", [dict(infer_batch_size=1), dict(infer_batch_size=1500), dict(infer_batch_size=4000, speed=0.8),
                                dict(infer_batch_size=2500, use_truth_duration=True, num_buckets=50),
                                dict(infer_batch_size=100000, tokenizer="char", min_secs=1, max_secs=60)])
def test_batches_equal_the_reference_function(corpus, kw):
    kw = dict(dict(min_secs=1, max_secs=40), **kw)
    want = ref_get_inference_prompt()(corpus, **kw)
    got = EB.get_inference_prompt(corpus, fake_mel, **kw)
    assert len(got) == len(want) and len(got) > 0
    for g, w in zip(got, want):  # same batches, same order (the shuffle is seeded), same tensors
        assert g[0] == w[0] and g[3] == w[3] and g[4] == w[4] and g[5] == w[5]
        assert all(torch.equal(a, b) for a, b in zip(g[1], w[1]))
        assert g[2].shape == w[2].shape and torch.equal(g[2], w[2])
    if kw["infer_batch_size"] == 1:
        assert all(len(p[0]) == 1 for p in got)


def test_frame_budget_classes_and_padding(corpus):
    got = EB.get_inference_prompt(corpus, fake_mel, infer_batch_size=3000, min_secs=1, max_secs=40)
    assert sorted(u for p in got for u in p[0]) == sorted(m[0] for m in corpus)  # every utterance exactly once
    span = (40 - 1) * 24000 // HOP + 1
    for utts, rms, mels, lens, totals, texts in got:
        assert mels.shape == (len(utts), max(lens), 100) and len(texts) == len(utts)
        assert max(totals) - min(totals) <= math.ceil(span / 200)  # one length class per batch
        assert sum(totals[:-1]) < 3000  # flushed as soon as the budget is reached
    one = EB.get_inference_prompt(corpus, fake_mel, infer_batch_size=10 ** 9, num_buckets=1, min_secs=1, max_secs=40)
    assert len(one) == 1 and EB.padding_fraction(got) < 0.05 < EB.padding_fraction(one)  # what the classes buy
    with pytest.raises(AssertionError):
        EB.get_inference_prompt(corpus, fake_mel, infer_batch_size=3000, min_secs=3, max_secs=4)  # out-of-range durations are refused


def test_deal_batches(corpus):
    got = EB.get_inference_prompt(corpus, fake_mel, infer_batch_size=2000, min_secs=1, max_secs=40)
    for world in (1, 2, 3, 8):
        for balanced in (True, False):
            deal = EB.deal_batches(got, world, balanced)
            assert sorted(i for r in deal for i in r) == list(range(len(got)))
        bal = [sum(EB.batch_cost(got[i]) for i in r) for r in EB.deal_batches(got, world, True)]
        con = [sum(EB.batch_cost(got[i]) for i in r) for r in EB.deal_batches(got, world, False)]
        assert max(bal) <= max(con)  # the greedy deal is never worse than the contiguous split of the shuffled list


@needs_ref
def test_metainfo_parsers_match_the_reference(tmp_path):
    path = os.path.join(REF, "eval", "utils_eval.py")
    lst = tmp_path / "meta.lst"
    lst.write_text("u1|hello there.|wavs/p1.wav|generate this\nu2|second|/abs/p2.wav|and this|/abs/g2.wav\nu3|third one|p3.wav|more text\n")
    assert EB.get_seedtts_testset_metainfo(str(lst)) == lift(path, "get_seedtts_testset_metainfo", {"os": os})(str(lst))
    ls = tmp_path / "ls.lst"
    ls.write_text("1-2-3\t4.0\tRef text.\t5-6-7\t3.0\tGen text.\n8-9-10\t2.5\tAnother ref\t8-9-11\t6.1\tAnother gen\n")
    assert EB.get_librispeech_test_clean_metainfo(str(ls), "/ls") == lift(path, "get_librispeech_test_clean_metainfo", {"os": os})(str(ls), "/ls")


def test_metainfo_parsers(tmp_path):
    lst = tmp_path / "meta.lst"
    lst.write_text("u1|hello there.|wavs/p1.wav|generate this\nu2|second|/abs/p2.wav|and this|/abs/g2.wav\n")
    m = EB.get_seedtts_testset_metainfo(str(lst))
    assert m[0] == ("u1", "hello there.", str(tmp_path / "wavs/p1.wav"), "generate this", str(tmp_path / "wavs" / "u1.wav"))
    assert m[1] == ("u2", "second", "/abs/p2.wav", "and this", "/abs/g2.wav")
    ls = tmp_path / "ls.lst"
    ls.write_text("1-2-3\t4.0\tRef text.\t5-6-7\t3.0\tGen text.\n")
    assert EB.get_librispeech_test_clean_metainfo(str(ls), "/ls") == [("5-6-7", "Ref text.", "/ls/1/2/1-2-3.flac", " Gen text.", "/ls/5/6/5-6-7.flac")]


@pytest.mark.skipif(not os.path.exists(CLANG), reason="needs the ROCm host clang++")
def test_batch_loop_on_the_host_build_of_the_engine(emu_engine, tmp_path):  # noqa: F811
    """eval_infer_batch.py:179-210 on the product's own classes over the emulated library: the loop equals one ragged sample() per batch
    + per-row slice + vocoder + RMS restore, bit for bit; every utterance comes out once, with the length it has alone."""
    from f5_tts_amd import config, synth
    from f5_tts_amd.engine import F5HipCFM, F5HipVocos

    cfg, vcfg = config.DIT_TINY, config.VOCOS_TINY
    eng = emu_engine(cfg, vcfg)
    eng.load_state_dict({**synth.synth_dit_state_dict(cfg, seed=1), **synth.synth_vocos_state_dict(vcfg, seed=1)})
    vocab = {chr(c): 1 + (c % (cfg.text_num_embeds - 2)) for c in range(32, 127)}
    model, voc = F5HipCFM(eng, precision="fp32", vocab_char_map=vocab), F5HipVocos(eng)
    meta = []
    for i, (secs, words) in enumerate([(0.45, 3), (0.5, 4), (0.47, 3), (0.9, 6)]):
        pw = tmp_path / f"p{i}.wav"
        write_wav(pw, int(secs * 24000), 24000, i, 0.02 if i == 1 else 0.2)
        meta.append((f"u{i}", "ab cd.", str(pw), " ".join(["w"] * words), ""))
    mel_fn = lambda wav: model.mel_spec(wav)  # noqa: E731  [1, n] -> [1, 100, T]
    kw = dict(tokenizer="char", min_secs=0, max_secs=3, num_buckets=3, hop_length=HOP)
    batches = EB.get_inference_prompt(meta, mel_fn, infer_batch_size=150, **kw)
    singles = EB.get_inference_prompt(meta, mel_fn, infer_batch_size=1, **kw)
    assert any(len(p[0]) > 1 for p in batches) and all(len(p[0]) == 1 for p in singles)
    run = dict(nfe_step=2, cfg_strength=2.0, sway_sampling_coef=-1.0, seed=3)
    got = dict(EB.run_prompt_batches(model, voc, batches, **run))
    alone = dict(EB.run_prompt_batches(model, voc, singles, **run))
    assert set(got) == set(alone) == {m[0] for m in meta}
    for u in got:
        assert got[u].shape == alone[u].shape and got[u].shape[0] == 1 and got[u].shape[1] > 0
    # the loop is exactly: one ragged sample() per batch, the generated slice of every row through the vocoder, RMS restored
    for utts, rms, mels, lens, totals, texts in batches:
        out, _ = model.sample(cond=mels, text=texts, duration=torch.tensor(totals), lens=torch.tensor(lens), steps=2, cfg_strength=2.0,
                              sway_sampling_coef=-1.0, seed=3)
        for i, u in enumerate(utts):
            w = voc.decode(out[i, lens[i]:totals[i]].unsqueeze(0).permute(0, 2, 1)).cpu()
            if rms[i] < 0.1:
                w = w * rms[i] / 0.1
            assert torch.equal(w, got[u]), u
    assert any(float(r) < 0.1 for p in batches for r in p[1])  # the quiet prompt went through the restore
    # (a batched row is NOT bit-equal to the row alone: without attn_mask_enabled the padded frames of shorter rows are keys too —
    # the reference's behaviour, dit.py:171-192 — which is one more reason to batch within a length class; they stay close)
    for u in got:
        assert (got[u] - alone[u]).abs().max().item() < 5e-2 * max(1e-3, alone[u].abs().max().item()) + 5e-3, u
