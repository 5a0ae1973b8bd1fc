"""CPU tests of the reference-audio preparation (f5-tts_amd/refaudio.py; reference utils_infer.py:279-378).  pydub is absent, so the
checks are semantic: silence detection on synthetic tone / silence layouts, the 12 s clipping rules, edge trimming, the cache and the
transcript punctuation rule."""
import os
import sys
import wave

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import f5_tts_amd  # noqa: E402,F401
from f5_tts_amd import refaudio as R  # noqa: E402

SR = 24000


def tone(seconds, amp=0.3, f=220.0):
    t = np.arange(int(seconds * SR)) / SR
    return (amp * 32767 * np.sin(2 * np.pi * f * t)).astype(np.int16)


def silence(seconds):
    return np.zeros(int(seconds * SR), np.int16)


def write_wav(path, pieces):
    a = np.concatenate(pieces)
    with wave.open(str(path), "wb") as w:
        w.setnchannels(1); w.setsampwidth(2); w.setframerate(SR); w.writeframes(a.tobytes())
    return str(path)


def read_len_ms(path):
    with wave.open(path, "rb") as w:
        return 1000.0 * w.getnframes() / w.getframerate()


def test_segment_semantics():
    seg = R.PcmSegment(np.concatenate([silence(0.5), tone(1.0), silence(0.25)])[:, None], SR)
    assert len(seg) == 1750 and len(seg[100:350]) == 250 and len(seg[-250:]) == 250
    assert seg[:500].dBFS == -float("inf") and abs(seg[500:1500].dBFS - 20 * np.log10(0.3 / np.sqrt(2))) < 0.05
    assert R.detect_leading_silence(seg, -42) == 500 and R.detect_leading_silence(seg.reverse(), -42) == 250
    trimmed = R.remove_silence_edges(seg)
    assert len(trimmed) == 1000
    assert R.detect_nonsilent(seg, min_silence_len=200, silence_thresh=-40, seek_step=10) == [[500, 1500]]
    assert R.detect_silence(R.PcmSegment(tone(1.0)[:, None], SR), 100, -40, 10) == []


def test_split_on_silence_keeps_padding_and_splits_overlap():
    seg = R.PcmSegment(np.concatenate([tone(1.0), silence(1.5), tone(1.0)])[:, None], SR)
    parts = R.split_on_silence(seg, min_silence_len=1000, silence_thresh=-50, keep_silence=1000, seek_step=10)
    assert len(parts) == 2
    # the two 1 s pads overlap inside the 1.5 s gap: the gap is split in the middle
    assert len(parts[0]) == 1750 and len(parts[1]) == 1750


def test_preprocess_short_clip_is_trimmed_and_padded(tmp_path):
    p = write_wav(tmp_path / "a.wav", [silence(0.4), tone(3.0), silence(0.7)])
    msgs = []
    out, text = R.preprocess_ref_audio_text(p, "Some call me nature", show_info=msgs.append)
    assert abs(read_len_ms(out) - 3050) <= 20  # edges trimmed (10 ms steps), 50 ms of silence appended
    assert text == "Some call me nature. "
    out2, text2 = R.preprocess_ref_audio_text(p, "ends with a dot.", show_info=msgs.append)
    assert out2 == out and any("cached" in m for m in msgs) and text2 == "ends with a dot. "
    assert R.preprocess_ref_audio_text(p, "中文。", show_info=msgs.append)[1] == "中文。"


def test_preprocess_clips_long_audio_at_a_long_silence(tmp_path):
    # 7 s speech, 1.5 s pause, 7 s speech: the second piece would take the clip past 12 s with more than 6 s collected -> rule (1)
    p = write_wav(tmp_path / "b.wav", [tone(7.0), silence(1.5), tone(7.0)])
    msgs = []
    out, _ = R.preprocess_ref_audio_text(p, "x", show_info=msgs.append)
    assert any("(1)" in m for m in msgs)
    assert 6900 <= read_len_ms(out) <= 7200


def test_preprocess_falls_back_to_short_silences_then_hard_cut(tmp_path):
    # no 1 s pause anywhere: rule (1) returns everything (> 12 s), rule (2) finds the 0.3 s pauses
    p = write_wav(tmp_path / "c.wav", [tone(6.5), silence(0.3), tone(6.5), silence(0.3), tone(3.0)])
    msgs = []
    out, _ = R.preprocess_ref_audio_text(p, "x", show_info=msgs.append)
    assert any("(2)" in m for m in msgs) and read_len_ms(out) <= 12100
    # one uninterrupted 14 s tone: nothing to split on -> hard cut at 12 s
    q = write_wav(tmp_path / "d.wav", [tone(14.0)])
    msgs = []
    out, _ = R.preprocess_ref_audio_text(q, "x", show_info=msgs.append)
    assert any("(3)" in m for m in msgs) and abs(read_len_ms(out) - 12050) <= 20


def test_missing_transcript_needs_a_transcriber(tmp_path):
    p = write_wav(tmp_path / "e.wav", [tone(1.0)])
    with pytest.raises(ValueError):
        R.preprocess_ref_audio_text(p, "  ", show_info=lambda m: None)
    assert R.preprocess_ref_audio_text(p, "", show_info=lambda m: None, transcribe=lambda path: "hello world")[1] == "hello world. "


def test_api_helpers_cpu(tmp_path):
    """The CPU-side pieces of the F5TTS API class (reference api.py): constructor argument errors, wav export, silence removal."""
    from f5_tts_amd import api as A

    with pytest.raises(ValueError):
        A.F5TTS(model="nope", ckpt_file="x")
    with pytest.raises(ValueError):
        A.F5TTS(model="tiny")  # no checkpoint and no network
    tts = object.__new__(A.F5TTS)
    tts.target_sample_rate = SR
    wav = np.concatenate([tone(1.0), silence(2.5), tone(1.0)]).astype(np.float32) / 32768.0
    out = str(tmp_path / "o.wav")
    tts.export_wav(wav, out)
    assert abs(read_len_ms(out) - 4500) < 1
    tts.export_wav(wav, out, remove_silence=True)  # the 2.5 s pause shrinks to 0.5 s + 0.5 s of kept silence
    assert abs(read_len_ms(out) - 3000) <= 20
    saved = tts.export_spectrogram(np.zeros((100, 7), np.float32), str(tmp_path / "s.png"))
    assert os.path.isfile(saved)
