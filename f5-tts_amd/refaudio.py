"""Reference-audio preparation of the reference's inference glue: ``preprocess_ref_audio_text`` and ``remove_silence_edges``
(reference ``src/f5_tts/infer/utils_infer.py:279-378``).  Host-only, no kernels.

The reference does this with ``pydub`` (``AudioSegment``, ``silence.split_on_silence``, ``silence.detect_leading_silence``) on top of
ffmpeg, and falls back to a Whisper ASR pipeline when no transcript is given.  Neither package exists offline, so the pieces of pydub the
call sites use are restated here from its published algorithm (pydub 0.25 ``audio_segment.py`` / ``silence.py``) on 16-bit PCM ``.wav``
files read with the standard library — **parity unpinned** (nothing in the reference tree holds pydub outputs); millisecond slicing,
``len()`` rounding, integer RMS (``audioop.rms``) and dBFS follow pydub so the clipping decisions match on the same samples.  ASR is not
built: an empty ``ref_text`` needs a ``transcribe`` callable from the caller.
"""
from __future__ import annotations

import hashlib
import math
import os
import tempfile
import wave
from typing import Callable, Dict, List, Optional, Tuple

import numpy as np


class PcmSegment:
    """The slice of pydub's ``AudioSegment`` the reference uses: 16-bit PCM frames [n, channels] + frame rate, millisecond slicing."""

    def __init__(self, frames: np.ndarray, frame_rate: int):
        assert frames.dtype == np.int16 and frames.ndim == 2
        self.frames, self.frame_rate = frames, int(frame_rate)

    # -- construction -------------------------------------------------------------------------------------------------------------
    @classmethod
    def from_file(cls, path: str) -> "PcmSegment":
        with wave.open(path, "rb") as w:
            sr, ch, width, n = w.getframerate(), w.getnchannels(), w.getsampwidth(), w.getnframes()
            raw = w.readframes(n)
        if width == 2:
            a = np.frombuffer(raw, dtype="<i2")
        elif width == 4:  # pydub keeps the width; the reference's thresholds are relative (dBFS), so 16 bits lose nothing that matters
            a = (np.frombuffer(raw, dtype="<i4") >> 16).astype(np.int16)
        else:
            raise ValueError(f"unsupported sample width {width} (only PCM .wav is read here; the reference goes through ffmpeg)")
        return cls(a.reshape(-1, ch).copy(), sr)

    @classmethod
    def silent(cls, duration: int = 1000, frame_rate: int = 11025, channels: int = 1) -> "PcmSegment":
        return cls(np.zeros((int(frame_rate * (duration / 1000.0)), channels), np.int16), frame_rate)

    # -- pydub semantics ----------------------------------------------------------------------------------------------------------
    def __len__(self) -> int:  # milliseconds, rounded (audio_segment.py: round(1000 * frame_count / frame_rate))
        return int(round(1000.0 * self.frames.shape[0] / self.frame_rate))

    def _pos(self, ms: float) -> int:
        return int(ms * (self.frame_rate / 1000.0))

    def __getitem__(self, sl: slice) -> "PcmSegment":
        n = len(self)
        start = 0 if sl.start is None else sl.start
        end = n if sl.stop is None else sl.stop
        start, end = min(max(start + n if start < 0 else start, 0), n), min(max(end + n if end < 0 else end, 0), n)
        a, b = self._pos(start), self._pos(end)
        return PcmSegment(self.frames[a:max(a, b)], self.frame_rate)

    def __add__(self, other: "PcmSegment") -> "PcmSegment":
        # pydub syncs the two formats before appending; the call sites only ever append to an empty segment or to one of the same
        # format, so: an empty side adopts the other side's format, anything else must match
        if self.frames.shape[0] == 0:
            return PcmSegment(other.frames.copy(), other.frame_rate)
        if other.frames.shape[0] == 0:
            return PcmSegment(self.frames.copy(), self.frame_rate)
        if other.frame_rate != self.frame_rate or other.frames.shape[1] != self.frames.shape[1]:
            raise ValueError("segments of different formats")
        return PcmSegment(np.concatenate([self.frames, other.frames], axis=0), self.frame_rate)

    def reverse(self) -> "PcmSegment":
        return PcmSegment(self.frames[::-1].copy(), self.frame_rate)

    @property
    def rms(self) -> int:  # audioop.rms: integer square root of the mean square over every sample of every channel
        if self.frames.size == 0:
            return 0
        return int(math.sqrt(float(np.mean(self.frames.astype(np.float64) ** 2))))

    max_possible_amplitude = 32768.0

    @property
    def dBFS(self) -> float:
        r = self.rms
        return -float("inf") if r == 0 else 20.0 * math.log10(r / self.max_possible_amplitude)

    def export(self, path: str, format: str = "wav") -> str:
        assert format == "wav"
        with wave.open(path, "wb") as w:
            w.setnchannels(self.frames.shape[1])
            w.setsampwidth(2)
            w.setframerate(self.frame_rate)
            w.writeframes(self.frames.astype("<i2").tobytes())
        return path


# ---- pydub.silence ---------------------------------------------------------------------------------------------------------------
def detect_silence(seg: PcmSegment, min_silence_len: int = 1000, silence_thresh: float = -16, seek_step: int = 1) -> List[List[int]]:
    seg_len = len(seg)
    if seg_len < min_silence_len:
        return []
    thresh = (10.0 ** (silence_thresh / 20.0)) * seg.max_possible_amplitude
    last = seg_len - min_silence_len
    starts = list(range(0, last + 1, seek_step))
    if last % seek_step:
        starts.append(last)
    # window RMS via a prefix sum of squares over all channels (same integer RMS as slicing each window)
    sq = np.concatenate([[0.0], np.cumsum(np.sum(seg.frames.astype(np.float64) ** 2, axis=1))])
    ch = seg.frames.shape[1]
    silent = []
    for i in starts:
        a, b = seg._pos(i), seg._pos(min(i + min_silence_len, seg_len))
        n = (b - a) * ch
        r = int(math.sqrt((sq[b] - sq[a]) / n)) if n > 0 else 0
        if r <= thresh:
            silent.append(i)
    if not silent:
        return []
    ranges = []
    prev = silent.pop(0)
    cur = prev
    for s in silent:
        if s != prev + seek_step and s > prev + min_silence_len:
            ranges.append([cur, prev + min_silence_len])
            cur = s
        prev = s
    ranges.append([cur, prev + min_silence_len])
    return ranges


def detect_nonsilent(seg: PcmSegment, min_silence_len: int = 1000, silence_thresh: float = -16, seek_step: int = 1) -> List[List[int]]:
    silent = detect_silence(seg, min_silence_len, silence_thresh, seek_step)
    n = len(seg)
    if not silent:
        return [[0, n]]
    if silent[0][0] == 0 and silent[0][1] == n:
        return []
    prev_end, out, end = 0, [], 0
    for start, end in silent:
        out.append([prev_end, start])
        prev_end = end
    if end != n:
        out.append([prev_end, n])
    if out[0] == [0, 0]:
        out.pop(0)
    return out


def split_on_silence(seg: PcmSegment, min_silence_len: int = 1000, silence_thresh: float = -16, keep_silence: int = 100,
                     seek_step: int = 1) -> List[PcmSegment]:
    if isinstance(keep_silence, bool):
        keep_silence = len(seg) if keep_silence else 0
    ranges = [[s - keep_silence, e + keep_silence] for s, e in detect_nonsilent(seg, min_silence_len, silence_thresh, seek_step)]
    for a, b in zip(ranges, ranges[1:]):
        if b[0] < a[1]:
            a[1] = (a[1] + b[0]) // 2
            b[0] = a[1]
    return [seg[max(s, 0):min(e, len(seg))] for s, e in ranges]


def detect_leading_silence(sound: PcmSegment, silence_threshold: float = -50.0, chunk_size: int = 10) -> int:
    trim = 0
    while sound[trim:trim + chunk_size].dBFS < silence_threshold and trim < len(sound):
        trim += chunk_size
    return min(trim, len(sound))


# ---- the reference's functions ---------------------------------------------------------------------------------------------------
def remove_silence_edges(audio: PcmSegment, silence_threshold: float = -42) -> PcmSegment:
    """reference utils_infer.py:279-292."""
    audio = audio[detect_leading_silence(audio, silence_threshold):]
    end = detect_leading_silence(audio.reverse(), silence_threshold)
    return audio[:len(audio) - end] if end > 0 else audio


_ref_audio_cache: Dict[str, str] = {}
_ref_text_cache: Dict[str, str] = {}


def _clip_by_silence(aseg: PcmSegment, show_info, min_silence_len: int, silence_thresh: float, tag: str) -> PcmSegment:
    out = PcmSegment.silent(duration=0)
    for piece in split_on_silence(aseg, min_silence_len=min_silence_len, silence_thresh=silence_thresh, keep_silence=1000, seek_step=10):
        if len(out) > 6000 and len(out + piece) > 12000:
            show_info(f"Audio is over 12s, clipping short. ({tag})")
            break
        out = out + piece
    return out


def preprocess_ref_audio_text(ref_audio_orig: str, ref_text: str, show_info=print,
                              transcribe: Optional[Callable[[str], str]] = None) -> Tuple[str, str]:
    """reference utils_infer.py:298-378: clip the prompt to <= 12 s at a silence, trim the silent edges, append 50 ms of silence, write
    a temporary .wav (cached by the md5 of the input file); make the transcript end with sentence punctuation."""
    show_info("Converting audio...")
    with open(ref_audio_orig, "rb") as f:
        audio_hash = hashlib.md5(f.read()).hexdigest()
    if audio_hash in _ref_audio_cache and os.path.isfile(_ref_audio_cache[audio_hash]):
        show_info("Using cached preprocessed reference audio...")
        ref_audio = _ref_audio_cache[audio_hash]
    else:
        aseg = PcmSegment.from_file(ref_audio_orig)
        wave_ = _clip_by_silence(aseg, show_info, 1000, -50, "1")      # 1. long silences
        if len(wave_) > 12000:
            wave_ = _clip_by_silence(aseg, show_info, 100, -40, "2")   # 2. short silences
        aseg = wave_
        if len(aseg) > 12000:                                          # 3. no usable silence
            aseg = aseg[:12000]
            show_info("Audio is over 12s, clipping short. (3)")
        aseg = remove_silence_edges(aseg) + PcmSegment.silent(duration=50, frame_rate=aseg.frame_rate, channels=aseg.frames.shape[1])
        fd, ref_audio = tempfile.mkstemp(suffix=".wav")
        os.close(fd)
        aseg.export(ref_audio, format="wav")
        _ref_audio_cache[audio_hash] = ref_audio
    if not ref_text.strip():
        if audio_hash in _ref_text_cache:
            show_info("Using cached reference text...")
            ref_text = _ref_text_cache[audio_hash]
        else:
            if transcribe is None:
                raise ValueError("no reference text and no `transcribe` callable: the reference's Whisper ASR fallback "
                                 "(utils_infer.py:150-186) needs packages and weights that are not available offline")
            show_info("No reference text provided, transcribing reference audio...")
            ref_text = transcribe(ref_audio)
            _ref_text_cache[audio_hash] = ref_text
    else:
        show_info("Using custom reference text...")
    if not ref_text.endswith(". ") and not ref_text.endswith("。"):
        ref_text += " " if ref_text.endswith(".") else ". "
    return ref_audio, ref_text
