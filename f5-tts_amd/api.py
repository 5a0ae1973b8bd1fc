"""``F5TTS`` — the reference's one-class Python API (reference ``src/f5_tts/api.py:23-149``) on top of the HIP engine.

Same constructor / ``infer`` / ``export_wav`` / ``export_spectrogram`` surface.  Differences, all forced by the offline box and stated
where they bite: the architecture comes from ``config.PRESETS`` (the reference reads the same numbers from its Hydra yaml,
``api.py:35-40``); checkpoints are never downloaded (``api.py:77-80``): pass ``ckpt_file`` / ``vocoder_local_path`` (or in-memory state
dicts); ``transcribe`` needs a caller-supplied ASR callable; ``.wav`` files are written with the standard library as 16-bit PCM
(the reference uses ``soundfile``); the spectrogram picture needs matplotlib, otherwise the array is saved as ``.npy``.
"""
from __future__ import annotations

import os
import random
import sys
import wave
from typing import Callable, Dict, Optional

import numpy as np
import torch

from . import infer as I
from .config import PRESETS, VOCOS_MEL_24K, VocosConfig
from .refaudio import PcmSegment, preprocess_ref_audio_text, split_on_silence


def seed_everything(seed: int = 0) -> None:
    """reference src/f5_tts/model/utils.py:19-26 (the cudnn switches have no counterpart here)."""
    random.seed(seed)
    os.environ["PYTHONHASHSEED"] = str(seed)
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)


def remove_silence_for_generated_wav(filename: str) -> None:
    """reference utils_infer.py:599-608: drop every pause longer than 1 s (keeping 0.5 s on each side), in place."""
    aseg = PcmSegment.from_file(filename)
    out = PcmSegment.silent(duration=0)
    for seg in split_on_silence(aseg, min_silence_len=1000, silence_thresh=-50, keep_silence=500, seek_step=10):
        out = out + seg
    out.export(filename, format="wav")


def save_spectrogram(spectrogram, path: str) -> str:
    """reference utils_infer.py:614-619; without matplotlib the mel is saved next to the requested path as .npy."""
    try:
        import matplotlib

        matplotlib.use("Agg")
        import matplotlib.pyplot as plt
    except Exception:
        out = os.path.splitext(path)[0] + ".npy"
        np.save(out, np.asarray(spectrogram))
        return out
    plt.figure(figsize=(12, 4))
    plt.imshow(spectrogram, origin="lower", aspect="auto")
    plt.colorbar()
    plt.savefig(path)
    plt.close()
    return path


class F5TTS:
    def __init__(self, model: str = "F5TTS_v1_Base", ckpt_file: str = "", vocab_file: str = "", ode_method: str = "euler", use_ema: bool = True,
                 vocoder_local_path: Optional[str] = None, device=None, hf_cache_dir=None, *, precision: str = "fp16m",
                 mel_spec_type: str = "vocos", vocos_cfg: VocosConfig = VOCOS_MEL_24K,
                 state_dict: Optional[Dict[str, torch.Tensor]] = None, vocoder_state_dict: Optional[Dict[str, torch.Tensor]] = None,
                 transcribe: Optional[Callable[[str], str]] = None, bigvgan_cfg=None):
        if model not in PRESETS:
            raise ValueError(f"unknown model {model!r}; known: {sorted(PRESETS)}")
        self.mel_spec_type = mel_spec_type
        self.target_sample_rate = I.target_sample_rate
        self.ode_method, self.use_ema = ode_method, use_ema
        self.device = 0 if device is None else device  # the engine only runs on a HIP device (reference api.py:45-58 falls back to cpu)
        self._transcribe = transcribe
        if not ckpt_file and state_dict is None:
            raise ValueError("no network here: pass ckpt_file (the reference would download hf://SWivid/..., api.py:77-80)")
        if mel_spec_type == "vocos" and vocoder_local_path is None and vocoder_state_dict is None:
            raise ValueError("no network here: pass vocoder_local_path (a directory with pytorch_model.bin, utils_infer.py:110-118)")
        self.ema_model = I.load_model(model, ckpt_file or None, mel_spec_type, vocab_file, ode_method, use_ema, self.device, precision=precision,
                                      vocos_cfg=vocos_cfg if mel_spec_type == "vocos" else None, state_dict=state_dict)
        if mel_spec_type == "vocos":
            self.vocoder = I.load_vocoder("vocos", vocoder_local_path is not None, vocoder_local_path or "", engine=self.ema_model.engine,
                                          state_dict=vocoder_state_dict)
        elif vocoder_local_path is not None or vocoder_state_dict is not None:  # reference api.py:60-62 -> load_vocoder("bigvgan", ...)
            self.vocoder = I.load_vocoder("bigvgan", vocoder_local_path is not None, vocoder_local_path or "", state_dict=vocoder_state_dict,
                                          device=self.device, precision=precision, bigvgan_cfg=bigvgan_cfg)
        else:  # no generator weights given: the caller assigns .vocoder (any module with the reference's `vocoder(mel)` call)
            self.vocoder = None
        self.seed = None

    def transcribe(self, ref_audio, language=None):
        if self._transcribe is None:
            raise ValueError("the Whisper ASR pipeline of the reference (utils_infer.py:150-186) is not available offline: pass transcribe=")
        return self._transcribe(ref_audio)

    def export_wav(self, wav, file_wave: str, remove_silence: bool = False) -> None:
        pcm = (np.clip(np.asarray(wav, dtype=np.float32), -1.0, 1.0) * 32767.0).astype("<i2")
        with wave.open(file_wave, "wb") as w:
            w.setnchannels(1)
            w.setsampwidth(2)
            w.setframerate(self.target_sample_rate)
            w.writeframes(pcm.tobytes())
        if remove_silence:
            remove_silence_for_generated_wav(file_wave)

    def export_spectrogram(self, spec, file_spec: str) -> str:
        return save_spectrogram(spec, file_spec)

    def infer(self, ref_file, ref_text, gen_text, show_info=print, progress=None, target_rms=0.1, cross_fade_duration=0.15,
              sway_sampling_coef=-1, cfg_strength=2, nfe_step=32, speed=1.0, fix_duration=None, remove_silence=False, file_wave=None,
              file_spec=None, seed=None, vocoder=None):
        """One request, the call surface of reference api.py:98-149: clip / transcribe the prompt, synthesise chunk by chunk, optionally write
        the wave and the spectrogram.  Returns ``(wave, sample_rate, mel)``.  Unlike the reference, the seed is also handed to every chunk's
        ``sample`` call — the chunks of one request run on a thread pool, and the global generator alone would not make them reproducible."""
        self.seed = self._reseed(seed)
        prompt_file, prompt_text = preprocess_ref_audio_text(ref_file, ref_text, show_info=show_info, transcribe=self._transcribe)
        sampler_options = dict(target_rms=target_rms, cross_fade_duration=cross_fade_duration, nfe_step=nfe_step, cfg_strength=cfg_strength,
                               sway_sampling_coef=sway_sampling_coef, speed=speed, fix_duration=fix_duration)
        wav, sr, spec = I.infer_process(prompt_file, prompt_text, gen_text, self.ema_model, self.vocoder if vocoder is None else vocoder,
                                        self.mel_spec_type, show_info=show_info, progress=progress, device=self.device,
                                        seed=self.seed % (2**63), **sampler_options)
        self._write_outputs(wav, spec, file_wave, file_spec, remove_silence)
        return wav, sr, spec

    @staticmethod
    def _reseed(seed: Optional[int]) -> int:
        """A caller-given seed, or a fresh one; either way every generator is reset to it (reference api.py:117-120)."""
        chosen = random.randint(0, sys.maxsize) if seed is None else seed
        seed_everything(chosen)
        return chosen

    def _write_outputs(self, wav, spec, file_wave, file_spec, remove_silence: bool) -> None:
        for path, write in ((file_wave, lambda p: self.export_wav(wav, p, remove_silence)), (file_spec, lambda p: self.export_spectrogram(spec, p))):
            if path is not None:
                write(path)
