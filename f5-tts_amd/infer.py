"""Host-side mirror of the reference's inference glue around the hot path.

Same names, argument meaning and call-site behaviour as ``src/f5_tts/infer/utils_infer.py`` so that ``api.F5TTS`` / the CLI keep
working when ``load_model`` / ``load_vocoder`` hand back the HIP adapters:

* ``load_model`` / ``load_checkpoint`` / ``load_vocoder`` / ``get_tokenizer`` — reference ``utils_infer.py:106-145,190-276``,
  ``model/utils.py:112-142``: checkpoint files (.safetensors / .pt, EMA key mapping) -> packed device blob.
* ``chunk_text`` (``utils_infer.py:73-102``), ``convert_char_to_pinyin`` (``model/utils.py:148-185``).
* ``infer_process`` / ``infer_batch_process`` (``utils_infer.py:384-593``): RMS normalisation, resampling, per-chunk duration
  heuristic, ``sample`` -> slice at ``ref_audio_len = n_samples // hop`` -> ``vocoder.decode`` -> RMS restore, thread-pool fan-out,
  cross-fade, streaming generator.

No numerics of the hot path live here: every ``sample`` / ``decode`` call goes to ``libf5hip.so`` through ``engine.py``.
``preprocess_ref_audio_text`` / ``remove_silence_edges`` (``utils_infer.py:279-378``) live in ``refaudio.py`` (pydub's silence logic
restated on PCM .wav files; the Whisper ASR fallback is a caller-supplied callable) and are re-exported here.
"""
from __future__ import annotations

import math
import os
import re
import wave
from concurrent.futures import ThreadPoolExecutor
from typing import Dict, Iterable, List, Optional, Sequence, Tuple

import numpy as np
import torch

from .config import HOP_LENGTH, PRESETS, TARGET_SAMPLE_RATE, VOCOS_MEL_24K, DiTConfig, VocosConfig
from .engine import F5HipCFM, F5HipEngine, F5HipVocos, filter_vocos_keys, map_checkpoint_keys
from .refaudio import preprocess_ref_audio_text, remove_silence_edges  # noqa: F401  (same import surface as utils_infer.py)

# defaults of the reference module (utils_infer.py:52-65)
target_sample_rate = TARGET_SAMPLE_RATE
hop_length = HOP_LENGTH
target_rms = 0.1
cross_fade_duration = 0.15
ode_method = "euler"
nfe_step = 32
cfg_strength = 2.0
sway_sampling_coef = -1.0
speed = 1.0
fix_duration = None


# ---- text ---------------------------------------------------------------------------------------------------------------------
_SENTENCE_SPLIT = re.compile(r"(?<=[;:,.!?])\s+|(?<=[；：，。！？])")


def _nbytes(s: str) -> int:
    return len(s.encode("utf-8"))


def chunk_text(text: str, max_chars: int = 135) -> List[str]:
    """Greedy sentence packing by UTF-8 byte budget (reference utils_infer.py:73-102): split after ;:,.!? + whitespace or after a
    full-width punctuation mark; a piece ending in a single-byte character gets a trailing space."""
    chunks: List[str] = []
    cur = ""
    for sent in _SENTENCE_SPLIT.split(text):
        if not sent:
            continue
        piece = sent + " " if _nbytes(sent[-1]) == 1 else sent
        if _nbytes(cur) + _nbytes(sent) <= max_chars:
            cur += piece
        else:
            if cur:
                chunks.append(cur.strip())
            cur = piece
    if cur:
        chunks.append(cur.strip())
    return chunks


_OOV_TRANS = str.maketrans({";": ",", "“": '"', "”": '"', "‘": "'", "’": "'"})
_ASCII_TOKENS = re.compile(r"[A-Za-z0-9]+|.", re.S)


def convert_char_to_pinyin(text_list: Sequence[str], polyphone: bool = True) -> List[List[str]]:
    """reference model/utils.py:148-185.  Pure single-byte text needs no G2P: segments are character runs, and a multi-character
    segment that follows anything but space/colon/quote gets a separating space.  East-asian text needs ``rjieba`` + ``pypinyin``
    (absent offline): used when importable, otherwise a ValueError says so."""
    try:  # pragma: no cover - optional dependencies
        import rjieba
        from pypinyin import Style, lazy_pinyin
    except Exception:
        rjieba = None
    out: List[List[str]] = []
    for text in text_list:
        text = text.translate(_OOV_TRANS)
        chars: List[str] = []
        if rjieba is None:
            if _nbytes(text) != len(text):
                raise ValueError("non-single-byte text needs the rjieba and pypinyin packages (text front-end, SURVEY.md 8f rank 4)")
            segs: Iterable[str] = _ASCII_TOKENS.findall(text)
        else:  # pragma: no cover
            segs = rjieba.cut(text)
        for seg in segs:
            nb = _nbytes(seg)
            if nb == len(seg):  # single-byte characters only
                if chars and nb > 1 and chars[-1] not in " :'\"":
                    chars.append(" ")
                chars.extend(seg)
            elif polyphone and nb == 3 * len(seg):  # pragma: no cover
                py = lazy_pinyin(seg, style=Style.TONE3, tone_sandhi=True)
                for i, c in enumerate(seg):
                    if "㄀" <= c <= "鿿":
                        chars.append(" ")
                    chars.append(py[i])
            else:  # pragma: no cover
                for c in seg:
                    if ord(c) < 256:
                        chars.extend(c)
                    elif "㄀" <= c <= "鿿":
                        chars.append(" ")
                        chars.extend(lazy_pinyin(c, style=Style.TONE3, tone_sandhi=True))
                    else:
                        chars.append(c)
        out.append(chars)
    return out


def get_tokenizer(vocab_file: str, tokenizer: str = "custom") -> Tuple[Optional[Dict[str, int]], int]:
    """reference model/utils.py:112-142 ("custom": path to vocab.txt, one token per line, line index = id; "byte": 256)."""
    if tokenizer == "byte":
        return None, 256
    if tokenizer != "custom":
        raise ValueError("pass the vocab.txt path with tokenizer='custom' (the packaged dataset vocabularies are not shipped here)")
    with open(vocab_file, "r", encoding="utf-8") as f:
        vocab = {line[:-1]: i for i, line in enumerate(f)}
    return vocab, len(vocab)


# ---- weights ------------------------------------------------------------------------------------------------------------------
def _read_checkpoint(path: str) -> Dict[str, torch.Tensor]:
    if path.endswith(".safetensors"):
        from safetensors.torch import load_file

        return load_file(path, device="cpu")
    return torch.load(path, map_location="cpu", weights_only=True)


def load_checkpoint(engine: F5HipEngine, ckpt_path: str, use_ema: bool = True, finalize: bool = False) -> F5HipEngine:
    """reference utils_infer.py:190-232: .safetensors or .pt; EMA checkpoints carry ``ema_model.``-prefixed keys plus ``initted`` /
    ``step``; legacy mel-STFT buffers are dropped.  Only ``transformer.*`` tensors exist in the engine (the mel front-end has no
    learned state)."""
    ckpt = _read_checkpoint(ckpt_path)
    if ckpt_path.endswith(".safetensors") and use_ema:
        ckpt = {"ema_model_state_dict": ckpt}
    elif ckpt_path.endswith(".safetensors"):
        ckpt = {"model_state_dict": ckpt}
    sd = map_checkpoint_keys(ckpt, use_ema=use_ema)
    sd = {k: v for k, v in sd.items() if k.startswith("transformer.")}
    engine.load_state_dict(sd, strict=True, finalize=finalize)
    return engine


def load_model(model_cfg, ckpt_path: Optional[str], mel_spec_type: str = "vocos", vocab_file: str = "", ode_method: str = ode_method,
               use_ema: bool = True, device=0, precision: str = "fp16m", vocos_cfg: Optional[VocosConfig] = VOCOS_MEL_24K,
               state_dict: Optional[Dict[str, torch.Tensor]] = None, model_cls=None) -> F5HipCFM:
    """reference utils_infer.py:238-276 — ``load_model(model_cls, model_cfg, ckpt_path, ...)``.  The reference's leading ``model_cls``
    argument (the backbone CLASS: DiT / UNetT / MMDiT) is the keyword ``model_cls`` here — a class or its name; it is REQUIRED when
    ``model_cfg`` is the reference's ``model.arch`` dict, which does not say which backbone it describes (and UNetT / MMDiT have
    constructor defaults — text_dim = mel_dim, no text conv layers — that a DiT default would silently get wrong).  ``model_cfg`` may
    also be a preset name ("F5TTS_v1_Base", "F5TTS_Base", "E2TTS_Base") or a ``DiTConfig``.  The returned object quacks like the
    reference's ``CFM`` on the inference path; with ``mel_spec_type="vocos"`` its engine also hosts the vocoder
    (``load_vocoder(engine=model.engine, ...)``); with ``"bigvgan"`` the generator is its own context and this one is finalised here."""
    if mel_spec_type not in ("vocos", "bigvgan"):
        raise ValueError("mel_spec_type must be vocos or bigvgan (modules.py:127)")
    if mel_spec_type == "bigvgan":
        vocos_cfg = None  # no Vocos tensors to wait for: load_vocoder("bigvgan") builds a separate F5HipBigVGAN context
    vocab_char_map, vocab_size = (get_tokenizer(vocab_file, "custom") if vocab_file else (None, None))
    if isinstance(model_cfg, str):
        cfg = PRESETS[model_cfg]
    elif isinstance(model_cfg, DiTConfig):
        cfg = model_cfg
    else:
        name = model_cls if isinstance(model_cls, str) or model_cls is None else getattr(model_cls, "__name__", str(model_cls))
        if name not in ("DiT", "UNetT", "MMDiT"):
            raise ValueError("load_model(model.arch dict): pass model_cls='DiT' | 'UNetT' | 'MMDiT' (the reference's first argument, "
                             "utils_infer.py:238) — the arch dict alone does not name its backbone")
        fields = DiTConfig.__dataclass_fields__
        arch = {k: v for k, v in dict(model_cfg).items() if k in fields}
        mel_dim = arch.get("mel_dim", 100)
        if name == "UNetT":  # unett.py:108-128: text_dim defaults to mel_dim, there are no ConvNeXt text layers
            arch = {"text_dim": mel_dim, "conv_layers": 0, **arch}
        elif name == "MMDiT":  # mmdit.py:94-110: text embedded at model width, no ConvNeXt text layers
            arch = {"text_dim": arch.get("dim", 1024), "conv_layers": 0, **arch}
        cfg = DiTConfig(**{**arch, "backbone": name})
    if vocab_size is not None and vocab_size != cfg.text_num_embeds:
        from dataclasses import replace

        cfg = replace(cfg, text_num_embeds=vocab_size)
    engine = F5HipEngine(cfg, vocos_cfg, device=device)
    if state_dict is not None:
        engine.load_state_dict({k: v for k, v in state_dict.items() if k.startswith("transformer.")}, finalize=False)
    elif ckpt_path:
        load_checkpoint(engine, ckpt_path, use_ema=use_ema)
    if vocos_cfg is None:
        engine.finalize()
    # mel_spec_type="bigvgan": the BigVGAN-type mel front-end runs on this engine; the generator is its own context
    # (load_vocoder("bigvgan") -> F5HipBigVGAN) and infer_batch_process calls it as ``vocoder(mel)`` (utils_infer.py:512-513)
    return F5HipCFM(engine, vocab_char_map=vocab_char_map, ode_method=ode_method, precision=precision, mel_spec_type=mel_spec_type)


def load_vocoder(vocoder_name: str = "vocos", is_local: bool = True, local_path: str = "", engine: Optional[F5HipEngine] = None,
                 state_dict: Optional[Dict[str, torch.Tensor]] = None, device=0, precision: str = "fp16m", bigvgan_cfg=None, **_ignored):
    """reference utils_infer.py:106-145.

    ``"vocos"`` (:107-129): ``<local_path>/pytorch_model.bin`` (config.yaml is the fixed charactr/vocos-mel-24khz architecture =
    ``VOCOS_MEL_24K``).  The Vocos vocoder lives in the same context as the backbone: pass ``engine=model.engine``; loading its tensors
    finalises the context.
    ``"bigvgan"`` (:130-144): ``<local_path>/config.json`` + ``bigvgan_generator.pt`` (the files of nvidia/bigvgan_v2_24khz_100band_256x
    that the reference downloads), weight norm folded at load (the reference's ``remove_weight_norm()``), its own HIP context
    (``F5HipBigVGAN``); ``state_dict`` + ``bigvgan_cfg`` instead of files for synthetic weights."""
    if vocoder_name == "bigvgan":
        from .bigvgan import F5HipBigVGAN
        from .config import BIGVGAN_V2_24K_100B_256X

        if state_dict is not None:
            return F5HipBigVGAN(bigvgan_cfg or BIGVGAN_V2_24K_100B_256X, device=device, precision=precision).load_state_dict(state_dict)
        if not is_local:
            raise ValueError("no network here: pass is_local=True and local_path")
        return F5HipBigVGAN.from_pretrained(local_path, use_cuda_kernel=False, device=device, precision=precision)
    if vocoder_name != "vocos":
        raise ValueError("vocoder_name must be vocos or bigvgan (utils_infer.py:107,130)")
    if engine is None:
        raise ValueError("pass engine=model.engine (one HIP context hosts backbone + vocoder)")
    if state_dict is None:
        if not is_local:
            raise ValueError("no network here: pass is_local=True and local_path")
        state_dict = torch.load(os.path.join(local_path, "pytorch_model.bin"), map_location="cpu", weights_only=True)
    engine.load_state_dict(filter_vocos_keys(state_dict), strict=False, finalize=True)
    return F5HipVocos(engine)


# ---- audio helpers ------------------------------------------------------------------------------------------------------------
def load_wav(path: str) -> Tuple[torch.Tensor, int]:
    """16-bit / 32-bit PCM .wav -> float32 [channels, samples] in [-1, 1) (what torchaudio.load returns, utils_infer.py:403)."""
    with wave.open(path, "rb") as w:
        sr, ch, width, n = w.getframerate(), w.getnchannels(), w.getsampwidth(), w.getnframes()
        raw = w.readframes(n)
    if width == 2:
        a = np.frombuffer(raw, dtype="<i2").astype(np.float32) / 32768.0
    elif width == 4:
        a = np.frombuffer(raw, dtype="<i4").astype(np.float32) / 2147483648.0
    else:
        raise ValueError(f"unsupported sample width {width}")
    return torch.from_numpy(a.reshape(-1, ch).T.copy()), sr


def resample(wave_: torch.Tensor, orig_freq: int, new_freq: int, lowpass_filter_width: int = 6, rolloff: float = 0.99) -> torch.Tensor:
    """Windowed-sinc polyphase resampling with torchaudio's defaults (``transforms.Resample``: sinc_interp_hann, width 6, rolloff
    0.99; call site utils_infer.py:466-468).  Restated from the published algorithm — unpinned here (torchaudio is absent)."""
    if orig_freq == new_freq:
        return wave_
    g = math.gcd(int(orig_freq), int(new_freq))
    orig, new = int(orig_freq) // g, int(new_freq) // g
    base = min(orig, new) * rolloff
    width = math.ceil(lowpass_filter_width * orig / base)
    idx = torch.arange(-width, width + orig, dtype=torch.float64)[None, None] / orig
    t = torch.arange(0, -new, -1, dtype=torch.float64)[:, None, None] / new + idx
    t = (t * base).clamp(-lowpass_filter_width, lowpass_filter_width)
    window = torch.cos(t * math.pi / lowpass_filter_width / 2) ** 2
    t = t * math.pi
    kernel = torch.where(t == 0, torch.ones_like(t), t.sin() / t) * window * (base / orig)
    kernel = kernel.to(torch.float32)
    shape = wave_.shape
    x = wave_.reshape(-1, shape[-1])
    x = torch.nn.functional.pad(x, (width, width + orig))
    y = torch.nn.functional.conv1d(x[:, None], kernel, stride=orig)
    y = y.transpose(1, 2).reshape(x.shape[0], -1)
    target = math.ceil(new * shape[-1] / orig)
    return y[..., :target].reshape(shape[:-1] + (target,))


def cross_fade_concat(waves: Sequence[np.ndarray], fade_seconds: float, sr: int = TARGET_SAMPLE_RATE) -> np.ndarray:
    """Linear cross-fade of consecutive chunks (reference utils_infer.py:549-586)."""
    if fade_seconds <= 0:
        return np.concatenate(waves)
    final = waves[0]
    for nxt in waves[1:]:
        k = min(int(fade_seconds * sr), len(final), len(nxt))
        if k <= 0:
            final = np.concatenate([final, nxt])
            continue
        mixed = final[-k:] * np.linspace(1, 0, k) + nxt[:k] * np.linspace(0, 1, k)
        final = np.concatenate([final[:-k], mixed, nxt[k:]])
    return final


# ---- inference ----------------------------------------------------------------------------------------------------------------
def total_frames(prompt_frames: int, prompt_bytes: int, chunk: str, speed: float, fix_duration: Optional[float]) -> int:
    """Length of prompt + generation in mel frames (reference utils_infer.py:487-496): a fixed total when `fix_duration` is given, otherwise
    the prompt's frames-per-byte rate applied to the chunk's UTF-8 length and divided by the speed — which drops to 0.3 for chunks under
    10 bytes so that very short texts are not rushed."""
    if fix_duration is not None:
        return int(fix_duration * target_sample_rate / hop_length)
    chunk_bytes = len(chunk.encode("utf-8"))
    pace = 0.3 if chunk_bytes < 10 else speed
    return prompt_frames + int(prompt_frames / prompt_bytes * chunk_bytes / pace)


# mel -> waveform per mel_spec_type (reference utils_infer.py:510-513): Vocos exposes .decode, a BigVGAN generator is called directly
# (F5HipBigVGAN from load_vocoder("bigvgan"), or any caller-supplied generator)
_VOCODERS = {"vocos": lambda voc, mel: voc.decode(mel), "bigvgan": lambda voc, mel: voc(mel)}


def infer_batch_process(ref_audio, ref_text: str, gen_text_batches: Sequence[str], model_obj, vocoder, mel_spec_type: str = "vocos",
                        progress=None, target_rms: float = target_rms, cross_fade_duration: float = cross_fade_duration,
                        nfe_step: int = nfe_step, cfg_strength: float = cfg_strength, sway_sampling_coef: Optional[float] = sway_sampling_coef,
                        speed: float = speed, fix_duration: Optional[float] = None, device=None, streaming: bool = False,
                        chunk_size: int = 2048, seed: Optional[int] = None):
    """Generator with the contract of reference utils_infer.py:440-593.  ``seed`` is an extension: the reference seeds torch globally
    in ``api.F5TTS.infer`` (``seed_everything``, api.py:117-120) and ``sample`` then draws from that stream; passing it here makes a
    multi-chunk request deterministic under the thread pool."""
    audio, sr = ref_audio
    audio = torch.as_tensor(audio, dtype=torch.float32)
    if audio.ndim == 1:
        audio = audio[None]
    if audio.shape[0] > 1:
        audio = audio.mean(dim=0, keepdim=True)
    rms = torch.sqrt(torch.mean(torch.square(audio)))
    if rms < target_rms:
        audio = audio * target_rms / rms
    if sr != target_sample_rate:
        audio = resample(audio, sr, target_sample_rate)
    audio = audio.to(model_obj.device)
    if len(ref_text[-1].encode("utf-8")) == 1:
        ref_text = ref_text + " "

    prompt_frames = audio.shape[-1] // hop_length
    prompt_bytes = len(ref_text.encode("utf-8"))
    quiet_prompt = bool(rms < target_rms)  # the prompt was amplified to target_rms above: the output is brought back to its level
    vocode = _VOCODERS.get(mel_spec_type)
    if vocode is None:
        raise ValueError("mel_spec_type must be vocos or bigvgan")

    def synthesize(chunk: str):
        """One text chunk -> (waveform as numpy, generated mel [1, n_mel, frames]) — the body of the reference's `process_batch`."""
        total = total_frames(prompt_frames, prompt_bytes, chunk, speed, fix_duration)
        mel, _ = model_obj.sample(cond=audio, text=convert_char_to_pinyin([ref_text + chunk]), duration=total, steps=nfe_step,
                                  cfg_strength=cfg_strength, sway_sampling_coef=sway_sampling_coef, seed=seed)
        gen = mel.to(torch.float32)[:, prompt_frames:, :].permute(0, 2, 1)  # drop the prompt, channels first for the vocoder
        wave = vocode(vocoder, gen)
        if quiet_prompt:
            wave = wave * rms / target_rms
        return wave.squeeze().cpu().numpy(), gen

    seq = progress.tqdm(gen_text_batches) if progress is not None else gen_text_batches
    if streaming:
        for gen_text in seq:
            w, _ = synthesize(gen_text)
            for j in range(0, len(w), chunk_size):
                yield w[j:j + chunk_size], target_sample_rate
        return
    waves, specs = [], []
    with ThreadPoolExecutor() as ex:  # the context serialises its entry points; the pool only overlaps host work
        futures = [ex.submit(synthesize, g) for g in gen_text_batches]
        for fut in (progress.tqdm(futures) if progress is not None else futures):
            w, spec = fut.result()
            waves.append(w)
            specs.append(spec[0].cpu().numpy())
    if not waves:
        yield None, target_sample_rate, None
        return
    yield cross_fade_concat(waves, cross_fade_duration), target_sample_rate, np.concatenate(specs, axis=1)


def infer_process(ref_audio, ref_text: str, gen_text: str, model_obj, vocoder, mel_spec_type: str = "vocos", show_info=print, progress=None,
                  target_rms: float = target_rms, cross_fade_duration: float = cross_fade_duration, nfe_step: int = nfe_step,
                  cfg_strength: float = cfg_strength, sway_sampling_coef: Optional[float] = sway_sampling_coef, speed: float = speed,
                  fix_duration: Optional[float] = None, device=None, seed: Optional[int] = None):
    """reference utils_infer.py:384-434: ``ref_audio`` is a .wav path or an ``(audio[channels, n], sr)`` pair."""
    audio, sr = load_wav(ref_audio) if isinstance(ref_audio, str) else ref_audio
    audio = torch.as_tensor(audio, dtype=torch.float32)
    if audio.ndim == 1:
        audio = audio[None]
    seconds = audio.shape[-1] / sr
    max_chars = int(len(ref_text.encode("utf-8")) / seconds * (22 - seconds) * speed)
    batches = chunk_text(gen_text, max_chars=max_chars)
    show_info(f"Generating audio in {len(batches)} batches...")
    if not batches:
        show_info("No text batches to generate.")
        return None, target_sample_rate, None
    return next(infer_batch_process((audio, sr), ref_text, batches, model_obj, vocoder, mel_spec_type=mel_spec_type, progress=progress,
                                    target_rms=target_rms, cross_fade_duration=cross_fade_duration, nfe_step=nfe_step,
                                    cfg_strength=cfg_strength, sway_sampling_coef=sway_sampling_coef, speed=speed,
                                    fix_duration=fix_duration, device=device, seed=seed))


# ---- speech editing (reference src/f5_tts/infer/speech_edit.py:139-235) --------------------------------------------------------------
def build_edit_condition(original_mel: torch.Tensor, parts_to_edit: Sequence[Sequence[float]], fix_duration: Optional[Sequence[float]] = None,
                         sample_rate: int = target_sample_rate, hop: int = hop_length) -> Tuple[torch.Tensor, torch.Tensor]:
    """Frame-level conditioning of an edit: the original mel with every [start, end] span (seconds) replaced by zero frames of the wanted
    length (``fix_duration`` per span, else the span's own length), and the mask of frames to KEEP (speech_edit.py:154-200).
    original_mel [1, frames, mel] -> (mel_cond [1, frames', mel], edit_mask bool [1, frames'])."""
    dev, mel = original_mel.device, original_mel.shape[-1]
    mel_cond = torch.zeros(1, 0, mel, device=dev)
    edit_mask = torch.zeros(1, 0, dtype=torch.bool, device=dev)
    fix = list(fix_duration) if fix_duration is not None else None
    offset = 0
    for start, end in parts_to_edit:
        dur = end - start if fix is None else fix.pop(0)
        start_f, end_f, dur_f = round(start * sample_rate / hop), round(end * sample_rate / hop), round(dur * sample_rate / hop)
        mel_cond = torch.cat((mel_cond, original_mel[:, offset:start_f, :], torch.zeros(1, dur_f, mel, device=dev)), dim=1)
        edit_mask = torch.cat((edit_mask, torch.ones(1, start_f - offset, dtype=torch.bool, device=dev),
                               torch.zeros(1, dur_f, dtype=torch.bool, device=dev)), dim=-1)
        offset = end_f
    mel_cond = torch.cat((mel_cond, original_mel[:, offset:, :]), dim=1)
    edit_mask = torch.nn.functional.pad(edit_mask, (0, mel_cond.shape[1] - edit_mask.shape[-1]), value=True)
    return mel_cond, edit_mask


def speech_edit(model_obj, vocoder, audio, sr: int, target_text: str, parts_to_edit: Sequence[Sequence[float]],
                fix_duration: Optional[Sequence[float]] = None, mel_spec_type: str = "vocos", nfe_step: int = 32, cfg_strength: float = 2.0,
                sway_sampling_coef: Optional[float] = -1.0, seed: Optional[int] = None, target_rms: float = target_rms,
                tokenizer: str = "pinyin"):
    """The body of the reference's speech_edit.py script as a function: regenerate the given time spans of ``audio`` so that the whole
    utterance reads ``target_text``; everything outside the spans is kept (``edit_mask``).  Returns (wave [1, n], mel [1, mel, frames])."""
    audio = torch.as_tensor(audio, dtype=torch.float32)
    if audio.ndim == 1:
        audio = audio[None]
    if audio.shape[0] > 1:
        audio = audio.mean(dim=0, keepdim=True)
    rms = torch.sqrt(torch.mean(torch.square(audio)))
    if rms < target_rms:
        audio = audio * target_rms / rms
    if sr != target_sample_rate:
        audio = resample(audio, sr, target_sample_rate)
    original_mel = model_obj.mel_spec(audio.to(model_obj.device)).permute(0, 2, 1)  # mel of the clean original first (speech_edit.py:148-152)
    mel_cond, edit_mask = build_edit_condition(original_mel, parts_to_edit, fix_duration)
    text_list = convert_char_to_pinyin([target_text]) if tokenizer == "pinyin" else [[target_text]]
    generated, _ = model_obj.sample(cond=mel_cond, text=text_list, duration=mel_cond.shape[1], steps=nfe_step, cfg_strength=cfg_strength,
                                    sway_sampling_coef=sway_sampling_coef, seed=seed, edit_mask=edit_mask)
    gen_mel = generated.to(torch.float32).permute(0, 2, 1)
    wave_out = vocoder.decode(gen_mel) if mel_spec_type == "vocos" else vocoder(gen_mel).squeeze(0)
    if rms < target_rms:
        wave_out = wave_out * rms / target_rms
    return wave_out.cpu(), gen_mel.cpu()
