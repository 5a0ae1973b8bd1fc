"""Length-bucketed batch formation and the batch inference loop of the reference's evaluation driver, for the HIP engine.

Mirrors reference ``src/f5_tts/eval/utils_eval.py``: the test-list parsers (:19-52), ``padded_mel_batch`` (:56-66) and
``get_inference_prompt`` (:72-205) — utterances are dropped into ``num_buckets`` length classes of their TOTAL (prompt + generated)
frame count, a class is flushed as one batch as soon as its frames reach ``infer_batch_size`` (a frame budget, not an utterance count),
leftovers become batches at the end and the batch list is shuffled with the fixed seed 666 — and the loop of
``src/f5_tts/eval/eval_infer_batch.py:178-214`` (``split_between_processes`` -> ``model.sample(cond, text, duration, lens, ...)`` ->
per-utterance slice -> vocoder -> RMS restore).  Same batches in the same order as the reference for the same inputs
(tests/test_eval_batching.py runs the reference's own function, lifted out of its module, next to this one).

Why it matters on this hardware: a batch costs (2 * B * max_frames) packed rows whatever the individual lengths (padded rows are masked,
not skipped), so forming batches from ONE length class keeps the padding below 1 / num_buckets of the range, and the frame budget keeps the
GEMMs of every batch in the same tile regime.  ``deal_batches`` then spreads the batches over the ranks by their padded cost
(``dist.shard_balanced``) instead of the reference's contiguous split of the shuffled list.
"""
from __future__ import annotations

import math
import os
import random
from typing import Callable, List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

from . import dist as fdist
from . import infer as I

Prompt = Tuple[List[str], List[torch.Tensor], torch.Tensor, List[int], List[int], List]
MetaRow = Tuple[str, str, str, str, str]  # (utt, prompt_text, prompt_wav, gt_text, gt_wav)


# ---- test lists --------------------------------------------------------------------------------------------------------------------
def _rows(path: str, sep: str):
    with open(path) as f:
        for line in f:
            yield line.strip().split(sep)


def get_seedtts_testset_metainfo(metalst: str) -> List[MetaRow]:
    """Seed-TTS list (utils_eval.py:19-34): ``utt|prompt_text|prompt_wav|gt_text`` with an optional fifth ``gt_wav`` field; without it the
    target recording is ``<list dir>/wavs/<utt>.wav``; a relative prompt path is taken from the list's directory; other lines are skipped."""
    here = os.path.dirname(metalst)
    out: List[MetaRow] = []
    for f in _rows(metalst, "|"):
        if len(f) not in (4, 5):
            continue
        utt, prompt_text, prompt_wav, gt_text = f[:4]
        gt_wav = f[4] if len(f) == 5 else os.path.join(here, "wavs", utt + ".wav")
        out.append((utt, prompt_text, prompt_wav if os.path.isabs(prompt_wav) else os.path.join(here, prompt_wav), gt_text, gt_wav))
    return out


def get_librispeech_test_clean_metainfo(metalst: str, librispeech_test_clean_path: str) -> List[MetaRow]:
    """LibriSpeech-PC test-clean cross-sentence list (utils_eval.py:38-52): six tab-separated fields ``ref_utt ref_dur ref_txt gen_utt
    gen_dur gen_txt``; an utterance ``spk-chapter-n`` lives at ``<root>/spk/chapter/spk-chapter-n.flac``; the text to generate gets a
    leading space."""
    def flac(utt: str) -> str:
        spk, chapter, _ = utt.split("-")
        return os.path.join(librispeech_test_clean_path, spk, chapter, utt + ".flac")

    return [(gen_utt, ref_txt, flac(ref_utt), " " + gen_txt, flac(gen_utt)) for ref_utt, _, ref_txt, gen_utt, _, gen_txt in _rows(metalst, "\t")]


def padded_mel_batch(ref_mels: Sequence[torch.Tensor]) -> torch.Tensor:
    """[n_mel, T_i] prompts -> zero-padded frame-major batch [B, max T, n_mel] (utils_eval.py:56-66)."""
    longest = max(int(m.shape[-1]) for m in ref_mels)
    return torch.stack([F.pad(m, (0, longest - m.shape[-1]), value=0) for m in ref_mels]).permute(0, 2, 1)


# ---- length classes ----------------------------------------------------------------------------------------------------------------
class _LengthClass:
    """The utterances waiting in one length class, column by column as the reference's batch tuple wants them."""

    __slots__ = ("utts", "rms", "mels", "ref_lens", "total_lens", "texts", "frames")

    def __init__(self):
        self.utts, self.rms, self.mels, self.ref_lens, self.total_lens, self.texts, self.frames = [], [], [], [], [], [], 0

    def add(self, utt, rms, mel, ref_len, total_len, text_list) -> None:
        self.utts.append(utt); self.rms.append(rms); self.mels.append(mel); self.ref_lens.append(ref_len)
        self.total_lens.append(total_len); self.texts.extend(text_list); self.frames += total_len

    def take(self) -> Prompt:
        batch = (self.utts, self.rms, padded_mel_batch(self.mels), self.ref_lens, self.total_lens, self.texts)
        self.__init__()
        return batch


def _prompt_audio(path: str, load_audio, resample, target_rms: float, target_sample_rate: int):
    """Prompt recording as the sampler wants it: quiet prompts raised to the target RMS (the measured RMS is kept for the way back),
    resampled to the model rate (utils_eval.py:110-118)."""
    audio, sr = load_audio(path)
    rms = torch.sqrt(torch.mean(torch.square(audio)))
    if rms < target_rms:
        audio = audio * target_rms / rms
    assert audio.shape[-1] > 5000, f"Empty prompt wav: {path}, or audio loader issue."
    return (resample(audio, sr, target_sample_rate) if sr != target_sample_rate else audio), rms


def get_inference_prompt(metainfo, mel_fn: Callable[[torch.Tensor], torch.Tensor], speed: float = 1.0, tokenizer: str = "pinyin",
                         polyphone: bool = True, target_sample_rate: int = 24000, hop_length: int = 256, target_rms: float = 0.1,
                         use_truth_duration: bool = False, infer_batch_size: int = 1, num_buckets: int = 200, min_secs: int = 3,
                         max_secs: int = 40, load_audio: Callable = I.load_wav, resample: Callable = I.resample,
                         shuffle_seed: Optional[int] = 666) -> List[Prompt]:
    """The batch list of utils_eval.py:72-205 for ``metainfo`` rows.  ``mel_fn(wave[1, n]) -> [1, n_mel, T]`` is the model's mel front-end
    (``F5HipCFM.mel_spec``; the reference builds a ``MelSpec``, :99-106); ``load_audio`` / ``resample`` stand for ``torchaudio.load`` /
    ``transforms.Resample``.  Every utterance goes to class ``floor((total - lo) / (hi - lo + 1) * num_buckets)`` of its TOTAL frame count
    (``lo``, ``hi`` = ``min_secs``, ``max_secs`` in frames); a class is emitted as one batch ``(utts, rms, padded mels [B, T, n_mel], prompt
    frames, total frames, token lists)`` the moment its frames reach ``infer_batch_size``, what is left is emitted class by class at the
    end, and the list is shuffled with the reference's fixed seed."""
    assert infer_batch_size > 0, "infer_batch_size should be greater than 0."
    lo, hi = (secs * target_sample_rate // hop_length for secs in (min_secs, max_secs))
    classes = [_LengthClass() for _ in range(num_buckets)]
    batches: List[Prompt] = []
    for utt, prompt_text, prompt_wav, gt_text, gt_wav in metainfo:
        audio, rms = _prompt_audio(prompt_wav, load_audio, resample, target_rms, target_sample_rate)
        if len(prompt_text[-1].encode("utf-8")) == 1:  # a single-byte last character is followed by a separating space (:121-122)
            prompt_text += " "
        joined = [prompt_text + gt_text]
        mel = mel_fn(audio).squeeze(0)  # [n_mel, T]
        ref_len = int(mel.shape[-1])
        if use_truth_duration:  # the target recording's own length, scaled by the speed (:132-137)
            gt_audio, gt_sr = load_audio(gt_wav)
            if gt_sr != target_sample_rate:
                gt_audio = resample(gt_audio, gt_sr, target_sample_rate)
            total = ref_len + int(gt_audio.shape[-1] / hop_length / speed)
        else:               # prompt frames per prompt byte, times the bytes to generate (:139-142)
            total = ref_len + int(ref_len / len(prompt_text.encode("utf-8")) * len(gt_text.encode("utf-8")) / speed)
        assert lo <= total <= hi, f"Audio {utt} has duration {total * hop_length // target_sample_rate}s out of range [{min_secs}, {max_secs}]."
        cls = classes[math.floor((total - lo) / (hi - lo + 1) * num_buckets)]
        cls.add(utt, rms, mel, ref_len, total, I.convert_char_to_pinyin(joined, polyphone=polyphone) if tokenizer == "pinyin" else joined)
        if cls.frames >= infer_batch_size:
            batches.append(cls.take())
    batches.extend(cls.take() for cls in classes if cls.frames > 0)
    if shuffle_seed is not None:  # long and short batches mixed over the ranks (:201-203); the reference reseeds the global generator
        random.seed(shuffle_seed)
        random.shuffle(batches)
    return batches


def batch_cost(prompt: Prompt) -> int:
    """Packed rows the backbone processes for a batch: utterances x the longest total length (padded rows are masked, not skipped)."""
    return len(prompt[0]) * max(prompt[4])


def padding_fraction(prompts: Sequence[Prompt]) -> float:
    """Share of the processed frame rows that are padding — what the length classes keep small."""
    used = sum(sum(p[4]) for p in prompts)
    paid = sum(batch_cost(p) for p in prompts)
    return 1.0 - used / paid if paid else 0.0


def deal_batches(prompts_all: Sequence[Prompt], world: int, balanced: bool = True) -> List[List[int]]:
    """Indices of the batches each rank runs.  ``balanced``: longest-first greedy deal by padded cost (``dist.shard_balanced``);
    otherwise the contiguous split of ``accelerator.split_between_processes`` (eval_infer_batch.py:178)."""
    if balanced:
        return fdist.shard_balanced([float(batch_cost(p)) for p in prompts_all], world)
    return [list(fdist.shard_contiguous(len(prompts_all), r, world)) for r in range(world)]


@torch.no_grad()
def run_prompt_batches(model, vocoder, prompts: Sequence[Prompt], mel_spec_type: str = "vocos", nfe_step: int = 32, cfg_strength: float = 2.0,
                       sway_sampling_coef: Optional[float] = -1.0, no_ref_audio: bool = False, seed: Optional[int] = None,
                       target_rms: float = 0.1, on_wave: Optional[Callable[[str, torch.Tensor], None]] = None):
    """The inference loop of eval_infer_batch.py:179-210 over this rank's batches: one ragged ``model.sample`` per batch (prompt lengths as
    ``lens``, total lengths as ``duration``), the generated part of every row through the vocoder, RMS restored.  Yields
    ``(utt, wave[1, samples])`` in batch order (``on_wave(utt, wave)`` is called instead when given — the reference saves a file there)."""
    out = []
    for utts, ref_rms_list, ref_mels, ref_mel_lens, total_mel_lens, final_text_list in prompts:
        lens = torch.tensor(ref_mel_lens, dtype=torch.long)
        duration = torch.tensor(total_mel_lens, dtype=torch.long)
        generated, _ = model.sample(cond=ref_mels, text=final_text_list, duration=duration, lens=lens, steps=nfe_step,
                                    cfg_strength=cfg_strength, sway_sampling_coef=sway_sampling_coef, no_ref_audio=no_ref_audio, seed=seed)
        for i, gen in enumerate(generated):
            gen = gen[ref_mel_lens[i]:total_mel_lens[i], :].unsqueeze(0)
            gen_mel_spec = gen.permute(0, 2, 1).to(torch.float32)
            if mel_spec_type == "vocos":
                wave = vocoder.decode(gen_mel_spec).cpu()
            else:  # bigvgan: vocoder(mel) -> [b, 1, samples]
                wave = vocoder(gen_mel_spec).squeeze(0).cpu()
            if ref_rms_list[i] < target_rms:
                wave = wave * ref_rms_list[i] / target_rms
            if on_wave is not None:
                on_wave(utts[i], wave)
            else:
                out.append((utts[i], wave))
    return out
