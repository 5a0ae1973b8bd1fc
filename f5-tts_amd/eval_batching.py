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


def get_seedtts_testset_metainfo(metalst: str) -> List[Tuple[str, str, str, str, str]]:
    """``utt|prompt_text|prompt_wav|gt_text[|gt_wav]`` lines (utils_eval.py:19-34)."""
    metainfo = []
    with open(metalst) as f:
        for line in f.readlines():
            parts = line.strip().split("|")
            if len(parts) == 5:
                utt, prompt_text, prompt_wav, gt_text, gt_wav = parts
            elif len(parts) == 4:
                utt, prompt_text, prompt_wav, gt_text = parts
                gt_wav = os.path.join(os.path.dirname(metalst), "wavs", utt + ".wav")
            else:
                continue
            if not os.path.isabs(prompt_wav):
                prompt_wav = os.path.join(os.path.dirname(metalst), prompt_wav)
            metainfo.append((utt, prompt_text, prompt_wav, gt_text, gt_wav))
    return metainfo


def get_librispeech_test_clean_metainfo(metalst: str, librispeech_test_clean_path: str) -> List[Tuple[str, str, str, str, str]]:
    """tab-separated ``ref_utt ref_dur ref_txt gen_utt gen_dur gen_txt`` lines (utils_eval.py:38-52)."""
    metainfo = []
    with open(metalst) as f:
        for line in f.readlines():
            ref_utt, _ref_dur, ref_txt, gen_utt, _gen_dur, gen_txt = line.strip().split("\t")
            ref_spk_id, ref_chaptr_id, _ = ref_utt.split("-")
            ref_wav = os.path.join(librispeech_test_clean_path, ref_spk_id, ref_chaptr_id, ref_utt + ".flac")
            gen_spk_id, gen_chaptr_id, _ = gen_utt.split("-")
            gen_wav = os.path.join(librispeech_test_clean_path, gen_spk_id, gen_chaptr_id, gen_utt + ".flac")
            metainfo.append((gen_utt, ref_txt, ref_wav, " " + gen_txt, gen_wav))
    return metainfo


def padded_mel_batch(ref_mels: Sequence[torch.Tensor]) -> torch.Tensor:
    """[n_mel, T_i] prompts -> zero-padded frame-major batch [B, max T, n_mel] (utils_eval.py:56-66)."""
    max_mel_length = max(int(mel.shape[-1]) for mel in ref_mels)
    padded = [F.pad(mel, (0, max_mel_length - mel.shape[-1]), value=0) for mel in ref_mels]
    return torch.stack(padded).permute(0, 2, 1)


def get_inference_prompt(metainfo, mel_fn: Callable[[torch.Tensor], torch.Tensor], speed: float = 1.0, tokenizer: str = "pinyin",
                         polyphone: bool = True, target_sample_rate: int = 24000, hop_length: int = 256, target_rms: float = 0.1,
                         use_truth_duration: bool = False, infer_batch_size: int = 1, num_buckets: int = 200, min_secs: int = 3,
                         max_secs: int = 40, load_audio: Callable = I.load_wav, resample: Callable = I.resample,
                         shuffle_seed: Optional[int] = 666) -> List[Prompt]:
    """utils_eval.py:72-205.  ``metainfo``: (utt, prompt_text, prompt_wav, gt_text, gt_wav) tuples; ``mel_fn(wave[1, n]) -> [1, n_mel, T]``
    is the model's mel front-end (``F5HipCFM.mel_spec``; the reference builds a ``MelSpec`` here, :99-106); ``load_audio`` /
    ``resample`` stand for ``torchaudio.load`` / ``transforms.Resample`` (:110,116-118).  Returns the reference's list of
    ``(utts, ref_rms_list, padded_ref_mels[B, T, n_mel], ref_mel_lens, total_mel_lens, final_text_list)`` batches."""
    prompts_all: List[Prompt] = []
    min_tokens = min_secs * target_sample_rate // hop_length
    max_tokens = max_secs * target_sample_rate // hop_length
    batch_accum = [0] * num_buckets
    utts, ref_rms_list, ref_mels, ref_mel_lens, total_mel_lens, final_text_list = ([[] for _ in range(num_buckets)] for _ in range(6))

    def flush(b: int) -> None:
        prompts_all.append((utts[b], ref_rms_list[b], padded_mel_batch(ref_mels[b]), ref_mel_lens[b], total_mel_lens[b], final_text_list[b]))
        batch_accum[b] = 0
        utts[b], ref_rms_list[b], ref_mels[b], ref_mel_lens[b], total_mel_lens[b], final_text_list[b] = [], [], [], [], [], []

    for utt, prompt_text, prompt_wav, gt_text, gt_wav in metainfo:
        ref_audio, ref_sr = load_audio(prompt_wav)
        ref_rms = torch.sqrt(torch.mean(torch.square(ref_audio)))
        if ref_rms < target_rms:
            ref_audio = ref_audio * target_rms / ref_rms
        assert ref_audio.shape[-1] > 5000, f"Empty prompt wav: {prompt_wav}, or audio loader issue."
        if ref_sr != target_sample_rate:
            ref_audio = resample(ref_audio, ref_sr, target_sample_rate)
        if len(prompt_text[-1].encode("utf-8")) == 1:  # a single-byte last character gets a separating space (:121-122)
            prompt_text = prompt_text + " "
        text = [prompt_text + gt_text]
        text_list = I.convert_char_to_pinyin(text, polyphone=polyphone) if tokenizer == "pinyin" else text
        ref_mel = mel_fn(ref_audio).squeeze(0)  # [n_mel, T]
        ref_mel_len = int(ref_mel.shape[-1])
        if use_truth_duration:
            gt_audio, gt_sr = load_audio(gt_wav)
            if gt_sr != target_sample_rate:
                gt_audio = resample(gt_audio, gt_sr, target_sample_rate)
            total_mel_len = ref_mel_len + int(gt_audio.shape[-1] / hop_length / speed)
        else:
            ref_text_len = len(prompt_text.encode("utf-8"))
            gen_text_len = len(gt_text.encode("utf-8"))
            total_mel_len = ref_mel_len + int(ref_mel_len / ref_text_len * gen_text_len / speed)
        assert infer_batch_size > 0, "infer_batch_size should be greater than 0."
        assert min_tokens <= total_mel_len <= max_tokens, (
            f"Audio {utt} has duration {total_mel_len * hop_length // target_sample_rate}s out of range [{min_secs}, {max_secs}].")
        b = math.floor((total_mel_len - min_tokens) / (max_tokens - min_tokens + 1) * num_buckets)
        utts[b].append(utt)
        ref_rms_list[b].append(ref_rms)
        ref_mels[b].append(ref_mel)
        ref_mel_lens[b].append(ref_mel_len)
        total_mel_lens[b].append(total_mel_len)
        final_text_list[b].extend(text_list)
        batch_accum[b] += total_mel_len
        if batch_accum[b] >= infer_batch_size:
            flush(b)
    for b, frames in enumerate(batch_accum):  # residual batches (:186-199)
        if frames > 0:
            flush(b)
    if shuffle_seed is not None:  # "not only leave easy work for last workers" (:201-203); the reference reseeds the global generator
        random.seed(shuffle_seed)
        random.shuffle(prompts_all)
    return prompts_all


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
