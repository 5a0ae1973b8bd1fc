#!/usr/bin/env python3
"""Utterance-sharded batch inference on N GPUs of one node — the reference's only multi-GPU inference pattern
(``accelerate launch src/f5_tts/eval/eval_infer_batch.py``, eval_infer_batch.py:178-214: one process per GPU, a disjoint slice of the
utterance list each, barriers around the loop), on the HIP engine:

    python tools/infer_batch.py --list utts.lst --out out_dir --ckpt model.safetensors --vocab vocab.txt --vocos vocos_dir
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 tools/infer_batch.py ...

``utts.lst``: one utterance per line, ``utt_id|ref_wav_path|ref_text|gen_text`` (the layout of the reference's LibriSpeech-PC test
list, data/librispeech_pc_test_clean_cross_sentence.lst, minus the duration columns).  Rank 0 reads the checkpoint; the packed weight
blob travels once over RCCL; utterances are dealt longest-first to balance the ranks; every rank writes its own ``<utt_id>.wav``.
Without ``--ckpt`` seeded synthetic weights are used (smoke / throughput runs; ``--synthetic N`` makes N utterances up as well).
"""
import argparse
import os
import sys
import time
import wave

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import f5_tts_amd  # noqa: E402,F401
from f5_tts_amd import config, synth  # noqa: E402
from f5_tts_amd import dist as fdist  # noqa: E402
from f5_tts_amd import infer as I  # noqa: E402
from f5_tts_amd.engine import F5HipCFM, F5HipEngine, F5HipVocos, filter_vocos_keys, map_checkpoint_keys  # noqa: E402


def write_wav(path, samples, sr=24000):
    pcm = (np.clip(samples, -1.0, 1.0) * 32767.0).astype("<i2")
    with wave.open(path, "wb") as w:
        w.setnchannels(1); w.setsampwidth(2); w.setframerate(sr); w.writeframes(pcm.tobytes())


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--list"); ap.add_argument("--out", default="infer_out")
    ap.add_argument("--model", default="F5TTS_v1_Base"); ap.add_argument("--ckpt"); ap.add_argument("--vocab"); ap.add_argument("--vocos")
    ap.add_argument("--synthetic", type=int, default=0); ap.add_argument("--nfe", type=int, default=16)
    ap.add_argument("--precision", default="fp16m"); ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--testset", choices=["seedtts", "ls_pc_test_clean"], default=None,
                    help="read --list with the reference's test-list parsers (eval/utils_eval.py:19-52) instead of the utt|wav|ref|gen layout")
    ap.add_argument("--librispeech-path", default="", help="root of LibriSpeech test-clean for --testset ls_pc_test_clean")
    ap.add_argument("--frames-per-batch", type=int, default=0,
                    help="> 0: form length-bucketed batches with this frame budget (the reference's infer_batch_size, eval/utils_eval.py:72-205; "
                         "f5-tts_amd/eval_batching.py) instead of running the utterances one by one")
    ap.add_argument("--tokenizer", choices=["pinyin", "char"], default=None,
                    help="text front-end of BOTH modes: 'pinyin' = convert_char_to_pinyin (what the F5-TTS presets were trained with and what "
                         "infer_process applies; ASCII text only here, the Chinese G2P packages are absent), 'char' = the characters as they are. "
                         "Default: pinyin when --vocab is given (the reference's eval_infer_batch.py takes the tokenizer from the model config), "
                         "char for the built-in ASCII smoke vocabulary")
    ap.add_argument("--num-buckets", type=int, default=200, help="length classes of the batch formation (eval/utils_eval.py:72 default 200; 1 = no "
                    "bucketing: batches in list order, as a server without a choice of batches would see them)")
    ap.add_argument("--attn-mask", action="store_true", help="attn_mask_enabled=True (the model must have been trained with it; the presets ship False): "
                    "padded keys are masked out of the attention — the precondition of --packed")
    ap.add_argument("--packed", action="store_true", help="engine option packed_rows: the block loop runs over the valid rows of a ragged batch only")
    ap.add_argument("--min-secs", type=int, default=3, help="shortest total length the length classes cover (eval/utils_eval.py:72 default)")
    ap.add_argument("--max-secs", type=int, default=40, help="longest total length the length classes cover (eval/utils_eval.py:72 default)")
    a = ap.parse_args()
    if a.tokenizer is None:
        a.tokenizer = "pinyin" if a.vocab else "char"
    if a.testset == "ls_pc_test_clean":
        # the list points at LibriSpeech .flac files; this tool reads PCM .wav through the standard library only (no torchaudio / soundfile here)
        ap.error("--testset ls_pc_test_clean needs a FLAC decoder, which this image does not have: convert test-clean to 16-bit .wav and pass the "
                 "utterances with --list in the utt|wav|ref_text|gen_text layout (or --testset seedtts, whose prompts are .wav)")
    rank, local, world = fdist.init_distributed()
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)
    cfg, vcfg = config.PRESETS[a.model], config.VOCOS_MEL_24K
    vocab = None
    if a.vocab:
        vocab, vsize = I.get_tokenizer(a.vocab)
        from dataclasses import replace
        cfg = replace(cfg, text_num_embeds=vsize)
    if a.attn_mask:
        from dataclasses import replace as _replace
        cfg = _replace(cfg, attn_mask_enabled=True)
    if vocab is None:  # smoke / throughput runs without a vocab.txt: printable ASCII, id 0 = space (the reference's unknown id)
        vocab = {" ": 0, **{chr(c): c - 32 for c in range(33, 127)}}
    eng = F5HipEngine(cfg, vcfg, device=dev)
    if rank == 0:  # rank 0 owns the checkpoint files
        sd = I._read_checkpoint(a.ckpt) if a.ckpt else None
        if sd is not None:
            sd = map_checkpoint_keys({"ema_model_state_dict": sd} if a.ckpt.endswith(".safetensors") else sd)
            sd = {k: v for k, v in sd.items() if k.startswith("transformer.")}
        else:
            sd = synth.synth_dit_state_dict(cfg, seed=0)
        vsd = (filter_vocos_keys(torch.load(os.path.join(a.vocos, "pytorch_model.bin"), map_location="cpu", weights_only=True)) if a.vocos
               else synth.synth_vocos_state_dict(vcfg, seed=0))
        eng.load_state_dict({**sd, **vsd}, strict=False, finalize=False)
    fdist.broadcast_engine_weights(eng, src=0)
    model, voc = F5HipCFM(eng, vocab_char_map=vocab, precision=a.precision), F5HipVocos(eng)
    if a.packed:
        eng.set_option("packed_rows", 1)

    if a.synthetic:
        utts = [(f"syn{i:04d}", (synth.synth_wave(24000 * (3 + i % 4), seed=i), 24000), "some call me nature, others call me mother nature.",
                 "i have been here for over four and a half billion years. " * (1 + i % 3)) for i in range(a.synthetic)]
    elif a.testset:  # the reference's evaluation lists: (utt, prompt_text, prompt_wav, gt_text, gt_wav) -> (utt, wav, ref text, gen text)
        from f5_tts_amd import eval_batching as EB

        meta = (EB.get_seedtts_testset_metainfo(a.list) if a.testset == "seedtts"
                else EB.get_librispeech_test_clean_metainfo(a.list, a.librispeech_path))
        utts = [(utt, prompt_wav, prompt_text, gt_text) for utt, prompt_text, prompt_wav, gt_text, _gt_wav in meta]
    else:
        utts = []
        for line in open(a.list, encoding="utf-8"):
            uid, ref_wav, ref_text, gen_text = line.rstrip("\n").split("|")[:4]
            utts.append((uid, ref_wav, ref_text, gen_text))
    os.makedirs(a.out, exist_ok=True)
    audio_s = 0.0
    if a.frames_per_batch > 0:  # the reference's evaluation batching: one ragged sample() per length-class batch
        from f5_tts_amd import eval_batching as EB

        def load_audio(x):  # a path, or the (wave, sr) pair of the synthetic utterances
            w, sr = x if isinstance(x, tuple) else I.load_wav(x)
            return (w if w.ndim == 2 else w[None]).float().cpu(), sr

        meta = [(uid, ref_text, ref, gen_text, "") for uid, ref, ref_text, gen_text in utts]
        # same tokens as the one-by-one mode below (infer_process -> convert_char_to_pinyin) and the reference's bucket range unless asked otherwise
        batches = EB.get_inference_prompt(meta, lambda w: model.mel_spec(w.to(dev)).cpu(), tokenizer=a.tokenizer, infer_batch_size=a.frames_per_batch,
                                          min_secs=a.min_secs, max_secs=a.max_secs, num_buckets=a.num_buckets, load_audio=load_audio)
        mine = EB.deal_batches(batches, world)[rank]
        if rank == 0:
            print(f"{len(batches)} batches of <= {max(len(b[0]) for b in batches)} utterances, padding {100 * EB.padding_fraction(batches):.1f} % of the rows")
        if world > 1:
            torch.distributed.barrier()  # eval_infer_batch.py:178
        t0 = time.perf_counter()

        def save(uid, wave_):
            nonlocal audio_s
            write_wav(os.path.join(a.out, uid + ".wav"), wave_[0].numpy())
            audio_s += wave_.shape[-1] / 24000

        EB.run_prompt_batches(model, voc, [batches[i] for i in mine], nfe_step=a.nfe, seed=a.seed, on_wave=save)
    else:
        costs = [len(u[3].encode()) + len(u[2].encode()) for u in utts]  # text bytes ~ frames to generate
        mine = fdist.shard_balanced(costs, world)[rank]
        if world > 1:
            torch.distributed.barrier()  # eval_infer_batch.py:178
        t0 = time.perf_counter()
        for i in mine:
            uid, ref, ref_text, gen_text = utts[i]
            wav, sr, _ = I.infer_process(ref, ref_text, gen_text, model, voc, show_info=lambda *_: None, nfe_step=a.nfe, seed=a.seed)
            write_wav(os.path.join(a.out, uid + ".wav"), wav, sr)
            audio_s += len(wav) / sr
    torch.cuda.synchronize(dev)
    if world > 1:
        torch.distributed.barrier()  # eval_infer_batch.py:214
    dt = fdist.barrier_max_seconds(time.perf_counter() - t0, dev)
    tot = torch.tensor([audio_s], dtype=torch.float64, device=dev)
    if world > 1:
        torch.distributed.all_reduce(tot)
    if rank == 0:
        print(f"{len(utts)} utterances, {float(tot):.1f} s of audio in {dt:.2f} s on {world} GPU(s): RTF {dt / max(float(tot), 1e-9):.4f}")


if __name__ == "__main__":
    main()
