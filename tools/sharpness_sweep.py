"""The SHARPNESS SWEEP: generated-mel error of every operand mode against reference-minted goldens whose attention
logits are multiplied by 1, 2 and 4 (to_q and to_k of the trained-like weights x 1, sqrt 2, 2: synth.sharpen_attention_state_dict), next to
the fp32 FLOOR of each golden — the difference between the reference's own fp32 CFM.sample and the fp32 restatement of the same arithmetic
in another operation order (tests/golden/pins.json, measured when the golden was minted): what no implementation can be held below.

    python tools/sharpness_sweep.py            (GPU box; markdown table on stdout)
"""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import f5_tts_amd  # noqa: E402,F401
from f5_tts_amd.engine import F5HipCFM, F5HipEngine  # noqa: E402
from oracle import make_golden as MG  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")
PINS = json.load(open(os.path.join(GOLD, "pins.json")))
FAMILIES = {
    "base_v1_trained_like (F5-TTS v1 Base, B = 1, N = 1406, NFE 16)": ["base_v1_trained_like", "base_v1_trained_like_sharp1p4", "base_v1_trained_like_sharp1p7", "base_v1_trained_like_sharp2", "base_v1_trained_like_sharp4"],
    "small_mask_ragged_b3_trained_like (Small, ragged masked batch of 3, NFE 8)": ["small_mask_ragged_b3_trained_like", "small_mask_ragged_b3_trained_like_sharp1p4", "small_mask_ragged_b3_trained_like_sharp1p7",
                                                                                  "small_mask_ragged_b3_trained_like_sharp2", "small_mask_ragged_b3_trained_like_sharp4"],
    "tiny_v1_trained_like (tiny DiT, NFE 16)": ["tiny_v1_trained_like", "tiny_v1_trained_like_sharp2", "tiny_v1_trained_like_sharp4"],
}
MODES = [("fp32", -1), ("fp16x3", 0), ("fp16m", 0), ("fp16x3", 6), ("fp16m", 6), ("fp16x3", 7), ("fp16m", 7), ("fp16x3", 2), ("fp16m", 2), ("fp16m", 3)]
cases = {**MG.CASES, **MG.SWEEP_CASES, **MG.FULL_CASES}
only = set(sys.argv[1:])


def gen_err(out, g, durs):
    return float(torch.cat([(out[b, :durs[b]].cpu() - torch.from_numpy(g[b, :durs[b]])).abs().reshape(-1) for b in range(g.shape[0])]).max())


for fam, names in FAMILIES.items():
    if only and not any(n in only for n in names):
        continue
    print(f"\n**{fam}**\n")
    print("| logits x | fp32 floor (reference vs its fp32 restatement) | " + " | ".join(f"`{p}`" + ("" if i < 0 else f" attn_impl {i}") for p, i in MODES) + " |")
    print("|---|---:|" + "---:|" * len(MODES))
    for name in names:
        c = cases[name]
        sharp = float(c.get("sharp", 1.0))
        floor = PINS.get(name, {}).get("oracle_vs_reference_out")
        if c.get("record_only") or not os.path.exists(os.path.join(GOLD, name + ".npz")):
            print(f"| {sharp * sharp:.1f} | {floor:.2e} (chaotic: no fixture) |" + " — |" * len(MODES), flush=True)
            continue
        cfg, wav, text, duration, lens = MG.case_inputs(c)
        eng = F5HipEngine(cfg, None, device=0)
        eng.load_state_dict(MG.case_weights(c))
        g = np.load(os.path.join(GOLD, name + ".npz"))["out"]
        durs = duration.tolist() if torch.is_tensor(duration) else [int(duration)] * g.shape[0]
        row = []
        for prec, impl in MODES:
            eng.set_option("attn_impl", max(impl, 0))
            out, _ = F5HipCFM(eng, precision=prec, ode_method=c.get("method", "euler")).sample(wav.cuda(), text, duration, lens=lens, **c["kw"])
            row.append(gen_err(out, g, durs))
        eng.close()
        print(f"| {sharp * sharp:.1f} | {floor:.2e} | " + " | ".join(f"{e:.2e}" for e in row) + " |", flush=True)
