"""Accuracy of each operand-precision mode on the FULL-SIZE F5-TTS Base model (BASELINE.json configs[1]) against the golden
minted from the reference's CPU path (tests/golden/base_v1_cfg1.npz).  Run on the GPU box: python tools/precision_study.py"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import f5_tts_amd  # noqa: E402,F401
from f5_tts_amd import synth  # noqa: E402
from f5_tts_amd.engine import F5HipCFM, F5HipEngine  # noqa: E402
from oracle import make_golden as MG  # noqa: E402


def main():
    c = MG.FULL_CASES["base_v1_cfg1"]
    cfg, wav, text, duration, lens = MG.case_inputs(c)
    g = np.load(os.path.join(ROOT, "tests", "golden", "base_v1_cfg1.npz"))
    gold = torch.as_tensor(g["out"])
    eng = F5HipEngine(cfg, None, device=0)
    eng.load_state_dict(synth.synth_dit_state_dict(cfg, seed=c["wseed"]))
    print("golden generated-mel range", float(gold[:, 468:].min()), float(gold[:, 468:].max()), "std", float(gold[:, 468:].std()))
    modes = [("fp32", 0), ("fp16x3", 0), ("fp16x3", 2), ("fp16x3", 3), ("fp16", 0)]
    for prec, attn in modes:
        eng.set_option("attn_impl", attn)
        model = F5HipCFM(eng, precision=prec)
        out, traj = model.sample(wav.cuda(), text, duration, **c["kw"])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out, traj = model.sample(wav.cuda(), text, duration, **c["kw"])
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        d = (out.cpu() - gold)[:, 468:].abs()
        d1 = (traj[1].cpu() - torch.as_tensor(g["traj_1"])).abs().max().item()
        print(f"{prec:8s} attn_impl={attn}: gen-mel max-abs {d.max().item():.3e}  mean-abs {d.mean().item():.3e}  "
              f"p99.9 {d.flatten().kthvalue(int(d.numel() * 0.999)).values.item():.3e}  traj[1] max-abs {d1:.3e}  sample {dt * 1e3:.1f} ms",
              flush=True)
    eng.close()


if __name__ == "__main__":
    main()
