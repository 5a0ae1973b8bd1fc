"""Generated-mel error of the fp16x3 mode against the reference-minted goldens with the attention scores computed from hi/lo-split q, k
(attn_impl 4, 3 MFMAs per product) and from plain fp16 q, k (attn_impl 0, the default: 1 MFMA): python tools/attn_precision_check.py  (GPU box)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import f5_tts_amd  # noqa: E402,F401
from f5_tts_amd import config, synth  # noqa: E402
from f5_tts_amd.engine import F5HipCFM, F5HipEngine  # noqa: E402
from oracle import make_golden as MG  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")
cases = {**MG.CASES, **MG.FULL_CASES}
names = sys.argv[1:] or sorted(cases)
for name in names:
    c = cases[name]
    if not os.path.exists(os.path.join(GOLD, name + ".npz")):
        continue
    cfg, wav, text, duration, lens = MG.case_inputs(c)
    eng = F5HipEngine(cfg, None, device=0)
    eng.load_state_dict(MG.case_weights(c))
    g = np.load(os.path.join(GOLD, name + ".npz"))["out"]
    res = []
    for impl in (4, 0):
        eng.set_option("attn_impl", impl)
        model = F5HipCFM(eng, precision="fp16x3", ode_method=c.get("method", "euler"))
        out, _ = model.sample(wav.cuda(), text, duration, lens=lens, **c["kw"])
        d = (out.cpu() - torch.from_numpy(g)).abs()
        res.append((float(d.max()), float(d.mean())))
    print(f"{name:28s} split q,k: max {res[0][0]:.2e} mean {res[0][1]:.2e}   plain fp16 q,k: max {res[1][0]:.2e} mean {res[1][1]:.2e}   |mel| max {np.abs(g).max():.2f}", flush=True)
    eng.close()
