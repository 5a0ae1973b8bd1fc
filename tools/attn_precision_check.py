"""Generated-mel error of a half-precision parity mode (ATTN_PREC = fp16m (default) | fp16x3) against the reference-minted goldens with
the attention scores computed from fp16 hi.hi + MX-fp6 corrections (attn_impl 0, the default since round 5: 1.5 MFMA-equivalents per
product), from hi/lo-split q, k (attn_impl 4: 3 MFMAs), from everything split (attn_impl 2) and from plain fp16 q, k, P, V (attn_impl 3, the
default of rounds 2-4): python tools/attn_precision_check.py [golden names]  (GPU box)."""
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
cases = {**MG.CASES, **MG.SWEEP_CASES, **MG.FULL_CASES}
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
    prec = os.environ.get("ATTN_PREC", "fp16m")
    durs = duration.tolist() if torch.is_tensor(duration) else [int(duration)] * g.shape[0]
    for impl in (0, 4, 2, 3):
        eng.set_option("attn_impl", impl)
        model = F5HipCFM(eng, precision=prec, ode_method=c.get("method", "euler"))
        out, _ = model.sample(wav.cuda(), text, duration, lens=lens, **c["kw"])
        d = torch.cat([(out[b, :durs[b]].cpu() - torch.from_numpy(g[b, :durs[b]])).abs().reshape(-1) for b in range(g.shape[0])])
        res.append((float(d.max()), float(d.mean())))
    print(f"{name:28s} {prec} MX-corrected scores: max {res[0][0]:.2e} mean {res[0][1]:.2e}   split q,k: max {res[1][0]:.2e} mean {res[1][1]:.2e}   all split: max {res[2][0]:.2e}"
          f"   plain fp16 q,k: max {res[3][0]:.2e} mean {res[3][1]:.2e}"
          f"   |mel| max {np.abs(g).max():.2f}", flush=True)
    eng.close()
