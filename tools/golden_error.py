"""Generated-mel max-abs error of ONE golden in chosen (precision, attn_impl) pairs — the quick A/B probe behind a library bisect:

    F5HIP_LIB=/path/to/libf5hip.so python tools/golden_error.py base_v1_trained_like fp16m:0 fp16x3:2     (GPU box)
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import f5_tts_amd  # noqa: E402,F401
from f5_tts_amd.engine import F5HipCFM, F5HipEngine  # noqa: E402
from oracle import make_golden as MG  # noqa: E402

name = sys.argv[1]
c = {**MG.CASES, **MG.SWEEP_CASES, **MG.FULL_CASES}[name]
cfg, wav, text, duration, lens = MG.case_inputs(c)
eng = F5HipEngine(cfg, None, device=0)
eng.load_state_dict(MG.case_weights(c))
g = np.load(os.path.join(ROOT, "tests", "golden", name + ".npz"))["out"]
durs = duration.tolist() if torch.is_tensor(duration) else [int(duration)] * g.shape[0]
res = []
for spec in sys.argv[2:] or ["fp16m:0"]:
    prec, impl = spec.split(":")
    eng.set_option("attn_impl", int(impl))
    out, _ = F5HipCFM(eng, precision=prec, ode_method=c.get("method", "euler")).sample(wav.cuda(), text, duration, lens=lens, **c["kw"])
    e = float(torch.cat([(out[b, :durs[b]].cpu() - torch.from_numpy(g[b, :durs[b]])).abs().reshape(-1) for b in range(g.shape[0])]).max())
    res.append(f"{spec} {e:.2e}")
print(os.environ.get("F5HIP_LIB", "in-tree library"), name, "  ".join(res), flush=True)
eng.close()
