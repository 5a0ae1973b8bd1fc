"""How sharp is a set of weights' attention — and what does the default half-precision attention cost on it?  For each golden named on the
command line: one fp32 pass with option "attn_stats" (the materialised-score attention publishes every softmax row's largest probability:
include/f5hip.h f5hip_attention_stats), then the generated-mel max-abs error of chosen (precision:attn_impl) pairs against the golden.  The
table behind INTEGRATION.md, "Which attention form does my checkpoint need?".

    python tools/attention_sharpness.py base_v1_trained_like base_v1_trained_like_sharp1p4 ... [-- fp16m:0 fp16m:6 fp16m:7]     (GPU box)
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

args = sys.argv[1:]
specs = ["fp16m:0", "fp16m:6", "fp16m:7"]
if "--" in args:
    specs = args[args.index("--") + 1:]
    args = args[:args.index("--")]
print("| golden | rows | mean of the rows' largest probability | largest | rows above 1/2 | " + " | ".join(f"`{s}`" for s in specs) + " |")
print("|---|---:|---:|---:|---:|" + "---:|" * len(specs), flush=True)
ALL = {**MG.CASES, **MG.SWEEP_CASES, **MG.FULL_CASES}
for name in args:
    c = ALL[name]
    cfg, wav, text, duration, lens = MG.case_inputs(c)
    eng = F5HipEngine(cfg, None, device=0)
    eng.load_state_dict(MG.case_weights(c))
    run = lambda prec: F5HipCFM(eng, precision=prec, ode_method=c.get("method", "euler")).sample(wav.cuda(), text, duration, lens=lens, **c["kw"])[0]  # noqa: E731
    eng.set_option("attn_stats", 1)
    run("fp32")
    st = eng.attention_stats()
    eng.set_option("attn_stats", 0)
    path = os.path.join(ROOT, "tests", "golden", name + ".npz")
    errs = []
    if os.path.exists(path):
        g = np.load(path)["out"]
        durs = duration.tolist() if torch.is_tensor(duration) else [int(duration)] * g.shape[0]
        for spec in specs:
            prec, impl = spec.split(":")
            eng.set_option("attn_impl", int(impl))
            out = run(prec)
            errs.append("%.2e" % float(torch.cat([(out[b, :durs[b]].cpu() - torch.from_numpy(g[b, :durs[b]])).abs().reshape(-1) for b in range(g.shape[0])]).max()))
    else:
        errs = ["(no fixture)"] * len(specs)
    print(f"| `{name}` | {st['rows']} | {st['mean_max_prob']:.4f} | {st['max_prob']:.4f} | {st['frac_rows_above_half']:.4f} | " + " | ".join(errs) + " |", flush=True)
    eng.close()
