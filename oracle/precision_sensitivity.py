"""Which contractions of the path need the fp16 hi/lo split?  (test infrastructure, CPU only — never on the product path)

The HIP path has two half-precision operand modes: `fp16` (operands rounded to fp16, fp32 accumulate) misses
the 1e-3 max-abs bar on the full-size model by 3.4x, `fp16x3` (hi/lo split, 3 MFMAs per product, 2x the operand bytes) meets it at
1.1e-4.  This script measures, on the CPU oracle at full size (golden `base_v1_cfg1`, minted from the reference), how much of the
`fp16` error each CLASS of contraction contributes, by rounding the operands of only that class to fp16 (everything else stays
fp32).  Independent rounding errors add in quadrature, so the per-class numbers say which classes could run in plain fp16 inside a
mixed mode without leaving the tolerance — an experiment that needs no GPU.

    python oracle/precision_sensitivity.py [--classes qkv,out,...] [--sets "qkv+ff1,out+ff2"] [--threads 8]

Classes: qkv, out, ff1, ff2 (DiT block GEMMs), attn_qk (q, k operands of the score product), attn_pv (P, V operands), adaln (all
modulation linears), time (time MLP), inproj (input projection), convpos (grouped conv k=31), text (ConvNeXt pointwise), projout.
"""
import argparse
import json
import os
import re
import sys
import time

import numpy as np
import torch
import torch.nn.functional as TF

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import f5_tts_amd  # noqa: E402,F401
from f5_tts_amd import synth  # noqa: E402
from oracle import f5_oracle as O  # noqa: E402
from oracle import make_golden as MG  # noqa: E402

CLASSES = ["qkv", "out", "ff1", "ff2", "attn_qk", "attn_pv", "adaln", "time", "inproj", "convpos", "text", "projout"]
_PAT = [
    (re.compile(r"attn\.to_[qkv]\.weight$"), "qkv"),
    (re.compile(r"attn\.to_out\.0\.weight$"), "out"),
    (re.compile(r"ff\.ff\.0\.0\.weight$"), "ff1"),
    (re.compile(r"ff\.ff\.2\.weight$"), "ff2"),
    (re.compile(r"(attn_norm|norm_out)\.linear\.weight$"), "adaln"),
    (re.compile(r"time_mlp\.\d\.weight$"), "time"),
    (re.compile(r"input_embed\.proj\.weight$"), "inproj"),
    (re.compile(r"text_blocks\.\d+\.pwconv\d\.weight$"), "text"),
    (re.compile(r"proj_out\.weight$"), "projout"),
    (re.compile(r"conv_pos_embed\.conv1d\.\d\.weight$"), "convpos"),
]


def r16(x):
    return x.half().float()


class Rounder:
    """Stands in for `torch.nn.functional` inside the oracle module: rounds the operands of the selected classes to fp16."""

    def __init__(self, sd, active):
        self.active = set(active)
        self.cls = {}
        for k, v in sd.items():
            for pat, c in _PAT:
                if pat.search(k):
                    self.cls[id(v)] = c
        self.w16 = {}

    def __getattr__(self, name):
        return getattr(TF, name)

    def _w(self, w):
        k = id(w)
        if k not in self.w16:
            self.w16[k] = r16(w)
        return self.w16[k]

    def linear(self, x, w, b=None):
        if self.cls.get(id(w)) in self.active:
            return TF.linear(r16(x), self._w(w), b)
        return TF.linear(x, w, b)

    def conv1d(self, x, w, b=None, *a, **kw):
        if self.cls.get(id(w)) in self.active:
            return TF.conv1d(r16(x), self._w(w), b, *a, **kw)
        return TF.conv1d(x, w, b, *a, **kw)

    def scaled_dot_product_attention(self, q, k, v, attn_mask=None, dropout_p=0.0, is_causal=False):
        qk, pv = "attn_qk" in self.active, "attn_pv" in self.active
        if not (qk or pv):
            return TF.scaled_dot_product_attention(q, k, v, attn_mask=attn_mask, dropout_p=dropout_p, is_causal=is_causal)
        q = q * (q.shape[-1] ** -0.5)  # the QKV epilogue folds the scale into q before the operand is emitted (gemm.h EpiQKV)
        if qk:
            q, k = r16(q), r16(k)
        s = q @ k.transpose(-1, -2)
        if attn_mask is not None:
            s = s.masked_fill(~attn_mask, float("-inf"))
        p = torch.softmax(s, dim=-1)
        if pv:
            # the flash kernel rounds exp(s - running max) (in (0, 1]) before the row sum is divided out: same relative rounding
            m = s.amax(-1, keepdim=True)
            e = torch.exp(s - m)
            return (r16(e) @ r16(v)) / e.sum(-1, keepdim=True)
        return p @ v


def run(sd, cfg, wav, text, duration, lens, kw, active):
    O.F = Rounder(sd, active)
    try:
        out, _ = O.cfm_sample(sd, cfg, wav, text, duration, lens=lens, **kw)
    finally:
        O.F = TF
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--case", default="base_v1_cfg1")
    ap.add_argument("--classes", default=",".join(CLASSES))
    ap.add_argument("--sets", default="", help="comma-separated unions of classes joined by '+', e.g. 'qkv+ff1,all'")
    ap.add_argument("--threads", type=int, default=os.cpu_count() or 1)
    ap.add_argument("--out", default="")
    a = ap.parse_args()
    torch.set_num_threads(a.threads)
    c = MG.FULL_CASES[a.case]
    cfg, wav, text, duration, lens = MG.case_inputs(c)
    sd = synth.synth_dit_state_dict(cfg, seed=c["wseed"])
    gold = torch.as_tensor(np.load(os.path.join(ROOT, "tests", "golden", a.case + ".npz"))["out"])
    ref_len = wav.shape[-1] // 256
    runs = [("none", [])]
    runs += [(s, CLASSES if s == "all" else s.split("+")) for s in a.sets.split(",") if s]
    runs += [(x, [x]) for x in a.classes.split(",") if x]
    res = {}
    for name, active in runs:
        t0 = time.perf_counter()
        out = run(sd, cfg, wav, text, duration, lens, c["kw"], active)
        d = (out - gold)[:, ref_len:].abs()
        res[name] = dict(max_abs=d.max().item(), mean_abs=d.mean().item(), rms=d.pow(2).mean().sqrt().item())
        print(f"{name:28s} max-abs {res[name]['max_abs']:.3e}  mean-abs {res[name]['mean_abs']:.3e}  rms {res[name]['rms']:.3e}"
              f"   ({time.perf_counter() - t0:.0f} s)", flush=True)
        if a.out:
            json.dump(res, open(a.out, "w"), indent=1)


if __name__ == "__main__":
    main()
