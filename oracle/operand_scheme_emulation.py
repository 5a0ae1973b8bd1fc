"""How many matrix-core products does a block GEMM need?  (test infrastructure, CPU only — never on the product path)

The `fp16x3` operand mode of the HIP path computes every block-GEMM product as `A_hi W_hi + A_hi W_lo + A_lo W_hi` (fp16 planes, fp32
accumulate): three fp16 MFMAs per product.  VERDICT r03 asks whether the q|k|v and out-projection GEMMs — whose results (q, k, v) or
inputs (the attention output) pass through fp16 anyway — need the third MFMA.  This script answers on the CPU, at the full model size,
against goldens minted from the reference, by EMULATING candidate operand schemes inside the oracle's `F.linear` (fp32 accumulate, exactly
the planes the kernel would multiply), with the attention operands rounded to fp16 as the engine's default does:

    x3    A_hi W_hi + A_hi W_lo + A_lo W_hi        (the shipped scheme: calibrates the emulation against the GPU's measured error)
    ahi   A_hi (W_hi + W_lo)                        (activations at 11 bits: 2 MFMAs)
    whi   (A_hi + A_lo) W_hi                        (weights at 11 bits: 2 MFMAs)
    x1    A_hi W_hi                                 (plain fp16)
    mx6   A_hi W_hi + mx6(A_hi) mx6(W_lo) + mx6(A_lo) mx6(W_hi)
          the two correction terms as MX-fp6 (e2m3, one power-of-two scale per 32 consecutive k): gfx950 runs
          v_mfma_scale_f32_32x32x64_f8f6f4 with fp6 operands at 4x the fp16 rate, so this product costs 1 + 2/4 = 1.5 fp16 MFMAs
    mx8   the same with e4m3 elements (2x the fp16 rate: 2.0 fp16 MFMAs; same 4-bit significand, wider exponent)
    mx4   the same with e2m1 elements (fp4)
    mx6k  mx6 with the KERNEL's blocks (round 5): one scale per 16 values — the k-set a lane owns, k = 8 q + 4 h + e of a 32-k block, not 16
          consecutive k —, scale exponent floored at MX_MIN_EXP, the remainder taken x 2^11 before the conversion (common.h mx_pack16)
    exact the fp32 product (isolates the other error sources)

`--attn` chooses what the attention products round (round 5, the error decomposition of VERDICT r04 item 4): fp16 (the engine's default:
q, k, P, V in fp16), fp32 (nothing), qk (q and k only), pv (P and V only); round 6: p16 / v16 (P only / V only), vs (P fp16, V as hi + lo),
vmx (P fp16, the V_lo P term as MX-fp6 over blocks of 32 keys).

    python oracle/operand_scheme_emulation.py --case base_v1_cfg1 --runs "all=x3;qkv=ahi;qkv=whi;out=ahi;all=mx6"

A run is `class[+class..]=scheme[,class=scheme..]`; classes qkv, out, ff1, ff2, all; classes a run does not name use x3.
Weights are conditioned like the engine's (row n scaled by a power of two so that its largest entry sits in [2^12, 2^13), undone exactly).
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
from oracle import f5_oracle as O  # noqa: E402
from oracle import make_golden as MG  # noqa: E402

_PAT = [
    (re.compile(r"attn\.to_[qkv]\.weight$"), "qkv"),
    (re.compile(r"attn\.to_out\.0\.weight$"), "out"),
    (re.compile(r"ff\.ff\.0\.0\.weight$"), "ff1"),
    (re.compile(r"ff\.ff\.2\.weight$"), "ff2"),
    (re.compile(r"\.4\.ff\.0\.0\.weight$"), "ff1"),  # UNetT layers
    (re.compile(r"\.4\.ff\.2\.weight$"), "ff2"),
]
GEMMS = ["qkv", "out", "ff1", "ff2"]


def r16(x):
    return x.half().float()


def mx_quant(x, fmt):
    """Round to an MX block format: blocks of 32 along the last axis share a power-of-two scale; elements e2m3 / e4m3 / e2m1."""
    k = x.shape[-1]
    assert k % 32 == 0
    xb = x.reshape(*x.shape[:-1], k // 32, 32)
    amax = xb.abs().amax(-1, keepdim=True)
    top = dict(e2m3=7.5, e4m3=448.0, e2m1=6.0)[fmt]
    emax = dict(e2m3=2, e4m3=8, e2m1=2)[fmt]           # exponent of the largest binade of the element format
    e = torch.floor(torch.log2(amax.clamp_min(1e-38))) - emax
    y = xb / torch.exp2(e)
    # the block maximum may round above the format's largest value: move such blocks one binade down
    bump = (y.abs().amax(-1, keepdim=True) > top * (1 + 2.0 ** -(dict(e2m3=5, e4m3=5, e2m1=3)[fmt])))
    e = e + bump.to(e.dtype)
    s = torch.exp2(e)
    y = xb / s
    a = y.abs()
    mant = dict(e2m3=3, e4m3=3, e2m1=1)[fmt]
    emin = dict(e2m3=0, e4m3=-6, e2m1=0)[fmt]          # exponent of the smallest normal binade
    be = torch.floor(torch.log2(a.clamp_min(2.0 ** (emin - 40)))).clamp_min(emin)
    step = torch.exp2(be - mant)
    q = (torch.round(a / step) * step).clamp_max(top)
    out = torch.sign(y) * q * s
    out = torch.where(amax > 0, out, torch.zeros_like(out))
    return out.reshape(x.shape)


def mx6_kernel(x, remainder):
    """The kernel's own MX-fp6 rounding of an operand plane (common.h mx_pack16): per 32-k block the two lane sets {8 q + 4 h + e}, h = 0, 1, of
    16 values each take their scale from the largest |VALUE| of the set (S = 2^(E - 2)); `remainder` planes (x - fp16(x)) are multiplied by
    2^11 first (exact) and carry the scale byte E - 2 - 11, i.e. the same relative grid.  Here: x is the plane to encode, given as a pair
    (values the scale is taken from, values to round)."""
    vals, plane = x
    k = vals.shape[-1]
    assert k % 32 == 0
    idx = torch.arange(32).reshape(4, 2, 4)            # (q, h, e) -> k = 8 q + 4 h + e
    perm = torch.cat([idx[:, 0, :].reshape(-1), idx[:, 1, :].reshape(-1)])  # h = 0 set, then h = 1 set
    vb = vals.reshape(*vals.shape[:-1], k // 32, 32)[..., perm].reshape(*vals.shape[:-1], k // 32, 2, 16)
    pb = plane.reshape(*plane.shape[:-1], k // 32, 32)[..., perm].reshape(*plane.shape[:-1], k // 32, 2, 16)
    amax = vb.abs().amax(-1, keepdim=True)
    ex = torch.floor(torch.log2(amax.clamp_min(2.0 ** (16 - 127))))          # biased exponent floored at MX_MIN_EXP = 16
    S = torch.exp2(ex - 2)
    y = (pb * (2048.0 if remainder else 1.0)) / S
    a = y.abs()
    be = torch.floor(torch.log2(a.clamp_min(2.0 ** -40))).clamp_min(0)      # e2m3: smallest normal binade 2^0, subnormal step 2^-3
    step = torch.exp2(be - 3)
    q = (torch.round(a / step) * step).clamp_max(7.5)                        # v_cvt_scalef32 saturates at 7.5
    out = torch.sign(y) * q * S / (2048.0 if remainder else 1.0)
    inv = torch.argsort(perm)
    return out.reshape(*plane.shape[:-1], k // 32, 32)[..., inv].reshape(plane.shape)


class Schemes:
    """Stands in for `torch.nn.functional` inside the oracle module."""

    def __init__(self, sd, plan, attn="fp16"):
        self.attn = attn
        self.plan = plan  # class -> scheme
        self.cls = {}
        for k, v in sd.items():
            for pat, c in _PAT:
                if pat.search(k):
                    self.cls[id(v)] = c
        self.wc = {}

    def __getattr__(self, name):
        return getattr(TF, name)

    def _w(self, w):
        k = id(w)
        if k not in self.wc:
            amax = w.abs().amax(1, keepdim=True).clamp_min(1e-30)
            alpha = torch.exp2(12 - torch.floor(torch.log2(amax)))      # row maximum -> [2^12, 2^13)
            ws = w * alpha
            hi = r16(ws)
            lo = r16(ws - hi)
            self.wc[k] = dict(alpha=alpha.reshape(-1), hi=hi, lo=lo)
        return self.wc[k]

    def _wq(self, w, name, fmt):
        c = self._w(w)
        key = name + fmt
        if key not in c:
            c[key] = mx_quant(c[name], fmt)
        return c[key]

    def linear(self, x, w, b=None):
        cl = self.cls.get(id(w))
        if cl is None:
            return TF.linear(x, w, b)
        sch = self.plan.get(cl, "x3")
        c = self._w(w)
        xh = r16(x)
        xl = r16(x - xh)
        if sch == "x3":
            y = TF.linear(xh, c["hi"]) + (TF.linear(xh, c["lo"]) + TF.linear(xl, c["hi"]))
        elif sch == "ahi":
            y = TF.linear(xh, c["hi"]) + TF.linear(xh, c["lo"])
        elif sch == "whi":
            y = TF.linear(xh, c["hi"]) + TF.linear(xl, c["hi"])
        elif sch == "x1":
            y = TF.linear(xh, c["hi"])
        elif sch == "exact":
            return TF.linear(x, w, b)
        elif sch == "mx6k":
            ws = c["hi"] + c["lo"]   # the conditioned weight values (the scale of a weight set is taken from them)
            xv = xh + xl
            y = TF.linear(xh, c["hi"]) + (TF.linear(mx6_kernel((xv, xh), False), mx6_kernel((ws, c["lo"]), True)) +
                                          TF.linear(mx6_kernel((xv, xl), True), mx6_kernel((ws, c["hi"]), False)))
        elif sch in ("mx6", "mx8", "mx4"):
            fmt = dict(mx6="e2m3", mx8="e4m3", mx4="e2m1")[sch]
            y = TF.linear(xh, c["hi"]) + (TF.linear(mx_quant(xh, fmt), self._wq(w, "lo", fmt)) + TF.linear(mx_quant(xl, fmt), self._wq(w, "hi", fmt)))
        else:
            raise ValueError(sch)
        y = y / c["alpha"]
        return y if b is None else y + b

    def scaled_dot_product_attention(self, q, k, v, attn_mask=None, dropout_p=0.0, is_causal=False):
        # the engine's default: plain fp16 q (pre-scaled), k, P, V; fp32 softmax and accumulators
        if self.attn == "fp32":
            return TF.scaled_dot_product_attention(q, k, v, attn_mask=attn_mask)
        mode = self.attn
        rp = r16 if mode in ("fp16", "pv", "qkm", "qkc", "q1", "qks") else (lambda t: t)
        # round 6 (the sharpness sweep: the error of fp16 P.V grows with the logits): which of the two operands carries it?
        #   p16  P rounded to fp16 (numerator and row sum over the rounded values), V exact     v16  V rounded, P exact
        #   vs   P rounded, V as hi + lo halves (V_hi P + V_lo P: one more fp16 MFMA per product)   vmx  the V_lo P term as MX-fp6 over 32 keys
        rP = r16 if mode in ("p16", "vs", "vmx") else rp
        rV = r16 if mode == "v16" else rp
        q = q * (q.shape[-1] ** -0.5)
        if mode == "qkc":   # softmax is invariant under a shift of every key by one vector: centre the keys of a sequence before rounding them
            if attn_mask is not None:
                w = attn_mask[:, :1, :1, :].transpose(-1, -2).to(k.dtype)   # [B,1,N,1] valid keys
                k = k - (k * w).sum(-2, keepdim=True) / w.sum(-2, keepdim=True)
            else:
                k = k - k.mean(-2, keepdim=True)
        if mode in ("fp16", "qk", "qkc"):
            s = r16(q) @ r16(k).transpose(-1, -2)
        elif mode == "qks":  # attn_impl 4: hi/lo-split q and k, three fp16 products
            qh, kh = r16(q), r16(k)
            ql, kl = r16(q - qh), r16(k - kh)
            s = qh @ kh.transpose(-1, -2) + (ql @ kh.transpose(-1, -2) + qh @ kl.transpose(-1, -2))
        elif mode == "q1":   # q split only (the operand a flash kernel holds in registers), k rounded
            qh, kh = r16(q), r16(k)
            s = qh @ kh.transpose(-1, -2) + r16(q - qh) @ kh.transpose(-1, -2)
        elif mode == "qkm":  # hi.hi in fp16, the two correction products as MX-fp6 over blocks of 32 head dims (what the block GEMMs do)
            qh, kh = r16(q), r16(k)
            ql, kl = q - qh, k - kh
            s = qh @ kh.transpose(-1, -2) + (mx_quant(ql, "e2m3") @ mx_quant(kh, "e2m3").transpose(-1, -2)
                                             + mx_quant(qh, "e2m3") @ mx_quant(kl, "e2m3").transpose(-1, -2))
        else:
            s = q @ k.transpose(-1, -2)
        if attn_mask is not None:
            s = s.masked_fill(~attn_mask, float("-inf"))
        m = s.amax(-1, keepdim=True)
        self.max_logit = max(getattr(self, "max_logit", 0.0), float(s[torch.isfinite(s)].abs().max()))
        e = torch.exp(s - m)
        e16 = rP(e)
        if mode == "vs":
            vh = r16(v)
            return (e16 @ vh + e16 @ r16(v - vh)) / e16.sum(-1, keepdim=True)
        if mode == "vmx":  # blocks of 32 consecutive keys share a scale (P block of a query row; V block of a channel)
            vh = r16(v)
            n = e16.shape[-1]
            pad = (-n) % 32
            ep = TF.pad(e16, (0, pad))
            vl = TF.pad((v - vh).transpose(-1, -2), (0, pad))  # [.., d, keys]
            return (e16 @ vh + mx_quant(ep, "e2m3") @ mx_quant(vl, "e2m3").transpose(-1, -2)) / e16.sum(-1, keepdim=True)
        return (e16 @ rV(v)) / e16.sum(-1, keepdim=True)


def parse_run(spec):
    plan = {}
    for part in spec.split(","):
        cl, sch = part.split("=")
        for c in (GEMMS if cl == "all" else cl.split("+")):
            plan[c] = sch
    return plan


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--case", default="base_v1_cfg1")
    ap.add_argument("--runs", default="all=x3;qkv=ahi;qkv=whi;out=ahi;out=whi;all=mx6")
    ap.add_argument("--threads", type=int, default=os.cpu_count() or 1)
    ap.add_argument("--out", default="")
    ap.add_argument("--attn", default="fp16", help="fp16 | fp32 | qk | pv, or a ;-separated list as long as --runs")
    a = ap.parse_args()
    torch.set_num_threads(a.threads)
    c = MG.FULL_CASES[a.case]
    cfg, wav, text, duration, lens = MG.case_inputs(c)
    sd = MG.case_weights(c)
    gold = torch.as_tensor(np.load(os.path.join(ROOT, "tests", "golden", a.case + ".npz"))["out"])
    ref_len = wav.shape[-1] // 256
    res = {}
    specs = a.runs.split(";")
    attns = a.attn.split(";") if ";" in a.attn else [a.attn] * len(specs)
    durs = duration.tolist() if torch.is_tensor(duration) else [int(duration)] * gold.shape[0]
    lens_l = lens.tolist() if lens is not None else [ref_len] * gold.shape[0]
    for spec, attn in zip(specs, attns):
        t0 = time.perf_counter()
        O_F = O.F = Schemes(sd, parse_run(spec), attn)
        try:
            out, _ = O.cfm_sample(sd, cfg, wav, text, duration, lens=lens, **c["kw"])
        finally:
            O.F = TF
        d = torch.cat([(out[b, lens_l[b]:durs[b]] - gold[b, lens_l[b]:durs[b]]).abs().reshape(-1) for b in range(gold.shape[0])])  # the generated frames of every row
        spec = f"{spec} attn={attn}"
        res[spec] = dict(max_abs=d.max().item(), mean_abs=d.mean().item(), rms=d.pow(2).mean().sqrt().item())
        tag = spec + (f" |s|max {O_F.max_logit:.0f}" if hasattr(O_F, "max_logit") else "")
        print(f"{a.case:16s} {tag:44s} max-abs {res[spec]['max_abs']:.3e}  mean-abs {res[spec]['mean_abs']:.3e}  rms {res[spec]['rms']:.3e}"
              f"   ({time.perf_counter() - t0:.0f} s)", flush=True)
        if a.out:
            json.dump(res, open(a.out, "w"), indent=1)


if __name__ == "__main__":
    main()
