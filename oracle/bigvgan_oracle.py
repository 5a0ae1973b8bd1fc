"""CPU restatement (torch fp32) of the BigVGAN-v2 generator — TEST INFRASTRUCTURE, never on the product path.

PARITY UNPINNED.  The reference calls `bigvgan.BigVGAN.from_pretrained(...)`, `.remove_weight_norm()`, `vocoder(mel)`
(/root/reference/src/f5_tts/infer/utils_infer.py:130-144, 512-513; BASELINE.json configs[4]) but the generator's source is an
un-vendored git submodule (`.gitmodules:1-3`, https://github.com/NVIDIA/BigVGAN.git, commit unknown; `src/third_party/BigVGAN/`
is empty) and no checkpoint is reachable offline.  What follows restates the PUBLISHED upstream algorithm (NVIDIA/BigVGAN v2:
`bigvgan.py` BigVGAN / AMPBlock1 / AMPBlock2, `activations.py` Snake / SnakeBeta, `alias_free_activation/torch/{act,resample,
filter}.py`) from its public description; there is no artefact under /root/reference to check it against, so every test that uses
this file checks the HIP path against THIS restatement and the restatement against independently written formulations of the same
arithmetic (tests/test_bigvgan_oracle.py), nothing more.

Tensors are channel-major [b, C, L] as in the upstream module.  `sd` is the generator state dict AFTER weight-norm removal
(`fold_weight_norm` does that for a raw checkpoint): keys `conv_pre.{weight,bias}`, `ups.{i}.0.{weight,bias}`,
`resblocks.{i*nk+j}.convs1.{m}.{weight,bias}`, `.convs2.{m}.*`, `.activations.{q}.act.{alpha,beta}`, `activation_post.act.*`,
`conv_post.weight` (+ `.bias` when `use_bias_at_final`).
"""
from __future__ import annotations

import math
from typing import Dict, List, Tuple

import torch
import torch.nn.functional as F

from f5_tts_amd.config import BigVGANConfig

Tensor = torch.Tensor
SD = Dict[str, Tensor]


# ---- alias_free_activation/torch/filter.py -----------------------------------------------------------------------------------------
def kaiser_sinc_filter1d(cutoff: float, half_width: float, kernel_size: int) -> Tensor:
    even = kernel_size % 2 == 0
    half_size = kernel_size // 2
    delta_f = 4 * half_width
    A = 2.285 * (half_size - 1) * math.pi * delta_f + 7.95
    if A > 50.0:
        beta = 0.1102 * (A - 8.7)
    elif A >= 21.0:
        beta = 0.5842 * (A - 21) ** 0.4 + 0.07886 * (A - 21.0)
    else:
        beta = 0.0
    window = torch.kaiser_window(kernel_size, beta=beta, periodic=False, dtype=torch.float32)
    time = (torch.arange(-half_size, half_size, dtype=torch.float32) + 0.5) if even else (torch.arange(kernel_size, dtype=torch.float32) - half_size)
    if cutoff == 0:
        return torch.zeros_like(time)
    x = 2 * cutoff * time
    sinc = torch.where(x == 0, torch.ones_like(x), torch.sin(math.pi * x) / (math.pi * x))
    filt = 2 * cutoff * window * sinc
    return filt / filt.sum()


def aa_filter(ratio: int = 2) -> Tensor:
    """The one 12-tap filter both resamplers of Activation1d use (up_ratio = down_ratio = 2, kernel 12)."""
    k = int(6 * ratio // 2) * 2
    return kaiser_sinc_filter1d(0.5 / ratio, 0.6 / ratio, k)


# ---- alias_free_activation/torch/resample.py ---------------------------------------------------------------------------------------
def upsample1d(x: Tensor, filt: Tensor, ratio: int = 2) -> Tensor:
    k = filt.numel()
    pad = k // ratio - 1
    pad_left = pad * ratio + (k - ratio) // 2
    pad_right = pad * ratio + (k - ratio + 1) // 2
    c = x.shape[1]
    x = F.pad(x, (pad, pad), mode="replicate")
    x = ratio * F.conv_transpose1d(x, filt.view(1, 1, k).expand(c, -1, -1), stride=ratio, groups=c)
    return x[..., pad_left:-pad_right]


def downsample1d(x: Tensor, filt: Tensor, ratio: int = 2) -> Tensor:
    k = filt.numel()
    pad_left = k // 2 - int(k % 2 == 0)
    pad_right = k // 2
    c = x.shape[1]
    x = F.pad(x, (pad_left, pad_right), mode="replicate")
    return F.conv1d(x, filt.view(1, 1, k).expand(c, -1, -1), stride=ratio, groups=c)


# ---- activations.py ----------------------------------------------------------------------------------------------------------------
def snake(x: Tensor, alpha: Tensor, beta: Tensor, logscale: bool) -> Tensor:
    """SnakeBeta: x + 1/(beta + 1e-9) * sin^2(alpha x); Snake is the same with beta = alpha.  alpha, beta: [C]."""
    a, b = alpha.view(1, -1, 1), beta.view(1, -1, 1)
    if logscale:
        a, b = torch.exp(a), torch.exp(b)
    return x + (1.0 / (b + 1e-9)) * torch.sin(x * a).pow(2)


def activation1d(sd: SD, pfx: str, x: Tensor, cfg: BigVGANConfig, filt: Tensor) -> Tensor:
    """alias_free_activation/torch/act.py Activation1d: upsample x2 -> activation -> downsample x2."""
    alpha = sd[pfx + "act.alpha"]
    beta = sd[pfx + "act.beta"] if cfg.activation == "snakebeta" else alpha
    return downsample1d(snake(upsample1d(x, filt), alpha, beta, cfg.snake_logscale), filt)


# ---- bigvgan.py --------------------------------------------------------------------------------------------------------------------
def amp_block(sd: SD, pfx: str, x: Tensor, cfg: BigVGANConfig, k: int, dilations: Tuple[int, ...], filt: Tensor) -> Tensor:
    if cfg.resblock == "1":
        for m, d in enumerate(dilations):
            xt = activation1d(sd, f"{pfx}activations.{2 * m}.", x, cfg, filt)
            xt = F.conv1d(xt, sd[f"{pfx}convs1.{m}.weight"], sd[f"{pfx}convs1.{m}.bias"], dilation=d, padding=(k * d - d) // 2)
            xt = activation1d(sd, f"{pfx}activations.{2 * m + 1}.", xt, cfg, filt)
            xt = F.conv1d(xt, sd[f"{pfx}convs2.{m}.weight"], sd[f"{pfx}convs2.{m}.bias"], dilation=1, padding=(k - 1) // 2)
            x = xt + x
        return x
    for m, d in enumerate(dilations):  # AMPBlock2
        xt = activation1d(sd, f"{pfx}activations.{m}.", x, cfg, filt)
        xt = F.conv1d(xt, sd[f"{pfx}convs.{m}.weight"], sd[f"{pfx}convs.{m}.bias"], dilation=d, padding=(k * d - d) // 2)
        x = xt + x
    return x


def bigvgan_forward(sd: SD, cfg: BigVGANConfig, mel: Tensor, return_stages: bool = False):
    """BigVGAN.forward: mel [b, num_mels, T] -> wav [b, 1, T * hop]."""
    filt = aa_filter()
    stages: List[Tensor] = []
    x = F.conv1d(mel, sd["conv_pre.weight"], sd["conv_pre.bias"], padding=3)
    stages.append(x)
    nk = len(cfg.resblock_kernel_sizes)
    for i, (u, k) in enumerate(zip(cfg.upsample_rates, cfg.upsample_kernel_sizes)):
        x = F.conv_transpose1d(x, sd[f"ups.{i}.0.weight"], sd[f"ups.{i}.0.bias"], stride=u, padding=(k - u) // 2)
        xs = None
        for j in range(nk):
            y = amp_block(sd, f"resblocks.{i * nk + j}.", x, cfg, cfg.resblock_kernel_sizes[j], cfg.resblock_dilation_sizes[j], filt)
            xs = y if xs is None else xs + y
        x = xs / nk
        stages.append(x)
    x = activation1d(sd, "activation_post.", x, cfg, filt)
    x = F.conv1d(x, sd["conv_post.weight"], sd.get("conv_post.bias") if cfg.use_bias_at_final else None, padding=3)
    x = torch.tanh(x) if cfg.use_tanh_at_final else torch.clamp(x, min=-1.0, max=1.0)
    return (x, stages) if return_stages else x


# ---- checkpoint handling -----------------------------------------------------------------------------------------------------------
def fold_weight_norm(sd: SD) -> SD:
    """`remove_weight_norm()` on a raw generator state dict: w = g * v / ||v|| with the norm over every dim but 0 (torch.nn.utils.
    weight_norm, dim = 0 — also for ConvTranspose1d, whose dim 0 is the INPUT channel).  Accepts the old (`weight_g`/`weight_v`) and
    the parametrised (`parametrizations.weight.original0/1`) spellings; resampling-filter buffers are dropped (recomputed)."""
    out: SD = {}
    for k, v in sd.items():
        if k.endswith(".filter"):
            continue
        if k.endswith(".weight_g") or k.endswith(".parametrizations.weight.original0"):
            base = k[: -len(".weight_g")] if k.endswith(".weight_g") else k[: -len(".parametrizations.weight.original0")]
            vv = sd.get(base + ".weight_v", sd.get(base + ".parametrizations.weight.original1"))
            g = v.float()
            vv = vv.float()
            norm = vv.reshape(vv.shape[0], -1).norm(dim=1).view(-1, *([1] * (vv.ndim - 1)))
            out[base + ".weight"] = g * vv / norm
        elif k.endswith(".weight_v") or k.endswith(".parametrizations.weight.original1"):
            continue
        else:
            out[k] = v.float()
    return out
