"""Seeded synthetic weights and inputs (no pretrained checkpoints are reachable offline).

The state-dict keys follow the reference's on-disk contract exactly (reference
``src/f5_tts/infer/utils_infer.py:190-232`` loads them with ``load_state_dict``), so a dict made
here loads into the reference ``CFM`` with ``strict=True`` — the golden generator does exactly that.

The reference zero-initialises AdaLN, the output projection (``model/backbones/dit.py:264-274``)
and the GRN gamma/beta (``model/modules.py:239-240``); a fresh model therefore predicts zero
velocity.  All those tensors get seeded NON-zero values here, otherwise parity would be vacuous.
Everything is drawn from one CPU ``torch.Generator`` in a fixed order, so the same
``(config, seed)`` gives bit-identical tensors on every box with the same torch build.
"""
from __future__ import annotations

import math
import re
from typing import Dict

import torch

from .config import BigVGANConfig, DiTConfig, VocosConfig


def _normal(g: torch.Generator, shape, std: float) -> torch.Tensor:
    return torch.randn(shape, generator=g, dtype=torch.float32) * std


def synth_dit_state_dict(cfg: DiTConfig, seed: int = 0) -> Dict[str, torch.Tensor]:
    """Random-init weights of the DiT backbone with the reference's key names (``transformer.*``)."""
    g = torch.Generator(device="cpu")
    g.manual_seed(1000003 * seed + 17)
    sd: Dict[str, torch.Tensor] = {}
    D, T, mel = cfg.dim, cfg.text_dim, cfg.mel_dim
    p = "transformer."

    def linear(name, out_f, in_f, w_std=None, b_std=0.02):
        sd[p + name + ".weight"] = _normal(g, (out_f, in_f), w_std if w_std is not None else 1.0 / math.sqrt(in_f))
        sd[p + name + ".bias"] = _normal(g, (out_f,), b_std)

    if cfg.backbone == "MMDiT":
        return _synth_mmdit(cfg, g, sd, linear)
    linear("time_embed.time_mlp.0", D, 256)
    linear("time_embed.time_mlp.2", D, D)
    sd[p + "text_embed.text_embed.weight"] = _normal(g, (cfg.text_num_embeds + 1, T), 1.0)
    for i in range(cfg.conv_layers):
        b = f"text_embed.text_blocks.{i}."
        sd[p + b + "dwconv.weight"] = _normal(g, (T, 1, 7), 1.0 / math.sqrt(7))
        sd[p + b + "dwconv.bias"] = _normal(g, (T,), 0.02)
        sd[p + b + "norm.weight"] = 1.0 + _normal(g, (T,), 0.05)
        sd[p + b + "norm.bias"] = _normal(g, (T,), 0.05)
        linear(b + "pwconv1", 2 * T, T)
        sd[p + b + "grn.gamma"] = _normal(g, (1, 1, 2 * T), 0.2)
        sd[p + b + "grn.beta"] = _normal(g, (1, 1, 2 * T), 0.05)
        linear(b + "pwconv2", T, 2 * T)
    linear("input_embed.proj", D, 2 * mel + T)
    cpg = D // cfg.conv_pos_groups
    for j in (0, 2):
        b = f"input_embed.conv_pos_embed.conv1d.{j}."
        sd[p + b + "weight"] = _normal(g, (D, cpg, cfg.conv_pos_kernel), 1.0 / math.sqrt(cpg * cfg.conv_pos_kernel))
        sd[p + b + "bias"] = _normal(g, (D,), 0.02)
    sd[p + "rotary_embed.inv_freq"] = 1.0 / (10000.0 ** (torch.arange(0, cfg.dim_head, 2).float() / cfg.dim_head))
    inner = cfg.heads * cfg.dim_head
    if cfg.backbone == "UNetT":  # reference src/f5_tts/model/backbones/unett.py:147-186 key layout: layers.{i}.{0..4}
        for i in range(cfg.depth):
            b = f"layers.{i}."
            if i >= cfg.depth // 2 and cfg.skip_connect_type == "concat":
                sd[p + b + "0.weight"] = _normal(g, (D, 2 * D), 1.0 / math.sqrt(2 * D))  # skip_proj, no bias
            sd[p + b + "1.g"] = 1.0 + _normal(g, (D,), 0.05)
            linear(b + "2.to_q", inner, D)
            linear(b + "2.to_k", inner, D)
            linear(b + "2.to_v", inner, D)
            linear(b + "2.to_out.0", D, inner, w_std=0.3 / math.sqrt(inner))
            sd[p + b + "3.g"] = 1.0 + _normal(g, (D,), 0.05)
            linear(b + "4.ff.0.0", cfg.ff_inner, D)
            linear(b + "4.ff.2", D, cfg.ff_inner, w_std=0.3 / math.sqrt(cfg.ff_inner))
        sd[p + "norm_out.g"] = 1.0 + _normal(g, (D,), 0.05)
        linear("proj_out", mel, D, w_std=0.04, b_std=0.02)
        if cfg.qk_norm == "rms_norm":  # drawn last: the tensors above do not depend on the switch
            for i in range(cfg.depth):
                sd[p + f"layers.{i}.2.q_norm.weight"] = 1.0 + _normal(g, (cfg.dim_head,), 0.1)
                sd[p + f"layers.{i}.2.k_norm.weight"] = 1.0 + _normal(g, (cfg.dim_head,), 0.1)
        return sd
    for i in range(cfg.depth):
        b = f"transformer_blocks.{i}."
        linear(b + "attn_norm.linear", 6 * D, D, w_std=0.02, b_std=0.05)  # zero-init in the reference
        linear(b + "attn.to_q", inner, D)
        linear(b + "attn.to_k", inner, D)
        linear(b + "attn.to_v", inner, D)
        linear(b + "attn.to_out.0", D, inner)
        linear(b + "ff.ff.0.0", cfg.ff_inner, D)
        linear(b + "ff.ff.2", D, cfg.ff_inner)
    linear("norm_out.linear", 2 * D, D, w_std=0.02, b_std=0.05)  # zero-init in the reference
    linear("proj_out", mel, D, w_std=0.04, b_std=0.02)  # zero-init in the reference
    # optional variants: drawn last so the tensors above are bit-identical with or without them
    if cfg.qk_norm == "rms_norm":
        for i in range(cfg.depth):
            sd[p + f"transformer_blocks.{i}.attn.q_norm.weight"] = 1.0 + _normal(g, (cfg.dim_head,), 0.1)
            sd[p + f"transformer_blocks.{i}.attn.k_norm.weight"] = 1.0 + _normal(g, (cfg.dim_head,), 0.1)
    if cfg.long_skip_connection:
        sd[p + "long_skip_connection.weight"] = _normal(g, (D, 2 * D), 1.0 / math.sqrt(2 * D))
    return sd


def _synth_mmdit(cfg: DiTConfig, g: torch.Generator, sd: Dict[str, torch.Tensor], linear) -> Dict[str, torch.Tensor]:
    """MMDiT key layout (reference src/f5_tts/model/backbones/mmdit.py:112-134, modules.py:773-814): per block attn_norm_c / attn_norm_x,
    attn.{to_q,to_k,to_v,to_q_c,to_k_c,to_v_c,to_out.0,to_out_c}, ff_c, ff_x; the last block is context_pre_only (AdaLayerNorm_Final for
    the text stream, no to_out_c / ff_c).  The zero-initialised tensors (mmdit.py:160-171) get non-zero values, as for the DiT."""
    D, mel, p = cfg.dim, cfg.mel_dim, "transformer."
    inner = cfg.heads * cfg.dim_head
    linear("time_embed.time_mlp.0", D, 256)
    linear("time_embed.time_mlp.2", D, D)
    sd[p + "text_embed.text_embed.weight"] = _normal(g, (cfg.text_num_embeds + 1, D), 1.0)
    linear("audio_embed.linear", D, 2 * mel)
    cpg = D // cfg.conv_pos_groups
    for j in (0, 2):
        b = f"audio_embed.conv_pos_embed.conv1d.{j}."
        sd[p + b + "weight"] = _normal(g, (D, cpg, cfg.conv_pos_kernel), 1.0 / math.sqrt(cpg * cfg.conv_pos_kernel))
        sd[p + b + "bias"] = _normal(g, (D,), 0.02)
    sd[p + "rotary_embed.inv_freq"] = 1.0 / (10000.0 ** (torch.arange(0, cfg.dim_head, 2).float() / cfg.dim_head))
    for i in range(cfg.depth):
        b = f"transformer_blocks.{i}."
        last = i == cfg.depth - 1
        linear(b + "attn_norm_c.linear", (2 if last else 6) * D, D, w_std=0.02, b_std=0.05)
        linear(b + "attn_norm_x.linear", 6 * D, D, w_std=0.02, b_std=0.05)
        for name in ("to_q", "to_k", "to_v", "to_q_c", "to_k_c", "to_v_c"):
            linear(b + "attn." + name, inner, D)
        linear(b + "attn.to_out.0", D, inner)
        if not last:
            linear(b + "attn.to_out_c", D, inner)
            linear(b + "ff_c.ff.0.0", cfg.ff_inner, D)
            linear(b + "ff_c.ff.2", D, cfg.ff_inner)
        linear(b + "ff_x.ff.0.0", cfg.ff_inner, D)
        linear(b + "ff_x.ff.2", D, cfg.ff_inner)
        if cfg.qk_norm == "rms_norm":
            for name in ("q_norm", "k_norm", "c_q_norm", "c_k_norm"):
                sd[p + b + f"attn.{name}.weight"] = 1.0 + _normal(g, (cfg.dim_head,), 0.1)
    linear("norm_out.linear", 2 * D, D, w_std=0.02, b_std=0.05)
    linear("proj_out", mel, D, w_std=0.04, b_std=0.02)
    return sd


def synth_vocos_state_dict(cfg: VocosConfig, seed: int = 0) -> Dict[str, torch.Tensor]:
    """Random-init Vocos (mel -> iSTFT head) weights with the ``vocos`` package's key names."""
    g = torch.Generator(device="cpu")
    g.manual_seed(7000003 * seed + 29)
    sd: Dict[str, torch.Tensor] = {}
    C, I, inp = cfg.dim, cfg.intermediate_dim, cfg.input_channels
    sd["backbone.embed.weight"] = _normal(g, (C, inp, 7), 1.0 / math.sqrt(inp * 7))
    sd["backbone.embed.bias"] = _normal(g, (C,), 0.02)
    sd["backbone.norm.weight"] = 1.0 + _normal(g, (C,), 0.05)
    sd["backbone.norm.bias"] = _normal(g, (C,), 0.05)
    for i in range(cfg.num_layers):
        b = f"backbone.convnext.{i}."
        sd[b + "dwconv.weight"] = _normal(g, (C, 1, 7), 1.0 / math.sqrt(7))
        sd[b + "dwconv.bias"] = _normal(g, (C,), 0.02)
        sd[b + "norm.weight"] = 1.0 + _normal(g, (C,), 0.05)
        sd[b + "norm.bias"] = _normal(g, (C,), 0.05)
        sd[b + "pwconv1.weight"] = _normal(g, (I, C), 1.0 / math.sqrt(C))
        sd[b + "pwconv1.bias"] = _normal(g, (I,), 0.02)
        sd[b + "pwconv2.weight"] = _normal(g, (C, I), 1.0 / math.sqrt(I))
        sd[b + "pwconv2.bias"] = _normal(g, (C,), 0.02)
        sd[b + "gamma"] = 1.0 / cfg.num_layers + _normal(g, (C,), 0.02)
    sd["backbone.final_layer_norm.weight"] = 1.0 + _normal(g, (C,), 0.05)
    sd["backbone.final_layer_norm.bias"] = _normal(g, (C,), 0.05)
    # head: keep log-magnitudes modest so exp() stays well below the 1e2 clip on most bins
    sd["head.out.weight"] = _normal(g, (cfg.n_fft + 2, C), 0.5 / math.sqrt(C))
    sd["head.out.bias"] = _normal(g, (cfg.n_fft + 2,), 0.1)
    sd["head.istft.window"] = torch.hann_window(cfg.n_fft)
    return sd


def synth_bigvgan_state_dict(cfg: BigVGANConfig, seed: int = 0, raw_weight_norm: bool = False) -> Dict[str, torch.Tensor]:
    """Random-init BigVGAN generator weights with upstream's key names (NVIDIA/BigVGAN ``bigvgan.py``), activations of O(1) through
    all stages.  ``raw_weight_norm``: the checkpoint spelling BEFORE ``remove_weight_norm()`` (``weight_g`` / ``weight_v``)."""
    g = torch.Generator(device="cpu")
    g.manual_seed(9000011 * seed + 31)
    sd: Dict[str, torch.Tensor] = {}

    def conv(name, cout, cin, k, bias=True, transpose=False, gain=1.0):
        fan = cin * k if not transpose else cin * k / max(1, cfg.upsample_rates[int(name.split(".")[1])])
        w = torch.randn((cin, cout, k) if transpose else (cout, cin, k), generator=g) * (gain / math.sqrt(fan))
        if raw_weight_norm:
            norm = w.reshape(w.shape[0], -1).norm(dim=1).view(-1, 1, 1)
            sd[name + ".weight_g"] = norm * (1.0 + 0.1 * torch.randn(norm.shape, generator=g))
            sd[name + ".weight_v"] = w * (0.5 + torch.rand(norm.shape, generator=g))
        else:
            sd[name + ".weight"] = w
        if bias:
            sd[name + ".bias"] = 0.1 * torch.randn(cout, generator=g)

    def act(name, ch):
        sd[name + ".act.alpha"] = 0.3 * torch.randn(ch, generator=g) if cfg.snake_logscale else 1.0 + 0.2 * torch.randn(ch, generator=g).abs()
        if cfg.activation == "snakebeta":
            sd[name + ".act.beta"] = 0.3 * torch.randn(ch, generator=g) if cfg.snake_logscale else 1.0 + 0.2 * torch.randn(ch, generator=g).abs()

    c0 = cfg.upsample_initial_channel
    conv("conv_pre", c0, cfg.num_mels, 7)
    nk = len(cfg.resblock_kernel_sizes)
    for i, (u, k) in enumerate(zip(cfg.upsample_rates, cfg.upsample_kernel_sizes)):
        cin, ch = c0 // (2 ** i), c0 // (2 ** (i + 1))
        conv(f"ups.{i}.0", ch, cin, k, transpose=True)
        for j in range(nk):
            pfx = f"resblocks.{i * nk + j}"
            nd = len(cfg.resblock_dilation_sizes[j])
            if cfg.resblock == "1":
                for m in range(nd):
                    conv(f"{pfx}.convs1.{m}", ch, ch, cfg.resblock_kernel_sizes[j], gain=0.7)
                    conv(f"{pfx}.convs2.{m}", ch, ch, cfg.resblock_kernel_sizes[j], gain=0.5)
                for q in range(2 * nd):
                    act(f"{pfx}.activations.{q}", ch)
            else:
                for m in range(nd):
                    conv(f"{pfx}.convs.{m}", ch, ch, cfg.resblock_kernel_sizes[j], gain=0.5)
                    act(f"{pfx}.activations.{m}", ch)
    chl = c0 // (2 ** len(cfg.upsample_rates))
    act("activation_post", chl)
    conv("conv_post", 1, chl, 7, bias=cfg.use_bias_at_final, gain=0.5)
    return sd


def stress_dit_state_dict(sd: Dict[str, torch.Tensor], cfg: DiTConfig, seed: int = 0) -> Dict[str, torch.Tensor]:
    """Dynamic-range stress (DiT backbones): what a trained checkpoint has and N(0, 1/sqrt(in)) weights do not — per-tensor scales spread over
    decades and a few outlier channels.  Every block gets three log-uniform factors s in [1e-3, 1]: to_v is scaled by s_v and to_out by 1/s_v
    (attention output at 1e-3 of its usual size, re-amplified by weights of up to 1000 times theirs), to_q by s_q and to_k by 1/s_q (same
    scores, operands decades apart), FF linear 1 by s_f and FF linear 2 by 1/s_f; 2 % of the AdaLN modulation outputs (scale / shift / gate
    channels) are amplified 4-12 times (heavy-tailed gains -> outlier activation channels).  Same keys, same shapes: loads into the reference
    unchanged.  Drawn from its own generator, so the unstressed tensors of (cfg, seed) are not disturbed."""
    assert cfg.backbone == "DiT"
    g = torch.Generator(device="cpu")
    g.manual_seed(5000011 * seed + 101)
    out = {k: v.clone() for k, v in sd.items()}
    p = "transformer."

    def logu():
        return float(10.0 ** (-3.0 * torch.rand((), generator=g)))

    for i in range(cfg.depth):
        b = p + f"transformer_blocks.{i}."
        sv, sq, sf = logu(), logu(), logu()
        for key, f in (("attn.to_v", sv), ("attn.to_q", sq), ("ff.ff.0.0", sf)):
            out[b + key + ".weight"] *= f
            out[b + key + ".bias"] *= f
        out[b + "attn.to_out.0.weight"] *= 1.0 / sv
        out[b + "attn.to_k.weight"] *= 1.0 / sq
        out[b + "attn.to_k.bias"] *= 1.0 / sq
        out[b + "ff.ff.2.weight"] *= 1.0 / sf
        w = out[b + "attn_norm.linear.weight"]
        pick = torch.rand(w.shape[0], generator=g) < 0.02
        gain = 4.0 + 8.0 * torch.rand(w.shape[0], generator=g)
        w[pick] *= gain[pick, None]
        out[b + "attn_norm.linear.bias"][pick] *= gain[pick]
    return out


def trained_like_dit_state_dict(sd: Dict[str, torch.Tensor], cfg: DiTConfig, seed: int = 0) -> Dict[str, torch.Tensor]:
    """Weight STATISTICS of a trained checkpoint on the seeded tensors (DiT / UNetT) — no checkpoint is reachable from the build container, and
    Gaussian N(0, 1 / sqrt(in)) matrices flatter every quantiser: (1) heavy-tailed entries — every matrix of the per-step path (block
    projections, AdaLN and time-MLP linears, input / output projections, text pointwise convs) is redrawn as Student-t (4 degrees of freedom:
    kurtosis far above a Gaussian's, a few entries 6-10 sigma out) at its tensor's own standard deviation; (2) per-output-channel gains exp(N(0,
    0.5^2)) and per-input-channel gains exp(N(0, 0.3^2)) (what learned norm gains and residual scaling leave behind in a checkpoint: rows and
    columns whose scales differ by factors of 3-5); (3) biases redrawn with the same tails; (4) norm gains far from 1: text ConvNeXt LayerNorm
    weights N(1, 0.3^2) clipped to [0.3, 2], their biases N(0, 0.2^2), GRN gamma N(0, 0.5^2), UNetT RMSNorm gains N(1, 0.3^2).  Same keys, same
    shapes: loads into the reference unchanged; drawn from its own generator."""
    g = torch.Generator(device="cpu")
    g.manual_seed(7000003 * seed + 313)
    out = {k: v.clone() for k, v in sd.items()}

    def student_t(shape):  # t_4 = N / sqrt(chi2_4 / 4), unit variance after / sqrt(2)
        z = torch.randn(shape, generator=g)
        chi = torch.randn((4,) + tuple(shape), generator=g).square().sum(0)
        return z / torch.sqrt(chi / 4.0) / math.sqrt(2.0)

    for k, v in sd.items():
        if not v.is_floating_point() or k.endswith("inv_freq") or "text_embed.text_embed.weight" in k:
            continue
        if k.endswith(("norm.weight", ".g")):  # LayerNorm weights of the text ConvNeXt blocks, RMSNorm gains of the UNetT
            out[k] = (1.0 + 0.3 * torch.randn(v.shape, generator=g)).clamp(0.3, 2.0)
        elif k.endswith("norm.bias"):
            out[k] = 0.2 * torch.randn(v.shape, generator=g)
        elif k.endswith("grn.gamma"):
            out[k] = 0.5 * torch.randn(v.shape, generator=g)
        elif v.ndim == 2 and k.endswith(".weight"):
            w = student_t(v.shape) * float(v.std())
            w = w * torch.exp(0.5 * torch.randn(v.shape[0], 1, generator=g)) * torch.exp(0.3 * torch.randn(1, v.shape[1], generator=g))
            out[k] = w * float(v.std() / w.std())  # the tensor keeps its overall scale: the model stays in its operating range
        elif v.ndim == 1 and k.endswith(".bias"):
            out[k] = student_t(v.shape) * float(v.std())
    return out


def sharpen_attention_state_dict(sd: Dict[str, torch.Tensor], s: float) -> Dict[str, torch.Tensor]:
    """Every query and key projection (``to_q`` / ``to_k`` and MMDiT's ``to_q_c`` / ``to_k_c``: weight AND bias) multiplied by ``s``, so every
    attention logit is multiplied by ``s * s`` — the sharpness sweep of DESIGN.md section 2: how the half-precision modes' error moves as a
    checkpoint's softmax rows approach one-hot.  Same keys, same shapes."""
    out = {k: v.clone() for k, v in sd.items()}
    for k in sd:
        if re.search(r"\.to_[qk](_c)?\.(weight|bias)$", k):
            out[k] = out[k] * float(s)
    return out


def synth_loud_wave(n_samples: int, seed: int = 0, batch: int = 1) -> torch.Tensor:
    """A prompt that hits the rails: ``0.6 * N(0,1)`` clipped to +-1 (about 10 % of the samples clip; rms ~= 0.55, so the RMS rule of
    ``utils_infer.py:463-465`` does not rescale it)."""
    g = torch.Generator(device="cpu")
    out = []
    for b in range(batch):
        g.manual_seed(424243 * (seed + b) + 7)
        out.append((0.6 * torch.randn(n_samples, generator=g)).clamp(-1.0, 1.0))
    return torch.stack(out, 0)


def synth_wave(n_samples: int, seed: int = 0, batch: int = 1) -> torch.Tensor:
    """``0.1 * N(0,1)`` clipped to +-1: rms ~= 0.1, so the RMS-normalise branch
    (reference ``src/f5_tts/infer/utils_infer.py:463-465``) is a no-op."""
    g = torch.Generator(device="cpu")
    out = []
    for b in range(batch):
        g.manual_seed(424243 * (seed + b) + 5)
        out.append((0.1 * torch.randn(n_samples, generator=g)).clamp(-1.0, 1.0))
    return torch.stack(out, 0)


def synth_text_ids(batch: int, nt: int, vocab: int, seed: int = 0) -> torch.Tensor:
    """int64 ``[batch, nt]`` ids uniform in ``[1, vocab-1]`` (0 is "space"/unknown, -1 is batch padding)."""
    g = torch.Generator(device="cpu")
    g.manual_seed(99991 * seed + 3)
    return torch.randint(1, vocab, (batch, nt), generator=g, dtype=torch.int64)
