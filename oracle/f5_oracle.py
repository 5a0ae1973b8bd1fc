"""TEST INFRASTRUCTURE — CPU oracle for the F5-TTS inference hot path.  NOT product code.

A self-contained restatement (plain torch fp32 on CPU, functional style over a flat
state-dict) of the reference's algorithm for the path named by BASELINE.json's north_star:
``CFM.sample`` + ``DiT.forward`` + Vocos mel front-end + Vocos decode.  Each function cites the
reference file:line it follows (paths relative to ``/root/reference/``).

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this module, and only as the CHECKER / the timed CPU baseline — the product path
(``f5-tts_amd``) never routes through it and fails loudly when the HIP library is missing.

Pinning status.  The reference has no tests and no golden vectors.  This
restatement is pinned against OUTPUTS OF THE REFERENCE ITSELF, run in the build container by
``oracle/make_golden.py`` (reference classes imported verbatim through ``oracle/ref_shims.py``)
and committed as fixtures under ``tests/golden/``; ``tests/test_oracle.py`` re-checks it against
the live reference whenever ``/root/reference`` exists.  The arithmetic that lives in
un-vendored third-party packages (torchdiffeq, x_transformers, torchaudio mel, the ``vocos``
package) is restated from the published algorithms — for those pieces the pin is the reference's
own call sites plus the in-repo cross-checks (rope: ``runtime/triton_trtllm/.../f5_tts_trtllm.py:232-237``;
STFT/iSTFT: ``runtime/triton_trtllm/scripts/conv_stft.py``, runnable, agreement checked by
``make_golden.py``); the Vocos *backbone* has no in-repo check: parity for it is "unpinned by the
reference" and anchored on the upstream definition cited in ``vocos_decode``.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor
SD = Dict[str, Tensor]


# ---------------------------------------------------------------------------------------------
# model/utils.py
# ---------------------------------------------------------------------------------------------
def lens_to_mask(t: Tensor, length: Optional[int] = None) -> Tensor:
    """src/f5_tts/model/utils.py:53-58"""
    if length is None:
        length = int(t.amax())
    seq = torch.arange(length)
    return seq[None, :] < t[:, None]


_EPSS = {  # src/f5_tts/model/utils.py:205-218
    5: [0, 2, 4, 8, 16, 32],
    6: [0, 2, 4, 6, 8, 16, 32],
    7: [0, 2, 4, 6, 8, 16, 24, 32],
    10: [0, 2, 4, 6, 8, 12, 16, 20, 24, 28, 32],
    12: [0, 2, 4, 6, 8, 10, 12, 14, 16, 20, 24, 28, 32],
    16: [0, 1, 2, 3, 4, 5, 6, 7, 8, 10, 12, 14, 16, 20, 24, 28, 32],
}


def time_grid(steps: int, sway_sampling_coef: Optional[float], use_epss: bool = True, t_start: float = 0.0) -> Tensor:
    """src/f5_tts/model/cfm.py:211-216 (+ utils.py:205-218)."""
    if t_start == 0 and use_epss and steps in _EPSS:
        t = (1 / 32) * torch.tensor(_EPSS[steps], dtype=torch.float32)
    else:
        t = torch.linspace(t_start, 1, steps + 1, dtype=torch.float32)
    if sway_sampling_coef is not None:
        t = t + sway_sampling_coef * (torch.cos(torch.pi / 2 * t) - 1 + t)
    return t


# ---------------------------------------------------------------------------------------------
# mel front-end  (model/modules.py:80-109 -> torchaudio.transforms.MelSpectrogram)
# ---------------------------------------------------------------------------------------------
def htk_fbanks(n_freqs=513, f_min=0.0, f_max=12000.0, n_mels=100, sample_rate=24000) -> Tensor:
    """torchaudio.functional.melscale_fbanks(norm=None, mel_scale='htk') -> [n_freqs, n_mels]."""
    all_freqs = torch.linspace(0, sample_rate // 2, n_freqs)
    m_min = 2595.0 * math.log10(1.0 + f_min / 700.0)
    m_max = 2595.0 * math.log10(1.0 + f_max / 700.0)
    m_pts = torch.linspace(m_min, m_max, n_mels + 2)
    f_pts = 700.0 * (10.0 ** (m_pts / 2595.0) - 1.0)
    f_diff = f_pts[1:] - f_pts[:-1]
    slopes = f_pts.unsqueeze(0) - all_freqs.unsqueeze(1)
    down = (-1.0 * slopes[:, :-2]) / f_diff[:-1]
    up = slopes[:, 2:] / f_diff[1:]
    return torch.max(torch.zeros(1), torch.min(down, up))


def vocos_mel(wav: Tensor, n_fft=1024, hop=256, win=1024, n_mels=100, sr=24000) -> Tensor:
    """src/f5_tts/model/modules.py:80-109: MelSpectrogram(power=1, center=True, norm=None) ->
    clamp(1e-5).log().  wav [b, nw] -> [b, n_mels, 1 + nw // hop]."""
    spec = torch.stft(wav, n_fft, hop_length=hop, win_length=win, window=torch.hann_window(win), center=True,
                      pad_mode="reflect", normalized=False, onesided=True, return_complex=True).abs()
    fb = htk_fbanks(n_fft // 2 + 1, 0.0, float(sr // 2), n_mels, sr)
    mel = torch.matmul(spec.transpose(-1, -2), fb).transpose(-1, -2)
    return mel.clamp(min=1e-5).log()


def slaney_mel_basis(sr=24000, n_fft=1024, n_mels=100, fmin=0.0, fmax=None) -> Tensor:
    """``librosa.filters.mel(sr, n_fft, n_mels, fmin, fmax)`` with its defaults (htk=False, norm="slaney", float32) -> [n_mels, 1+n_fft/2].
    librosa is a third-party dependency that is absent here and un-vendored in the reference (call site model/modules.py:50):
    restated from its published algorithm — Slaney's Auditory-Toolbox mel scale (linear below 1 kHz at 200/3 Hz per mel,
    logarithmic above with step ln(6.4)/27), triangular filters on the FFT bin centres, each scaled by 2 / (its bandwidth in Hz).
    PARITY UNPINNED for this table: nothing in the reference tree holds its values."""
    import numpy as np

    if fmax is None:
        fmax = sr / 2.0
    f_sp, min_log_hz = 200.0 / 3.0, 1000.0
    min_log_mel, logstep = min_log_hz / f_sp, math.log(6.4) / 27.0

    def hz_to_mel(f):
        return min_log_mel + math.log(f / min_log_hz) / logstep if f >= min_log_hz else f / f_sp

    mels = np.linspace(hz_to_mel(fmin), hz_to_mel(fmax), n_mels + 2)
    mel_f = np.where(mels >= min_log_mel, min_log_hz * np.exp(logstep * (mels - min_log_mel)), f_sp * mels)
    fftfreqs = np.arange(0, n_fft // 2 + 1) / (n_fft * (1.0 / sr))  # np.fft.rfftfreq(n_fft, 1 / sr)
    fdiff = np.diff(mel_f)
    ramps = np.subtract.outer(mel_f, fftfreqs)
    w = np.zeros((n_mels, n_fft // 2 + 1), dtype=np.float32)
    for i in range(n_mels):
        w[i] = np.maximum(0, np.minimum(-ramps[i] / fdiff[i], ramps[i + 2] / fdiff[i + 1]))
    w *= (2.0 / (mel_f[2: n_mels + 2] - mel_f[:n_mels]))[:, None]
    return torch.from_numpy(w)


def bigvgan_mel(wav: Tensor, n_fft=1024, hop=256, win=1024, n_mels=100, sr=24000) -> Tensor:
    """src/f5_tts/model/modules.py:35-77 (``mel_spec_type="bigvgan"``): reflect-pad (n_fft-hop)/2 on both sides, STFT without centring,
    sqrt(re^2 + im^2 + 1e-9), slaney mel basis, log(clamp 1e-5).  wav [b, nw] -> [b, n_mels, nw // hop]."""
    pad = (n_fft - hop) // 2
    x = F.pad(wav.unsqueeze(1), (pad, pad), mode="reflect").squeeze(1)
    spec = torch.stft(x, n_fft, hop_length=hop, win_length=win, window=torch.hann_window(win), center=False, pad_mode="reflect",
                      normalized=False, onesided=True, return_complex=True)
    spec = torch.sqrt(torch.view_as_real(spec).pow(2).sum(-1) + 1e-9)
    return torch.log(torch.clamp(torch.matmul(slaney_mel_basis(sr, n_fft, n_mels), spec), min=1e-5))


# ---------------------------------------------------------------------------------------------
# modules
# ---------------------------------------------------------------------------------------------
def precompute_freqs_cis(dim: int, end: int, theta: float = 10000.0) -> Tensor:
    """src/f5_tts/model/modules.py:207-218 -> [end, dim] = cat(cos, sin)."""
    freqs = 1.0 / (theta ** (torch.arange(0, dim, 2)[: (dim // 2)].float() / dim))
    t = torch.arange(end)
    freqs = torch.outer(t, freqs).float()
    return torch.cat([torch.cos(freqs), torch.sin(freqs)], dim=-1)


def sinus_time_embedding(t: Tensor, dim: int = 256, scale: float = 1000.0) -> Tensor:
    """src/f5_tts/model/modules.py:157-169 (divides by half_dim-1, cat(sin, cos))."""
    half = dim // 2
    emb = math.log(10000) / (half - 1)
    emb = torch.exp(torch.arange(half).float() * -emb)
    emb = scale * t.unsqueeze(1) * emb.unsqueeze(0)
    return torch.cat((emb.sin(), emb.cos()), dim=-1)


def timestep_embedding(sd: SD, t: Tensor) -> Tensor:
    """src/f5_tts/model/modules.py:852-862."""
    h = sinus_time_embedding(t)
    h = F.linear(h, sd["transformer.time_embed.time_mlp.0.weight"], sd["transformer.time_embed.time_mlp.0.bias"])
    h = F.silu(h)
    return F.linear(h, sd["transformer.time_embed.time_mlp.2.weight"], sd["transformer.time_embed.time_mlp.2.bias"])


def grn(x: Tensor, gamma: Tensor, beta: Tensor) -> Tensor:
    """src/f5_tts/model/modules.py:242-245 — L2 norm over the SEQUENCE axis (dim=1)."""
    gx = torch.norm(x, p=2, dim=1, keepdim=True)
    nx = gx / (gx.mean(dim=-1, keepdim=True) + 1e-6)
    return gamma * (x * nx) + beta + x


def convnext_v2_block(sd: SD, pfx: str, x: Tensor) -> Tensor:
    """src/f5_tts/model/modules.py:270-280."""
    res = x
    h = F.conv1d(x.transpose(1, 2), sd[pfx + "dwconv.weight"], sd[pfx + "dwconv.bias"], padding=3, groups=x.shape[-1])
    h = h.transpose(1, 2)
    h = F.layer_norm(h, (h.shape[-1],), sd[pfx + "norm.weight"], sd[pfx + "norm.bias"], eps=1e-6)
    h = F.linear(h, sd[pfx + "pwconv1.weight"], sd[pfx + "pwconv1.bias"])
    h = F.gelu(h)
    h = grn(h, sd[pfx + "grn.gamma"], sd[pfx + "grn.beta"])
    h = F.linear(h, sd[pfx + "pwconv2.weight"], sd[pfx + "pwconv2.bias"])
    return res + h


def average_upsample_text_by_mask(text: Tensor, text_mask: Tensor, target_lens: Tensor) -> Tensor:
    """src/f5_tts/model/backbones/dit.py:55-84 — spread the valid text tokens over the first ``target_lens[b]`` frames: token j is
    repeated ``base`` times, the last ``remainder`` tokens ``base + 1`` times; everything behind is zero."""
    out = torch.zeros_like(text)
    for b in range(text.shape[0]):
        valid = torch.nonzero(text_mask[b]).flatten()
        tl, al = int(valid.numel()), int(target_lens[b])
        if tl == 0 or al <= 0:
            continue
        base, rem = al // tl, al % tl
        reps = torch.tensor([base + (1 if j >= tl - rem else 0) for j in range(tl)], dtype=torch.long)
        idx = torch.repeat_interleave(torch.arange(tl), reps)[:al]
        out[b, :al] = text[b, valid[idx]]
    return out


def text_embedding(sd: SD, cfg, text: Tensor, seq_len, drop_text: bool) -> Tensor:
    """src/f5_tts/model/backbones/dit.py:86-139.
    ``seq_len``: int (no mask, batch==1) or int64 [b] (per-sample valid length)."""
    text = text + 1
    valid = None
    if torch.is_tensor(seq_len):
        max_len = int(seq_len.max())
    else:
        max_len = int(seq_len)
    text = text[:, :max_len]
    text = F.pad(text, (0, max_len - text.shape[1]), value=0)
    if torch.is_tensor(seq_len):
        valid = torch.arange(max_len).unsqueeze(0) < seq_len.unsqueeze(1)
        text = text.masked_fill(~valid, 0)
    text_mask = (text == 0) if cfg.text_mask_padding else None
    if drop_text:
        text = torch.zeros_like(text)
    h = F.embedding(text, sd["transformer.text_embed.text_embed.weight"])
    if valid is not None:
        h = h.masked_fill(~valid.unsqueeze(-1), 0.0)
    if cfg.conv_layers > 0:
        freqs = precompute_freqs_cis(cfg.text_dim, 8192)[:max_len, :]
        if valid is not None:
            freqs = freqs.unsqueeze(0) * valid.unsqueeze(-1).to(freqs.dtype)
        h = h + freqs
        if cfg.text_mask_padding:
            m = text_mask.unsqueeze(-1)
            h = h.masked_fill(m, 0.0)
            for i in range(cfg.conv_layers):
                h = convnext_v2_block(sd, f"transformer.text_embed.text_blocks.{i}.", h)
                h = h.masked_fill(m, 0.0)
        else:
            for i in range(cfg.conv_layers):
                h = convnext_v2_block(sd, f"transformer.text_embed.text_blocks.{i}.", h)
    if getattr(cfg, "text_embedding_average_upsampling", False):  # dit.py:131-137 (requires text_mask_padding, dit.py:42-43)
        tl = seq_len.long() if torch.is_tensor(seq_len) else torch.full((text.shape[0],), int(seq_len), dtype=torch.long)
        h = average_upsample_text_by_mask(h, ~text_mask, tl)
    return h


def mish(x: Tensor) -> Tensor:
    return x * torch.tanh(F.softplus(x))


def conv_position_embedding(sd: SD, cfg, x: Tensor, mask: Optional[Tensor], prefix: str = "transformer.input_embed.conv_pos_embed.") -> Tensor:
    """src/f5_tts/model/modules.py:187-201."""
    pfx = prefix + "conv1d."
    k, g = cfg.conv_pos_kernel, cfg.conv_pos_groups
    m = mask.unsqueeze(1) if mask is not None else None
    h = x.permute(0, 2, 1)
    if m is not None:
        h = h.masked_fill(~m, 0.0)
    for j in (0, 2):
        h = F.conv1d(h, sd[pfx + f"{j}.weight"], sd[pfx + f"{j}.bias"], padding=k // 2, groups=g)
        if m is not None:
            h = h.masked_fill(~m, 0.0)
        h = mish(h)
    return h.permute(0, 2, 1)


def input_embedding(sd: SD, cfg, x: Tensor, cond: Tensor, text_embed: Tensor, drop_audio_cond: bool,
                    mask: Optional[Tensor]) -> Tensor:
    """src/f5_tts/model/backbones/dit.py:151-164."""
    if drop_audio_cond:
        cond = torch.zeros_like(cond)
    h = F.linear(torch.cat((x, cond, text_embed), dim=-1), sd["transformer.input_embed.proj.weight"],
                 sd["transformer.input_embed.proj.bias"])
    return conv_position_embedding(sd, cfg, h, mask) + h


def rotary_freqs(dim_head: int, n: int) -> Tensor:
    """x_transformers RotaryEmbedding.forward_from_seq_len (call: dit.py:352): [1, n, dim_head],
    every frequency duplicated into adjacent lanes [f0,f0,f1,f1,...]."""
    inv_freq = 1.0 / (10000.0 ** (torch.arange(0, dim_head, 2).float() / dim_head))
    freqs = torch.einsum("i,j->ij", torch.arange(n).float(), inv_freq)
    return torch.stack((freqs, freqs), dim=-1).reshape(1, n, dim_head)


def apply_rope(t: Tensor, freqs: Tensor) -> Tensor:
    """x_transformers apply_rotary_pos_emb (calls: modules.py:503-509): interleaved pairs
    (x0,x1)->(x0 cos - x1 sin, x1 cos + x0 sin).  t [b,h,n,d], freqs [1,n,d]."""
    f = freqs[:, None]
    tp = t.reshape(*t.shape[:-1], t.shape[-1] // 2, 2)
    x1, x2 = tp.unbind(dim=-1)
    rot = torch.stack((-x2, x1), dim=-1).reshape(t.shape)
    return t * f.cos() + rot * f.sin()


def attention(sd: SD, cfg, pfx: str, x: Tensor, mask: Optional[Tensor], freqs: Tensor) -> Tensor:
    """src/f5_tts/model/modules.py:471-556 (torch backend; qk_norm None or "rms_norm")."""
    b, n, _ = x.shape
    hds, dh = cfg.heads, cfg.dim_head
    q = F.linear(x, sd[pfx + "to_q.weight"], sd[pfx + "to_q.bias"]).view(b, n, hds, dh).transpose(1, 2)
    k = F.linear(x, sd[pfx + "to_k.weight"], sd[pfx + "to_k.bias"]).view(b, n, hds, dh).transpose(1, 2)
    v = F.linear(x, sd[pfx + "to_v.weight"], sd[pfx + "to_v.bias"]).view(b, n, hds, dh).transpose(1, 2)
    if getattr(cfg, "qk_norm", None) == "rms_norm":  # modules.py:402-409,493-496; RMSNorm :286-305 with eps 1e-6 over dim_head
        q = q * torch.rsqrt(q.pow(2).mean(-1, keepdim=True) + 1e-6) * sd[pfx + "q_norm.weight"]
        k = k * torch.rsqrt(k.pow(2).mean(-1, keepdim=True) + 1e-6) * sd[pfx + "k_norm.weight"]
    if cfg.pe_attn_head is not None:
        pn = cfg.pe_attn_head
        q = torch.cat((apply_rope(q[:, :pn], freqs), q[:, pn:]), dim=1)
        k = torch.cat((apply_rope(k[:, :pn], freqs), k[:, pn:]), dim=1)
    else:
        q = apply_rope(q, freqs)
        k = apply_rope(k, freqs)
    attn_mask = None
    if cfg.attn_mask_enabled and mask is not None:
        attn_mask = mask[:, None, None, :].expand(b, hds, n, n)
    o = F.scaled_dot_product_attention(q, k, v, attn_mask=attn_mask, dropout_p=0.0, is_causal=False)
    o = o.transpose(1, 2).reshape(b, n, hds * dh)
    o = F.linear(o, sd[pfx + "to_out.0.weight"], sd[pfx + "to_out.0.bias"])
    if mask is not None:
        o = o.masked_fill(~mask.unsqueeze(-1), 0.0)
    return o


def dit_block(sd: SD, cfg, i: int, x: Tensor, t: Tensor, mask: Optional[Tensor], freqs: Tensor) -> Tensor:
    """src/f5_tts/model/modules.py:743-757 (+ AdaLayerNorm :321-326, FeedForward :353-364)."""
    pfx = f"transformer.transformer_blocks.{i}."
    d = x.shape[-1]
    emb = F.linear(F.silu(t), sd[pfx + "attn_norm.linear.weight"], sd[pfx + "attn_norm.linear.bias"])
    shift_msa, scale_msa, gate_msa, shift_mlp, scale_mlp, gate_mlp = torch.chunk(emb, 6, dim=1)
    norm = F.layer_norm(x, (d,), eps=1e-6) * (1 + scale_msa[:, None]) + shift_msa[:, None]
    x = x + gate_msa.unsqueeze(1) * attention(sd, cfg, pfx + "attn.", norm, mask, freqs)
    norm = F.layer_norm(x, (d,), eps=1e-6) * (1 + scale_mlp[:, None]) + shift_mlp[:, None]
    h = F.linear(norm, sd[pfx + "ff.ff.0.0.weight"], sd[pfx + "ff.ff.0.0.bias"])
    h = F.gelu(h, approximate="tanh")
    h = F.linear(h, sd[pfx + "ff.ff.2.weight"], sd[pfx + "ff.ff.2.bias"])
    return x + gate_mlp.unsqueeze(1) * h


def dit_forward_cfg(sd: SD, cfg, x: Tensor, cond: Tensor, text_cond: Tensor, text_uncond: Tensor, time: Tensor,
                    mask: Optional[Tensor], return_hidden: bool = False):
    """src/f5_tts/model/backbones/dit.py:319-370 with cfg_infer=True, cache=True -> [2b, n, mel]."""
    b, n = x.shape[0], x.shape[1]
    if time.ndim == 0:
        time = time.repeat(b)
    t = timestep_embedding(sd, time)
    x_c = input_embedding(sd, cfg, x, cond, text_cond, False, mask)
    x_u = input_embedding(sd, cfg, x, cond, text_uncond, True, mask)
    h = torch.cat((x_c, x_u), dim=0)
    t = torch.cat((t, t), dim=0)
    m2 = torch.cat((mask, mask), dim=0) if mask is not None else None
    freqs = rotary_freqs(cfg.dim_head, n)
    hidden = [h]
    residual = h  # dit.py:354-355
    for i in range(cfg.depth):
        h = dit_block(sd, cfg, i, h, t, m2, freqs)
        if return_hidden:
            hidden.append(h)
    if getattr(cfg, "long_skip_connection", False):  # dit.py:228,364-365
        h = F.linear(torch.cat((h, residual), dim=-1), sd["transformer.long_skip_connection.weight"])
    emb = F.linear(F.silu(t), sd["transformer.norm_out.linear.weight"], sd["transformer.norm_out.linear.bias"])
    scale, shift = torch.chunk(emb, 2, dim=1)  # NOTE (scale, shift) order: modules.py:344
    h = F.layer_norm(h, (h.shape[-1],), eps=1e-6) * (1 + scale)[:, None, :] + shift[:, None, :]
    out = F.linear(h, sd["transformer.proj_out.weight"], sd["transformer.proj_out.bias"])
    return (out, hidden) if return_hidden else out


# ---------------------------------------------------------------------------------------------
# UNetT backbone (E2-TTS): src/f5_tts/model/backbones/unett.py
# ---------------------------------------------------------------------------------------------
def x_rmsnorm(x: Tensor, g: Tensor) -> Tensor:
    """x_transformers RMSNorm (import unett.py:19; upstream: F.normalize(x, dim=-1) * sqrt(dim) * g)."""
    return F.normalize(x, dim=-1) * (x.shape[-1] ** 0.5) * g


def unett_text_embedding(sd: SD, cfg, text: Tensor, seq_len: int, drop_text: bool) -> Tensor:
    """unett.py:54-84 with conv_layers == 0 (E2TTS_Base.yaml): +1, curtail/pad to seq_len, embed.  mask_padding only acts inside
    the extra-modeling branch, so it is inert here."""
    assert cfg.conv_layers == 0, "UNetT text ConvNeXt blocks are not built (no shipped config uses them)"
    text = text + 1
    text = text[:, :seq_len]
    text = F.pad(text, (0, seq_len - text.shape[1]), value=0)
    if drop_text:
        text = torch.zeros_like(text)
    return F.embedding(text, sd["transformer.text_embed.text_embed.weight"])


def unett_forward_cfg(sd: SD, cfg, x: Tensor, cond: Tensor, text_cond: Tensor, text_uncond: Tensor, time: Tensor,
                      mask: Optional[Tensor]):
    """unett.py:244-307 with cfg_infer=True, cache=True -> [2b, n, mel]."""
    b, n = x.shape[0], x.shape[1]
    if time.ndim == 0:
        time = time.repeat(b)
    t = timestep_embedding(sd, time)
    # InputEmbedding.forward (unett.py:96-102): conv_pos_embed is called WITHOUT the mask
    x_c = input_embedding(sd, cfg, x, cond, text_cond, False, None)
    x_u = input_embedding(sd, cfg, x, cond, text_uncond, True, None)
    h = torch.cat((x_c, x_u), dim=0)
    t = torch.cat((t, t), dim=0)
    m2 = torch.cat((mask, mask), dim=0) if mask is not None else None
    h = torch.cat([t.unsqueeze(1), h], dim=1)  # time token first (:272)
    if m2 is not None:
        m2 = F.pad(m2, (1, 0), value=True)
    freqs = rotary_freqs(cfg.dim_head, n + 1)
    skips = []
    for i in range(cfg.depth):
        pfx = f"transformer.layers.{i}."
        sct = getattr(cfg, "skip_connect_type", "concat")  # unett.py:127,289-295
        if i < cfg.depth // 2:
            skips.append(h)
        else:
            skip = skips.pop()
            if sct == "concat":
                h = F.linear(torch.cat((h, skip), dim=-1), sd[pfx + "0.weight"])
            elif sct == "add":
                h = h + skip
        h = attention(sd, cfg, pfx + "2.", x_rmsnorm(h, sd[pfx + "1.g"]), m2, freqs) + h
        f = F.linear(x_rmsnorm(h, sd[pfx + "3.g"]), sd[pfx + "4.ff.0.0.weight"], sd[pfx + "4.ff.0.0.bias"])
        f = F.linear(F.gelu(f, approximate="tanh"), sd[pfx + "4.ff.2.weight"], sd[pfx + "4.ff.2.bias"])
        h = f + h
    h = x_rmsnorm(h, sd["transformer.norm_out.g"])[:, 1:, :]
    return F.linear(h, sd["transformer.proj_out.weight"], sd["transformer.proj_out.bias"])


# ---------------------------------------------------------------------------------------------
# MMDiT backbone (SD3-style two-stream blocks with joint attention): src/f5_tts/model/backbones/mmdit.py
# ---------------------------------------------------------------------------------------------
def mmdit_text_embedding(sd: SD, cfg, text: Tensor, drop_text: bool) -> Tensor:
    """mmdit.py:43-66: +1, embed (dim = model dim), absolute sinusoid positions 0..nt-1 (clipped to 1023), padding rows zeroed.
    The text keeps its own length nt — it is a second token stream, not upsampled to the frame count."""
    text = text + 1
    text_mask = text == 0
    if drop_text:
        text = torch.zeros_like(text)
    h = F.embedding(text, sd["transformer.text_embed.text_embed.weight"])
    pos = torch.arange(text.shape[1]).clamp(max=1023)  # get_pos_embed_indices(start 0, max_pos 1024), modules.py:221-230
    h = h + precompute_freqs_cis(cfg.dim, 1024)[pos]
    if cfg.text_mask_padding:
        h = h.masked_fill(text_mask.unsqueeze(-1), 0.0)
    return h


def _ada6(sd: SD, pfx: str, h: Tensor, t: Tensor):
    """AdaLayerNorm (modules.py:312-326): modulated LayerNorm + the four values the block uses later."""
    emb = F.linear(F.silu(t), sd[pfx + "linear.weight"], sd[pfx + "linear.bias"])
    shift_msa, scale_msa, gate_msa, shift_mlp, scale_mlp, gate_mlp = torch.chunk(emb, 6, dim=1)
    return (F.layer_norm(h, (h.shape[-1],), eps=1e-6) * (1 + scale_msa[:, None]) + shift_msa[:, None], gate_msa, shift_mlp, scale_mlp,
            gate_mlp)


def _ff(sd: SD, pfx: str, h: Tensor) -> Tensor:
    """FeedForward (modules.py:353-364), tanh-GELU."""
    h = F.gelu(F.linear(h, sd[pfx + "ff.0.0.weight"], sd[pfx + "ff.0.0.bias"]), approximate="tanh")
    return F.linear(h, sd[pfx + "ff.2.weight"], sd[pfx + "ff.2.bias"])


def mmdit_joint_attention(sd: SD, cfg, pfx: str, x: Tensor, c: Tensor, mask: Optional[Tensor], c_mask: Tensor, freqs_x: Tensor,
                          freqs_c: Tensor, last: bool):
    """JointAttnProcessor.__call__ (modules.py:581-705, torch backend): both streams are projected with their own weights, rotated with
    their own positions (each from 0), concatenated along the sequence ([audio | text]) for ONE softmax attention, then split again."""
    b, n, _ = x.shape
    nt = c.shape[1]
    hds, dh = cfg.heads, cfg.dim_head

    def proj(h, name, length):
        return F.linear(h, sd[pfx + name + ".weight"], sd[pfx + name + ".bias"]).view(b, length, hds, dh).transpose(1, 2)

    q, k, v = proj(x, "to_q", n), proj(x, "to_k", n), proj(x, "to_v", n)
    cq, ck, cv = proj(c, "to_q_c", nt), proj(c, "to_k_c", nt), proj(c, "to_v_c", nt)
    if getattr(cfg, "qk_norm", None) == "rms_norm":  # modules.py:616-624
        def rms(h, name):
            return h * torch.rsqrt(h.pow(2).mean(-1, keepdim=True) + 1e-6) * sd[pfx + name + ".weight"]
        q, k, cq, ck = rms(q, "q_norm"), rms(k, "k_norm"), rms(cq, "c_q_norm"), rms(ck, "c_k_norm")
    q, k = apply_rope(q, freqs_x), apply_rope(k, freqs_x)
    cq, ck = apply_rope(cq, freqs_c), apply_rope(ck, freqs_c)
    q, k, v = torch.cat((q, cq), dim=2), torch.cat((k, ck), dim=2), torch.cat((v, cv), dim=2)
    attn_mask = None
    if cfg.attn_mask_enabled and mask is not None:  # modules.py:643-657: audio key-padding mask + text mask
        attn_mask = torch.cat((mask, c_mask), dim=1)[:, None, None, :].expand(b, hds, n + nt, n + nt)
    o = F.scaled_dot_product_attention(q, k, v, attn_mask=attn_mask, dropout_p=0.0, is_causal=False)
    o = o.transpose(1, 2).reshape(b, n + nt, hds * dh)
    ox, oc = o[:, :n], o[:, n:]
    ox = F.linear(ox, sd[pfx + "to_out.0.weight"], sd[pfx + "to_out.0.bias"])
    if not last:
        oc = F.linear(oc, sd[pfx + "to_out_c.weight"], sd[pfx + "to_out_c.bias"])
    if mask is not None:
        ox = ox.masked_fill(~mask.unsqueeze(-1), 0.0)
    oc = oc.masked_fill(~c_mask.unsqueeze(-1), 0.0)
    return ox, oc


def mmdit_forward_cfg(sd: SD, cfg, x: Tensor, cond: Tensor, text_cond: Tensor, text_uncond: Tensor, time: Tensor,
                      mask: Optional[Tensor], c_mask: Tensor):
    """mmdit.py:213-262 with cfg_infer=True, cache=True -> [2b, n, mel].  ``c_mask`` = (text + 1) != 0 (mmdit.py:232)."""
    b, n = x.shape[0], x.shape[1]
    if time.ndim == 0:
        time = time.repeat(b)
    t = timestep_embedding(sd, time)

    def audio_embed(drop_audio_cond):  # AudioEmbedding.forward (mmdit.py:79-85); conv_pos_embed is called WITHOUT a mask
        cd = torch.zeros_like(cond) if drop_audio_cond else cond
        h = F.linear(torch.cat((x, cd), dim=-1), sd["transformer.audio_embed.linear.weight"], sd["transformer.audio_embed.linear.bias"])
        return conv_position_embedding(sd, cfg, h, None, prefix="transformer.audio_embed.conv_pos_embed.") + h

    h = torch.cat((audio_embed(False), audio_embed(True)), dim=0)
    c = torch.cat((text_cond, text_uncond), dim=0)
    t = torch.cat((t, t), dim=0)
    m2 = torch.cat((mask, mask), dim=0) if mask is not None else None
    cm2 = torch.cat((c_mask, c_mask), dim=0)
    freqs_x, freqs_c = rotary_freqs(cfg.dim_head, n), rotary_freqs(cfg.dim_head, c.shape[1])
    for i in range(cfg.depth):  # MMDiTBlock.forward (modules.py:816-845)
        pfx = f"transformer.transformer_blocks.{i}."
        last = i == cfg.depth - 1  # context_pre_only (mmdit.py:118)
        if last:
            emb = F.linear(F.silu(t), sd[pfx + "attn_norm_c.linear.weight"], sd[pfx + "attn_norm_c.linear.bias"])
            sc, sh = torch.chunk(emb, 2, dim=1)  # AdaLayerNorm_Final: (scale, shift)
            norm_c = F.layer_norm(c, (c.shape[-1],), eps=1e-6) * (1 + sc)[:, None, :] + sh[:, None, :]
        else:
            norm_c, c_gate_msa, c_shift_mlp, c_scale_mlp, c_gate_mlp = _ada6(sd, pfx + "attn_norm_c.", c, t)
        norm_x, x_gate_msa, x_shift_mlp, x_scale_mlp, x_gate_mlp = _ada6(sd, pfx + "attn_norm_x.", h, t)
        ox, oc = mmdit_joint_attention(sd, cfg, pfx + "attn.", norm_x, norm_c, m2, cm2, freqs_x, freqs_c, last)
        if not last:
            c = c + c_gate_msa.unsqueeze(1) * oc
            nc = F.layer_norm(c, (c.shape[-1],), eps=1e-6) * (1 + c_scale_mlp[:, None]) + c_shift_mlp[:, None]
            c = c + c_gate_mlp.unsqueeze(1) * _ff(sd, pfx + "ff_c.", nc)
        h = h + x_gate_msa.unsqueeze(1) * ox
        nx = F.layer_norm(h, (h.shape[-1],), eps=1e-6) * (1 + x_scale_mlp[:, None]) + x_shift_mlp[:, None]
        h = h + x_gate_mlp.unsqueeze(1) * _ff(sd, pfx + "ff_x.", nx)
    emb = F.linear(F.silu(t), sd["transformer.norm_out.linear.weight"], sd["transformer.norm_out.linear.bias"])
    scale, shift = torch.chunk(emb, 2, dim=1)
    h = F.layer_norm(h, (h.shape[-1],), eps=1e-6) * (1 + scale)[:, None, :] + shift[:, None, :]
    return F.linear(h, sd["transformer.proj_out.weight"], sd["transformer.proj_out.bias"])


# ---------------------------------------------------------------------------------------------
# sampler
# ---------------------------------------------------------------------------------------------
def make_noise(duration: Tensor, mel_dim: int, seed: Optional[int]) -> Tensor:
    """src/f5_tts/model/cfm.py:196-201 — per-sample reseed of torch's CPU generator, pad with 0."""
    y0 = []
    for dur in duration:
        if seed is not None:
            torch.manual_seed(seed)
        y0.append(torch.randn(int(dur), mel_dim, dtype=torch.float32))
    return torch.nn.utils.rnn.pad_sequence(y0, padding_value=0, batch_first=True)


@torch.no_grad()
def cfm_sample(sd: SD, cfg, cond: Tensor, text: Tensor, duration, *, lens: Optional[Tensor] = None, steps: int = 32,
               cfg_strength: float = 1.0, sway_sampling_coef: Optional[float] = None, seed: Optional[int] = None,
               max_duration: int = 65536, use_epss: bool = True, edit_mask: Optional[Tensor] = None,
               return_steps: bool = False, method: str = "euler", no_ref_audio: bool = False, duplicate_test: bool = False,
               t_inter: float = 0.1, mel_spec_type: str = "vocos"):
    """src/f5_tts/model/cfm.py:83-229 with a DiT or UNetT backbone (cfg.backbone), with or without CFG, fixed-grid euler / midpoint.
    cond: wave [b, nw] or mel [b, n, mel]; text: int64 [b, nt] (already tokenised, -1 padded)."""
    if cond.ndim == 2:
        cond = (bigvgan_mel(cond) if mel_spec_type == "bigvgan" else vocos_mel(cond)).permute(0, 2, 1)
        assert cond.shape[-1] == cfg.mel_dim
    cond = cond.float()
    batch, cond_seq_len = cond.shape[:2]
    if lens is None:
        lens = torch.full((batch,), cond_seq_len, dtype=torch.long)
    cond_mask = lens_to_mask(lens)
    if edit_mask is not None:
        cond_mask = cond_mask & edit_mask
    if isinstance(duration, int):
        duration = torch.full((batch,), duration, dtype=torch.long)
    duration = torch.maximum(torch.maximum((text != -1).sum(dim=-1), lens) + 1, duration)
    duration = duration.clamp(max=max_duration)
    n = int(duration.amax())
    if duplicate_test:  # cfm.py:141-143
        test_cond = F.pad(cond, (0, 0, cond_seq_len, n - 2 * cond_seq_len), value=0.0)
    cond = F.pad(cond, (0, 0, 0, n - cond_seq_len), value=0.0)
    if no_ref_audio:  # cfm.py:146-147
        cond = torch.zeros_like(cond)
    cond_mask = F.pad(cond_mask, (0, n - cond_mask.shape[-1]), value=False).unsqueeze(-1)
    step_cond = torch.where(cond_mask, cond, torch.zeros_like(cond))
    mask = lens_to_mask(duration) if batch > 1 else None

    unett = getattr(cfg, "backbone", "DiT") == "UNetT"
    mmdit = getattr(cfg, "backbone", "DiT") == "MMDiT"
    if mmdit:  # mmdit.py:190-206: the text stream keeps its own length
        text_cond = mmdit_text_embedding(sd, cfg, text, drop_text=False)
        text_uncond = mmdit_text_embedding(sd, cfg, text, drop_text=True)
        c_mask = (text + 1) != 0  # mmdit.py:232
    elif unett:  # unett.py:218-228: seq_len is the padded frame count for every sample
        text_cond = unett_text_embedding(sd, cfg, text, n, drop_text=False)
        text_uncond = unett_text_embedding(sd, cfg, text, n, drop_text=True)
    else:
        seq_len = n if mask is None else mask.sum(dim=1)  # dit.py:295-298
        text_cond = text_embedding(sd, cfg, text, seq_len, drop_text=False)
        text_uncond = text_embedding(sd, cfg, text, seq_len, drop_text=True)
    forward_cfg = unett_forward_cfg if unett else dit_forward_cfg
    if mmdit:
        def forward_cfg(sd_, cfg_, x_, sc_, tc_, tu_, ti_, mask_):
            return mmdit_forward_cfg(sd_, cfg_, x_, sc_, tc_, tu_, ti_, mask_, c_mask)

    y = make_noise(duration, cfg.mel_dim, seed)
    t_start = 0.0
    if duplicate_test:  # cfm.py:205-209: start from a noised copy of the prompt placed behind it, at t = t_inter
        t_start = t_inter
        y = (1 - t_start) * y + t_start * test_cond
        steps = int(steps * (1 - t_start))
    t = time_grid(steps, sway_sampling_coef, use_epss, t_start)
    traj = [y]
    vel = []
    def fn(ti, x):  # cfm.py:162-191
        pred_cfg = forward_cfg(sd, cfg, x, step_cond, text_cond, text_uncond, ti, mask)
        pred, null = torch.chunk(pred_cfg, 2, dim=0)
        if cfg_strength < 1e-5:  # single conditional branch (cfm.py:166-177); rows are independent, so the packed cond half is it
            return pred
        return pred + (pred - null) * cfg_strength  # cfm.py:190-191

    for i in range(steps):  # torchdiffeq fixed-grid solvers on the supplied grid (cfm.py:218)
        dt = t[i + 1] - t[i]
        if method == "euler":
            v = fn(t[i], y)
        elif method == "midpoint":  # y_mid = y + f(t, y) dt/2;  y += dt f(t + dt/2, y_mid)
            half = 0.5 * dt
            v = fn(t[i] + half, y + fn(t[i], y) * half)
        else:
            raise ValueError(method)
        y = y + dt * v
        traj.append(y)
        if return_steps:
            vel.append(v)
    trajectory = torch.stack(traj, 0)
    out = torch.where(cond_mask, cond, trajectory[-1])  # cfm.py:221-223
    if return_steps:
        return out, trajectory, dict(text_cond=text_cond, text_uncond=text_uncond, velocity=torch.stack(vel, 0), t=t,
                                     step_cond=step_cond, y0=traj[0])
    return out, trajectory


# ---------------------------------------------------------------------------------------------
# Vocos (third-party ``vocos`` package, absent here — restated from upstream
# vocos/models.py::VocosBackbone, vocos/modules.py::ConvNeXtBlock, vocos/heads.py::ISTFTHead,
# vocos/spectral_ops.py::ISTFT(padding="center"); head math cross-checked in-repo by
# src/f5_tts/runtime/triton_trtllm/scripts/export_vocoder_to_onnx.py:45-59)
# ---------------------------------------------------------------------------------------------
def vocos_backbone(sd: SD, mel: Tensor, num_layers: int) -> Tensor:
    """mel [b, 100, T] -> [b, T, dim]."""
    x = F.conv1d(mel, sd["backbone.embed.weight"], sd["backbone.embed.bias"], padding=3)
    c = x.shape[1]
    x = F.layer_norm(x.transpose(1, 2), (c,), sd["backbone.norm.weight"], sd["backbone.norm.bias"], eps=1e-6).transpose(1, 2)
    for i in range(num_layers):
        p = f"backbone.convnext.{i}."
        res = x
        h = F.conv1d(x, sd[p + "dwconv.weight"], sd[p + "dwconv.bias"], padding=3, groups=c).transpose(1, 2)
        h = F.layer_norm(h, (c,), sd[p + "norm.weight"], sd[p + "norm.bias"], eps=1e-6)
        h = F.linear(h, sd[p + "pwconv1.weight"], sd[p + "pwconv1.bias"])
        h = F.gelu(h)
        h = F.linear(h, sd[p + "pwconv2.weight"], sd[p + "pwconv2.bias"])
        h = sd[p + "gamma"] * h
        x = res + h.transpose(1, 2)
    return F.layer_norm(x.transpose(1, 2), (c,), sd["backbone.final_layer_norm.weight"],
                        sd["backbone.final_layer_norm.bias"], eps=1e-6)


def vocos_head_spectrum(sd: SD, x: Tensor) -> Tensor:
    """ISTFTHead up to the complex spectrum: [b, T, dim] -> complex [b, n_fft/2+1, T]."""
    x = F.linear(x, sd["head.out.weight"], sd["head.out.bias"]).transpose(1, 2)
    mag, p = x.chunk(2, dim=1)
    mag = torch.exp(mag)
    mag = torch.clip(mag, max=1e2)
    return mag * (torch.cos(p) + 1j * torch.sin(p))


def vocos_decode(sd: SD, mel: Tensor, num_layers: int = 8, n_fft: int = 1024, hop: int = 256) -> Tensor:
    """``Vocos.decode`` (call site src/f5_tts/infer/utils_infer.py:510-511): mel [b,100,T] -> wav [b, hop*(T-1)]."""
    x = vocos_backbone(sd, mel, num_layers)
    spec = vocos_head_spectrum(sd, x)
    return torch.istft(spec, n_fft, hop_length=hop, win_length=n_fft, window=sd["head.istft.window"], center=True)


def vocos_head(sd: SD, hidden: Tensor, n_fft: int = 1024, hop: int = 256) -> Tensor:
    """``ISTFTHead.forward`` alone (the reference's runnable copy: runtime/triton_trtllm/scripts/export_vocoder_to_onnx.py:43-59):
    hidden [b, T, dim] -> wav [b, hop*(T-1)].  Pinned by tests/golden/vocos_head_ref*.npz, which the reference class itself produced."""
    return torch.istft(vocos_head_spectrum(sd, hidden), n_fft, hop_length=hop, win_length=n_fft, window=sd["head.istft.window"], center=True)


def istft_manual(spec: Tensor, n_fft: int = 1024, hop: int = 256) -> Tensor:
    """torch.istft(center=True) spelled out: irfft -> x window -> overlap-add -> / sum(window^2) ->
    trim n_fft/2 each side.  Used to pin the semantics the HIP kernel implements."""
    win = torch.hann_window(n_fft)
    frames = torch.fft.irfft(spec, n=n_fft, dim=1) * win[None, :, None]  # [b, n_fft, T]
    b, _, T = frames.shape
    total = n_fft + hop * (T - 1)
    y = F.fold(frames, output_size=(1, total), kernel_size=(1, n_fft), stride=(1, hop))[:, 0, 0, :]
    env = F.fold((win * win)[None, :, None].expand(1, n_fft, T), output_size=(1, total), kernel_size=(1, n_fft),
                 stride=(1, hop))[:, 0, 0, :]
    s = n_fft // 2
    return y[:, s:total - s] / env[:, s:total - s]


# ---------------------------------------------------------------------------------------------
# L4 call-site glue around the path (src/f5_tts/infer/utils_infer.py:477-520), fixed-duration form
# ---------------------------------------------------------------------------------------------
def infer_basic(sd_dit: SD, cfg, sd_vocos: SD, vocos_layers: int, audio: Tensor, text: Tensor, duration: int, *,
                steps: int, cfg_strength: float, sway_sampling_coef: Optional[float], seed: Optional[int]):
    ref_audio_len = audio.shape[-1] // 256  # :486  (one less than the mel frame count)
    out, _ = cfm_sample(sd_dit, cfg, audio, text, duration, steps=steps, cfg_strength=cfg_strength,
                        sway_sampling_coef=sway_sampling_coef, seed=seed)
    gen = out[:, ref_audio_len:, :].permute(0, 2, 1)  # :507-509
    wav = vocos_decode(sd_vocos, gen, vocos_layers)  # :510-511
    return wav, out
