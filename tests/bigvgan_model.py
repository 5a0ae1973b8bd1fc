"""Host model of the formulations libf5hip's BigVGAN path uses (f5-tts_amd/csrc/bigvgan.{hip,cpp}) — TEST INFRASTRUCTURE.

The kernels do not evaluate the generator the way the upstream module does; they use channels-last tensors [b, L, C] and
  * every Conv1d / ConvTranspose1d as ONE GEMM over a tap-gathered ("im2col") operand whose channel block is padded to a multiple of
    32 (`im2col`, `conv_weight_matrix`, `convt_weight_matrix`: a stride-u transposed conv is a 3-tap conv with u*Cout output columns,
    column block r holding output phase r);
  * Activation1d (x2 kaiser-sinc upsample -> snake -> x2 low-pass downsample, replicate padding at both levels) as closed-form index
    arithmetic on the un-padded signal (`aa_snake`).
This file restates exactly that index arithmetic with torch gathers so that it can be checked against the oracle's F.conv1d /
F.conv_transpose1d formulation on the CPU (tests/test_bigvgan_oracle.py); the HIP code is a transcription of these functions.
"""
from __future__ import annotations

import torch


def rup(x: int, m: int) -> int:
    return (x + m - 1) // m * m


def aa_snake(x: torch.Tensor, alpha: torch.Tensor, beta: torch.Tensor, f: torch.Tensor, logscale: bool) -> torch.Tensor:
    """x [b, L, C] -> [b, L, C].  f: the 12-tap filter.  u = upsampled signal (length 2L), v = snake(u), z = low-pass + decimate.
         u[2q]   = 2 * sum_t x[clamp(q - 3 + t)] * f[11 - 2t]      t = 0..5
         u[2q+1] = 2 * sum_t x[clamp(q - 2 + t)] * f[10 - 2t]
         z[l]    = sum_j f[j] * v[clamp(2l + j - 5, 0, 2L - 1)]     j = 0..11"""
    b, L, C = x.shape
    a = torch.exp(alpha) if logscale else alpha
    bb = torch.exp(beta) if logscale else beta
    invb = 1.0 / (bb + 1e-9)
    q = torch.arange(L)
    ue = torch.zeros(b, L, C)
    uo = torch.zeros(b, L, C)
    for t in range(6):
        ue = ue + x[:, (q - 3 + t).clamp(0, L - 1), :] * f[11 - 2 * t]
        uo = uo + x[:, (q - 2 + t).clamp(0, L - 1), :] * f[10 - 2 * t]
    u = torch.stack((2.0 * ue, 2.0 * uo), dim=2).reshape(b, 2 * L, C)
    v = u + invb * torch.sin(u * a).pow(2)
    l = torch.arange(L)
    z = torch.zeros(b, L, C)
    for j in range(12):
        z = z + f[j] * v[:, (2 * l + j - 5).clamp(0, 2 * L - 1), :]
    return z


def im2col(y: torch.Tensor, ntaps: int, shift0: int, dstep: int, cpad: int) -> torch.Tensor:
    """y [b, L, C] -> [b, L, ntaps * cpad]: column j * cpad + c of row l = y[l + shift0 + j * dstep, c] (0 outside [0, L) / c >= C)."""
    b, L, C = y.shape
    out = torch.zeros(b, L, ntaps * cpad)
    l = torch.arange(L)
    for j in range(ntaps):
        src = l + shift0 + j * dstep
        ok = (src >= 0) & (src < L)
        out[:, ok, j * cpad:j * cpad + C] = y[:, src[ok], :]
    return out


def conv_weight_matrix(w: torch.Tensor, cpad: int) -> torch.Tensor:
    """Conv1d weight [Cout, Cin, k] -> GEMM weight [Cout, k * cpad], column j * cpad + ci."""
    cout, cin, k = w.shape
    m = torch.zeros(cout, k, cpad)
    m[:, :, :cin] = w.permute(0, 2, 1)
    return m.reshape(cout, k * cpad)


def convt_taps(k: int, u: int):
    """Input shifts s (output row l*u + r reads input rows l + s) a ConvTranspose1d(k, stride u, padding (k-u)//2) needs."""
    pad = (k - u) // 2
    ss = [s for s in range(-k, k + 1) if any(0 <= r + pad - s * u < k for r in range(u))]
    return min(ss), max(ss) - min(ss) + 1


def convt_weight_matrix(w: torch.Tensor, u: int, cpad: int) -> torch.Tensor:
    """ConvTranspose1d weight [Cin, Cout, k] -> GEMM weight [u * Cout, ntaps * cpad]: row r * Cout + co, column t * cpad + ci holds
    w[ci, co, r + pad - (shift0 + t) * u] (0 where that tap index is outside [0, k))."""
    cin, cout, k = w.shape
    pad = (k - u) // 2
    shift0, ntaps = convt_taps(k, u)
    m = torch.zeros(u, cout, ntaps, cpad)
    for r in range(u):
        for t in range(ntaps):
            j = r + pad - (shift0 + t) * u
            if 0 <= j < k:
                m[r, :, t, :cin] = w[:, :, j].t()
    return m.reshape(u * cout, ntaps * cpad)


def conv_cl(y, w, bias, dilation, cpad=None):
    """Conv1d ("same" padding) on channels-last y via im2col + matmul."""
    cout, cin, k = w.shape
    cpad = cpad or rup(cin, 32)
    col = im2col(y, k, -(k // 2) * dilation, dilation, cpad)
    out = col @ conv_weight_matrix(w, cpad).t()
    return out + bias if bias is not None else out


def conv_implicit_cl(y, wmat, ntaps, shift0, dstep, cpad):
    """The implicit-GEMM form (csrc/conv_gemm.h): ONE padded operand copy [b, L, cpad]; for tap j the activation rows are read shifted
    by shift0 + j * dstep (zero outside [0, L)) against columns [j * cpad, (j + 1) * cpad) of the same weight matrix."""
    b, L, C = y.shape
    yp = torch.zeros(b, L, cpad)
    yp[:, :, :C] = y
    out = torch.zeros(b, L, wmat.shape[0])
    l = torch.arange(L)
    for j in range(ntaps):
        src = l + shift0 + j * dstep
        ok = (src >= 0) & (src < L)
        rows = torch.zeros(b, L, cpad)
        rows[:, ok] = yp[:, src[ok]]
        out = out + rows @ wmat[:, j * cpad:(j + 1) * cpad].t()
    return out


def convt_cl(y, w, bias, u):
    cin, cout, k = w.shape
    cpad = rup(cin, 32)
    shift0, ntaps = convt_taps(k, u)
    col = im2col(y, ntaps, shift0, 1, cpad)
    out = col @ convt_weight_matrix(w, u, cpad).t() + bias.repeat(u)
    b, L, _ = y.shape
    return out.reshape(b, L * u, cout)


def bigvgan_forward_cl(sd, cfg, mel, f):
    """The whole generator in the kernels' formulation: mel [b, num_mels, T] -> wav [b, T * hop]."""
    def act(pfx, x):
        a = sd[pfx + "act.alpha"]
        return aa_snake(x, a, sd[pfx + "act.beta"] if cfg.activation == "snakebeta" else a, f, cfg.snake_logscale)

    x = conv_cl(mel.transpose(1, 2), sd["conv_pre.weight"], sd["conv_pre.bias"], 1)
    nk = len(cfg.resblock_kernel_sizes)
    for i, u in enumerate(cfg.upsample_rates):
        x = convt_cl(x, sd[f"ups.{i}.0.weight"], sd[f"ups.{i}.0.bias"], u)
        acc = None
        for j in range(nk):
            pfx = f"resblocks.{i * nk + j}."
            r = x
            for m, d in enumerate(cfg.resblock_dilation_sizes[j]):
                if cfg.resblock == "1":
                    t = conv_cl(act(f"{pfx}activations.{2 * m}.", r), sd[f"{pfx}convs1.{m}.weight"], sd[f"{pfx}convs1.{m}.bias"], d)
                    r = conv_cl(act(f"{pfx}activations.{2 * m + 1}.", t), sd[f"{pfx}convs2.{m}.weight"], sd[f"{pfx}convs2.{m}.bias"], 1) + r
                else:
                    r = conv_cl(act(f"{pfx}activations.{m}.", r), sd[f"{pfx}convs.{m}.weight"], sd[f"{pfx}convs.{m}.bias"], d) + r
            acc = r if acc is None else acc + r
        x = acc / nk
    y = act("activation_post.", x)
    w = sd["conv_post.weight"]  # [1, C, 7]
    out = torch.zeros(y.shape[0], y.shape[1])
    L = y.shape[1]
    l = torch.arange(L)
    for j in range(7):
        src = l + j - 3
        ok = (src >= 0) & (src < L)
        out[:, ok] += (y[:, src[ok], :] * w[0, :, j]).sum(-1)
    if cfg.use_bias_at_final:
        out = out + sd["conv_post.bias"]
    return torch.tanh(out) if cfg.use_tanh_at_final else out.clamp(-1.0, 1.0)
