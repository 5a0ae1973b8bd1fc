"""TEST INFRASTRUCTURE — not product code.

Import the reference's *own* sampler/backbone code (``/root/reference/src/f5_tts/model``)
verbatim on CPU, by registering small ``sys.modules`` shims for the third-party
packages that are not installed in this container.  Used ONLY by
``oracle/make_golden.py`` (fixture generation) and by the ``not gpu`` tests that
validate ``oracle/f5_oracle.py`` against the reference when ``/root/reference``
exists.  Nothing here can travel to the GPU box (the reference tree is absent
there); the committed fixtures under ``tests/golden/`` do.

What the shims restate (un-vendored third-party arithmetic):

* ``torchdiffeq.odeint``      fixed-grid euler / midpoint on exactly the supplied grid
                              (call site: reference ``src/f5_tts/model/cfm.py:218``).
* ``x_transformers``          ``RotaryEmbedding.forward_from_seq_len``, ``apply_rotary_pos_emb``
                              (interleaved pairs), ``RMSNorm`` (call sites
                              ``model/backbones/dit.py:207,352``, ``model/modules.py:500-509``,
                              ``model/backbones/unett.py:154``).  The interleaved convention is
                              cross-checked in-repo by
                              ``runtime/triton_trtllm/model_repo_f5_tts/f5_tts/1/f5_tts_trtllm.py:232-237``.
* ``torchaudio.transforms.MelSpectrogram``  torch.stft + HTK triangular filterbank
                              (call site ``model/modules.py:91-101``).
* ``librosa.filters.mel``     slaney-scale, slaney-normalised triangular filterbank (bigvgan-type mel only, call site
                              ``model/modules.py:50``), delegated to ``oracle/f5_oracle.py::slaney_mel_basis``.
* ``rjieba``, ``pypinyin``    import-time stubs only.

These restatements have no golden vectors in the reference (it has no tests):
parity of those pieces is "unpinned by the reference" and anchored on the
in-repo cross-checks listed above plus ``scripts/conv_stft.py`` for STFT/iSTFT.
"""
from __future__ import annotations

import importlib
import math
import os
import sys
import types

import torch
import torch.nn.functional as F
from torch import nn

REFERENCE_SRC = os.environ.get("F5_REFERENCE_SRC", "/root/reference/src")


def reference_available() -> bool:
    return os.path.isfile(os.path.join(REFERENCE_SRC, "f5_tts", "model", "cfm.py"))


# ----------------------------------------------------------------------------------------------
# torchdiffeq
# ----------------------------------------------------------------------------------------------
def _odeint(fn, y0, t, method="euler", **_kw):
    """Fixed-grid solver on exactly the supplied ``t`` (torchdiffeq ``FixedGridODESolver``
    with ``step_size=None``): returns all states stacked, ``t_i`` passed as a 0-d tensor."""
    ys = [y0]
    y = y0
    for i in range(t.shape[0] - 1):
        t0, t1 = t[i], t[i + 1]
        dt = t1 - t0
        if method == "euler":
            dy = dt * fn(t0, y)
        elif method == "midpoint":
            half = 0.5 * dt
            k = fn(t0, y)
            dy = dt * fn(t0 + half, y + k * half)
        else:  # pragma: no cover
            raise ValueError(f"unsupported ode method {method}")
        y = y + dy
        ys.append(y)
    return torch.stack(ys, dim=0)


# ----------------------------------------------------------------------------------------------
# x_transformers
# ----------------------------------------------------------------------------------------------
class _RotaryEmbedding(nn.Module):
    def __init__(self, dim, use_xpos=False, scale_base=512, interpolation_factor=1.0, base=10000, base_rescale_factor=1.0):
        super().__init__()
        base *= base_rescale_factor ** (dim / (dim - 2))
        inv_freq = 1.0 / (base ** (torch.arange(0, dim, 2).float() / dim))
        self.register_buffer("inv_freq", inv_freq)
        self.interpolation_factor = interpolation_factor
        self.scale = None

    def forward_from_seq_len(self, seq_len):
        t = torch.arange(seq_len, device=self.inv_freq.device)
        return self.forward(t)

    def forward(self, t):
        max_pos = t.max() + 1
        if t.ndim == 1:
            t = t[None, :]
        freqs = torch.einsum("b i , j -> b i j", t.type_as(self.inv_freq), self.inv_freq) / self.interpolation_factor
        freqs = torch.stack((freqs, freqs), dim=-1)
        freqs = freqs.reshape(*freqs.shape[:-2], -1)  # [f0,f0,f1,f1,...]
        return freqs, 1.0


def _rotate_half(x):
    x = x.reshape(*x.shape[:-1], x.shape[-1] // 2, 2)
    x1, x2 = x.unbind(dim=-1)
    x = torch.stack((-x2, x1), dim=-1)
    return x.reshape(*x.shape[:-2], -1)


def _apply_rotary_pos_emb(t, freqs, scale=1):
    rot_dim, seq_len, orig_dtype = freqs.shape[-1], t.shape[-2], t.dtype
    freqs = freqs[:, -seq_len:, :]
    scale = scale[:, -seq_len:, :] if isinstance(scale, torch.Tensor) else scale
    if t.ndim == 4 and freqs.ndim == 3:
        freqs = freqs[:, None]  # b 1 n d
    t, t_unrotated = t[..., :rot_dim], t[..., rot_dim:]
    t = (t * freqs.cos() * scale) + (_rotate_half(t) * freqs.sin() * scale)
    out = torch.cat((t, t_unrotated), dim=-1)
    return out.type(orig_dtype)


class _XRMSNorm(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.scale = dim**0.5
        self.g = nn.Parameter(torch.ones(dim))

    def forward(self, x):
        return F.normalize(x, dim=-1) * self.scale * self.g


# ----------------------------------------------------------------------------------------------
# torchaudio
# ----------------------------------------------------------------------------------------------
def _hz_to_mel_htk(f):
    return 2595.0 * math.log10(1.0 + f / 700.0)


def melscale_fbanks_htk(n_freqs, f_min, f_max, n_mels, sample_rate):
    """torchaudio.functional.melscale_fbanks(norm=None, mel_scale="htk") -> [n_freqs, n_mels]."""
    all_freqs = torch.linspace(0, sample_rate // 2, n_freqs)
    m_min = _hz_to_mel_htk(f_min)
    m_max = _hz_to_mel_htk(f_max)
    m_pts = torch.linspace(m_min, m_max, n_mels + 2)
    f_pts = 700.0 * (10.0 ** (m_pts / 2595.0) - 1.0)
    f_diff = f_pts[1:] - f_pts[:-1]
    slopes = f_pts.unsqueeze(0) - all_freqs.unsqueeze(1)
    down_slopes = (-1.0 * slopes[:, :-2]) / f_diff[:-1]
    up_slopes = slopes[:, 2:] / f_diff[1:]
    fb = torch.max(torch.zeros(1), torch.min(down_slopes, up_slopes))
    return fb


class _MelSpectrogram(nn.Module):
    def __init__(self, sample_rate=16000, n_fft=400, win_length=None, hop_length=None, f_min=0.0, f_max=None,
                 pad=0, n_mels=128, window_fn=torch.hann_window, power=2.0, normalized=False, wkwargs=None,
                 center=True, pad_mode="reflect", onesided=None, norm=None, mel_scale="htk"):
        super().__init__()
        assert norm is None and mel_scale == "htk" and not normalized and pad == 0
        self.n_fft = n_fft
        self.win_length = win_length or n_fft
        self.hop_length = hop_length or self.win_length // 2
        self.power = power
        self.center = center
        self.pad_mode = pad_mode
        self.register_buffer("window", window_fn(self.win_length), persistent=False)
        f_max = f_max if f_max is not None else float(sample_rate // 2)
        self.register_buffer("fb", melscale_fbanks_htk(n_fft // 2 + 1, f_min, f_max, n_mels, sample_rate), persistent=False)

    def forward(self, waveform):
        spec = torch.stft(waveform, self.n_fft, hop_length=self.hop_length, win_length=self.win_length,
                          window=self.window, center=self.center, pad_mode=self.pad_mode, normalized=False,
                          onesided=True, return_complex=True)
        spec = spec.abs()
        if self.power != 1:
            spec = spec.pow(self.power)
        # [..., freq, time] -> mel
        return torch.matmul(spec.transpose(-1, -2), self.fb).transpose(-1, -2)


# ----------------------------------------------------------------------------------------------
def install():
    """Register shims + bare ``f5_tts`` packages (bypasses ``model/__init__.py`` -> trainer imports)."""
    if "f5_tts.model.cfm" in sys.modules:
        return
    if not reference_available():
        raise RuntimeError(f"reference tree not found under {REFERENCE_SRC}")

    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    if "torchdiffeq" not in sys.modules:
        mod("torchdiffeq", odeint=_odeint)
    if "x_transformers" not in sys.modules:
        xt = mod("x_transformers")
        xtx = mod("x_transformers.x_transformers", RotaryEmbedding=_RotaryEmbedding,
                  apply_rotary_pos_emb=_apply_rotary_pos_emb, RMSNorm=_XRMSNorm)
        xt.x_transformers = xtx
        xt.RMSNorm = _XRMSNorm
    if "torchaudio" not in sys.modules:
        ta = mod("torchaudio")
        tr = mod("torchaudio.transforms", MelSpectrogram=_MelSpectrogram)
        ta.transforms = tr
    if "librosa" not in sys.modules:
        lb = mod("librosa")

        def _slaney_mel(sr, n_fft, n_mels=128, fmin=0.0, fmax=None, **_k):
            # bigvgan-type mel only (model/modules.py:50).  librosa is absent: the filterbank comes from the restatement of its published
            # algorithm in oracle/f5_oracle.py (parity unpinned for the table; everything around it is the reference's own code).
            from oracle.f5_oracle import slaney_mel_basis

            return slaney_mel_basis(sr, n_fft, n_mels, fmin, fmax).numpy()

        lf = mod("librosa.filters", mel=_slaney_mel)
        lb.filters = lf
    g2p_stubs = []  # import-time only: taken out of sys.modules again below, so that nothing else in the process mistakes them for the packages
    if "rjieba" not in sys.modules:  # (f5_tts_amd.infer.convert_char_to_pinyin uses rjieba when it is importable: with the stub left behind,
        mod("rjieba", cut=lambda s: [s])  # tests/test_infer_host.py failed whenever an oracle test had run earlier in the same process)
        g2p_stubs.append("rjieba")
    if "pypinyin" not in sys.modules:
        mod("pypinyin", Style=types.SimpleNamespace(TONE3=0), lazy_pinyin=lambda s, **k: list(s))
        g2p_stubs.append("pypinyin")

    base = os.path.join(REFERENCE_SRC, "f5_tts")
    for name, path in (("f5_tts", base), ("f5_tts.model", os.path.join(base, "model")),
                       ("f5_tts.model.backbones", os.path.join(base, "model", "backbones"))):
        if name not in sys.modules:
            pkg = types.ModuleType(name)
            pkg.__path__ = [path]
            sys.modules[name] = pkg
    for sub in ("f5_tts.model.utils", "f5_tts.model.modules", "f5_tts.model.backbones.dit",
                "f5_tts.model.backbones.unett", "f5_tts.model.backbones.mmdit", "f5_tts.model.cfm"):
        importlib.import_module(sub)
    for name in g2p_stubs:  # the reference's modules keep their own bindings
        sys.modules.pop(name, None)


def reference_classes():
    """Return (CFM, DiT, UNetT) — the reference's own classes."""
    install()
    from f5_tts.model.backbones.dit import DiT
    from f5_tts.model.backbones.unett import UNetT
    from f5_tts.model.cfm import CFM

    return CFM, DiT, UNetT


MEL_KW = dict(n_fft=1024, hop_length=256, win_length=1024, n_mel_channels=100, target_sample_rate=24000, mel_spec_type="vocos")


def build_reference_cfm(cfg, sd, method="euler"):
    """The reference's own CFM around its own backbone class, loaded (strict) with a synth.py state dict — what oracle/make_golden.py runs
    to mint the goldens and what bench.py's cpu_baseline leg times in the build container (kind "reference")."""
    CFM, DiT, UNetT = reference_classes()
    backbone = {"UNetT": UNetT, "MMDiT": reference_mmdit() if cfg.backbone == "MMDiT" else None}.get(cfg.backbone, DiT)
    model = CFM(transformer=backbone(**cfg.arch_kwargs()), mel_spec_kwargs=MEL_KW, odeint_kwargs=dict(method=method))
    model.load_state_dict(sd, strict=True)  # proves the key contract of synth.py == the reference's
    return model.eval()


def reference_mmdit():
    """The reference's own MMDiT class (src/f5_tts/model/backbones/mmdit.py)."""
    install()
    from f5_tts.model.backbones.mmdit import MMDiT

    return MMDiT


def reference_conv_stft():
    """The reference's runnable conv-STFT (``runtime/triton_trtllm/scripts/conv_stft.py``)."""
    path = os.path.join(REFERENCE_SRC, "f5_tts", "runtime", "triton_trtllm", "scripts", "conv_stft.py")
    spec = importlib.util.spec_from_file_location("_ref_conv_stft", path)
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m
