"""Host-side mirror of the BigVGAN vocoder object on the inference path, over the C ABI (``f5hip_bigvgan_*``).

``F5HipBigVGAN`` quacks like what the reference's ``load_vocoder(vocoder_name="bigvgan")`` returns
(reference ``src/f5_tts/infer/utils_infer.py:130-144``: ``bigvgan.BigVGAN.from_pretrained(path, use_cuda_kernel=False)``,
``.remove_weight_norm()``, ``.eval().to(device)``) and what ``infer_batch_process`` does with it (``:512-513``:
``vocoder(mel[b, 100, T]) -> wav[b, 1, 256 T]``).  The generator's source is an un-vendored submodule of the reference
(``.gitmodules:1-3``); tensor names and ``config.json`` fields are upstream's (NVIDIA/BigVGAN ``bigvgan.py``).  Every numeric
step runs in ``libf5hip.so``; torch is used for device memory and streams only.  There is no CPU path.
"""
from __future__ import annotations

import ctypes as C
import json
import os
from typing import Dict, Union

import torch

from . import binding
from .binding import PRECISIONS, BigVGANConfigC, check, load_library
from .config import BIGVGAN_V2_24K_100B_256X, BigVGANConfig


def fold_weight_norm(sd: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    """What ``remove_weight_norm()`` leaves behind, computed on the state dict: ``weight = g * v / ||v||`` with the norm over every
    dim but 0 (``torch.nn.utils.weight_norm`` default ``dim=0``; for ``ConvTranspose1d`` dim 0 is the input channel).  Accepts the
    ``weight_g`` / ``weight_v`` spelling of the published checkpoints and the ``parametrizations.weight.original0/1`` one; the
    resampling-filter buffers (``*.filter``) are dropped — the library recomputes the one kaiser-sinc filter they all hold."""
    out: Dict[str, torch.Tensor] = {}
    for k, v in sd.items():
        if k.endswith(".filter"):
            continue
        for g_sfx, v_sfx in ((".weight_g", ".weight_v"), (".parametrizations.weight.original0", ".parametrizations.weight.original1")):
            if k.endswith(g_sfx):
                base = k[: -len(g_sfx)]
                vv = sd[base + v_sfx].to(torch.float32)
                norm = vv.reshape(vv.shape[0], -1).norm(dim=1).reshape(-1, *([1] * (vv.ndim - 1)))
                out[base + ".weight"] = v.to(torch.float32) * vv / norm
                break
            if k.endswith(v_sfx):
                break
        else:
            out[k] = v.to(torch.float32)
    return out


def config_from_json(h: dict) -> BigVGANConfig:
    """``config.json`` of a published BigVGAN checkpoint -> ``BigVGANConfig`` (only the generator fields matter here)."""
    return BigVGANConfig(num_mels=int(h["num_mels"]), upsample_rates=tuple(h["upsample_rates"]),
                         upsample_kernel_sizes=tuple(h["upsample_kernel_sizes"]), upsample_initial_channel=int(h["upsample_initial_channel"]),
                         resblock=str(h["resblock"]), resblock_kernel_sizes=tuple(h["resblock_kernel_sizes"]),
                         resblock_dilation_sizes=tuple(tuple(d) for d in h["resblock_dilation_sizes"]), activation=h.get("activation", "snakebeta"),
                         snake_logscale=bool(h.get("snake_logscale", True)), use_tanh_at_final=bool(h.get("use_tanh_at_final", True)),
                         use_bias_at_final=bool(h.get("use_bias_at_final", True)))


def _config_c(cfg: BigVGANConfig) -> BigVGANConfigC:
    if cfg.activation not in ("snake", "snakebeta"):
        raise NotImplementedError("activation incorrectly specified. check the config file and look for 'activation'.")  # upstream's message
    if cfg.resblock not in ("1", "2"):
        raise ValueError(f"Incorrect resblock class specified in hyperparameters. Got {cfg.resblock}")
    nu, nk = len(cfg.upsample_rates), len(cfg.resblock_kernel_sizes)
    if nu > 8 or nk > 4 or any(len(d) > 4 for d in cfg.resblock_dilation_sizes) or len(cfg.upsample_kernel_sizes) != nu \
            or len(cfg.resblock_dilation_sizes) != nk:
        raise ValueError("unsupported BigVGAN shape: at most 8 upsampling stages, 4 parallel resblocks, 4 dilations each")
    c = BigVGANConfigC(num_mels=cfg.num_mels, num_upsamples=nu, upsample_initial_channel=cfg.upsample_initial_channel,
                       resblock=int(cfg.resblock), num_kernels=nk, activation=int(cfg.activation == "snakebeta"),
                       snake_logscale=int(cfg.snake_logscale), use_tanh_at_final=int(cfg.use_tanh_at_final),
                       use_bias_at_final=int(cfg.use_bias_at_final))
    for i in range(nu):
        c.upsample_rates[i], c.upsample_kernel_sizes[i] = cfg.upsample_rates[i], cfg.upsample_kernel_sizes[i]
    for j in range(nk):
        c.resblock_kernel_sizes[j] = cfg.resblock_kernel_sizes[j]
        c.resblock_num_dilations[j] = len(cfg.resblock_dilation_sizes[j])
        for m, d in enumerate(cfg.resblock_dilation_sizes[j]):
            c.resblock_dilation_sizes[j][m] = d
    return c


class F5HipBigVGAN:
    """Owns one ``f5hip_bigvgan`` context (weights + workspace on one GPU).  ``vocoder(mel)`` as in the reference."""

    def __init__(self, cfg: BigVGANConfig = BIGVGAN_V2_24K_100B_256X, device: Union[int, str, torch.device] = 0, precision: str = "fp16x3"):
        self.lib = load_library()
        dev = torch.device(device) if not isinstance(device, int) else torch.device("cuda", device)
        if dev.type != "cuda":
            raise binding.F5HipError("F5HipBigVGAN needs a HIP device (torch device type 'cuda'); there is no CPU path")
        self.device = torch.device("cuda", dev.index if dev.index is not None else torch.cuda.current_device())
        if precision not in PRECISIONS:
            raise ValueError(f"precision must be one of {sorted(PRECISIONS)}")
        self.cfg, self.h, self.precision = cfg, cfg, precision
        self._ctx = C.c_void_p()
        c = _config_c(cfg)
        check(self.lib, None, self.lib.f5hip_bigvgan_create(C.byref(c), self.device.index, C.byref(self._ctx)), self.lib.f5hip_bigvgan_last_error)
        self.finalized = False

    # -- lifetime ------------------------------------------------------------------------------
    def close(self):
        if getattr(self, "_ctx", None) is not None and self._ctx.value:
            self.lib.f5hip_bigvgan_destroy(self._ctx)
            self._ctx = C.c_void_p()

    def __del__(self):  # pragma: no cover
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, st):
        check(self.lib, self._ctx, st, self.lib.f5hip_bigvgan_last_error)

    # -- weights -------------------------------------------------------------------------------
    def tensor_table(self):
        out = []
        name, numel = C.c_char_p(), C.c_int64()
        for i in range(self.lib.f5hip_bigvgan_num_tensors(self._ctx)):
            self._chk(self.lib.f5hip_bigvgan_tensor_info(self._ctx, i, C.byref(name), C.byref(numel)))
            out.append((name.value.decode(), numel.value))
        return out

    def load_state_dict(self, sd: Dict[str, torch.Tensor], strict: bool = True):
        """Generator state dict, raw (``weight_g`` / ``weight_v``: the published ``bigvgan_generator.pt["generator"]``) or after
        ``remove_weight_norm()``."""
        sd = fold_weight_norm(sd)
        expected = dict(self.tensor_table())
        missing = [k for k in expected if k not in sd]
        unexpected = [k for k in sd if k not in expected]
        if strict and (missing or unexpected):
            raise RuntimeError(f"Error(s) in loading state_dict for BigVGAN: missing {missing[:4]}, unexpected {unexpected[:4]}")
        for k in expected:
            if k in sd:
                t = sd[k].detach().to(device="cpu", dtype=torch.float32).contiguous()
                self._chk(self.lib.f5hip_bigvgan_load_tensor(self._ctx, k.encode(), C.c_void_p(t.data_ptr()), t.numel()))
        with torch.cuda.device(self.device):
            self._chk(self.lib.f5hip_bigvgan_finalize(self._ctx))
        self.finalized = True
        return self

    @classmethod
    def from_pretrained(cls, local_path: str, use_cuda_kernel: bool = False, device=0, precision: str = "fp16x3", **_ignored) -> "F5HipBigVGAN":
        """``<local_path>/config.json`` + ``<local_path>/bigvgan_generator.pt`` (the layout of nvidia/bigvgan_v2_24khz_100band_256x that
        reference utils_infer.py:136-137 points at).  Local directories only: there is no network here."""
        if not os.path.isdir(local_path):
            raise ValueError("no network here: pass a local directory holding config.json and bigvgan_generator.pt")
        cfg = config_from_json(json.load(open(os.path.join(local_path, "config.json"))))
        ck = torch.load(os.path.join(local_path, "bigvgan_generator.pt"), map_location="cpu", weights_only=True)
        return cls(cfg, device=device, precision=precision).load_state_dict(ck.get("generator", ck))

    # -- the nn.Module surface the reference touches ---------------------------------------------
    def remove_weight_norm(self):
        """Folded at load time (``fold_weight_norm``); kept because the reference calls it (utils_infer.py:143)."""

    def eval(self):
        return self

    def to(self, *a, **k):
        return self

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        """mel ``[b, num_mels, T]`` -> wav ``[b, 1, T * hop]`` (BigVGAN.forward)."""
        if x.ndim != 3 or x.shape[1] != self.cfg.num_mels:
            raise ValueError(f"expected mel [b, {self.cfg.num_mels}, T], got {tuple(x.shape)}")
        x = x.to(device=self.device, dtype=torch.float32).contiguous()
        b, _, t = x.shape
        out = torch.empty((b, 1, t * self.cfg.hop), device=self.device, dtype=torch.float32)
        with torch.cuda.device(self.device):
            stream = C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
            self._chk(self.lib.f5hip_bigvgan_forward(self._ctx, C.c_void_p(x.data_ptr()), b, t, 1, PRECISIONS[self.precision],
                                                     C.c_void_p(out.data_ptr()), stream))
        return out

    __call__ = forward

    def set_option(self, key: str, value: int):
        self._chk(self.lib.f5hip_bigvgan_set_option(self._ctx, key.encode(), int(value)))

    def kernel_stats(self) -> Dict[str, dict]:
        """Per-kernel-class calls / milliseconds / algorithmic FLOPs and bytes accumulated while option "profile" is on."""
        out = {}
        name, calls = C.c_char_p(), C.c_int64()
        ms, fl, by = C.c_double(), C.c_double(), C.c_double()
        for i in range(self.lib.f5hip_bigvgan_num_kernel_stats(self._ctx)):
            self._chk(self.lib.f5hip_bigvgan_kernel_stat(self._ctx, i, C.byref(name), C.byref(calls), C.byref(ms), C.byref(fl), C.byref(by)))
            out[name.value.decode()] = dict(calls=calls.value, ms=ms.value, flops=fl.value, bytes=by.value)
        return out

    def reset_kernel_stats(self):
        self._chk(self.lib.f5hip_bigvgan_reset_kernel_stats(self._ctx))

    def stage_tensor(self, x: torch.Tensor, stage: int) -> torch.Tensor:
        """Parity tap (tests): the channels-last tensor [b, L_k, C_k] after conv_pre (stage 0) / after upsampling stage k."""
        x = x.to(device=self.device, dtype=torch.float32).contiguous()
        b, _, t = x.shape
        L = t
        for u in self.cfg.upsample_rates[:stage]:
            L *= u
        out = torch.empty((b, L, self.cfg.upsample_initial_channel >> stage), device=self.device, dtype=torch.float32)
        self._chk(self.lib.f5hip_bigvgan_set_option(self._ctx, b"stop_after_stage", stage))
        try:
            with torch.cuda.device(self.device):
                stream = C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
                self._chk(self.lib.f5hip_bigvgan_forward(self._ctx, C.c_void_p(x.data_ptr()), b, t, 1, PRECISIONS[self.precision],
                                                         C.c_void_p(out.data_ptr()), stream))
        finally:
            self._chk(self.lib.f5hip_bigvgan_set_option(self._ctx, b"stop_after_stage", -1))
        return out
