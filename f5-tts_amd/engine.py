"""Host-side mirror of the reference's inference surface for the hot path, over the C ABI.

* ``F5HipCFM``   quacks like the object ``load_model`` returns (reference
  ``src/f5_tts/infer/utils_infer.py:238-276``): ``.sample(...)`` with the signature of
  ``CFM.sample`` (reference ``src/f5_tts/model/cfm.py:84-102``), ``.mel_spec(wav)``, ``.eval()``,
  ``.to()``, ``.device``, ``.transformer.clear_cache()`` / ``.transformer.dim``.
* ``F5HipVocos`` quacks like the ``vocos.Vocos`` object ``load_vocoder`` returns
  (``utils_infer.py:106-129``): ``.decode(mel[b,100,T]) -> wav[b, nw]``.

torch is used here for device memory, streams and the CPU random generator only (the reference's
noise comes from torch's CPU generator stream, cfm.py:196-201 — it must be reproduced bit-for-bit, so
it is generated on the host exactly as the reference does and uploaded).  Every numeric step of the
path runs in ``libf5hip.so``.
"""
from __future__ import annotations

import ctypes as C
from typing import Callable, Dict, Iterable, List, Optional, Sequence, Union

import torch

from . import binding
from .binding import PRECISIONS, DitConfigC, VocosConfigC, check, load_bench_library, load_library
from .config import DiTConfig, VocosConfig

# reference src/f5_tts/model/utils.py:205-218
_EPSS = {
    5: [0, 2, 4, 8, 16, 32],
    6: [0, 2, 4, 6, 8, 16, 32],
    7: [0, 2, 4, 6, 8, 16, 24, 32],
    10: [0, 2, 4, 6, 8, 12, 16, 20, 24, 28, 32],
    12: [0, 2, 4, 6, 8, 10, 12, 14, 16, 20, 24, 28, 32],
    16: [0, 1, 2, 3, 4, 5, 6, 7, 8, 10, 12, 14, 16, 20, 24, 28, 32],
}


def get_epss_timesteps(n: int, dtype=torch.float32) -> torch.Tensor:
    t = _EPSS.get(n, [])
    if not t:
        return torch.linspace(0, 1, n + 1, dtype=dtype)
    return (1 / 32) * torch.tensor(t, dtype=dtype)


def lens_to_mask(t: torch.Tensor, length: Optional[int] = None) -> torch.Tensor:
    """reference src/f5_tts/model/utils.py:53-58"""
    if length is None:
        length = int(t.amax())
    seq = torch.arange(length, device=t.device)
    return seq[None, :] < t[:, None]


def list_str_to_idx(text: Sequence[Union[str, Sequence[str]]], vocab_char_map: Dict[str, int], padding_value=-1) -> torch.Tensor:
    """reference src/f5_tts/model/utils.py:99-106"""
    idx = [torch.tensor([vocab_char_map.get(c, 0) for c in t]) for t in text]
    return torch.nn.utils.rnn.pad_sequence(idx, padding_value=padding_value, batch_first=True)


def list_str_to_tensor(text: Sequence[str], padding_value=-1) -> torch.Tensor:
    """reference src/f5_tts/model/utils.py:92-95"""
    t = [torch.tensor([*bytes(s, "UTF-8")]) for s in text]
    return torch.nn.utils.rnn.pad_sequence(t, padding_value=padding_value, batch_first=True)


def _ptr(t: Optional[torch.Tensor]):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


class F5HipEngine:
    """Owns one ``f5hip_ctx``: the packed weight blob, derived layouts and workspace on one GPU."""

    def __init__(self, dit_cfg: DiTConfig, vocos_cfg: Optional[VocosConfig] = None, device: Union[int, str, torch.device] = 0):
        self.lib = load_library()
        dev = torch.device(device) if not isinstance(device, int) else torch.device("cuda", device)
        if dev.type != "cuda":
            raise binding.F5HipError("F5HipEngine needs a HIP device (torch device type 'cuda'); there is no CPU path")
        self.device = torch.device("cuda", dev.index if dev.index is not None else torch.cuda.current_device())
        self.dit_cfg, self.vocos_cfg = dit_cfg, vocos_cfg
        c = DitConfigC(dim=dit_cfg.dim, depth=dit_cfg.depth, heads=dit_cfg.heads, dim_head=dit_cfg.dim_head,
                       ff_inner=dit_cfg.ff_inner, mel_dim=dit_cfg.mel_dim, text_num_embeds=dit_cfg.text_num_embeds,
                       text_dim=dit_cfg.text_dim, conv_layers=dit_cfg.conv_layers,
                       text_mask_padding=int(dit_cfg.text_mask_padding),
                       pe_attn_head=-1 if dit_cfg.pe_attn_head is None else int(dit_cfg.pe_attn_head),
                       attn_mask_enabled=int(dit_cfg.attn_mask_enabled), conv_pos_kernel=dit_cfg.conv_pos_kernel,
                       conv_pos_groups=dit_cfg.conv_pos_groups, backbone={"DiT": 0, "UNetT": 1, "MMDiT": 2}[dit_cfg.backbone],
                       qk_norm={None: 0, "rms_norm": 1}[dit_cfg.qk_norm],  # KeyError == the reference's ValueError (modules.py:409)
                       long_skip_connection=int(dit_cfg.long_skip_connection),
                       text_average_upsampling=int(dit_cfg.text_embedding_average_upsampling),
                       skip_connect_type={"concat": 0, "add": 1, "none": 2}[dit_cfg.skip_connect_type])
        v = None
        if vocos_cfg is not None:
            v = VocosConfigC(input_channels=vocos_cfg.input_channels, dim=vocos_cfg.dim,
                             intermediate_dim=vocos_cfg.intermediate_dim, num_layers=vocos_cfg.num_layers,
                             n_fft=vocos_cfg.n_fft, hop_length=vocos_cfg.hop_length)
        self._ctx = C.c_void_p()
        st = self.lib.f5hip_create(C.byref(c), C.byref(v) if v is not None else None, self.device.index, C.byref(self._ctx))
        check(self.lib, None, st)
        self.finalized = False

    # -- lifetime ------------------------------------------------------------------------------
    @property
    def bench_lib(self):
        """libf5hip_bench.so (include/f5hip_bench.h): microbenchmarks and format checks over this context — tools and tests only."""
        return load_bench_library()

    def close(self):
        if getattr(self, "_ctx", None) is not None and self._ctx.value:
            self.lib.f5hip_destroy(self._ctx)
            self._ctx = C.c_void_p()

    def __del__(self):  # pragma: no cover
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, st):
        check(self.lib, self._ctx, st)

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    # -- weights -------------------------------------------------------------------------------
    def tensor_table(self):
        out = []
        name, numel, off = C.c_char_p(), C.c_int64(), C.c_int64()
        for i in range(self.lib.f5hip_num_tensors(self._ctx)):
            self._chk(self.lib.f5hip_tensor_info(self._ctx, i, C.byref(name), C.byref(numel), C.byref(off)))
            out.append((name.value.decode(), numel.value, off.value))
        return out

    def load_state_dict(self, sd: Dict[str, torch.Tensor], strict: bool = True, finalize: bool = True):
        """Load tensors by their reference state-dict keys (see ``load_checkpoint`` for the EMA key mapping)."""
        expected = {n for n, _, _ in self.tensor_table()}
        for k, v in sd.items():
            if k not in expected:
                if strict:
                    raise ValueError(f"unexpected key in state dict: {k}")
                continue
            t = v.detach().to(device="cpu", dtype=torch.float32).contiguous()
            self._chk(self.lib.f5hip_load_tensor(self._ctx, k.encode(), C.c_void_p(t.data_ptr()), t.numel()))
        if finalize:
            self.finalize()

    def weight_blob(self) -> torch.Tensor:
        """The packed fp32 device blob as a 1-D uint8-free view (float32) — used for the RCCL broadcast."""
        p, nbytes = C.c_void_p(), C.c_int64()
        self._chk(self.lib.f5hip_weight_blob(self._ctx, C.byref(p), C.byref(nbytes)))
        return _as_tensor(p.value, nbytes.value // 4, self.device)

    def mark_all_loaded(self):
        self._chk(self.lib.f5hip_mark_all_loaded(self._ctx))

    def loaded_mask(self) -> torch.Tensor:
        """uint8 [num_tensors] (tensor_table order): which state-dict entries this context has received."""
        n = self.lib.f5hip_num_tensors(self._ctx)
        m = torch.zeros(n, dtype=torch.uint8)
        self._chk(self.lib.f5hip_loaded_mask(self._ctx, C.c_void_p(m.data_ptr()), n))
        return m

    def set_loaded_mask(self, mask: torch.Tensor):
        """Receiver side of the weight broadcast: adopt the sender's mask (optional buffers included)."""
        m = mask.detach().to(device="cpu", dtype=torch.uint8).contiguous()
        self._chk(self.lib.f5hip_set_loaded_mask(self._ctx, C.c_void_p(m.data_ptr()), m.numel()))
        self.finalized = False

    def finalize(self):
        with torch.cuda.device(self.device):
            self._chk(self.lib.f5hip_finalize_weights(self._ctx))
        self.finalized = True

    def set_option(self, key: str, value: int):
        self._chk(self.lib.f5hip_set_option(self._ctx, key.encode(), int(value)))

    def kernel_stats(self) -> Dict[str, dict]:
        out = {}
        name, calls = C.c_char_p(), C.c_int64()
        ms, fl, by = C.c_double(), C.c_double(), C.c_double()
        for i in range(self.lib.f5hip_num_kernel_stats(self._ctx)):
            self._chk(self.lib.f5hip_kernel_stat(self._ctx, i, C.byref(name), C.byref(calls), C.byref(ms), C.byref(fl), C.byref(by)))
            out[name.value.decode()] = dict(calls=calls.value, ms=ms.value, flops=fl.value, bytes=by.value)
        return out

    def reset_kernel_stats(self):
        self._chk(self.lib.f5hip_reset_kernel_stats(self._ctx))

    def attention_stats(self, reset: bool = True) -> Dict[str, float]:
        """How sharp the softmax rows of the materialised-score attention were since ``set_option("attn_stats", 1)`` (include/f5hip.h
        f5hip_attention_stats): the mean / largest of the rows' largest probabilities and the share of rows above 1/2."""
        out = (C.c_double * 4)()
        self._chk(self.lib.f5hip_attention_stats(self._ctx, out, int(reset)))
        rows = out[2]
        return dict(rows=int(rows), mean_max_prob=(out[1] / rows if rows else 0.0), max_prob=out[0], frac_rows_above_half=(out[3] / rows if rows else 0.0))

    # -- compute -------------------------------------------------------------------------------
    def mel(self, wav: torch.Tensor, frame_major: bool = False, mel_spec_type: str = "vocos") -> torch.Tensor:
        if mel_spec_type not in ("vocos", "bigvgan"):
            raise AssertionError('We only support two extract mel backend: vocos or bigvgan')  # modules.py:127
        wav = wav.to(device=self.device, dtype=torch.float32).contiguous()
        if wav.ndim == 3:
            wav = wav.squeeze(1)
        assert wav.ndim == 2
        b, nw = wav.shape
        big = mel_spec_type == "bigvgan"
        frames = nw // 256 if big else 1 + nw // 256  # bigvgan: no centring, (1024-256)/2 reflect padding (modules.py:59-60)
        mel = self.dit_cfg.mel_dim
        out = torch.empty((b, frames, mel) if frame_major else (b, mel, frames), device=self.device, dtype=torch.float32)
        with torch.cuda.device(self.device):
            self._chk(self.lib.f5hip_mel(self._ctx, _ptr(wav), b, nw, _ptr(out), int(frame_major), int(big), self._stream()))
        return out

    def sample(self, cond: torch.Tensor, cond_mask: torch.Tensor, text: torch.Tensor, duration: torch.Tensor, use_mask: bool,
               y0: torch.Tensor, t: torch.Tensor, cfg_strength: float, precision: str = "fp32", want_trajectory: bool = True,
               ode_method: str = "euler"):
        b, n, mel = cond.shape
        cond = cond.to(device=self.device, dtype=torch.float32).contiguous()
        y0 = y0.to(device=self.device, dtype=torch.float32).contiguous()
        cm = cond_mask.to(device="cpu", dtype=torch.uint8).contiguous()
        tx = text.to(device="cpu", dtype=torch.int64).contiguous()
        du = duration.to(device="cpu", dtype=torch.int64).contiguous()
        tt = t.to(device="cpu", dtype=torch.float32).contiguous()
        steps = tt.numel() - 1
        out = torch.empty_like(cond)
        traj = torch.empty((steps + 1, b, n, mel), device=self.device, dtype=torch.float32) if want_trajectory else None
        with torch.cuda.device(self.device):
            self._chk(self.lib.f5hip_sample(self._ctx, b, n, _ptr(cond), _ptr(cm), _ptr(tx), tx.shape[1], _ptr(du), int(use_mask),
                                            _ptr(y0), _ptr(tt), steps, {"euler": 0, "midpoint": 1}[ode_method], float(cfg_strength),
                                            PRECISIONS[precision], _ptr(out),
                                            _ptr(traj), self._stream()))
        return out, traj

    def debug_tensor(self, which: int, shape) -> torch.Tensor:
        out = torch.empty(shape, device=self.device, dtype=torch.float32)
        with torch.cuda.device(self.device):
            self._chk(self.lib.f5hip_debug_tensor(self._ctx, which, _ptr(out), out.numel(), self._stream()))
        return out

    def vocos_decode(self, mel: torch.Tensor, channel_major: bool = True) -> torch.Tensor:
        mel = mel.to(device=self.device, dtype=torch.float32).contiguous()
        b = mel.shape[0]
        frames = mel.shape[2] if channel_major else mel.shape[1]
        out = torch.empty((b, 256 * (frames - 1)), device=self.device, dtype=torch.float32)
        with torch.cuda.device(self.device):
            self._chk(self.lib.f5hip_vocos_decode(self._ctx, _ptr(mel), b, frames, int(channel_major), _ptr(out), self._stream()))
        return out

    def istft(self, logits: torch.Tensor) -> torch.Tensor:
        """The Vocos head's inverse STFT alone: logits [batch, frames, ld >= 1026] = (log-magnitude | phase) -> wave [batch, 256 * (frames - 1)]."""
        logits = logits.to(device=self.device, dtype=torch.float32).contiguous()
        b, frames, ld = logits.shape
        out = torch.empty((b, 256 * (frames - 1)), device=self.device, dtype=torch.float32)
        with torch.cuda.device(self.device):
            self._chk(self.lib.f5hip_istft(self._ctx, _ptr(logits), ld, b, frames, _ptr(out), self._stream()))
        return out


    def vocos_head(self, hidden: torch.Tensor) -> torch.Tensor:
        """The Vocos ISTFTHead alone: hidden [batch, frames, dim] (after final_layer_norm) -> wave [batch, 256 * (frames - 1)]."""
        hidden = hidden.to(device=self.device, dtype=torch.float32).contiguous()
        b, frames, _ = hidden.shape
        out = torch.empty((b, 256 * (frames - 1)), device=self.device, dtype=torch.float32)
        with torch.cuda.device(self.device):
            self._chk(self.lib.f5hip_vocos_head(self._ctx, _ptr(hidden), b, frames, _ptr(out), self._stream()))
        return out


def _as_tensor(ptr: int, numel: int, device: torch.device) -> torch.Tensor:
    """Wrap a raw device pointer as a float32 torch tensor (no copy) via __cuda_array_interface__."""

    class _Holder:
        pass

    h = _Holder()
    h.__cuda_array_interface__ = {"shape": (numel,), "typestr": "<f4", "data": (ptr, False), "version": 2}
    t = torch.as_tensor(h, device=device)
    if t.data_ptr() != ptr or t.device != device:  # a silent copy would make a broadcast fill the copy, not the engine's blob
        raise binding.F5HipError(f"could not alias device memory at {ptr:#x} on {device} (got {t.device}, {t.data_ptr():#x})")
    return t


# -------------------------------------------------------------------------------------------------
class _MelSpecAdapter:
    """``model.mel_spec`` (reference model/modules.py:112-151): callable + the attributes callers read."""

    def __init__(self, engine: F5HipEngine, mel_spec_type: str = "vocos"):
        self._e = engine
        self.n_fft, self.hop_length, self.win_length = 1024, 256, 1024
        self.n_mel_channels = engine.dit_cfg.mel_dim
        self.target_sample_rate = 24000
        self.mel_spec_type = mel_spec_type

    def __call__(self, wav: torch.Tensor) -> torch.Tensor:
        return self._e.mel(wav, frame_major=False, mel_spec_type=self.mel_spec_type)  # [b, 100, T] like MelSpec.forward


class _TransformerAdapter:
    """``model.transformer``: only ``.dim`` and ``.clear_cache()`` are touched by callers (cfm.py:66-68,219)."""

    def __init__(self, cfg: DiTConfig):
        self.dim = cfg.dim
        self.depth = cfg.depth

    def clear_cache(self):  # the text-embedding cache lives inside one f5hip_sample call
        return None


class F5HipCFM:
    """Drop-in for the ``CFM`` object on the inference path (reference src/f5_tts/model/cfm.py:34-229)."""

    def __init__(self, engine: F5HipEngine, vocab_char_map: Optional[Dict[str, int]] = None, ode_method: str = "euler",
                 precision: str = "fp32", mel_spec_type: str = "vocos"):
        if ode_method not in ("euler", "midpoint"):
            raise ValueError("only the fixed-grid euler / midpoint solvers are built (reference utils_infer.py:60, eval_infer_batch.py:47)")
        self.ode_method = ode_method
        self.engine = engine
        self.vocab_char_map = vocab_char_map
        self.precision = precision
        self.num_channels = engine.dit_cfg.mel_dim
        self.mel_spec = _MelSpecAdapter(engine, mel_spec_type)
        self.transformer = _TransformerAdapter(engine.dit_cfg)
        self.dim = engine.dit_cfg.dim

    @property
    def device(self):
        return self.engine.device

    def eval(self):
        return self

    def to(self, *a, **k):
        return self

    @torch.no_grad()
    def sample(self, cond, text, duration, *, lens=None, steps=32, cfg_strength=1.0, sway_sampling_coef=None, seed=None,
               max_duration=65536, vocoder: Optional[Callable] = None, use_epss=True, no_ref_audio=False,
               duplicate_test=False, t_inter=0.1, edit_mask=None):
        """Same contract as ``CFM.sample`` (cfm.py:83-229): returns ``(out, trajectory)``."""
        dev = self.device
        if cond.ndim == 2:  # raw wave -> mel (cfm.py:106-109)
            cond = self.engine.mel(cond, frame_major=True, mel_spec_type=self.mel_spec.mel_spec_type)
            assert cond.shape[-1] == self.num_channels
        cond = cond.to(device=dev, dtype=torch.float32)
        batch, cond_seq_len = cond.shape[:2]
        if lens is None:
            lens = torch.full((batch,), cond_seq_len, dtype=torch.long)
        lens = lens.to("cpu", torch.long)
        if isinstance(text, list):  # cfm.py:119-124
            text = list_str_to_idx(text, self.vocab_char_map) if self.vocab_char_map is not None else list_str_to_tensor(text)
            assert text.shape[0] == batch
        text = text.to("cpu", torch.long)
        cond_mask = lens_to_mask(lens)  # cfm.py:128-130
        if edit_mask is not None:
            cond_mask = cond_mask & edit_mask.to("cpu")
        if isinstance(duration, int):
            duration = torch.full((batch,), duration, dtype=torch.long)
        duration = duration.to("cpu", torch.long)
        duration = torch.maximum(torch.maximum((text != -1).sum(dim=-1), lens) + 1, duration)  # cfm.py:135-137
        duration = duration.clamp(max=max_duration)
        n = int(duration.amax())
        if duplicate_test:  # cfm.py:141-143: a copy of the prompt placed right behind it
            test_cond = torch.nn.functional.pad(cond, (0, 0, cond_seq_len, n - 2 * cond_seq_len), value=0.0)
        cond = torch.nn.functional.pad(cond, (0, 0, 0, n - cond_seq_len), value=0.0)  # cfm.py:145
        if no_ref_audio:
            cond = torch.zeros_like(cond)
        cond_mask = torch.nn.functional.pad(cond_mask, (0, n - cond_mask.shape[-1]), value=False)
        use_mask = batch > 1  # cfm.py:155-158
        # noise from torch's CPU generator, exactly as the reference draws it (cfm.py:196-201)
        # (seeded: a private generator re-seeded per utterance — the same stream torch.manual_seed(seed) would give the global one, without
        # touching state that other threads of infer_batch_process' pool draw from)
        y0 = []
        gen = torch.Generator(device="cpu") if seed is not None else None
        for dur in duration:
            if gen is not None:
                gen.manual_seed(seed)
            y0.append(torch.randn(int(dur), self.num_channels, dtype=torch.float32, generator=gen))
        y0 = torch.nn.utils.rnn.pad_sequence(y0, padding_value=0, batch_first=True)
        t_start = 0
        if duplicate_test:  # cfm.py:205-209: start the solve at t_inter from a noised copy of the prompt
            t_start = t_inter
            y0 = (1 - t_start) * y0.to(dev) + t_start * test_cond
            steps = int(steps * (1 - t_start))
        if t_start == 0 and use_epss:  # cfm.py:211-214
            t = get_epss_timesteps(steps)
        else:
            t = torch.linspace(t_start, 1, steps + 1, dtype=torch.float32)
        if sway_sampling_coef is not None:
            t = t + sway_sampling_coef * (torch.cos(torch.pi / 2 * t) - 1 + t)
        out, trajectory = self.engine.sample(cond, cond_mask, text, duration, use_mask, y0, t, cfg_strength,
                                             precision=self.precision, want_trajectory=True, ode_method=self.ode_method)
        if vocoder is not None:  # cfm.py:225-227
            out = vocoder(out.permute(0, 2, 1))
        return out, trajectory


class F5HipVocos:
    """Drop-in for the ``vocos.Vocos`` object on the inference path: ``decode(mel[b,100,T]) -> wav[b,nw]``."""

    def __init__(self, engine: F5HipEngine):
        if engine.vocos_cfg is None:
            raise ValueError("engine was created without a vocoder config")
        self.engine = engine

    def eval(self):
        return self

    def to(self, *a, **k):
        return self

    def decode(self, mel: torch.Tensor) -> torch.Tensor:
        return self.engine.vocos_decode(mel, channel_major=True)


# -------------------------------------------------------------------------------------------------
def map_checkpoint_keys(checkpoint: Dict[str, torch.Tensor], use_ema: bool = True) -> Dict[str, torch.Tensor]:
    """The key mapping of reference ``load_checkpoint`` (src/f5_tts/infer/utils_infer.py:209-227)."""
    if use_ema:
        if "ema_model_state_dict" in checkpoint:
            checkpoint = checkpoint["ema_model_state_dict"]
        sd = {k.replace("ema_model.", ""): v for k, v in checkpoint.items() if k not in ["initted", "step"]}
    else:
        sd = checkpoint.get("model_state_dict", checkpoint)
    for key in ["mel_spec.mel_stft.mel_scale.fb", "mel_spec.mel_stft.spectrogram.window"]:
        sd.pop(key, None)
    return sd


def filter_vocos_keys(sd: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    """Keep the backbone/head tensors of a ``charactr/vocos-mel-24khz`` ``pytorch_model.bin``."""
    return {k: v for k, v in sd.items() if k.startswith("backbone.") or k.startswith("head.")}
