"""ctypes binding of ``csrc/libf5hip.so`` (C ABI declared in ``include/f5hip.h``).

This is the stub a reference maintainer would add next to ``src/f5_tts/infer/utils_infer.py`` to
call the engine.  There is NO fallback: if the shared library is missing or
fails to load, importing the engine raises — the product path never routes through CPU code.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

_HERE = os.path.dirname(os.path.abspath(__file__))
# F5HIP_LIB=/path/to/libf5hip.so: another build of the same ABI (A/B measurements of two builds on one box; tools/gpu_run.sh)
LIB_PATH = os.environ.get("F5HIP_LIB") or os.path.join(_HERE, "csrc", "libf5hip.so")
BENCH_LIB_PATH = os.environ.get("F5HIP_BENCH_LIB") or os.path.join(_HERE, "csrc", "libf5hip_bench.so")

ABI_VERSION = 10  # F5HIP_ABI_VERSION in include/f5hip.h
PREC_FP32, PREC_FP16X3, PREC_FP16, PREC_FP16M = 0, 1, 2, 3
PRECISIONS = {"fp32": PREC_FP32, "fp16x3": PREC_FP16X3, "fp16": PREC_FP16, "fp16m": PREC_FP16M}


class DitConfigC(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        "dim", "depth", "heads", "dim_head", "ff_inner", "mel_dim", "text_num_embeds", "text_dim", "conv_layers",
        "text_mask_padding", "pe_attn_head", "attn_mask_enabled", "conv_pos_kernel", "conv_pos_groups", "backbone",
        "qk_norm", "long_skip_connection", "text_average_upsampling", "skip_connect_type")]


class VocosConfigC(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("input_channels", "dim", "intermediate_dim", "num_layers", "n_fft", "hop_length")]


class BigVGANConfigC(C.Structure):
    _fields_ = [("num_mels", C.c_int32), ("num_upsamples", C.c_int32), ("upsample_initial_channel", C.c_int32),
                ("upsample_rates", C.c_int32 * 8), ("upsample_kernel_sizes", C.c_int32 * 8), ("resblock", C.c_int32),
                ("num_kernels", C.c_int32), ("resblock_kernel_sizes", C.c_int32 * 4), ("resblock_num_dilations", C.c_int32 * 4),
                ("resblock_dilation_sizes", (C.c_int32 * 4) * 4), ("activation", C.c_int32), ("snake_logscale", C.c_int32),
                ("use_tanh_at_final", C.c_int32), ("use_bias_at_final", C.c_int32)]


# every symbol include/f5hip.h declares: name -> (restype, argtypes)
_P = C.c_void_p
SYMBOLS = {
    "f5hip_abi_version": (C.c_int, []),
    "f5hip_create": (C.c_int, [C.POINTER(DitConfigC), C.POINTER(VocosConfigC), C.c_int, C.POINTER(_P)]),
    "f5hip_destroy": (C.c_int, [_P]),
    "f5hip_last_error": (C.c_char_p, [_P]),
    "f5hip_load_tensor": (C.c_int, [_P, C.c_char_p, _P, C.c_int64]),
    "f5hip_num_tensors": (C.c_int, [_P]),
    "f5hip_tensor_info": (C.c_int, [_P, C.c_int, C.POINTER(C.c_char_p), C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "f5hip_weight_blob": (C.c_int, [_P, C.POINTER(_P), C.POINTER(C.c_int64)]),
    "f5hip_mark_all_loaded": (C.c_int, [_P]),
    "f5hip_loaded_mask": (C.c_int, [_P, _P, C.c_int]),
    "f5hip_set_loaded_mask": (C.c_int, [_P, _P, C.c_int]),
    "f5hip_finalize_weights": (C.c_int, [_P]),
    "f5hip_mel": (C.c_int, [_P, _P, C.c_int, C.c_int64, _P, C.c_int, C.c_int, _P]),
    "f5hip_sample": (C.c_int, [_P, C.c_int, C.c_int, _P, _P, _P, C.c_int, _P, C.c_int, _P, _P, C.c_int, C.c_int, C.c_float,
                               C.c_int, _P, _P, _P]),
    "f5hip_debug_tensor": (C.c_int, [_P, C.c_int, _P, C.c_int64, _P]),
    "f5hip_vocos_decode": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, _P, _P]),
    "f5hip_istft": (C.c_int, [_P, _P, C.c_int64, C.c_int, C.c_int, _P, _P]),
    "f5hip_vocos_head": (C.c_int, [_P, _P, C.c_int, C.c_int, _P, _P]),
    "f5hip_set_option": (C.c_int, [_P, C.c_char_p, C.c_int64]),
    "f5hip_num_kernel_stats": (C.c_int, [_P]),
    "f5hip_kernel_stat": (C.c_int, [_P, C.c_int, C.POINTER(C.c_char_p), C.POINTER(C.c_int64), C.POINTER(C.c_double),
                                    C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "f5hip_reset_kernel_stats": (C.c_int, [_P]),
    "f5hip_attention_stats": (C.c_int, [_P, C.POINTER(C.c_double), C.c_int]),
    "f5hip_bigvgan_create": (C.c_int, [C.POINTER(BigVGANConfigC), C.c_int, C.POINTER(_P)]),
    "f5hip_bigvgan_destroy": (C.c_int, [_P]),
    "f5hip_bigvgan_last_error": (C.c_char_p, [_P]),
    "f5hip_bigvgan_num_tensors": (C.c_int, [_P]),
    "f5hip_bigvgan_tensor_info": (C.c_int, [_P, C.c_int, C.POINTER(C.c_char_p), C.POINTER(C.c_int64)]),
    "f5hip_bigvgan_load_tensor": (C.c_int, [_P, C.c_char_p, _P, C.c_int64]),
    "f5hip_bigvgan_finalize": (C.c_int, [_P]),
    "f5hip_bigvgan_forward": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, C.c_int, _P, _P]),
    "f5hip_bigvgan_set_option": (C.c_int, [_P, C.c_char_p, C.c_int64]),
    "f5hip_bigvgan_num_kernel_stats": (C.c_int, [_P]),
    "f5hip_bigvgan_kernel_stat": (C.c_int, [_P, C.c_int, C.POINTER(C.c_char_p), C.POINTER(C.c_int64), C.POINTER(C.c_double),
                                            C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "f5hip_bigvgan_reset_kernel_stats": (C.c_int, [_P]),
}

# include/f5hip_bench.h (libf5hip_bench.so: microbenchmarks, format checks, fault reproducers — tools and tests only)
BENCH_SYMBOLS = {
    "f5hip_bench_gemm": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double)]),
    "f5hip_bench_mx_pack": (C.c_int, [_P, C.c_int, C.c_int, C.POINTER(C.c_double)]),
    "f5hip_bench_qkv": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_int64)]),
    "f5hip_bench_attention": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double)]),
    "f5hip_bench_qkv_probe": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int64), C.c_char_p]),
}

_lib: Optional[C.CDLL] = None
_bench_lib: Optional[C.CDLL] = None


class F5HipError(RuntimeError):
    pass


def load_library(path: str = LIB_PATH) -> C.CDLL:
    """dlopen libf5hip.so and type every entry point.  Raises if the library or a symbol is missing."""
    global _lib
    if _lib is not None and path == LIB_PATH:
        return _lib
    if not os.path.isfile(path):
        raise F5HipError(f"{path} not found — build it with `python __graft_entry__.py` (or `make -C f5-tts_amd/csrc`); "
                         "there is no CPU fallback")
    lib = C.CDLL(path, mode=C.RTLD_GLOBAL)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)  # AttributeError if the .so does not export what the header declares
        fn.restype = res
        fn.argtypes = args
    if lib.f5hip_abi_version() != ABI_VERSION:
        raise F5HipError("libf5hip ABI version mismatch")
    if path == LIB_PATH:
        _lib = lib
    return lib


def type_symbols(lib: C.CDLL, symbols) -> C.CDLL:
    for name, (res, args) in symbols.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    return lib


def load_bench_library(path: str = None) -> C.CDLL:
    """dlopen libf5hip_bench.so (after libf5hip.so, whose internal launchers it calls) and type the f5hip_bench_* entry points."""
    global _bench_lib
    if _bench_lib is not None and path is None:
        return _bench_lib
    load_library()
    p = path or BENCH_LIB_PATH
    if not os.path.isfile(p):
        raise F5HipError(f"{p} not found — build it with `make -C f5-tts_amd/csrc` (tools and tests only; the product does not need it)")
    lib = type_symbols(C.CDLL(p, mode=C.RTLD_GLOBAL), BENCH_SYMBOLS)
    if path is None:
        _bench_lib = lib
    return lib


def check(lib: C.CDLL, ctx, status: int, last_error=None) -> None:
    if status == 0:
        return
    msg = (last_error or lib.f5hip_last_error)(ctx)
    text = msg.decode("utf-8", "replace") if msg else "unknown error"
    # mirror the reference's error behaviour: bad shapes/arguments are ValueError (asserts in cfm.py:109,124),
    # everything else RuntimeError
    if status in (1, 4):
        raise ValueError(f"f5hip: {text}")
    raise F5HipError(f"f5hip (status {status}): {text}")
