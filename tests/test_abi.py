"""CPU tests of the drop-in boundary: the C-ABI library loads and exports every symbol
include/f5hip.h declares; the binding types them all; no compute happens without a GPU and the
product path fails loudly (no CPU fallback)."""
import os
import re
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from f5_tts_amd import binding, config  # noqa: E402

HEADER = os.path.join(ROOT, "include", "f5hip.h")


def header_symbols():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(f5hip_[a-z_0-9]+)\s*\(", src)))


def test_header_and_binding_agree():
    assert header_symbols() == sorted(binding.SYMBOLS)


def test_library_exports_every_declared_symbol():
    assert os.path.isfile(binding.LIB_PATH), "build first: python __graft_entry__.py"
    lib = binding.load_library()
    for name in header_symbols():
        assert hasattr(lib, name)
    assert lib.f5hip_abi_version() == binding.ABI_VERSION == 10
    out = subprocess.run(["nm", "-D", "--defined-only", binding.LIB_PATH], capture_output=True, text=True).stdout
    exported = set(re.findall(r"\bT (f5hip_[a-z_0-9]+)", out))
    assert set(header_symbols()) <= exported
    # the microbenchmarks, format checks and the fault reproducer are a library of their own (include/f5hip_bench.h): not in the product
    assert not [s for s in exported if s.startswith("f5hip_bench_")]


def test_bench_library_exports_its_header():
    bench_header = os.path.join(ROOT, "include", "f5hip_bench.h")
    src = re.sub(r"/\*.*?\*/", "", open(bench_header).read(), flags=re.S)
    declared = sorted(set(re.findall(r"\b(f5hip_bench_[a-z_0-9]+)\s*\(", src)))
    assert declared == sorted(binding.BENCH_SYMBOLS)
    assert os.path.isfile(binding.BENCH_LIB_PATH), "build first: python __graft_entry__.py"
    out = subprocess.run(["nm", "-D", "--defined-only", binding.BENCH_LIB_PATH], capture_output=True, text=True).stdout
    assert set(declared) <= set(re.findall(r"\bT (f5hip_[a-z_0-9]+)", out))
    lib = binding.load_bench_library()
    for name in declared:
        assert hasattr(lib, name)


def test_header_cites_reference_interfaces():
    src = open(HEADER).read()
    for cite in ("utils_infer.py:238-276", "cfm.py:128-223", "modules.py:80-109", "utils_infer.py:510-511", "utils_infer.py:190-232"):
        assert cite in src


def test_missing_library_fails_loudly(tmp_path):
    with pytest.raises(binding.F5HipError):
        binding.load_library(str(tmp_path / "nope.so"))


@pytest.mark.skipif(torch.cuda.is_available(), reason="only meaningful without a GPU")
def test_no_cpu_fallback():
    from f5_tts_amd.engine import F5HipEngine

    with pytest.raises((binding.F5HipError, RuntimeError, AssertionError, ValueError)):
        F5HipEngine(config.DIT_TINY, None, device="cpu")
    lib = binding.load_library()
    import ctypes as C

    c = binding.DitConfigC(dim=256, depth=2, heads=4, dim_head=64, ff_inner=512, mel_dim=100, text_num_embeds=255, text_dim=128,
                           conv_layers=2, text_mask_padding=1, pe_attn_head=-1, attn_mask_enabled=0, conv_pos_kernel=31, conv_pos_groups=16)
    ctx = C.c_void_p()
    st = lib.f5hip_create(C.byref(c), None, 0, C.byref(ctx))
    assert st != 0 and not ctx.value
    assert b"CPU fallback" in lib.f5hip_last_error(None) or b"device" in lib.f5hip_last_error(None)


def test_create_rejects_bad_config():
    lib = binding.load_library()
    import ctypes as C

    c = binding.DitConfigC(dim=250, depth=2, heads=4, dim_head=64, ff_inner=512, mel_dim=100, text_num_embeds=255, text_dim=128,
                           conv_layers=2, text_mask_padding=1, pe_attn_head=-1, attn_mask_enabled=0, conv_pos_kernel=31, conv_pos_groups=16)
    ctx = C.c_void_p()
    assert lib.f5hip_create(C.byref(c), None, 0, C.byref(ctx)) == 1  # F5HIP_ERR_INVALID before touching any device


def test_config_structs_match_the_header():
    """Field order of the ctypes structs == field order of the C structs in include/f5hip.h (all int32)."""
    src = open(HEADER).read()
    for cname, pyc in (("f5hip_dit_config", binding.DitConfigC), ("f5hip_vocos_config", binding.VocosConfigC)):
        body = re.search(r"typedef struct " + cname + r" \{(.*?)\} " + cname + ";", src, re.S).group(1)
        body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
        fields = []
        for decl in re.findall(r"int32_t\s+([^;]+);", body):
            fields += [f.strip() for f in decl.split(",")]
        assert fields == [n for n, _ in pyc._fields_], cname


@pytest.mark.parametrize("bad", [dict(qk_norm=2), dict(text_average_upsampling=1, text_mask_padding=0), dict(skip_connect_type=1),
                                 dict(backbone=1, long_skip_connection=1, conv_layers=0), dict(backbone=2, conv_layers=0, text_dim=128),
                                 dict(backbone=2, conv_layers=0, text_dim=256, pe_attn_head=1), dict(backbone=3), dict(skip_connect_type=3, backbone=1, conv_layers=0)])
def test_create_rejects_bad_switches(bad):
    """The constructor-switch combinations the reference itself refuses (dit.py:43, modules.py:409) or that do not exist."""
    lib = binding.load_library()
    import ctypes as C

    kw = dict(dim=256, depth=2, heads=4, dim_head=64, ff_inner=512, mel_dim=100, text_num_embeds=255, text_dim=128, conv_layers=2,
              text_mask_padding=1, pe_attn_head=-1, attn_mask_enabled=0, conv_pos_kernel=31, conv_pos_groups=16)
    kw.update(bad)
    ctx = C.c_void_p()
    assert lib.f5hip_create(C.byref(binding.DitConfigC(**kw)), None, 0, C.byref(ctx)) == 1
    assert lib.f5hip_last_error(None)


def test_graft_entry_build_check_runs():
    """__graft_entry__.build() is what the driver runs on the CPU box every round: it must pass against the in-tree library."""
    import importlib

    g = importlib.import_module("__graft_entry__")
    g.build()


def build_c_smoke():
    """gcc (plain C11, no hipcc, no torch) over tests/c_abi/smoke.c against include/f5hip.h + libf5hip.so + the HIP runtime."""
    src = os.path.join(ROOT, "tests", "c_abi", "smoke.c")
    out_dir = os.path.join(ROOT, "tests", "c_abi", "_build")
    os.makedirs(out_dir, exist_ok=True)
    exe = os.path.join(out_dir, "smoke")
    lib_dir = os.path.dirname(binding.LIB_PATH)
    cmd = ["gcc", "-std=c11", "-O1", "-D__HIP_PLATFORM_AMD__", "-I", os.path.join(ROOT, "include"), "-I", "/opt/rocm/include", src,
           "-L", lib_dir, "-lf5hip", "-L", "/opt/rocm/lib", "-lamdhip64", f"-Wl,-rpath,{lib_dir}", "-Wl,-rpath,/opt/rocm/lib", "-lm", "-o", exe]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    return exe


@pytest.mark.skipif(torch.cuda.is_available(), reason="the GPU suite runs the program for real")
def test_c_program_builds_against_the_header_and_fails_loudly_without_a_gpu():
    r = subprocess.run([build_c_smoke()], capture_output=True, text=True)
    assert r.returncode == 77 and "no CPU fallback" in r.stderr  # f5hip_create -> F5HIP_ERR_HIP


@pytest.mark.gpu
def test_c_program_runs_the_path_through_the_c_abi():
    """mel -> sample (fp32 / fp16x3 / fp16) -> vocos_decode from plain C: finite, deterministic, prompt restored bit for bit."""
    r = subprocess.run([build_c_smoke()], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "c_abi smoke ok" in r.stdout, r.stderr


def test_bigvgan_config_struct_layout_matches_the_header():
    """f5hip_bigvgan_config holds arrays: compare every field's offset / size as gcc lays the C struct out with the ctypes mirror."""
    fields = [n for n, _ in binding.BigVGANConfigC._fields_]
    src = '#include <stdio.h>\n#include <stddef.h>\n#include "f5hip.h"\nint main(void) {\n'
    for f in fields:
        src += f'  printf("{f} %zu %zu\\n", offsetof(f5hip_bigvgan_config, {f}), sizeof(((f5hip_bigvgan_config*)0)->{f}));\n'
    src += '  printf("total %zu 0\\n", sizeof(f5hip_bigvgan_config));\n  return 0;\n}\n'
    out_dir = os.path.join(ROOT, "tests", "c_abi", "_build")
    os.makedirs(out_dir, exist_ok=True)
    cfile, exe = os.path.join(out_dir, "bv_layout.c"), os.path.join(out_dir, "bv_layout")
    open(cfile, "w").write(src)
    r = subprocess.run(["gcc", "-std=c11", "-I", os.path.join(ROOT, "include"), cfile, "-o", exe], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    got = {ln.split()[0]: (int(ln.split()[1]), int(ln.split()[2])) for ln in subprocess.run([exe], capture_output=True, text=True).stdout.splitlines()}
    import ctypes as C

    for f in fields:
        d = getattr(binding.BigVGANConfigC, f)
        assert got[f] == (d.offset, d.size), f
    assert got["total"][0] == C.sizeof(binding.BigVGANConfigC)
    # every C field is mirrored (the header's declaration order)
    body = re.search(r"typedef struct f5hip_bigvgan_config \{(.*?)\} f5hip_bigvgan_config;", open(HEADER).read(), re.S).group(1)
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    names = []
    for decl in re.findall(r"int32_t\s+([^;]+);", body):
        names += [re.sub(r"\[.*", "", f.strip()) for f in decl.split(",")]
    assert names == fields
