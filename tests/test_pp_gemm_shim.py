"""The pipelined block GEMM (csrc/gemm_pp.h) on the CPU shim: every tile variant and wave-tile epilogue against the generic 128x64 kernel
of gemm.h (GPU-verified in round 1), byte for byte — same MFMA order per accumulator, same epilogue arithmetic.  What the shim can
and cannot show is stated in tests/hipemu/README.md: it catches index / layout / hazard-by-ordering errors (a stage refilled while another
wave still reads it), not timing or an LDS-DMA that is read before it landed (the counted waits are argued in gemm_pp.h)."""
import ctypes as C
import os
import re

import pytest

from tests.test_hipemu import CLANG, emu_engine, engine_emu_lib  # noqa: F401  (fixtures)

pytestmark = pytest.mark.skipif(not os.path.exists(CLANG), reason="needs the ROCm host clang++")

PREC = {"fp16x3": 1, "fp16": 2}


def run(make_engine, capfd, prec, variant, epi, M, N, K):
    from f5_tts_amd import binding, config

    eng = make_engine(config.DIT_TINY)
    os.environ["KB_CHECK"] = "1"
    try:
        ms = C.c_double()
        st = eng.lib.f5hip_bench_gemm(eng._ctx, binding.PRECISIONS[prec], variant, epi, M, N, K, 1, C.byref(ms))
    finally:
        os.environ.pop("KB_CHECK", None)
    err = capfd.readouterr().err
    assert st == 0, err
    lines = re.findall(r"KB_CHECK variant (\d+) epi (\d+) rep (\d+): (\d+) of (\d+) bytes differ.*max \|diff\| (\S+) of max \|value\| (\S+)", err)
    assert len(lines) == 3, err
    return [(int(bad), int(nb), float(d), float(v)) for _, _, _, bad, nb, d, v in lines]


# (variant, M, N): ragged rows everywhere; N a multiple of 32 that is NOT a multiple of the tile width where the tile allows it.
# The default run covers the tiles pick_pp_variant() can choose plus one of every other family; F5HIP_SHIM_ALL_VARIANTS=1 runs them all
# (the microbenchmark-only tiles too: 21 variants, +2 minutes).
ALL_CASES = [(50, 300, 288), (51, 300, 160), (52, 200, 288), (53, 250, 96), (54, 250, 160), (55, 250, 160), (56, 250, 224), (57, 200, 160),
             (58, 200, 160), (59, 130, 160), (60, 300, 160), (61, 200, 224), (62, 250, 160), (63, 130, 160), (64, 250, 96),
             (65, 250, 160), (66, 130, 160), (67, 250, 96), (68, 250, 224), (69, 250, 160), (70, 250, 96)]
PRODUCTION = (50, 51, 55, 56, 59, 66, 68, 69)  # + 52 (128x256), 62 (two per CU), 65 (k-split of a wide tile) as family representatives
CASES = ALL_CASES if os.environ.get("F5HIP_SHIM_ALL_VARIANTS") else [c for c in ALL_CASES if c[0] in PRODUCTION + (52, 62, 65)]


KSPLIT = (65, 66, 67, 68, 69, 70)  # the k-split and k-step-split tiles


@pytest.mark.parametrize("variant,M,N", CASES)
@pytest.mark.parametrize("epi", [1, 2])
def test_pp_variant_equals_generic_kernel_fp16x3(emu_engine, capfd, variant, M, N, epi):  # noqa: F811
    if variant in KSPLIT:  # two partial sums per output (another fp32 summation order): values, not bytes; >= 3 k-tiles per group
        for bad, nb, d, v in run(emu_engine, capfd, "fp16x3", variant, epi, M, N, 256):
            assert v > 0 and d <= 4e-6 * v, (bad, nb, d, v)
        return
    for bad, nb, d, v in run(emu_engine, capfd, "fp16x3", variant, epi, M, N, 128):
        assert bad == 0 and v > 0, (bad, nb, d, v)


@pytest.mark.parametrize("variant,M,N", [(50, 300, 288), (53, 250, 96), (56, 250, 224)])
@pytest.mark.parametrize("epi", [0, 2])
def test_pp_variant_equals_generic_kernel_fp16(emu_engine, capfd, variant, M, N, epi):  # noqa: F811
    for bad, nb, d, v in run(emu_engine, capfd, "fp16", variant, epi, M, N, 320):
        assert bad == 0 and v > 0, (bad, nb, d, v)


def run_qkv(make_engine, capfd, prec, variant, seqs, nseq, K=128):
    from f5_tts_amd import binding, config

    eng = make_engine(config.DIT_TINY)
    ms, diff = C.c_double(), C.c_int64()
    st = eng.lib.f5hip_bench_qkv(eng._ctx, binding.PRECISIONS[prec], variant, seqs, nseq, K, 1, 1, C.byref(ms), C.byref(diff))
    err = capfd.readouterr().err
    assert st == 0, err
    return diff.value, err


# the fused q|k|v epilogue (rope, head scatter, V^T) of every tile variant against the generic kernel's, byte for byte: sequences that
# straddle row tiles, even (paired V^T stores) and odd (scalar stores) sequence lengths, several sequences per launch
@pytest.mark.parametrize("variant", [c[0] for c in CASES])
@pytest.mark.parametrize("seqs,nseq", [(3, 150), (2, 141)])
def test_pp_qkv_epilogue_equals_generic_kernel(emu_engine, capfd, variant, seqs, nseq):  # noqa: F811
    diff, err = run_qkv(emu_engine, capfd, "fp16x3", variant, seqs, nseq, K=256 if variant in KSPLIT else 128)
    assert diff == 0, err[-2000:]
