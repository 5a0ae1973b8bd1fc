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
        st = eng.bench_lib.f5hip_bench_gemm(eng._ctx, binding.PRECISIONS[prec], variant, epi, M, N, K, 1, C.byref(ms))
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


# ---- the ping-pong kernel of the many-round launches (csrc/gemm_p8.h, variant id 80) ---------------------------------------------------------
# Two groups of four waves half a phase apart, the LDS refilled by quarters two phases behind the reads: on the shim (LDS-DMA synchronous,
# fibers switched at barriers) a quarter refilled a phase too early, a fragment read from the wrong buffer or a quadrant paired with the
# wrong weight tile all show as differing bytes.  K = 256 .. 640: two pairs of k-tiles (one steady body + the last-pair body), three, and five pairs.
@pytest.mark.parametrize("M,N,K", [(300, 288, 256), (300, 288, 384), (520, 544, 640)])
@pytest.mark.parametrize("epi", [0, 1, 2])
def test_p8_equals_generic_kernel_fp16(emu_engine, capfd, M, N, K, epi):  # noqa: F811
    for bad, nb, d, v in run(emu_engine, capfd, "fp16", 80, epi, M, N, K):
        assert bad == 0 and v > 0, (bad, nb, d, v)


@pytest.mark.parametrize("M,N,K", [(300, 288, 128), (300, 288, 192), (520, 544, 320)])
def test_p8_mx_lines_product_close_to_the_three_term_product(emu_engine, capfd, M, N, K):  # noqa: F811
    err = run_mx(emu_engine, capfd, 80, 2, M, N, K)
    m = re.search(r"KB_CHECK fp16m variant \d+: max \|diff\| (\S+) mean (\S+) of max \|value\| (\S+)", err)
    assert m, err
    d, mean, v = (float(x) for x in m.groups())
    assert v > 0 and 0 < d <= 1e-4 * v and mean <= 1e-5 * v, err


@pytest.mark.parametrize("epi", [0, 1])
def test_p8_mx_output_rows_decode_to_the_fp16x3_rows(emu_engine, capfd, epi):  # noqa: F811
    M, N = 300, 288
    err = run_mx(emu_engine, capfd, 80, epi, M, N, 256)
    m = re.search(r"KB_CHECK fp16m variant \d+ epi \d: (\d+) lines, (\d+) hi halves differ .* errors (\S+) / (\S+) block steps, (\d+) out of bounds", err)
    assert m, err
    lines, hidiff, ec, el, bad = int(m.group(1)), int(m.group(2)), float(m.group(3)), float(m.group(4)), int(m.group(5))
    assert lines == M * N // 32 and bad == 0 and hidiff <= lines * 32 // 8 and 0 < ec <= 0.51 and 0 < el <= 1.5, err


@pytest.mark.parametrize("prec,K", [("fp16", 256), ("fp16m", 128)])
def test_p8_qkv_epilogue(emu_engine, capfd, prec, K):  # noqa: F811
    diff, err = run_qkv(emu_engine, capfd, prec, 80, 3, 150, K=K)
    assert diff == 0, err[-2000:]


# ---- fp16m: 2 fp16 MFMAs + 1 MX-fp6 MFMA per 32 k (common.h) ---------------------------------------------------------------------------
# The correction terms are rounded to ~4 bits per factor, so the results are close to — not bytes of — the three-term product: the bound
# is the scheme's own error (2^-16 per operand relative to the block maximum, summed over K), the check that it is non-zero shows the MX
# instruction ran.  Every tile instantiated for MX lines (gemm.hip F5_MX_TILES), ragged shapes, the k-split tile with >= 3 k-tiles per group.
MX_CASES = [(50, 300, 288), (54, 250, 160), (55, 250, 160), (56, 250, 224), (59, 130, 160), (61, 200, 224), (62, 250, 160), (63, 130, 160), (66, 130, 160),
            (68, 250, 224), (69, 250, 160)]  # 68 / 69: the k-step split (the groups take the fp6 correction of alternate k-tiles)


def run_mx(make_engine, capfd, variant, epi, M, N, K):
    from f5_tts_amd import binding, config

    eng = make_engine(config.DIT_TINY)
    os.environ["KB_CHECK"] = "1"
    try:
        ms = C.c_double()
        st = eng.bench_lib.f5hip_bench_gemm(eng._ctx, binding.PRECISIONS["fp16m"], variant, epi, M, N, K, 1, C.byref(ms))
    finally:
        os.environ.pop("KB_CHECK", None)
    err = capfd.readouterr().err
    assert st == 0, err
    return err


@pytest.mark.parametrize("variant,M,N", MX_CASES)
def test_pp_mx_lines_product_close_to_the_three_term_product(emu_engine, capfd, variant, M, N):  # noqa: F811
    err = run_mx(emu_engine, capfd, variant, 2, M, N, 256)
    m = re.search(r"KB_CHECK fp16m variant \d+: max \|diff\| (\S+) mean (\S+) of max \|value\| (\S+)", err)
    assert m, err
    d, mean, v = (float(x) for x in m.groups())
    assert v > 0 and 0 < d <= 1e-4 * v and mean <= 1e-5 * v, err


@pytest.mark.parametrize("variant,M,N", [(50, 300, 288), (56, 250, 224), (66, 130, 160)])
@pytest.mark.parametrize("epi", [0, 1])
def test_pp_mx_output_rows_decode_to_the_fp16x3_rows(emu_engine, capfd, variant, M, N, epi):  # noqa: F811
    """The FF1-type epilogue writes the NEXT GEMM's MX lines: hi halves equal the fp16x3 rows', coarse values and remainders within their steps."""
    err = run_mx(emu_engine, capfd, variant, epi, M, N, 256)
    m = re.search(r"KB_CHECK fp16m variant \d+ epi \d: (\d+) lines, (\d+) hi halves differ .* errors (\S+) / (\S+) block steps, (\d+) out of bounds", err)
    assert m, err
    lines, hidiff, ec, el, bad = int(m.group(1)), int(m.group(2)), float(m.group(3)), float(m.group(4)), int(m.group(5))
    # (hi halves: the two GEMMs' results differ by a few 1e-5 relative, which crosses an fp16 rounding boundary in a few % of the values)
    assert lines == M * N // 32 and bad == 0 and hidiff <= lines * 32 // 8 and 0 < ec <= 0.51 and 0 < el <= 1.5, err


@pytest.mark.parametrize("rows,K", [(37, 256), (5, 1024), (9, 768), (3, 2048)])
def test_mx_pack_and_layernorm_rows_against_the_format(emu_engine, rows, K):  # noqa: F811
    """pack_mx_rows_kernel (both operand sides) and the LayerNorm producer: every line decoded on the host against the fp32 values, with the
    format's own bounds (coarse value within half a block step, remainder within a quarter step, hi = fp16(value), zero padding words)."""
    from f5_tts_amd import config

    eng = emu_engine(config.DIT_TINY)
    out = (C.c_double * 9)()
    assert eng.bench_lib.f5hip_bench_mx_pack(eng._ctx, rows, K, out) == 0
    for which in range(3):
        wc, wl, bad = out[3 * which], out[3 * which + 1], out[3 * which + 2]
        assert bad == 0 and 0.2 < wc <= 0.5001 and 0.1 < wl <= 0.27, (which, wc, wl, bad)  # (remainders stay below 4 block steps: their top binade rounds to 0.125)


def run_qkv(make_engine, capfd, prec, variant, seqs, nseq, K=128, check=1):
    from f5_tts_amd import binding, config

    eng = make_engine(config.DIT_TINY)
    ms, diff = C.c_double(), C.c_int64()
    st = eng.bench_lib.f5hip_bench_qkv(eng._ctx, binding.PRECISIONS[prec], variant, seqs, nseq, K, 1, check, C.byref(ms), C.byref(diff))
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


@pytest.mark.parametrize("variant", [c[0] for c in MX_CASES])
def test_pp_qkv_epilogue_on_mx_lines(emu_engine, capfd, variant):  # noqa: F811
    """The fused q|k|v epilogue behind the MX k-loop: every q / k / V^T value within 1e-4 of the fp16x3 generic kernel's (another product
    form, so values — the index scheme itself is the byte-for-byte test above)."""
    diff, err = run_qkv(emu_engine, capfd, "fp16m", variant, 3, 150, K=256)
    assert diff == 0, err


@pytest.mark.parametrize("prec,variant", [("fp16x3", CASES[0][0]), ("fp16x3", CASES[-1][0]), ("fp16m", MX_CASES[0][0]), ("fp16m", 80)])
def test_qkv_epilogue_packs_the_score_corrections_and_the_flash_kernel_reads_them(emu_engine, capfd, prec, variant):  # noqa: F811
    """Round 5's attention default: the q|k|v epilogue leaves MX-fp6 P words in the second planes of q and k (EpiQKV::mx_qk) and the flash
    kernel adds both correction products of the scores as one MX MFMA per 32 head channels.  The P words are decoded against the generic
    kernel's hi + lo values with the format's bounds, and at logits of tens the output must follow the split-q,k kernel's where plain fp16
    scores visibly do not."""
    diff, err = run_qkv(emu_engine, capfd, prec, variant, 2, 150, K=256, check=2)
    assert diff == 0, err[-3000:]
    m = re.search(r"attention on the MX planes against split q, k: max \|diff\| (\S+) mean (\S+); plain fp16 scores: max (\S+) mean (\S+);", err)
    assert m, err[-3000:]
    print(err.strip().splitlines()[-3:])
    assert float(m.group(2)) < 0.35 * float(m.group(4)) and float(m.group(4)) > 0, m.group(0)


def test_tile_grouping_covers_every_tile_with_a_ragged_last_group():
    """GemmCore::group_m (row tiles per group, row tile fastest inside a group: gemm.hip default_group_m picks 5 for the wide one-round
    launches since round 4, 4 from 8192 rows) is read from F5HIP_GEMM_GROUPM once per process, so a child process runs a few of the
    byte-for-byte cases above under group sizes that do NOT divide the row-tile count (M = 250 / 300 rows: 3 row tiles of 96, 2 of 192 or
    256): every tile must still be computed exactly once — a tile skipped or done twice by the index arithmetic of gemm_pp.h / gemm.h
    shows as differing bytes against the generic kernel, which these runs group the same way but walk with its own 128x64 tiles."""
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for gm in ("2", "5"):
        r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", os.path.join(root, "tests", "test_pp_gemm_shim.py"), "-k",
                            "(test_pp_variant_equals_generic_kernel_fp16x3 and (59-130 or 55-250 or 50-300)) or (test_pp_mx_lines_product and 59)"],
                           capture_output=True, text=True, env=dict(os.environ, F5HIP_GEMM_GROUPM=gm), cwd=root, timeout=1800)
        assert r.returncode == 0 and " passed" in r.stdout and "no tests ran" not in r.stdout, (gm, r.stdout[-1500:], r.stderr[-500:])
