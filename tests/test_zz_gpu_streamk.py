"""GPU tests of the stream-K GEMM with the reduce-scattered epilogue (csrc/gemm_skrs.h, GEMM variants 42 / 43, engine option
"gemm_streamk") — default OFF in the product; nothing here touches the default path.

FIRST LIGHT: written without GPU minutes; the kernel source has been executed on the CPU through tests/hipemu (tests/test_hipemu.py:
results, self-cleaning flags, run-to-run determinism) but never on an MI355X, and it synchronises workgroups through global memory
(bounded spins).  Same isolation as tests/test_zz_gpu_bigvgan.py: the direct tests run only with F5HIP_STREAMK_GPU=1;
`test_first_light_in_a_subprocess` runs them in a child with a time limit and turns anything but a green child into an xfail."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import f5_tts_amd  # noqa: E402,F401
from f5_tts_amd import synth  # noqa: E402

pytestmark = [pytest.mark.gpu]
direct = pytest.mark.skipif(os.environ.get("F5HIP_STREAMK_GPU") != "1",
                            reason="run by test_first_light_in_a_subprocess (or directly with F5HIP_STREAMK_GPU=1)")


@pytest.mark.skipif(os.environ.get("F5HIP_STREAMK_GPU") == "1", reason="this IS the child / a direct run")
def test_first_light_in_a_subprocess():
    env = dict(os.environ, F5HIP_STREAMK_GPU="1")
    try:
        r = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-x", "-q", "-m", "gpu", "-p", "no:cacheprovider"], env=env, cwd=ROOT,
                           capture_output=True, text=True, timeout=900)
    except subprocess.TimeoutExpired as e:  # pragma: no cover
        pytest.xfail(f"stream-K first light: child timed out: {str(e.stdout)[-1500:]}")
    tail = (r.stdout + r.stderr)[-2500:]
    try:
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        open(os.path.join(ROOT, "gpurun_out", "streamk_first_light.log"), "w").write(r.stdout + r.stderr)
    except OSError:
        pass
    if r.returncode != 0:
        pytest.xfail(f"stream-K first light did not pass (exit {r.returncode}): {tail}")
    assert " passed" in r.stdout and "failed" not in r.stdout, tail


@direct
@pytest.mark.parametrize("variant", [42, 43])
@pytest.mark.parametrize("shape", [(2816, 1024, 1024), (2816, 1024, 2048), (2816, 2048, 1024), (2816, 3072, 1024), (1406, 1024, 1024)])
def test_microbenchmark_check_against_the_plain_tiling(variant, shape):
    """f5hip_bench_gemm with KB_CHECK: output of the variant against variant 1 on the same operands — numeric distance, the spin
    time-out word and the flags left set are printed by the library and parsed here (child process: it writes to stderr)."""
    M, N, K = shape
    env = dict(os.environ, KB_CHECK="1", KB_EPI="1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "kernel_bench.py"), "one", "fp16x3", str(variant), str(M), str(N), str(K), "3"],
                       env=env, cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "st=0" in r.stdout, r.stdout + r.stderr
    lines = [ln for ln in r.stderr.splitlines() if ln.startswith("KB_CHECK")]
    assert len(lines) == 3, r.stderr
    for ln in lines:  # "... max |diff| D of max |value| V, sk err word E, flags left set F"
        d = float(ln.split("max |diff| ")[1].split(" ")[0])
        v = float(ln.split("max |value| ")[1].split(",")[0])
        assert d <= 2e-3 * v, ln  # fp16 operand planes of the next GEMM: one ulp of fp16 at the largest value
        assert "sk err word 0" in ln and ln.rstrip().endswith("flags left set 0"), ln


@direct
def test_full_size_model_with_streamk_block_gemms():
    """F5-TTS Base at full size, packed schedule, every DiT block GEMM through gemm_skrs.h: against the reference-minted golden, against
    the default path, and twice in a row (self-cleaning workspace, graph replay)."""
    from f5_tts_amd.engine import F5HipCFM, F5HipEngine
    from oracle import make_golden as MG

    c = MG.FULL_CASES["base_v1_cfg1"]
    cfg, wav, text, duration, lens = MG.case_inputs(c)
    gold = torch.as_tensor(np.load(os.path.join(ROOT, "tests", "golden", "base_v1_cfg1.npz"))["out"])
    eng = F5HipEngine(cfg, None, device=0)
    eng.load_state_dict(synth.synth_dit_state_dict(cfg, seed=c["wseed"]))
    try:
        model = F5HipCFM(eng, precision="fp16x3")
        eng.set_option("branch_streams", 0)
        base, _ = model.sample(wav.cuda(), text, duration, **c["kw"])
        for variant in (42, 43):
            eng.set_option("gemm_streamk", variant)
            for use_graph in (0, 1):
                eng.set_option("use_graph", use_graph)
                a, _ = model.sample(wav.cuda(), text, duration, **c["kw"])
                b, _ = model.sample(wav.cuda(), text, duration, **c["kw"])
                assert torch.equal(a, b), "deterministic summation order + clean workspace"
                assert (a.cpu() - gold)[:, 468:].abs().max().item() < 1e-3
                assert (a - base).abs().max().item() < 5e-4
        # the two-chain schedule with half the grid per chain
        eng.set_option("branch_streams", 1)
        eng.set_option("gemm_streamk_split", 1)
        eng.set_option("gemm_streamk", 42)
        a, _ = model.sample(wav.cuda(), text, duration, **c["kw"])
        b, _ = model.sample(wav.cuda(), text, duration, **c["kw"])
        assert torch.equal(a, b) and (a.cpu() - gold)[:, 468:].abs().max().item() < 1e-3
        eng.set_option("gemm_streamk", 0)
    finally:
        eng.close()


@direct
def test_key_split_attention_on_the_full_size_model():
    """option "attn_kv_split" (flash attention with every query block cut into 2 / 3 key ranges + a merge kernel, attention_kernel.h):
    against the reference-minted golden and the unsplit path, both schedules."""
    from f5_tts_amd.engine import F5HipCFM, F5HipEngine
    from oracle import make_golden as MG

    c = MG.FULL_CASES["base_v1_cfg1"]
    cfg, wav, text, duration, lens = MG.case_inputs(c)
    gold = torch.as_tensor(np.load(os.path.join(ROOT, "tests", "golden", "base_v1_cfg1.npz"))["out"])
    eng = F5HipEngine(cfg, None, device=0)
    eng.load_state_dict(synth.synth_dit_state_dict(cfg, seed=c["wseed"]))
    try:
        model = F5HipCFM(eng, precision="fp16x3")
        base, _ = model.sample(wav.cuda(), text, duration, **c["kw"])
        for streams in (1, 0):
            eng.set_option("branch_streams", streams)
            for kvs in (2, 3):
                eng.set_option("attn_kv_split", kvs)
                a, _ = model.sample(wav.cuda(), text, duration, **c["kw"])
                assert (a.cpu() - gold)[:, 468:].abs().max().item() < 1e-3
                assert (a - base).abs().max().item() < 5e-4
        eng.set_option("attn_kv_split", 1)
    finally:
        eng.close()
