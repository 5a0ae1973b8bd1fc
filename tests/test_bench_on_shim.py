"""bench.py executed end to end without a GPU: its main() over libf5hip built for the host (tests/hipemu) on the --tiny workload — the
single-rank flow, and the PLAIN `bench.py --gpus 2` command: the file launches its own two ranks under torch.distributed.run (gloo here,
RCCL on the GPU box), rank 0 loads and broadcasts the weights, max-over-ranks timing, one JSON line from rank 0.  Numbers printed here
mean nothing; the contract fields and the rank protocol are what is checked."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from test_hipemu import CLANG, engine_emu_lib  # noqa: E402,F401

pytestmark = pytest.mark.skipif(not os.path.exists(CLANG), reason="needs the ROCm host clang++")
HARNESS = os.path.join(ROOT, "tests", "bench_shim_harness.py")
CONTRACT = ["metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config"]
LAUNCHER_ENV = ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "GROUP_RANK", "ROLE_RANK", "LOCAL_WORLD_SIZE", "ROLE_WORLD_SIZE")


def run(cmd, lib, timeout=900, expect_fail=False, **extra_env):
    env = {k: v for k, v in os.environ.items() if k not in LAUNCHER_ENV}
    env.update(dict(F5HIP_EMU_LIB=lib._name, OMP_NUM_THREADS="2"), **extra_env)
    r = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=timeout)
    if expect_fail:
        assert r.returncode != 0, r.stdout[-1000:]
        return r.stdout + r.stderr
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]  # ONE JSON line
    return json.loads(lines[0])


def test_single_rank(engine_emu_lib):  # noqa: F811
    d = run([sys.executable, HARNESS, "--tiny", "--nfe", "1", "--steps", "1", "--warmup", "0"], engine_emu_lib)
    assert all(k in d for k in CONTRACT) and d["n_gpus"] == 1 and d["steps"] == 1 and d["warmup"] == 0 and d["scaling"] == "weak"
    assert d["value"] > 0 and d["vs_baseline"] is None and "NOT A BENCHMARK" in d["config"]["workload"]
    assert d["config"]["rccl_ranks"] == 0 and len(d["config"]["per_rank_ms_per_step"]) == 1 and not d["config"]["self_launched"]
    assert d["roofline"]["bound"] == "mfma" and d["roofline"]["achieved"] > 0 and d["roofline"]["launches"] > 0
    assert d["roofline"]["traffic"] is None  # no PMC pass of the tiny workload is committed
    assert d["roofline_whole_path"]["achieved"] > 0
    assert d["cpu_baseline"]["kind"] in ("reference", "port") and d["cpu_baseline"]["value"] > 0 and d["cpu_baseline"]["cores"] >= 1


def test_single_rank_with_a_forced_group(engine_emu_lib):  # noqa: F811
    """F5HIP_DIST_FORCE=1: a ONE-rank process group (gloo here; RCCL in tests/test_gpu_rccl.py) — census, broadcast, barriers and reductions all
    run, and the line names the group."""
    import socket

    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    d = run([sys.executable, HARNESS, "--tiny", "--nfe", "1", "--steps", "1", "--warmup", "0", "--no-cpu-baseline"], engine_emu_lib,
            F5HIP_DIST_FORCE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", LOCAL_RANK="0", WORLD_SIZE="1")
    c = d["config"]
    assert d["n_gpus"] == 1 and c["rccl_ranks"] == 1 and c["rccl_ranks_seen"] == 1 and len(c["rccl_devices"]) == 1 and "rccl broadcast" in c["weights"], c


def test_plain_gpus_2_command_launches_its_own_two_ranks(engine_emu_lib):  # noqa: F811
    """`python bench.py --gpus 2 ...` with no launcher around it (what a user types; the driver wraps it in torch.distributed.run itself)."""
    d = run([sys.executable, HARNESS, "--gpus", "2", "--tiny", "--nfe", "1", "--steps", "2", "--warmup", "1"], engine_emu_lib, timeout=1200)
    assert d["n_gpus"] == 2 and d["config"]["global_batch"] == 2 and "rccl broadcast" in d["config"]["weights"]
    assert d["config"]["rccl_ranks"] == 2 and len(d["config"]["per_rank_ms_per_step"]) == 2 and d["config"]["self_launched"]
    assert d["ms_per_step"] >= max(d["config"]["per_rank_ms_per_step"]) * 0.999  # MAX over ranks (plus the closing barrier)
    assert d["config"]["weight_broadcast_plus_finalize_s"] > 0
    assert d["config"]["graph"] is True  # the NFE loop is captured and replayed (stream capture emulated by the shim)
    assert d["value"] > 0 and "cpu_baseline" not in d  # the CPU baseline is a single-rank leg
    assert d["config"]["rccl_ranks_seen"] == 2 and len(d["config"]["host_numa_pinning"]) == 2  # the all-reduce census and every rank's pinning report
    devs = d["config"]["rccl_devices"]  # the device census: one line per rank, two different devices
    assert len(devs) == 2 and devs[0].startswith("rank 0:") and devs[1].startswith("rank 1:") and "cuda:0" in devs[0] and "cuda:1" in devs[1]


def test_eight_ranks_name_the_sharded_configuration(engine_emu_lib):  # noqa: F811
    """World size 8 (BASELINE.json configs[3] is 8 ranks x 32 utterances): every rank takes part in the census, the line carries eight
    per-rank times, and the workload string says how many utterances over how many GPUs."""
    d = run([sys.executable, HARNESS, "--gpus", "8", "--tiny", "--nfe", "1", "--steps", "1", "--warmup", "0"], engine_emu_lib, timeout=2400,
            OMP_NUM_THREADS="1")
    assert d["n_gpus"] == 8 and d["config"]["rccl_ranks"] == d["config"]["rccl_ranks_seen"] == 8
    assert len(d["config"]["per_rank_ms_per_step"]) == 8 and d["config"]["global_batch"] == 8
    assert len(set(x.split(": ", 1)[1] for x in d["config"]["rccl_devices"])) == 8


def test_under_the_drivers_launcher(engine_emu_lib):  # noqa: F811
    port = 29911 + (os.getpid() % 80)
    d = run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(port),
             HARNESS, "--gpus", "2", "--tiny", "--nfe", "1", "--steps", "1", "--warmup", "0"], engine_emu_lib, timeout=1200)
    assert d["n_gpus"] == 2 and d["config"]["rccl_ranks"] == 2 and not d["config"]["self_launched"]


def test_a_launch_of_the_wrong_size_is_refused(engine_emu_lib):  # noqa: F811
    out = run([sys.executable, HARNESS, "--gpus", "2", "--tiny", "--nfe", "1", "--steps", "1", "--warmup", "0"], engine_emu_lib, expect_fail=True,
              WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    assert "refusing" in out


@pytest.mark.parametrize("n", [2, 8])
def test_ranks_that_share_a_device_are_refused(engine_emu_lib, n):  # noqa: F811
    """N ranks whose census names ONE device (a launcher that handed every rank the same LOCAL_RANK / a masked visible-device list) must not
    print an N-GPU line: every rank sees the same all-gathered census and exits before any measurement."""
    out = run([sys.executable, HARNESS, "--gpus", str(n), "--tiny", "--nfe", "1", "--steps", "1", "--warmup", "0"], engine_emu_lib, expect_fail=True,
              timeout=1200, SHIM_RANKS_SHARE_ONE_DEVICE="1", OMP_NUM_THREADS="1")
    assert "refusing" in out and f"{n} ranks name only 1 distinct device" in out
    assert not [ln for ln in out.splitlines() if ln.startswith("{")]
