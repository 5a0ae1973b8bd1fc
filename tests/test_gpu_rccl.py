"""The RCCL calls of f5-tts_amd/dist.py on real hardware.  A GPU box of this pool has ONE device, so the job is a one-rank process group —
but it is backend "nccl" (= RCCL on ROCm): communicator creation, the chunked broadcast of the packed weight blob out of the context's own
device memory, the loaded-mask broadcast, the all-gather of the device census, the MAX all-reduce of the bench protocol and the all-reduce
of ones all run through librccl with the tensor types and aliases the N-rank job uses (world 2 and 8 are covered on the CPU with gloo:
tests/test_dist_cpu.py, tests/test_bench_on_shim.py).  In a child process: the group must not leak into the other tests."""
import os
import subprocess
import sys
import textwrap

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu

CHILD = textwrap.dedent('''
    import os, socket, sys
    sys.path.insert(0, %r)
    import torch, torch.distributed as dist
    import f5_tts_amd
    from f5_tts_amd import config, synth, dist as fdist
    from f5_tts_amd.engine import F5HipCFM, F5HipEngine

    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", LOCAL_RANK="0", WORLD_SIZE="1")
    torch.cuda.set_device(0)
    dist.init_process_group(backend="nccl", rank=0, world_size=1)
    dev = torch.device("cuda", 0)
    cfg = config.DIT_TINY
    sd = synth.synth_dit_state_dict(cfg, seed=3)
    wav = synth.synth_wave(256 * 40, seed=1)
    text = synth.synth_text_ids(1, 30, cfg.text_num_embeds, seed=2)
    kw = dict(steps=4, cfg_strength=2.0, sway_sampling_coef=-1.0, seed=7)

    plain = F5HipEngine(cfg, None, device=dev)
    plain.load_state_dict(sd)                                   # no group involved
    want, _ = F5HipCFM(plain, precision="fp16m").sample(wav.cuda(), text, 100, **kw)

    eng = F5HipEngine(cfg, None, device=dev)
    eng.load_state_dict(sd, finalize=False)
    census = fdist.device_census(fdist.device_identity(0, "cuda"))   # all_gather_object over RCCL
    assert fdist.check_census(census, 1) is None, census
    assert census[0]["pci_bus_id"], census
    assert fdist.check_census(census + [dict(census[0], rank=1)], 2) is not None   # two ranks naming this one device are refused
    before = eng.weight_blob().clone()
    fdist.broadcast_engine_weights(eng, src=0)                  # chunked broadcast of the blob (an alias of the context's memory) + the mask
    assert torch.equal(eng.weight_blob(), before)
    got, _ = F5HipCFM(eng, precision="fp16m").sample(wav.cuda(), text, 100, **kw)
    assert torch.equal(got, want), float((got - want).abs().max())
    assert fdist.barrier_max_seconds(1.25, dev) == 1.25         # MAX all-reduce of a float64 on the device
    assert fdist.ranks_seen(dev) == 1
    dist.barrier()
    dist.destroy_process_group()
    print("rccl one-rank group ok:", census[0]["pci_bus_id"])
''') % ROOT


def test_rccl_collectives_of_the_multi_gpu_layer_on_a_one_rank_group():
    env = {**os.environ, "HSA_ENABLE_IPC_MODE_LEGACY": os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0")}
    r = subprocess.run([sys.executable, "-c", CHILD], capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert r.returncode == 0 and "rccl one-rank group ok" in r.stdout, (r.stdout[-2000:], r.stderr[-3000:])


def test_bench_rank_protocol_over_rccl_with_a_forced_one_rank_group():
    """bench.py itself under the driver's launcher with F5HIP_DIST_FORCE=1: census all-gather, blob broadcast, barriers, MAX all-reduce, the
    all-reduce of ones and the per-rank all_gather_object — over RCCL; the line names the group's one device."""
    import json
    import socket

    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = {**os.environ, "F5HIP_DIST_FORCE": "1", "HSA_ENABLE_IPC_MODE_LEGACY": os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0")}
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1", "--master-port", str(port),
                        os.path.join(ROOT, "bench.py"), "--gpus", "1", "--tiny", "--steps", "2", "--warmup", "1", "--no-cpu-baseline"],
                       capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert r.returncode == 0 and len(lines) == 1, (r.stdout[-2000:], r.stderr[-3000:])
    d = json.loads(lines[0])
    c = d["config"]
    assert d["n_gpus"] == 1 and d["value"] > 0 and c["rccl_ranks"] == 1 and c["rccl_ranks_seen"] == 1, c
    assert len(c["rccl_devices"]) == 1 and "pci" in c["rccl_devices"][0] and "rccl broadcast" in c["weights"], c
