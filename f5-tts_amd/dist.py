"""Multi-GPU layer: utterances shard by rank, weights travel once over RCCL, nothing inside a step.

The reference's only multi-GPU inference pattern is one process per GPU, each processing a disjoint
slice of the utterance list with barriers around the loop (reference
``src/f5_tts/eval/eval_infer_batch.py:178-214`` via ``accelerator.split_between_processes``;
``src/f5_tts/runtime/triton_trtllm/benchmark.py:199-212,340-344`` via ``DistributedSampler``), every rank
reading the checkpoint from disk itself.  Here rank 0 owns the checkpoint and the packed fp32 weight
blob is broadcast once (torch.distributed backend "nccl" == RCCL over xGMI on ROCm; "gloo" in the CPU tests).
"""
from __future__ import annotations

import datetime
import os
from typing import List, Sequence, Tuple

import torch
import torch.distributed as dist


def init_distributed(backend: str | None = None) -> Tuple[int, int, int]:
    """Read RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the env (torch.distributed.run contract)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # F5HIP_DIST_FORCE=1: a process group even for ONE rank — every collective of the job then really runs (RCCL on a one-GPU box, gloo on the shim)
    if (world > 1 or os.environ.get("F5HIP_DIST_FORCE") == "1") and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local)
        # generous collective timeout: rank 0 may spend minutes between two collectives (checkpoint reading, bench.py's schedule probing)
        dist.init_process_group(backend=backend, rank=rank, world_size=world, timeout=datetime.timedelta(minutes=30))
    return rank, local, world


def shard_contiguous(n_items: int, rank: int, world: int) -> range:
    """Contiguous slice per rank, sizes differing by at most one (accelerate's split_between_processes)."""
    base, rem = divmod(n_items, world)
    start = rank * base + min(rank, rem)
    return range(start, start + base + (1 if rank < rem else 0))


def shard_balanced(costs: Sequence[float], world: int) -> List[List[int]]:
    """Ragged utterances: longest-first greedy deal so every rank gets a similar frame budget
    (the reference shuffles buckets "not only leave easy work for last workers", eval/utils_eval.py:201-203)."""
    order = sorted(range(len(costs)), key=lambda i: -costs[i])
    loads = [0.0] * world
    out: List[List[int]] = [[] for _ in range(world)]
    for i in order:
        r = min(range(world), key=lambda k: (loads[k], k))
        out[r].append(i)
        loads[r] += costs[i]
    for lst in out:
        lst.sort()
    return out


def broadcast_blob(blob: torch.Tensor, src: int = 0, chunk_elems: int = 64 << 20) -> None:
    """Broadcast the packed weight blob in <=256 MiB chunks (few, large collectives; xGMI links are
    point-to-point so a ring/tree broadcast is per-link bound — ~1.4 GB fp32 takes ~10 ms)."""
    if not dist.is_initialized():  # (a one-rank group still goes through the collective: tests/test_gpu_rccl.py runs RCCL itself that way)
        return
    flat = blob.view(-1)
    for s in range(0, flat.numel(), chunk_elems):
        dist.broadcast(flat[s:s + chunk_elems], src=src)


def broadcast_engine_weights(engine, src: int = 0) -> None:
    """Rank `src` has loaded the state dict; everyone else receives the blob and finalises."""
    if dist.is_initialized():
        blob = engine.weight_blob()  # an alias of the context's device memory (engine._as_tensor refuses to hand out a copy)
        broadcast_blob(blob, src=src)
        # which entries the sender actually read (a checkpoint's optional buffers travel in the blob; the mask says to use them)
        mask = engine.loaded_mask().to(blob.device)
        dist.broadcast(mask, src=src)
        if dist.get_rank() != src:
            engine.set_loaded_mask(mask)
    engine.finalize()


def barrier_max_seconds(seconds: float, device: torch.device | None = None) -> float:
    """MAX over ranks of a per-rank wall time (the bench contract)."""
    if not dist.is_initialized():
        return seconds
    t = torch.tensor([seconds], dtype=torch.float64, device=device if device is not None else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def _cpulist(text: str) -> List[int]:
    out: List[int] = []
    for part in text.strip().split(","):
        if not part:
            continue
        lo, _, hi = part.partition("-")
        out.extend(range(int(lo), int(hi or lo) + 1))
    return out


def pin_to_gpu_numa(local_rank: int) -> dict:
    """Keep this rank's host threads on the NUMA node its GPU hangs off (staging copies and launches then stay off the socket
    interconnect).  Best effort, Linux sysfs only: the PCI address of ``cuda:<local_rank>`` -> ``/sys/bus/pci/devices/<addr>/numa_node`` ->
    that node's cpulist intersected with the CPUs the process may use.  Returns what it did (bench.py puts it into its line)."""
    info = {"numa_node": None, "cpus": None, "pinned": False}
    try:
        if not torch.cuda.is_available() or not hasattr(os, "sched_setaffinity"):
            return info
        bus = torch.cuda.get_device_properties(local_rank).pci_bus_id  # torch >= 2.x: "0000:0c:00.0"-style string or an int bus number
        addr = bus.lower() if isinstance(bus, str) else None
        if addr is None:
            p = torch.cuda.get_device_properties(local_rank)
            addr = "%04x:%02x:%02x.0" % (getattr(p, "pci_domain_id", 0), p.pci_bus_id, getattr(p, "pci_device_id", 0))
        node = int(open(f"/sys/bus/pci/devices/{addr}/numa_node").read())
        info["numa_node"] = node
        if node < 0:
            return info
        cpus = set(_cpulist(open(f"/sys/devices/system/node/node{node}/cpulist").read())) & set(os.sched_getaffinity(0))
        if cpus:
            os.sched_setaffinity(0, cpus)
            info.update(cpus=len(cpus), pinned=True)
    except Exception as e:  # containers often hide sysfs: not an error
        info["error"] = type(e).__name__
    return info


def device_identity(local_rank: int, device_type: str = "cuda") -> dict:
    """What THIS rank computes on: host name, the device index it set, and the PCI address (plus the UUID where torch exposes one) of that
    device — the key two ranks must not share.  ``bench.py`` refuses a line whose ranks do not name N distinct devices."""
    import socket

    ident = {"rank": int(os.environ.get("RANK", "0")), "host": socket.gethostname(), "device_index": int(local_rank), "pci_bus_id": None, "uuid": None}
    if device_type == "cuda" and torch.cuda.is_available():
        p = torch.cuda.get_device_properties(local_rank)
        bus = getattr(p, "pci_bus_id", None)
        ident["pci_bus_id"] = bus.lower() if isinstance(bus, str) else "%04x:%02x:%02x.0" % (getattr(p, "pci_domain_id", 0) or 0, bus or 0, getattr(p, "pci_device_id", 0) or 0)
        if getattr(p, "uuid", None) is not None:
            ident["uuid"] = str(p.uuid)
        ident["device_index"] = int(torch.cuda.current_device())  # the device the rank's context REALLY sits on, not what it was told
    return ident


def device_census(ident: dict) -> List[dict]:
    """All-gather of every rank's ``device_identity`` over the job's process group (the group the weight broadcast used), in rank order."""
    if not dist.is_initialized():
        return [ident]
    box: List[dict | None] = [None] * dist.get_world_size()
    dist.all_gather_object(box, ident)
    return sorted(box, key=lambda d: d["rank"])  # type: ignore[arg-type, index]


def distinct_devices(census: Sequence[dict]) -> int:
    """How many different physical devices a census names: (host, PCI address) where known, (host, device index) otherwise."""
    return len({(d["host"], d["pci_bus_id"] or d["uuid"] or ("index", d["device_index"])) for d in census})


def check_census(census: Sequence[dict], world: int) -> str | None:
    """None if the census is ``world`` ranks 0 .. world-1 on ``world`` distinct devices, else the reason a bench line must not be printed."""
    if sorted(d["rank"] for d in census) != list(range(world)):
        return f"the census holds ranks {sorted(d['rank'] for d in census)}, not 0..{world - 1}"
    n = distinct_devices(census)
    if n != world:
        return (f"{world} ranks name only {n} distinct device(s): " + ", ".join(f"rank {d['rank']} -> {d['host']}:{d['pci_bus_id'] or d['device_index']}" for d in census))
    return None


def ranks_seen(device) -> int:
    """An all-reduce of ones over the job's process group: how many ranks actually took part in a collective (1 without a group)."""
    if not dist.is_initialized():
        return 1
    one = torch.ones(1, dtype=torch.int32, device=device)
    dist.all_reduce(one)
    return int(one.item())
