"""world_size-2 gloo tests of the N>1 path (CPU): static sharding, weight-blob broadcast, max-over-ranks timing."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_shard_contiguous_partitions_everything():
    from f5_tts_amd.dist import shard_contiguous

    for n in (0, 1, 7, 32, 255, 256):
        for world in (1, 2, 3, 8):
            got = [i for r in range(world) for i in shard_contiguous(n, r, world)]
            assert got == list(range(n))
            sizes = [len(shard_contiguous(n, r, world)) for r in range(world)]
            assert max(sizes) - min(sizes) <= 1
    assert [len(shard_contiguous(256, r, 8)) for r in range(8)] == [32] * 8  # BASELINE config 4


def test_shard_balanced():
    from f5_tts_amd.dist import shard_balanced

    costs = [100, 1, 1, 1, 50, 50, 2, 97]
    parts = shard_balanced(costs, 2)
    assert sorted(i for p in parts for i in p) == list(range(len(costs)))
    loads = [sum(costs[i] for i in p) for p in parts]
    assert abs(loads[0] - loads[1]) <= 4


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    import f5_tts_amd  # noqa: F401
    from f5_tts_amd import dist as fd

    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    r, l, w = fd.init_distributed(backend="gloo")
    assert (r, w) == (rank, world)
    # rank 0 "loaded a checkpoint"; the others start from garbage and must end up identical
    g = torch.Generator().manual_seed(1234)
    blob = torch.randn(3_000_017, generator=g) if rank == 0 else torch.full((3_000_017,), float("nan"))
    fd.broadcast_blob(blob, src=0, chunk_elems=1 << 20)
    g2 = torch.Generator().manual_seed(1234)
    ok = torch.equal(blob, torch.randn(3_000_017, generator=g2))
    t = fd.barrier_max_seconds(1.0 + rank)
    mine = list(fd.shard_contiguous(5, rank, world))
    dist.barrier()
    q.put((rank, ok, t, mine))
    dist.destroy_process_group()


def test_world2_gloo_broadcast_and_timing():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29611 + (os.getpid() % 200)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=180) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert [r[1] for r in res] == [True, True]
    assert [r[2] for r in res] == [2.0, 2.0]  # MAX over ranks
    assert res[0][3] == [0, 1, 2] and res[1][3] == [3, 4]
