"""world_size-2 gloo tests of the N>1 path (CPU): static sharding, weight-blob broadcast, max-over-ranks timing."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


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


def test_census_check_names_distinct_devices():
    """bench.py's multi-GPU guard (dist.check_census): N ranks must name N distinct (host, PCI address) pairs."""
    from f5_tts_amd.dist import check_census, device_identity, distinct_devices

    def ident(rank, idx, pci, host="box"):
        return {"rank": rank, "host": host, "device_index": idx, "pci_bus_id": pci, "uuid": None}

    good = [ident(r, r, "0000:%02x:00.0" % (5 + 8 * r)) for r in range(8)]
    assert distinct_devices(good) == 8 and check_census(good, 8) is None
    same = [ident(r, 0, "0000:05:00.0") for r in range(8)]  # eight ranks on one GPU
    assert distinct_devices(same) == 1 and "8 ranks name only 1 distinct device" in check_census(same, 8)
    assert "only 7" in check_census(good[:7] + [ident(7, 6, good[6]["pci_bus_id"])], 8)  # two ranks share the seventh
    assert check_census([ident(0, 0, "a"), ident(0, 1, "b")], 2).startswith("the census holds ranks [0, 0]")  # a duplicated rank
    assert check_census([ident(0, 3, "0000:05:00.0", "n0"), ident(1, 3, "0000:05:00.0", "n1")], 2) is None  # same slot on two hosts: distinct
    assert check_census([ident(0, 0, None), ident(1, 1, None)], 2) is None and check_census([ident(0, 0, None), ident(1, 0, None)], 2)  # index fallback
    me = device_identity(0, "cpu")
    assert me["device_index"] == 0 and me["pci_bus_id"] is None and me["host"]


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


# ---- the real engine under the real broadcast: libf5hip's host code and kernels built for the CPU (tests/hipemu), two gloo ranks ---------
from test_hipemu import CLANG, engine_emu_lib  # noqa: E402,F401  (fixture: builds tests/c_abi/_build/engine_emu/libf5hip_engine_emu.so)


def _engine_worker(rank, world, port, lib_path, q):
    import contextlib
    import ctypes as C
    import types

    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import f5_tts_amd  # noqa: F401
    from f5_tts_amd import binding, config, synth
    from f5_tts_amd import dist as fd
    from f5_tts_amd import engine as E
    from test_hipemu import host_alias

    lib = C.CDLL(lib_path, mode=C.RTLD_LOCAL)
    for name, (res, args) in binding.SYMBOLS.items():
        if hasattr(lib, name):
            fn = getattr(lib, name)
            fn.restype, fn.argtypes = res, args
    E.load_library = lambda *a, **k: lib
    E._as_tensor = host_alias
    torch.cuda.device = lambda *_a, **_k: contextlib.nullcontext()
    torch.cuda.current_stream = lambda *_a, **_k: types.SimpleNamespace(cuda_stream=0)

    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    fd.init_distributed(backend="gloo")
    cfg, vcfg = config.DIT_TINY, config.VOCOS_TINY
    eng = E.F5HipEngine(cfg, vcfg, device="cuda:0")  # a descriptor only on the shim
    eng.device = torch.device("cpu")
    if rank == 0:  # bench.py's protocol: rank 0 reads the checkpoint (here one that carries an optional buffer), nobody else does
        sd = {**synth.synth_dit_state_dict(cfg, seed=5), **synth.synth_vocos_state_dict(vcfg, seed=5)}
        half = cfg.dim_head // 2
        sd["transformer.rotary_embed.inv_freq"] = 1.0 / (9000.0 ** (torch.arange(half).float() / half))
        eng.load_state_dict(sd, finalize=False)
    fd.broadcast_engine_weights(eng, src=0)
    wav = synth.synth_wave(256 * 30, seed=2, batch=1)
    text = synth.synth_text_ids(1, 25, cfg.text_num_embeds, seed=4)
    out, _ = E.F5HipCFM(eng, precision="fp32").sample(wav, text, 80, steps=2, cfg_strength=2.0, sway_sampling_coef=-1.0, seed=3)
    wave = eng.vocos_decode(out[:, 30:, :].contiguous(), channel_major=False)
    dist.barrier()
    q.put((rank, int(eng.loaded_mask().sum()), out.numpy(), wave.numpy()))
    eng.close()
    dist.destroy_process_group()


@pytest.mark.skipif(not os.path.exists(CLANG), reason="needs the ROCm host clang++")
def test_world2_gloo_engine_weight_broadcast(engine_emu_lib):  # noqa: F811
    """bench.py's N>1 start-up with real contexts: rank 0 loads, `broadcast_engine_weights` ships blob + loaded mask, rank 1 finalises
    from what it received; both ranks then run the sampler and the vocoder and must agree bit for bit."""
    import numpy as np

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29811 + (os.getpid() % 150)
    procs = [ctx.Process(target=_engine_worker, args=(r, 2, port, engine_emu_lib._name, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=600) for _ in procs), key=lambda r: r[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res[0][1] == res[1][1] > 0
    assert np.isfinite(res[0][2]).all() and np.abs(res[0][2]).max() > 0
    assert np.array_equal(res[0][2], res[1][2]) and np.array_equal(res[0][3], res[1][3])


def _bucket_worker(rank, world, port, lib_path, wav_dir, q):
    import contextlib
    import ctypes as C
    import types
    import wave as wave_mod

    import numpy as np

    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import f5_tts_amd  # noqa: F401
    from f5_tts_amd import binding, config, synth
    from f5_tts_amd import dist as fd
    from f5_tts_amd import engine as E
    from f5_tts_amd import eval_batching as EB
    from test_hipemu import host_alias

    lib = C.CDLL(lib_path, mode=C.RTLD_LOCAL)
    for name, (res, args) in binding.SYMBOLS.items():
        if hasattr(lib, name):
            fn = getattr(lib, name)
            fn.restype, fn.argtypes = res, args
    E.load_library = lambda *a, **k: lib
    E._as_tensor = host_alias
    torch.cuda.device = lambda *_a, **_k: contextlib.nullcontext()
    torch.cuda.current_stream = lambda *_a, **_k: types.SimpleNamespace(cuda_stream=0)
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    fd.init_distributed(backend="gloo")
    cfg, vcfg = config.DIT_TINY, config.VOCOS_TINY
    eng = E.F5HipEngine(cfg, vcfg, device="cuda:0")
    eng.device = torch.device("cpu")
    if rank == 0:
        eng.load_state_dict({**synth.synth_dit_state_dict(cfg, seed=5), **synth.synth_vocos_state_dict(vcfg, seed=5)}, finalize=False)
    fd.broadcast_engine_weights(eng, src=0)
    vocab = {chr(c): 1 + (c % (cfg.text_num_embeds - 2)) for c in range(32, 127)}
    model, voc = E.F5HipCFM(eng, precision="fp32", vocab_char_map=vocab), E.F5HipVocos(eng)
    meta = []
    for i, (secs, words) in enumerate([(0.45, 3), (0.5, 4), (0.47, 3), (0.9, 6), (0.52, 4), (0.95, 7), (0.6, 2)]):
        p = os.path.join(wav_dir, f"p{i}.wav")
        if rank == 0:
            x = 0.2 * np.random.default_rng(i).standard_normal(int(secs * 24000))
            with wave_mod.open(p, "wb") as w:
                w.setnchannels(1); w.setsampwidth(2); w.setframerate(24000); w.writeframes((x.clip(-1, 1) * 32767).astype("<i2").tobytes())
        meta.append((f"u{i}", "ab cd.", p, " ".join(["w"] * words), ""))
    dist.barrier()  # the files exist
    # every rank forms the SAME batch list (deterministic, seeded shuffle) and runs only its share
    batches = EB.get_inference_prompt(meta, lambda wv: model.mel_spec(wv), tokenizer="char", infer_batch_size=150, min_secs=0, max_secs=3,
                                      num_buckets=3)
    mine = EB.deal_batches(batches, world)[rank]
    got = EB.run_prompt_batches(model, voc, [batches[i] for i in mine], nfe_step=1, seed=1)
    box = [None] * world
    dist.all_gather_object(box, [(u, tuple(w.shape), float(w.abs().sum())) for u, w in got])
    q.put((rank, len(batches), mine, box))
    eng.close()
    dist.destroy_process_group()


@pytest.mark.skipif(not os.path.exists(CLANG), reason="needs the ROCm host clang++")
def test_world2_gloo_bucketed_eval_batches(engine_emu_lib, tmp_path):  # noqa: F811
    """eval_infer_batch.py:178-214 on two ranks: the same length-bucketed batch list on every rank, dealt by padded cost, every utterance
    synthesised exactly once across the ranks."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29611 + (os.getpid() % 150)
    procs = [ctx.Process(target=_bucket_worker, args=(r, 2, port, engine_emu_lib._name, str(tmp_path), q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=600) for _ in procs), key=lambda r: r[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res[0][1] == res[1][1] and sorted(res[0][2] + res[1][2]) == list(range(res[0][1]))  # same list, disjoint shares
    assert res[0][3] == res[1][3]  # the gathered report is the same on both ranks
    utts = [u for part in res[0][3] for u, _, _ in part]
    assert sorted(utts) == [f"u{i}" for i in range(7)]
    assert all(shape[0] == 1 and shape[1] > 0 and s > 0 for part in res[0][3] for _, shape, s in part)
