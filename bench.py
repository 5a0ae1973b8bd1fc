#!/usr/bin/env python3
"""bench.py — the driver's benchmark contract for the F5-TTS hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

A "step" is one pass of the hot path over one batch of synthetic utterances on every rank:
mel front-end of the 5 s prompt -> CFM.sample (text embed + NFE-step ODE loop, CFG) -> slice -> Vocos decode.
Default workload = BASELINE.json configs[1]: F5-TTS Base + Vocos, batch 1, NFE 16, sway sampling (N=1406 frames:
469 prompt + 937 generated -> 938 vocoded frames, 9.995 s of audio).  Weak scaling: every rank runs the same
per-GPU batch on its own utterances; weights are broadcast once from rank 0 over RCCL; no collective inside a step.

Prints ONE JSON line on rank 0 (see README / DESIGN.md for the fields).  `roofline` is measured live for the dominant
kernel (the DiT block GEMM) with HIP events on the launch stream in an extra, untimed, eager pass; `cpu_baseline` is
the oracle (a restatement of the reference's CPU path) timed on the host cores on a bounded sample.

Before the warm-up, `--schedule auto` (default, small batches only) lets rank 0 try the engine's off-by-default kernel schedules in child
processes on its GPU (parity against the default path + time), shares what verified with all ranks, re-checks it in-process, and runs the
timed region with a schedule only if it is verified on every rank and faster; the mel is compared with the default path's again afterwards.
Everything that happened is reported under `config.schedule`.  `--schedule default` = no probing.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402

import f5_tts_amd  # noqa: E402,F401
from f5_tts_amd import config, synth  # noqa: E402
from f5_tts_amd import dist as fdist  # noqa: E402

PROBE_CMD = [sys.executable, os.path.abspath(__file__)]  # the schedule-probing child (tests swap in a harness that runs this file on the CPU shim)
DEVICE_TYPE = "cuda"  # the engine refuses anything else; only the CPU-shim harness (which also swaps the library) changes it
PEAK_TFLOPS_FP16_DENSE = 2500.0  # /opt/skills/guides/MI355X_MICROARCH.md: ~2.5 PF dense bf16/fp16 MFMA
HOP, SR = 256, 24000


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=1, help="utterances per GPU per step (configs[2] uses 32)")
    ap.add_argument("--nfe", type=int, default=16)
    ap.add_argument("--precision", default="fp16x3", choices=["fp32", "fp16x3", "fp16"])
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--model", default="F5TTS_v1_Base")
    ap.add_argument("--branch-streams", type=int, default=-1, help="-1 auto / 0 / 1: cond and uncond branches on two streams")
    ap.add_argument("--vocoder", default="vocos", choices=["vocos", "bigvgan"],
                    help="bigvgan: BigVGAN-type mel front-end + the BigVGAN-v2 generator (BASELINE.json configs[4] pairs it with E2TTS_Base)")
    ap.add_argument("--schedule", default="auto", choices=["auto", "default"],
                    help="auto: at start-up, kernel schedules that are off by default (stream-K block GEMMs, key-split attention) are tried in a "
                         "child process on this GPU at this workload, parity-checked against the default path and timed; one is adopted only if "
                         "it is verified and faster, again checked in this process.  default: the default schedule, no probing")
    ap.add_argument("--tiny", action="store_true",
                    help="NOT a benchmark: the tiny model / Vocos at 120 frames (what smoke() runs), so that this file's own logic — rank protocol, "
                         "schedule probing, JSON assembly — can be executed end to end, including on the CPU shim (tests/test_bench_on_shim.py)")
    ap.add_argument("--probe", default=None, choices=["sk", "kv"], help="internal: run as the schedule-probing child for one option group")
    ap.add_argument("--probe-device", type=int, default=0, help="internal: HIP device of the probing child")
    return ap.parse_args()


def host_cores():
    """Threads the CPU baseline may use: the affinity mask, capped by the cgroup CPU quota (a 256-thread EPYC box with a
    16-CPU quota runs the oracle 80x slower with 256 OpenMP threads than with 16)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return n


def cpu_baseline(cfg, sd, vsd, vcfg, wav, text, duration, nfe, t_gen, bigvgan=None):
    """Oracle (port of the reference CPU path) on the host cores, bounded sample (~10-30 s of CPU work): mel + text-embed +
    1 ODE step, then 1 + `probe` steps (probe = 2..8, sized from the first run), + the vocoder, all at full size; per-step time
    extrapolated to `nfe` steps (every step does identical work)."""
    from oracle import f5_oracle as O

    cores = host_cores()
    torch.set_num_threads(cores)
    kw = dict(cfg_strength=2.0, sway_sampling_coef=-1.0, seed=0, use_epss=False)
    t0 = time.perf_counter()
    O.cfm_sample(sd, cfg, wav[:1], text[:1], duration, steps=1, **kw)
    t1 = time.perf_counter()
    probe = max(2, min(8, int(15.0 / max(t1 - t0, 1e-3)) - 2))  # sized from the first measurement so that the whole sample is ~10-30 s of CPU work
    out, _ = O.cfm_sample(sd, cfg, wav[:1], text[:1], duration, steps=1 + probe, **kw)
    t2 = time.perf_counter()
    gen = out[:, wav.shape[-1] // HOP:, :].permute(0, 2, 1)
    voc_scale = 1.0
    if bigvgan is not None:  # the generator is linear in the frame count: time 64 frames of it and scale
        from oracle import bigvgan_oracle as BO

        BO.bigvgan_forward(bigvgan[1], bigvgan[0], gen[:, :, :64])
        voc_scale = gen.shape[-1] / 64.0
    else:
        O.vocos_decode(vsd, gen, vcfg.num_layers)
    t3 = time.perf_counter()
    per_step = max(((t2 - t1) - (t1 - t0)) / probe, 0.0)
    setup = max((t1 - t0) - per_step, 0.0)
    total = setup + nfe * per_step + (t3 - t2) * voc_scale
    cpu_name = ""
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                cpu_name = line.split(":", 1)[1].strip()
                break
    except Exception:
        pass
    return {"value": t_gen / total, "unit": "frames/s", "cores": cores, "kind": "port", "cpu": cpu_name,
            "rtf": total / (HOP * (t_gen - 1) / SR), "seconds_per_utterance_extrapolated": total,
            "sample": f"full-size model, 1 utterance: mel + text-embed + {probe + 2} ODE steps + vocoder measured "
                      f"({t3 - t0:.1f} s of CPU on {cores} threads), per-step time ({per_step:.2f} s) extrapolated to NFE={nfe}"}


def pmc_traffic(a, B):
    """HBM bytes per launch of the dominant kernel from the committed PMC summary of this workload (profiles/*.json)."""
    import glob

    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*pmc*.json")), reverse=True):
        try:
            d = json.load(open(f))
        except Exception:
            continue
        if d.get("precision") == a.precision and d.get("batch") == B and d.get("nfe") == a.nfe and d.get("model") == a.model:
            return d.get("hbm_bytes_per_launch")
    return None


# ---- schedule selection by measurement ------------------------------------------------------------------------------------------------
# Option sets that are OFF by default in the engine (written after the GPU budget of their round was spent).  Each is tried in
# a child process first, so that a kernel that has never run on this GPU cannot take the benchmark down with it.
SCHEDULE_GROUPS = {
    "sk": [{"gemm_streamk": 42}, {"gemm_streamk": 43},
           {"gemm_streamk": 42, "gemm_streamk_split": 1, "branch_streams": 1}, {"gemm_streamk": 43, "gemm_streamk_split": 1, "branch_streams": 1}],
    "kv": [{"attn_kv_split": 2}, {"attn_kv_split": 3}, {"attn_kv_split": 4}],
}
SCHEDULE_OFF = {"gemm_streamk": 0, "gemm_streamk_split": 0, "attn_kv_split": 1}
ADOPT_RATIO = float(os.environ.get("F5HIP_BENCH_ADOPT_RATIO", "0.99"))  # adopt only below this fraction of the default schedule's time (tests raise it)
SCHEDULE_TOL = 5e-4  # max-abs on the mel between two schedules of the same precision mode (summation order only); parity mode bound is 1e-3


def set_schedule(eng, opts, branch_streams):
    for k, v in {**SCHEDULE_OFF, "branch_streams": branch_streams, **opts}.items():
        eng.set_option(k, v)


def try_schedule(eng, opts, branch_streams, run, dev, base_mel, reps=4):
    """Apply `opts`, check the generated mel against the default schedule's (and run-to-run), time `reps` passes.  Always restores the default."""
    res = {"options": opts, "ok": False}
    try:
        set_schedule(eng, opts, branch_streams)
        m1, m2 = run(mel=True), run(mel=True)
        torch.cuda.synchronize(dev)
        err = float((m1 - base_mel).abs().max())
        res.update(max_abs_vs_default=err, deterministic=bool(torch.equal(m1, m2)), finite=bool(torch.isfinite(m1).all()))
        res["ok"] = res["finite"] and res["deterministic"] and err < SCHEDULE_TOL
        if res["ok"]:
            t0 = time.perf_counter()
            for _ in range(reps):
                run()
            torch.cuda.synchronize(dev)
            res["ms"] = 1e3 * (time.perf_counter() - t0) / reps
    except Exception as e:  # an option the engine refuses, a failed launch
        res["error"] = repr(e)[:300]
    finally:
        set_schedule(eng, {}, branch_streams)
    return res


def time_default(eng, branch_streams, run, dev, reps=4):
    set_schedule(eng, {}, branch_streams)
    base = run(mel=True).clone()
    run()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(reps):
        run()
    torch.cuda.synchronize(dev)
    return base, 1e3 * (time.perf_counter() - t0) / reps


def probe_child(a, eng, run, dev):
    """--probe GROUP: the child's whole job.  One JSON line {"probe": ...} on stdout."""
    reps = 1 if a.tiny else 4
    base, t_base = time_default(eng, a.branch_streams, run, dev, reps)
    group = SCHEDULE_GROUPS[a.probe]
    if a.tiny:  # the shim's end-to-end test of this file: one packed and one two-chain stream-K candidate, one key-split candidate
        group = group[::2][:2] if a.probe == "sk" else group[:1]
    cands = [try_schedule(eng, o, a.branch_streams, run, dev, base, reps) for o in group]
    t_base = min(t_base, time_default(eng, a.branch_streams, run, dev, reps)[1])  # the default again after the candidates (clock ramp-up favours whoever runs later)
    out = {"group": a.probe, "default_ms": t_base, "candidates": cands}
    print(json.dumps({"probe": out}), flush=True)


def probe_in_children(a, local):
    """Rank 0: one child per option group on this rank's GPU (same workload flags), bounded in time; returns the verified winners."""
    import subprocess

    env = {k: v for k, v in os.environ.items()
           if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "GROUP_RANK", "ROLE_RANK", "LOCAL_WORLD_SIZE", "ROLE_WORLD_SIZE")
           and not k.startswith("TORCHELASTIC")}
    report, winners = {}, []
    for group in SCHEDULE_GROUPS:
        cmd = PROBE_CMD + (["--tiny"] if a.tiny else []) + (["--no-graph"] if a.no_graph else []) + ["--probe", group, "--probe-device", str(local), "--batch", str(a.batch), "--nfe", str(a.nfe),
               "--precision", a.precision, "--model", a.model, "--vocoder", a.vocoder, "--branch-streams", str(a.branch_streams), "--no-cpu-baseline"]
        try:
            r = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=120)
            line = next((ln for ln in reversed(r.stdout.splitlines()) if ln.startswith('{"probe"')), None)
            if line is None:
                report[group] = {"error": f"child exit {r.returncode}: {(r.stdout + r.stderr)[-400:]}"}
                continue
            pr = json.loads(line)["probe"]
        except subprocess.TimeoutExpired:
            report[group] = {"error": "child timed out"}
            continue
        except Exception as e:  # pragma: no cover
            report[group] = {"error": repr(e)[:300]}
            continue
        report[group] = pr
        good = [c for c in pr["candidates"] if c.get("ok") and c.get("ms") and c["ms"] < ADOPT_RATIO * pr["default_ms"]]
        if good:
            winners.append(min(good, key=lambda c: c["ms"])["options"])
    return report, winners


def main():
    a = parse()
    if a.probe:
        for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
            os.environ.pop(k, None)
        os.environ["LOCAL_RANK"] = str(a.probe_device)
    rank, local, world = fdist.init_distributed()
    assert world == max(a.gpus, 1) or world == 1, f"--gpus {a.gpus} but WORLD_SIZE={world}"
    from f5_tts_amd.engine import F5HipCFM, F5HipEngine, F5HipVocos

    dev = torch.device(DEVICE_TYPE, local) if DEVICE_TYPE == "cuda" else torch.device(DEVICE_TYPE)
    torch.cuda.set_device(dev)
    big = a.vocoder == "bigvgan"
    cfg, vcfg = config.PRESETS[a.model], (None if big else config.VOCOS_MEL_24K)
    if a.tiny:
        assert not big, "--tiny is the Vocos pairing only"
        a.model, cfg, vcfg = "tiny", config.DIT_TINY, config.VOCOS_TINY
    eng = F5HipEngine(cfg, vcfg, device=dev)
    sd = vsd = None
    if rank == 0:  # rank 0 "reads the checkpoint"; the packed blob travels over RCCL/xGMI
        sd = synth.synth_dit_state_dict(cfg, seed=0)
        vsd = {} if big else synth.synth_vocos_state_dict(vcfg, seed=0)
        eng.load_state_dict({**sd, **vsd}, finalize=False)
    weights_via = "rccl broadcast of the packed blob from rank 0" if world > 1 else "local (single rank)"
    fdist.broadcast_engine_weights(eng, src=0)
    if not a.no_graph:
        eng.set_option("use_graph", 1)
    eng.set_option("branch_streams", a.branch_streams)
    if os.environ.get("F5HIP_BENCH_KVSPLIT"):  # experiment switch: key-split flash attention
        eng.set_option("attn_kv_split", int(os.environ["F5HIP_BENCH_KVSPLIT"]))
    if os.environ.get("F5HIP_BENCH_STREAMK"):  # experiment switch (tools/r2_first_call.sh): DiT block GEMMs through gemm_skrs.h
        eng.set_option("gemm_streamk", int(os.environ["F5HIP_BENCH_STREAMK"]))
        eng.set_option("gemm_streamk_split", int(os.environ.get("F5HIP_BENCH_STREAMK_SPLIT", "0")))
    if big:  # the generator is a context of its own (as in the reference); every rank builds the same seeded weights
        from f5_tts_amd.bigvgan import F5HipBigVGAN

        bcfg = config.BIGVGAN_V2_24K_100B_256X
        model = F5HipCFM(eng, precision=a.precision, mel_spec_type="bigvgan")
        bsd = synth.synth_bigvgan_state_dict(bcfg, seed=0)
        voc = F5HipBigVGAN(bcfg, device=dev, precision=a.precision).load_state_dict(bsd)
    else:
        model, voc = F5HipCFM(eng, precision=a.precision), F5HipVocos(eng)

    B, nw, nt, duration = a.batch, 120000, 220, 1406
    if a.tiny:
        nw, nt, duration = 256 * 40, 30, 120
    wav = synth.synth_wave(nw, seed=1000 * rank, batch=B).to(dev)  # resident in HBM before the timed region
    text = synth.synth_text_ids(B, nt, cfg.text_num_embeds, seed=rank)
    ref_len = nw // HOP  # 468 (utils_infer.py:486)
    t_gen = duration - ref_len  # 938 vocoded frames per utterance
    kw = dict(steps=a.nfe, cfg_strength=2.0, sway_sampling_coef=-1.0, seed=0)

    def one_pass(mel=False):
        out, _ = model.sample(wav, text, duration, **kw)
        gen = out[:, ref_len:, :]  # [B, 938, 100] view; the engine takes frame-major directly
        if big:  # vocoder(mel[b, 100, T]) as at reference utils_infer.py:509-513
            wave = voc(gen.permute(0, 2, 1))[:, 0]
        else:
            wave = eng.vocos_decode(gen.contiguous(), channel_major=False)
        return out if mel else wave

    if a.probe:
        probe_child(a, eng, one_pass, dev)
        return
    forced = any(os.environ.get(k) for k in ("F5HIP_BENCH_KVSPLIT", "F5HIP_BENCH_STREAMK"))
    schedule = {"selected": {}, "how": "default schedule" + (" (switches forced by the environment)" if forced else "")}
    if a.schedule == "auto" and not forced and B * duration > 6144:
        # the probed schedules exist for latency-bound launches (few, short workgroups per kernel); at large batch every launch is many
        # rounds of tiles deep and they have nothing to offer — not worth minutes of probing
        schedule["how"] = "default schedule (probing applies to small batches only: B x N <= 6144 rows)"
    elif a.schedule == "auto" and not forced:
        report, cands = probe_in_children(a, local) if rank == 0 else ({}, [])
        if world > 1:  # every rank runs what rank 0's children verified
            box = [cands]
            torch.distributed.broadcast_object_list(box, src=0)
            cands = box[0]
        if len(cands) == 2:
            cands = [{**cands[0], **cands[1]}] + cands  # the combination first, then each alone
        schedule["probe"] = report
        if cands:  # second check, in this process: parity and time against the default, the same decision on every rank
            reps = 1 if a.tiny else 4
            sched_base, t_def = time_default(eng, a.branch_streams, one_pass, dev, reps)
            tried = [try_schedule(eng, o, a.branch_streams, one_pass, dev, sched_base, reps) for o in cands]
            t_def = min(t_def, time_default(eng, a.branch_streams, one_pass, dev, reps)[1])  # again after the candidates: clocks ramp, the first timing is the pessimistic one
            ms = torch.tensor([t_def] + [t["ms"] if t["ok"] else float("inf") for t in tried], dtype=torch.float64, device=dev)
            if world > 1:
                torch.distributed.all_reduce(ms, op=torch.distributed.ReduceOp.MAX)
            ms = ms.tolist()
            best = min(range(1, len(ms)), key=lambda i: ms[i])
            schedule.update(default_ms=ms[0], tried=[{**t, "ms_max_over_ranks": m} for t, m in zip(tried, ms[1:])])
            if ms[best] < ADOPT_RATIO * ms[0]:
                schedule["selected"] = cands[best - 1]
                schedule["how"] = ("measured at start-up: verified and timed in a child process, then parity-checked (max-abs on the mel < "
                                   f"{SCHEDULE_TOL}) and timed again here against the default schedule")
        set_schedule(eng, schedule["selected"], a.branch_streams)

    one_pass()  # set-up, like loading the weights: workspace allocation and graph capture happen here, whatever --warmup says

    def timed_region():
        for _ in range(a.warmup):
            one_pass()
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for _ in range(a.steps):
            wave = one_pass()
        torch.cuda.synchronize(dev)
        if world > 1:
            torch.distributed.barrier()
        dt = fdist.barrier_max_seconds(time.perf_counter() - t0, dev)
        assert wave.shape == (B, HOP * (t_gen if big else t_gen - 1)) and bool(torch.isfinite(wave).all())
        return dt

    dt = timed_region()
    if schedule["selected"]:  # an adopted schedule must still reproduce the default path's mel AFTER the timed steps, on every rank
        ok = torch.tensor([float((one_pass(mel=True) - sched_base).abs().max()) < SCHEDULE_TOL], dtype=torch.float64, device=dev)
        if world > 1:
            torch.distributed.all_reduce(ok, op=torch.distributed.ReduceOp.MIN)
        if not bool(ok.item()):  # never report a number from a schedule that drifted: back to the default, measured again
            schedule.update(selected={}, how="default schedule (the probed schedule failed the parity re-check after the timed region: discarded, default re-timed)")
            set_schedule(eng, {}, a.branch_streams)
            dt = timed_region()

    if rank != 0:
        if world > 1:
            torch.distributed.barrier()  # rank 0 is still profiling / printing; leave together
            torch.distributed.destroy_process_group()
        return
    ms_per_step = 1e3 * dt / a.steps
    frames = world * B * t_gen
    audio_s = world * B * HOP * (t_gen - 1) / SR
    res = {
        "metric": "gen_mel_frames_per_s", "value": frames / (dt / a.steps), "unit": "frames/s",
        "rtf": (dt / a.steps) / audio_s, "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": ms_per_step,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": {"fp16x3": "fp16x3 (fp16 hi/lo split MFMA operands, fp32 accumulate/state)", "fp16": "fp16 (fp32 accumulate/state)",
                  "fp32": "fp32"}[a.precision],
        "data": "synthetic (seeded 0.1*N(0,1) prompts, uniform token ids, random-init weights of the named architecture)",
        "config": {"workload": f"NOT A BENCHMARK (--tiny): tiny model + tiny Vocos, {duration} frames, NFE={a.nfe}" if a.tiny else
                               f"{a.model} + {'BigVGAN-v2 (24 kHz, 100 band, 256x)' if big else 'Vocos'}, batch {B}/GPU, 5 s ref + 10 s gen (N=1406 frames, 938 vocoded), NFE={a.nfe}, "
                               f"sway -1, CFG 2.0, euler (BASELINE.json configs[{1 if B == 1 else 2}])",
                   "batch_per_gpu": B, "global_batch": B * world, "frames": duration, "nfe": a.nfe, "graph": not a.no_graph,
                   "parallelism": f"utterance-sharded x{world}, RCCL weight broadcast, no in-step collective", "weights": weights_via,
                   "schedule": schedule},
    }
    # ---- roofline of the dominant kernel: DiT block GEMMs, HIP events on the launch stream, untimed eager pass -------
    # The profiled pass is serial (one chain).  An adopted two-chain stream-K schedule would therefore run its PACKED form here: allowed only
    # if that form itself verified in the probing child; otherwise this pass describes the default kernels and says so.
    sel = schedule["selected"]
    prof_sel = dict(sel)
    if sel.get("gemm_streamk_split"):
        packed = {"gemm_streamk": sel["gemm_streamk"]}
        if not any(c.get("ok") and c.get("options") == packed for c in schedule.get("probe", {}).get("sk", {}).get("candidates", [])):
            prof_sel = {k: v for k, v in sel.items() if not k.startswith("gemm_streamk") and k != "branch_streams"}
            set_schedule(eng, prof_sel, a.branch_streams)
    eng.set_option("profile", 1)
    eng.reset_kernel_stats()
    if big:
        voc.set_option("profile", 1)
        voc.reset_kernel_stats()
    one_pass()
    torch.cuda.synchronize(dev)
    stats = eng.kernel_stats()
    eng.set_option("profile", 0)
    if big:  # the generator's own kernel classes (its context is separate from the backbone's)
        voc.set_option("profile", 0)
        res["vocoder_classes"] = {
            k: {"calls": v["calls"], "ms": round(v["ms"], 3),
                **({"tflops": round(v["flops"] / (1e-3 * v["ms"]) / 1e12, 1)} if v["flops"] and v["ms"] > 0 else {}),
                **({"gbps": round(v["bytes"] / (1e-3 * v["ms"]) / 1e9, 1)} if v["bytes"] and v["ms"] > 0 else {})}
            for k, v in voc.kernel_stats().items() if v["calls"]}
    g = stats["gemm_block"]
    if g["calls"] and g["ms"] > 0:
        avg_s = 1e-3 * g["ms"] / g["calls"]
        ach = g["flops"] / g["calls"] / avg_s / 1e12
        res["roofline"] = {"kernel": ("gemm_skrs_kernel, stream-K" if prof_sel.get("gemm_streamk") else "gemm_kernel") + " (DiT block QKV/out/FF1/FF2)"
                                     + ("" if prof_sel == sel else " — the default kernels: the adopted two-chain stream-K schedule has no verified serial form to event-time"), "bound": "mfma", "achieved": ach,
                           "peak": PEAK_TFLOPS_FP16_DENSE, "unit": "TFLOP/s", "frac": ach / PEAK_TFLOPS_FP16_DENSE,
                           "traffic": None if schedule["selected"] else pmc_traffic(a, B), "avg_launch_us": 1e6 * avg_s, "launches": g["calls"],
                           "mfma_issue_tflops": ach * (3 if a.precision == "fp16x3" else 1),
                           "note": "achieved = algorithmic FLOPs (2MNK, SURVEY.md 8d) / avg launch duration, HIP events on the launch "
                                   "stream; fp16x3 issues 3 fp16 MFMAs per algorithmic product (mfma_issue_tflops = 3x achieved); "
                                   "traffic = HBM bytes per launch from the committed rocprofv3 PMC pass of the same command "
                                   "(profiles/, FETCH_SIZE x2 gfx950 correction + WRITE_SIZE), null if none matches"}
    res["kernel_classes_ms"] = {k: round(v["ms"], 3) for k, v in stats.items() if v["calls"]}
    # per kernel class: HIP-event time of the profiled pass with the algorithmic FLOPs / bytes the launch sites declare:
    # TFLOP/s for the MFMA-bound classes, GB/s (ideal-fusion bytes) for the HBM-bound ones
    res["kernel_classes"] = {
        k: {"calls": v["calls"], "ms": round(v["ms"], 3),
            **({"tflops": round(v["flops"] / (1e-3 * v["ms"]) / 1e12, 1)} if v["flops"] else {}),
            **({"gbps": round(v["bytes"] / (1e-3 * v["ms"]) / 1e9, 1)} if v["bytes"] and k not in ("gemm_block",) else {})}
        for k, v in stats.items() if v["calls"] and v["ms"] > 0}
    if not a.no_cpu_baseline and world == 1:
        try:
            res["cpu_baseline"] = cpu_baseline(cfg, sd, vsd, vcfg, wav.cpu(), text, duration, a.nfe, t_gen, (bcfg, bsd) if big else None)
        except Exception as e:  # pragma: no cover
            res["cpu_baseline"] = {"error": repr(e)}
    print(json.dumps(res), flush=True)
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
