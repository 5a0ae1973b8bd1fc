"""Aggregate rocprofv3 FETCH_SIZE / WRITE_SIZE passes (`tools/gpu_run.sh pmc` / `counters`) into HBM bytes per launch per kernel class.
FETCH_SIZE / WRITE_SIZE are reported in KiB-units of 1024 B by rocprofv3; on gfx950 FETCH_SIZE counts 128-B requests as 64 B
for wide coalesced reads, so it is doubled (/opt/skills/guides/MI355X_MICROARCH.md, HBM section); WRITE_SIZE is uncalibrated."""
import argparse
import collections
import csv
import glob
import json
import os
import re
import sys


def klass(name):
    # the block GEMMs: the pipelined kernel (gemm_pp.h) and the generic one (gemm.h), by operand mode
    if "gemm_p8_kernel" in name:  # the ping-pong 256x256 kernel (gemm_p8.h): NSPLIT 1 = plain fp16 rows, 2 = MX lines
        return "gemm_fp16m" if ("gemm_p8_kernelILi2E" in name or "gemm_p8_kernel<2," in name) else "gemm_fp16"
    if "gemm_pp_kernel" in name or ("gemm_kernel" in name and ("EpiQKV" in name or "EpiStore" in name)):
        if "DF16_Li3" in name or "_Float16, 3" in name:
            return "gemm_fp16x3"
        if "DF16_Li2" in name or "_Float16, 2" in name:  # MX lines (fp16m): 2 fp16 MFMAs + 1 MX-fp6 MFMA per 32 k
            return "gemm_fp16m"
        if "DF16_Li1" in name or "_Float16, 1" in name:
            return "gemm_fp16"
        return "gemm_fp32"
    if "flash_attn" in name or "flash_pipe" in name or "flash_pp" in name:
        return "flash_attn"
    if "layernorm" in name:
        return "layernorm"
    if "convpos_kernel" in name or "convpos_mx_kernel" in name:
        return "convpos"
    # the memory-bound kernels north_star names (ConvNeXt text blocks, Vocos blocks, mel front-end, iSTFT) and the per-step update
    for key in ("dwconv7_ln_kernel", "grn_finish_kernel", "grn_sumsq_kernel", "grn_apply_kernel", "text_embed_kernel", "mel_kernel",
                "istft_fused_kernel", "cfg_euler_kernel", "im2col7_kernel"):
        if key in name:
            return key[: -len("_kernel")]
    return None


def load(d, counter):
    agg = collections.defaultdict(lambda: [0.0, 0])
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            k = klass(r["Kernel_Name"])
            if k:
                agg[k][0] += float(r["Counter_Value"])
                agg[k][1] += 1
    return agg


def durations(d):
    """Average kernel duration per class (us) and launches from a `rocprofv3 --kernel-trace --output-format csv` pass of the same command."""
    agg = collections.defaultdict(lambda: [0.0, 0])
    for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = klass(r["Kernel_Name"])
            if k:
                agg[k][0] += (float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) * 1e-3
                agg[k][1] += 1
    return agg


def main():
    out = sys.argv[1]
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=1)
    ap.add_argument("--nfe", type=int, default=16)
    ap.add_argument("--precision", default="fp16m")  # bench.py's default
    ap.add_argument("--model", default="F5TTS_v1_Base")
    a, _ = ap.parse_known_args(sys.argv[2:])
    fetch, write = load(os.path.join(out, "fetch"), "FETCH_SIZE"), load(os.path.join(out, "write"), "WRITE_SIZE")
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench  # kernel_source_hash(): bench.py quotes this summary only while the kernel sources are the ones it was taken from

    res = {"precision": a.precision, "batch": a.batch, "nfe": a.nfe, "model": a.model, "kernel_source_hash": bench.kernel_source_hash(),
           "classes": {}}
    for k in sorted(set(fetch) | set(write)):
        f, w = fetch.get(k, [0, 0]), write.get(k, [0, 0])
        res["classes"][k] = {"launches": f[1] or w[1], "fetch_bytes_per_launch_x2": 2 * 1024 * f[0] / max(f[1], 1),
                             "write_bytes_per_launch": 1024 * w[0] / max(w[1], 1)}
    # MFMA pipe utilisation per class: SQ_VALU_MFMA_BUSY_CYCLES (cycles, summed over the chip's 1024 SIMDs) / (GRBM_GUI_ACTIVE / 8 XCDs
    # x 1024): the fraction of SIMD-cycles in which the matrix pipe was busy, at the clock the kernel actually ran at
    busy, act = load(os.path.join(out, "mfma"), "SQ_VALU_MFMA_BUSY_CYCLES"), load(os.path.join(out, "mfma"), "GRBM_GUI_ACTIVE")
    for k in busy:
        if k in res["classes"] and act.get(k, [0, 0])[0] > 0:
            res["classes"][k]["mfma_busy_frac"] = round(busy[k][0] / (act[k][0] / 8.0 * 1024.0), 4)
    # achieved HBM rate of every class: counter bytes per launch over the average duration of the un-instrumented kernel-trace pass
    dur = durations(os.path.join(out, "trace"))
    for k, c in res["classes"].items():
        if dur.get(k, [0, 0])[1]:
            c["avg_us"] = round(dur[k][0] / dur[k][1], 3)
            c["hbm_gbps"] = round((c["fetch_bytes_per_launch_x2"] + c["write_bytes_per_launch"]) / (c["avg_us"] * 1e-6) / 1e9, 1)
            c["hbm_frac_of_8TBps"] = round(c["hbm_gbps"] / 8000.0, 4)
    dom = "gemm_" + a.precision
    if dom in res["classes"]:
        c = res["classes"][dom]
        res["dominant"] = dom
        res["hbm_bytes_per_launch"] = c["fetch_bytes_per_launch_x2"] + c["write_bytes_per_launch"]
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
