"""Kernel microbenchmarks through the C ABI (f5hip_bench_gemm / f5hip_bench_attention): TFLOP/s per tile variant and shape.
Run on the GPU box:  python tools/kernel_bench.py [gemm|attn|all]"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402,F401  (initialises the HIP runtime the same way the product does)

import f5_tts_amd  # noqa: E402,F401
from f5_tts_amd import config  # noqa: E402
from f5_tts_amd.binding import PRECISIONS  # noqa: E402
from f5_tts_amd.engine import F5HipEngine  # noqa: E402

VARIANTS = {-1: "auto", 0: "64x128", 1: "128x64", 2: "128x128", 6: "glds128x64", 8: "64x64",
            50: "pp256x256w8", 51: "pp256x128w8", 52: "pp128x256w8", 53: "pp192x64w4", 54: "pp192x128w4", 55: "pp192x128w8", 56: "pp192x192w4",
            57: "pp128x128s3", 58: "pp128x128s2", 59: "pp96x128w4", 60: "pp256x128w4",
            61: "pp128x192s2", 62: "pp192x128s2", 63: "pp96x128s2", 64: "pp192x64s2",
            65: "pp192x128k2", 66: "pp96x128k2", 67: "pp192x64k2",
            68: "pp192x192ks", 69: "pp192x128ks", 70: "pp192x64ks", 80: "p8-256x256",
            1080: "p8:noEpi", 4080: "p8:noDMA", 8080: "p8:noMFMA", 9080: "p8:noEpi+MFMA"}
for _id in (50, 56, 59):
    for _c, _n in ((1, "noEpi"), (2, "noStore"), (3, "denseStore"), (4, "noDMA"), (8, "noMFMA"), (12, "noDMA+MFMA"), (13, "reads+barriers")):
        VARIANTS[1000 * _c + _id] = f"{VARIANTS[_id]}:{_n}"


def main():
    what = sys.argv[1] if len(sys.argv) > 1 else "all"
    global VSET, EPI
    EPI = int(os.environ.get("KB_EPI", "1"))
    VSET = tuple(int(x) for x in os.environ.get("KB_VARIANTS", "1,2,3,4,5").split(","))
    eng = F5HipEngine(config.DIT_TINY, None, device=0)
    lib, ctx = eng.bench_lib, eng._ctx  # include/f5hip_bench.h (libf5hip_bench.so)
    ms = C.c_double()
    if what == "one":  # one gemm config: one <prec> <variant> M N K [iters]   (for rocprofv3 --pmc runs)
        prec, v, M, N, K = sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]), int(sys.argv[6])
        iters = int(sys.argv[7]) if len(sys.argv) > 7 else 10
        st = lib.f5hip_bench_gemm(ctx, PRECISIONS[prec], v, EPI, M, N, K, iters, C.byref(ms))
        print(f"gemm {prec} {VARIANTS[v]} M={M} N={N} K={K}: st={st} {ms.value * 1e3:.1f} us {2.0 * M * N * K / ms.value / 1e9:.1f} TF")
        eng.close()
        return
    if what == "qkv":  # qkv <prec> <seqs> <nseq> <variants comma> [iters]: time + byte comparison with the generic kernel's planes
        prec, seqs, nseq = sys.argv[2], int(sys.argv[3]), int(sys.argv[4])
        iters = int(sys.argv[6]) if len(sys.argv) > 6 else 10
        diff = C.c_int64()
        for v in (int(x) for x in sys.argv[5].split(",")):
            for rep in range(3):
                st = lib.f5hip_bench_qkv(ctx, PRECISIONS[prec], v, seqs, nseq, 1024, iters, 1, C.byref(ms), C.byref(diff))
                print(f"qkv {prec} {VARIANTS.get(v, v)} seqs={seqs} nseq={nseq}: st={st} {ms.value * 1e3:.1f} us differing halves {diff.value}", flush=True)
        eng.close()
        return
    if what == "qkvprobe":  # qkvprobe <seqs> <nseq> <reps> <spec;spec;..>  spec = variant,expt[,abl[,lds_pad[,noise]]]  (csrc/race_probe.hip)
        seqs, nseq, reps = int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
        dump = os.environ.get("KB_PROBE_DUMP", "")
        bad = (C.c_int64 * reps)()
        for spec in sys.argv[5].split(";"):
            f = [int(x) for x in spec.split(",")] + [0, 0, 0]
            v, expt, abl, pad, noise = f[:5]
            path = f"{dump}.v{v}e{expt}a{abl}p{pad}n{noise}.bin" if dump else ""
            st = lib.f5hip_bench_qkv_probe(ctx, v, expt, abl, pad, noise, seqs, nseq, reps, bad, path.encode() if path else None)
            print(f"qkvprobe tile {v} expt {expt} abl {abl} lds_pad {pad} noise {noise} seqs={seqs} nseq={nseq}: st={st} wrong outputs per launch {list(bad)}", flush=True)
        eng.close()
        return
    if what == "oneattn":  # oneattn <prec> <batch2> <n> [iters]
        prec, b2, n = sys.argv[2], int(sys.argv[3]), int(sys.argv[4])
        iters = int(sys.argv[5]) if len(sys.argv) > 5 else 10
        st = lib.f5hip_bench_attention(ctx, PRECISIONS[prec], b2, 16, n, iters, C.byref(ms))
        print(f"attn {prec} B'={b2} n={n}: st={st} {ms.value * 1e3:.1f} us {4.0 * b2 * 16 * n * n * 64 / ms.value / 1e9:.1f} TF")
        eng.close()
        return
    if what in ("gemm", "all"):
        shapes = [(2812, 3072, 1024, "QKV  B=1"), (2812, 1024, 1024, "out  B=1"), (2812, 2048, 1024, "FF1  B=1"), (2812, 1024, 2048, "FF2  B=1"),
                  (22496, 3072, 1024, "QKV  B=8"), (22496, 1024, 2048, "FF2  B=8"), (89984, 2048, 1024, "FF1  B=32")]
        if os.environ.get("KB_SHAPES"):  # "M,N,K;M,N,K;..."
            shapes = [tuple(int(x) for x in sh.split(",")) + ("custom  ",) for sh in os.environ["KB_SHAPES"].split(";")]
        for prec in os.environ.get("KB_PRECS", "fp16x3,fp16").split(","):
            for M, N, K, tag in shapes:
                if prec == "fp32" and M > 30000:
                    continue
                row = []
                for v in VSET:
                    iters = 20 if M < 30000 else 5
                    st = lib.f5hip_bench_gemm(ctx, PRECISIONS[prec], v, EPI, M, N, K, iters, C.byref(ms))
                    row.append(f"{VARIANTS[v]} {ms.value * 1e3:7.1f}us {2.0 * M * N * K / ms.value / 1e9:6.1f}TF" if st == 0 else f"{VARIANTS[v]} ERR{st}")
                print(f"gemm {prec:7s} {tag} M={M:6d} N={N:5d} K={K:5d} | " + " | ".join(row), flush=True)
    if what in ("attn", "all"):
        for prec in ("fp16", "fp16x3"):
            for b2, n in ((2, 1406), (16, 1406), (64, 1406), (2, 3000)):
                st = lib.f5hip_bench_attention(ctx, PRECISIONS[prec], b2, 16, n, 10, C.byref(ms))
                fl = 4.0 * b2 * 16 * n * n * 64
                print(f"attn {prec:7s} B'={b2:3d} H=16 n={n}: {ms.value * 1e3:8.1f} us  {fl / ms.value / 1e9:7.1f} TF" if st == 0 else f"attn ERR {st}", flush=True)
    eng.close()


if __name__ == "__main__":
    main()
