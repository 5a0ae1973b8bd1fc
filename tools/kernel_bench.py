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

VARIANTS = {-1: "auto", 0: "64x128", 1: "128x64", 2: "128x128", 3: "256x128", 4: "128x256", 5: "256x256", 6: "glds128x64", 7: "glds128x128", 8: "64x64", 10: "128x192", 13: "glds256x128w4", 21: "glds256sq_w8a", 22: "glds256sq_w8b", 23: "glds256sq_prio", 26: "glds128x64w2", 27: "glds128x64w2s2", 28: "glds128x128s2", 29: "glds64x128w2", 30: "glds256x128w8", 31: "glds128x256w8", 40: "sk256x128", 41: "sk128x256", 42: "skrs256x128", 43: "skrs128x256", 24: "glds128x64_2of3", 25: "glds256sq_2of3", 14: "glds128x256w4", 9: "noLoad", 10: "noLdsSt", 11: "noLd+noSt", 12: "noMFMA", 15: "noAll", 17: "s:noEpi", 18: "s:noMFMA", 19: "s:noAll", 20: "s:noStage"}


def main():
    what = sys.argv[1] if len(sys.argv) > 1 else "all"
    global VSET, EPI
    EPI = int(os.environ.get("KB_EPI", "1"))
    VSET = tuple(int(x) for x in os.environ.get("KB_VARIANTS", "1,2,3,4,5").split(","))
    eng = F5HipEngine(config.DIT_TINY, None, device=0)
    lib, ctx = eng.lib, eng._ctx
    ms = C.c_double()
    if what == "one":  # one gemm config: one <prec> <variant> M N K [iters]   (for rocprofv3 --pmc runs)
        prec, v, M, N, K = sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]), int(sys.argv[6])
        iters = int(sys.argv[7]) if len(sys.argv) > 7 else 10
        st = lib.f5hip_bench_gemm(ctx, PRECISIONS[prec], v, EPI, M, N, K, iters, C.byref(ms))
        print(f"gemm {prec} {VARIANTS[v]} M={M} N={N} K={K}: st={st} {ms.value * 1e3:.1f} us {2.0 * M * N * K / ms.value / 1e9:.1f} TF")
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
