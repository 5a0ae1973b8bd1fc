#!/bin/bash
# Round 2, GPU call 3: ablations of the pipelined GEMM (where does a launch's time go) + end-to-end with the measured tile table.
set -u
out=$GRAFT_REPO_ROOT/gpurun_out/r2c3; mkdir -p $out
cd $GRAFT_REPO_ROOT
KB_SHAPES="2812,3072,1024;2812,2048,1024;1406,3072,1024" KB_PRECS=fp16x3 KB_EPI=1 KB_VARIANTS=56,1056,2056,4056,8056,12056,13056 timeout 300 python tools/kernel_bench.py gemm 2>&1 | grep ^gemm > $out/abl.log
KB_SHAPES="2812,1024,1024;2812,1024,2048;1406,1024,1024" KB_PRECS=fp16x3 KB_EPI=1 KB_VARIANTS=59,1059,2059,4059,8059,12059,13059 timeout 300 python tools/kernel_bench.py gemm 2>&1 | grep ^gemm >> $out/abl.log
KB_SHAPES="89984,2048,1024;22496,3072,1024" KB_PRECS=fp16x3 KB_EPI=1 KB_VARIANTS=50,1050,2050,4050,8050,12050,13050 timeout 300 python tools/kernel_bench.py gemm 2>&1 | grep ^gemm >> $out/abl.log
# pinned fragment reads: the tiles of the table again
KB_SHAPES="2812,3072,1024;2812,1024,1024;2812,2048,1024;2812,1024,2048;1406,3072,1024;1406,1024,1024;1406,2048,1024;1406,1024,2048" KB_PRECS=fp16x3 KB_EPI=1 KB_VARIANTS=53,55,56,58,59 timeout 300 python tools/kernel_bench.py gemm 2>&1 | grep ^gemm > $out/kb_b1.log
KB_SHAPES="22496,3072,1024;89984,2048,1024;89984,1024,2048" KB_PRECS=fp16x3,fp16 KB_EPI=1 KB_VARIANTS=50,51,58 timeout 300 python tools/kernel_bench.py gemm 2>&1 | grep ^gemm > $out/kb_big.log
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "base_v1 or e2_base or reference_golden" > $out/parity.log 2>&1; tail -3 $out/parity.log
timeout 300 python bench.py --schedule default --steps 10 --warmup 3 --no-cpu-baseline > $out/bench_b1.json 2> $out/bench_b1.err
timeout 300 python bench.py --schedule default --steps 10 --warmup 3 --no-cpu-baseline --branch-streams 0 > $out/bench_b1_packed.json 2> $out/bench_b1_packed.err
timeout 600 python bench.py --schedule default --batch 32 --nfe 32 --steps 2 --warmup 1 --no-cpu-baseline > $out/bench_b32.json 2> $out/bench_b32.err
cut -c1-600 $out/abl.log
python - <<PY
import json
for f in ("bench_b1","bench_b1_packed","bench_b32"):
    try:
        d=json.loads(open("$out/%s.json"%f).read().strip().splitlines()[-1]); print(f, d["ms_per_step"], d["kernel_classes_ms"])
    except Exception as e: print(f, "ERR", e, open("$out/%s.err"%f).read()[-500:])
PY
