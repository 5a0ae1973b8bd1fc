#!/bin/bash
# round 3, call 10: the library built with -packed-fp32-ops: GPU suite, B = 1 bench, the two-workgroups-per-CU tiles against the production tiles
set -u
out=$GRAFT_REPO_ROOT/gpurun_out/r3c10; mkdir -p $out
cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests -q -m gpu -x > $out/gpu_tests.log 2>&1; tail -4 $out/gpu_tests.log
timeout 600 python bench.py --no-cpu-baseline > $out/bench_b1.json 2> $out/bench_b1.err; python - <<PY
import json
d=json.loads(open("$out/bench_b1.json").read().strip().splitlines()[-1]); print("B=1", round(d["ms_per_step"],2), d["kernel_classes_ms"])
PY
{
for sq in "2 1406" "8 1406"; do timeout 300 python tools/kernel_bench.py qkv fp16x3 $sq -1,68,56,55,58,61,62,63,64 20 2>&1 | grep -E "^qkv|QKV_CHECK"; done
B1="2812,2048,1024;1406,2048,1024;1406,3072,1024"
KB_SHAPES=$B1 KB_PRECS=fp16x3 KB_EPI=1 KB_VARIANTS=-1,55,69,58,61,62,63,64 timeout 300 python tools/kernel_bench.py gemm 2>&1 | grep ^gemm | cut -c1-330
B2="2812,1024,1024;2812,1024,2048;1406,1024,1024;1406,1024,2048"
KB_SHAPES=$B2 KB_PRECS=fp16x3 KB_EPI=2 KB_VARIANTS=-1,59,66,58,62,63,64 timeout 300 python tools/kernel_bench.py gemm 2>&1 | grep ^gemm | cut -c1-330
} > $out/kb.log 2>&1
cat $out/kb.log
timeout 300 python tools/kernel_bench.py attn 2>&1 | grep -E "^attn" | head -12 > $out/attn.log; cat $out/attn.log
