#!/bin/bash
# Round 2, GPU call 4: two-per-CU tiles, end-to-end table choices under the one-chain / two-chain schedules, full GPU test suite.
set -u
out=$GRAFT_REPO_ROOT/gpurun_out/r2c4; mkdir -p $out
cd $GRAFT_REPO_ROOT
B1="2812,3072,1024;2812,1024,1024;2812,2048,1024;2812,1024,2048;1406,3072,1024;1406,1024,1024;1406,2048,1024;1406,1024,2048"
for epi in 1 2; do
KB_SHAPES=$B1 KB_PRECS=fp16x3 KB_EPI=$epi KB_VARIANTS=53,55,56,58,59,61,62,63,64 timeout 300 python tools/kernel_bench.py gemm 2>&1 | grep ^gemm | sed "s/^/epi$epi /"
done > $out/kb_b1.log
KB_SHAPES="22496,3072,1024;22496,1024,2048;89984,2048,1024;89984,1024,2048;89984,3072,1024" KB_PRECS=fp16x3,fp16 KB_EPI=1 KB_VARIANTS=50,51,58,61,62 timeout 300 python tools/kernel_bench.py gemm 2>&1 | grep ^gemm > $out/kb_big.log
run() { tag=$1; extra=$2; shift; shift; env "$@" timeout 300 python bench.py --steps 8 --warmup 2 --no-cpu-baseline $extra > $out/bench_$tag.json 2> $out/bench_$tag.err; python - <<PY
import json
try:
    d=json.loads(open("$out/bench_$tag.json").read().strip().splitlines()[-1]); print("$tag", round(d["ms_per_step"],2), d["kernel_classes_ms"]["gemm_block"])
except Exception as e: print("$tag", "ERR", e, open("$out/bench_$tag.err").read()[-300:])
PY
}
run default "" X=1
run packed "--branch-streams 0" X=1
run two_58 "--branch-streams 1" F5HIP_PP_VARIANT=58
run two_61 "--branch-streams 1" F5HIP_PP_VARIANT_N3072=61 F5HIP_PP_VARIANT_N2048=61 F5HIP_PP_VARIANT_N1024=63
run two_63 "--branch-streams 1" F5HIP_PP_VARIANT_N3072=61 F5HIP_PP_VARIANT_N2048=63 F5HIP_PP_VARIANT_N1024=63
run two_64 "--branch-streams 1" F5HIP_PP_VARIANT_N3072=62 F5HIP_PP_VARIANT_N2048=62 F5HIP_PP_VARIANT_N1024=64
timeout 900 python -m pytest tests -q -m gpu > $out/gpu_tests.log 2>&1; tail -5 $out/gpu_tests.log
cut -c1-400 $out/kb_b1.log; cut -c1-300 $out/kb_big.log
