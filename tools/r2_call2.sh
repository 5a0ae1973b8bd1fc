#!/bin/bash
# Round 2, GPU call 2: first light of the pipelined GEMM (gemm_pp.h): correctness against the generic kernel, then time per tile variant.
set -u
out=$GRAFT_REPO_ROOT/gpurun_out/r2c2; mkdir -p $out
cd $GRAFT_REPO_ROOT
for v in 50 51 52 53 54 55 56 57 58 59 60; do for epi in 1 2; do
  KB_CHECK=1 KB_EPI=$epi timeout 120 python tools/kernel_bench.py one fp16x3 $v 2812 2048 1024 3 2>&1 | grep -E "KB_CHECK|^gemm"
done; done > $out/check_x3.log 2>&1
for v in 50 53 56; do KB_CHECK=1 KB_EPI=2 timeout 120 python tools/kernel_bench.py one fp16 $v 2812 3072 1024 3 2>&1 | grep -E "KB_CHECK|^gemm"; done > $out/check_f16.log 2>&1
B1="2812,3072,1024;2812,1024,1024;2812,2048,1024;2812,1024,2048;1406,3072,1024;1406,1024,1024;1406,2048,1024;1406,1024,2048;5624,3072,1024;5624,1024,2048"
BIG="22496,3072,1024;22496,1024,2048;89984,2048,1024;89984,3072,1024;89984,1024,1024;89984,1024,2048"
for epi in 1 2; do
  KB_SHAPES=$B1 KB_PRECS=fp16x3,fp16 KB_EPI=$epi KB_VARIANTS=1,6,53,54,55,56,57,58,59,60,51,52 timeout 400 python tools/kernel_bench.py gemm 2>&1 | grep ^gemm | sed "s/^/epi$epi /"
done > $out/kb_b1.log 2>&1
for epi in 1 2; do
  KB_SHAPES=$BIG KB_PRECS=fp16x3,fp16 KB_EPI=$epi KB_VARIANTS=2,50,51,52,60,57,58 timeout 400 python tools/kernel_bench.py gemm 2>&1 | grep ^gemm | sed "s/^/epi$epi /"
done > $out/kb_big.log 2>&1
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "base_v1 or e2_base or reference_golden" > $out/parity.log 2>&1; tail -5 $out/parity.log
timeout 300 python bench.py --schedule default --steps 10 --warmup 3 --no-cpu-baseline > $out/bench_b1.json 2> $out/bench_b1.err
timeout 300 python bench.py --schedule default --steps 10 --warmup 3 --no-cpu-baseline --branch-streams 0 > $out/bench_b1_packed.json 2> $out/bench_b1_packed.err
timeout 600 python bench.py --schedule default --batch 32 --nfe 32 --steps 2 --warmup 1 --no-cpu-baseline > $out/bench_b32.json 2> $out/bench_b32.err
cat $out/check_x3.log $out/check_f16.log | grep -v "0 of" | head -40
python - <<PY
import json
for f in ("bench_b1","bench_b1_packed","bench_b32"):
    try:
        d=json.loads(open("$out/%s.json"%f).read().strip().splitlines()[-1]); print(f, d["ms_per_step"], d["kernel_classes_ms"])
    except Exception as e: print(f, "ERR", e, open("$out/%s.err"%f).read()[-500:])
PY
