#!/bin/bash
# k-step-split tiles (68-70) against the one-round tiles: correctness (KB_CHECK values, qkv check, 3 reps) and time
set -u
out=$GRAFT_REPO_ROOT/gpurun_out/r2c21; mkdir -p $out
cd $GRAFT_REPO_ROOT
{
echo "== KB_CHECK"
for epi in 1 2; do
KB_CHECK=1 KB_SHAPES="2812,3072,1024;2812,2048,1024;1406,1024,2048" KB_PRECS=fp16x3,fp16 KB_EPI=$epi KB_VARIANTS=68,69,70 timeout 300 python tools/kernel_bench.py gemm 2>&1 | grep -E "KB_CHECK|^gemm" | cut -c1-220
done
echo "== qkv check + time"
for sq in "2 1406" "1 1406" "3 1000"; do timeout 200 python tools/kernel_bench.py qkv fp16x3 $sq 55,56,68,69,70 30 2>&1 | grep -E "^qkv|QKV_CHECK" ; done
echo "== time"
for rep in 1 2; do
KB_SHAPES="2812,3072,1024;2812,2048,1024;1406,3072,1024;1406,2048,1024" KB_PRECS=fp16x3 KB_EPI=1 KB_VARIANTS=55,56,66,68,69,70 timeout 300 python tools/kernel_bench.py gemm 2>&1 | grep ^gemm | cut -c1-260
KB_SHAPES="2812,1024,1024;2812,1024,2048;1406,1024,1024;1406,1024,2048" KB_PRECS=fp16x3 KB_EPI=2 KB_VARIANTS=59,66,69,70 timeout 300 python tools/kernel_bench.py gemm 2>&1 | grep ^gemm | cut -c1-260
done
} > $out/kss.log 2>&1
cat $out/kss.log
