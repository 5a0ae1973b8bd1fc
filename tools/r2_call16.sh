#!/bin/bash
# k-split tiles (65-67) against the one-round tiles they would replace: correctness (KB_CHECK values, qkv check) and time
set -u
out=$GRAFT_REPO_ROOT/gpurun_out/r2c16; mkdir -p $out
cd $GRAFT_REPO_ROOT
{
echo "== KB_CHECK"
for epi in 1 2; do
KB_CHECK=1 KB_SHAPES="2812,2048,1024;1406,1024,2048" KB_PRECS=fp16x3 KB_EPI=$epi KB_VARIANTS=65,66,67 timeout 300 python tools/kernel_bench.py gemm 2>&1 | grep -E "KB_CHECK.*rep 0|^gemm" | cut -c1-220
done
echo "== qkv check"
for sq in "2 1406" "1 1406"; do timeout 200 python tools/kernel_bench.py qkv fp16x3 $sq 65,66,67 20 2>&1 | grep -E "^qkv|QKV_CHECK" ; done
echo "== time FF1 (epi 1: N 2048 K 1024), out/FF2 (epi 2: N 1024, K 1024 / 2048), QKV-like N 3072"
for rep in 1 2; do
KB_SHAPES="2812,2048,1024;1406,2048,1024;2812,3072,1024;1406,3072,1024" KB_PRECS=fp16x3 KB_EPI=1 KB_VARIANTS=55,56,59,65,66,67 timeout 300 python tools/kernel_bench.py gemm 2>&1 | grep ^gemm | cut -c1-260
KB_SHAPES="2812,1024,1024;2812,1024,2048;1406,1024,1024;1406,1024,2048" KB_PRECS=fp16x3 KB_EPI=2 KB_VARIANTS=55,59,65,66,67 timeout 300 python tools/kernel_bench.py gemm 2>&1 | grep ^gemm | cut -c1-260
done
echo "== qkv time"
for sq in "2 1406" "1 1406"; do timeout 200 python tools/kernel_bench.py qkv fp16x3 $sq 55,56,65,66,67 50 2>&1 | grep -E "^qkv"; done
} > $out/ksplit.log 2>&1
cat $out/ksplit.log
