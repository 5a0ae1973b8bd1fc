#!/bin/bash
# deeper DMA rings (71-74) against the 3-stage / k-split tiles: value check, then time
set -u
out=$GRAFT_REPO_ROOT/gpurun_out/r2c25; mkdir -p $out
cd $GRAFT_REPO_ROOT
{
KB_CHECK=1 KB_SHAPES="2812,2048,1024;1406,1024,2048" KB_PRECS=fp16x3 KB_EPI=2 KB_VARIANTS=71,72,73,74 timeout 300 python tools/kernel_bench.py gemm 2>&1 | grep -E "KB_CHECK.*rep 0|^gemm" | cut -c1-220
for rep in 1 2; do
KB_SHAPES="2812,2048,1024;2812,3072,1024;1406,2048,1024" KB_PRECS=fp16x3 KB_EPI=1 KB_VARIANTS=55,69,72,74,66,70,71,73 timeout 300 python tools/kernel_bench.py gemm 2>&1 | grep ^gemm | cut -c1-330
KB_SHAPES="2812,1024,1024;2812,1024,2048;1406,1024,1024;1406,1024,2048" KB_PRECS=fp16x3 KB_EPI=2 KB_VARIANTS=59,66,70,71,73 timeout 300 python tools/kernel_bench.py gemm 2>&1 | grep ^gemm | cut -c1-260
done
} > $out/deep.log 2>&1
cat $out/deep.log
