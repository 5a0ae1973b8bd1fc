#!/bin/bash
# what does the row-strided store pattern of the FF1 epilogue cost?  real / no stores / the same bytes stored lane-linearly
set -u
out=$GRAFT_REPO_ROOT/gpurun_out/r2c26; mkdir -p $out
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
KB_SHAPES="89984,2048,1024;22496,2048,1024;2812,2048,1024" KB_PRECS=fp16x3 KB_EPI=1 KB_VARIANTS=50,1050,2050,3050 timeout 300 python tools/kernel_bench.py gemm 2>&1 | grep ^gemm | cut -c1-260
KB_SHAPES="2812,2048,1024;2812,3072,1024" KB_PRECS=fp16x3 KB_EPI=1 KB_VARIANTS=56,1056,2056,3056 timeout 300 python tools/kernel_bench.py gemm 2>&1 | grep ^gemm | cut -c1-260
done > $out/dense.log 2>&1
cat $out/dense.log
