#!/bin/bash
set -u
out=$GRAFT_REPO_ROOT/gpurun_out/r2c13; mkdir -p $out
cd $GRAFT_REPO_ROOT
for st in -1 0 10000 20000 40000 80000; do
  echo "== stagger $st"
  if [ $st = -1 ]; then E="X=1"; else E="F5HIP_PP_STAGGER=$st"; fi
  env $E KB_SHAPES="89984,2048,1024;89984,3072,1024;89984,1024,2048;22496,3072,1024" KB_PRECS=fp16x3 KB_EPI=1 KB_VARIANTS=50,51 timeout 300 python tools/kernel_bench.py gemm 2>&1 | grep ^gemm | cut -c1-200
  env $E timeout 300 python tools/kernel_bench.py qkv fp16x3 64 1406 50 5 2>&1 | grep "^qkv" | tail -1
done > $out/stagger.log 2>&1
cat $out/stagger.log
