#!/bin/bash
# round 3, call 11: the two-workgroups-per-CU tiles (58, 61-64) against the production tiles at B = 2 .. 32 (fp16x3 and fp16), all four block GEMM shapes
set -u
out=$GRAFT_REPO_ROOT/gpurun_out/r3c11; mkdir -p $out
cd $GRAFT_REPO_ROOT
{
for sq in "4 1406" "16 1406" "64 1406"; do timeout 300 python tools/kernel_bench.py qkv fp16x3 $sq -1,55,50,58,61,62 5 2>&1 | grep -E "^qkv" | awk 'NR%3==1'; done
for M in 5624 11248 22496 44992 89984; do
KB_SHAPES="$M,2048,1024" KB_PRECS=fp16x3,fp16 KB_EPI=1 KB_VARIANTS=-1,51,50,58,61,62,63 timeout 300 python tools/kernel_bench.py gemm 2>&1 | grep ^gemm | cut -c1-300
KB_SHAPES="$M,1024,1024;$M,1024,2048" KB_PRECS=fp16x3,fp16 KB_EPI=2 KB_VARIANTS=-1,51,50,58,61,62,63 timeout 300 python tools/kernel_bench.py gemm 2>&1 | grep ^gemm | cut -c1-300
done
} > $out/kb.log 2>&1
cat $out/kb.log
