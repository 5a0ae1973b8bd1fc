#!/bin/bash
set -u
out=$GRAFT_REPO_ROOT/gpurun_out/r2c7; mkdir -p $out
cd $GRAFT_REPO_ROOT
for sq in "2 1406" "4 1406" "8 1406" "8 1407"; do set -- $sq
  timeout 300 python tools/kernel_bench.py qkv fp16x3 $1 $2 50,56,55,59,58,61,62,63,64,57,53 5 2>&1 | grep -E "^qkv|QKV_CHECK" 
done > $out/qkv.log 2>&1
grep -v "differing halves 0$" $out/qkv.log | head -60; echo; grep -c "differing halves 0$" $out/qkv.log
