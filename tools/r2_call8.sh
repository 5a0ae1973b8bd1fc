#!/bin/bash
set -u
out=$GRAFT_REPO_ROOT/gpurun_out/r2c8; mkdir -p $out
cd $GRAFT_REPO_ROOT
r() { echo "== $*"; env "$@" timeout 300 python tools/kernel_bench.py qkv fp16x3 8 1406 58,63 5 2>&1 | grep -E "^qkv" | awk '{print $3, $NF}'; }
{ r X=1; r F5HIP_PP_LDS_PAD=32768; r F5HIP_PP_EXP=1; r F5HIP_PP_EXP=2; r F5HIP_PP_EXP=3; } > $out/exp.log 2>&1
cat $out/exp.log
