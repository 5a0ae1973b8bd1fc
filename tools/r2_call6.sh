#!/bin/bash
set -u
out=$GRAFT_REPO_ROOT/gpurun_out/r2c6; mkdir -p $out
cd $GRAFT_REPO_ROOT
for v in 58 62 63 64 61 50; do echo "== N3072 via $v"; F5HIP_PP_VARIANT_N3072=$v F5HIP_PP_VARIANT_N2048=0 F5HIP_PP_VARIANT_N1024=0 timeout 300 python tools/bisect_b4.py 2>&1 | grep branch_streams; done > $out/bisect.log 2>&1
cat $out/bisect.log
