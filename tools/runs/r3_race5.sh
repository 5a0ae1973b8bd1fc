#!/bin/bash
# round 3, call 5: is the lost sine product tied to v_permlane32_swap (13) or to the packed fp32 instructions (14)?
set -u
out=$GRAFT_REPO_ROOT/gpurun_out/r3c5; mkdir -p $out
cd $GRAFT_REPO_ROOT
timeout 600 python tools/kernel_bench.py qkvprobe 2 1406 6 "58,1;58,13;58,14;62,13;62,14;63,13;63,14" > $out/probe.log 2>&1
grep -E "qkvprobe|QKV_PROBE|rror" $out/probe.log
