#!/bin/bash
# round 3, call 7: the packed rotation pinned in asm (failing form 18, + idle cycles 19, fresh copy 20, crossing on src0 21)
set -u
out=$GRAFT_REPO_ROOT/gpurun_out/r3c7; mkdir -p $out
cd $GRAFT_REPO_ROOT
timeout 600 python tools/kernel_bench.py qkvprobe 2 1406 6 "58,1;58,18;58,19;58,20;58,21;58,18,0,32768;62,18;62,21;63,18;63,21;61,18" > $out/probe.log 2>&1
grep -E "qkvprobe|QKV_PROBE|rror" $out/probe.log
