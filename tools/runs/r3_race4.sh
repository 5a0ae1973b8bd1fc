#!/bin/bash
# round 3, call 4: the fetch + wait + copy sequence pinned in asm: does a register read right after s_waitcnt vmcnt(0) see other data than 32 cycles later?
set -u
out=$GRAFT_REPO_ROOT/gpurun_out/r3c4; mkdir -p $out
cd $GRAFT_REPO_ROOT
export KB_PROBE_DUMP=$out/dump
timeout 600 python tools/kernel_bench.py qkvprobe 2 1406 6 "58,1;58,11;58,12;58,11,0,32768;62,11;62,12;63,11;63,12;61,11" > $out/probe.log 2>&1
grep -E "qkvprobe|QKV_PROBE|rror" $out/probe.log
for t in v58e11 v62e11; do f=$out/dump.${t}a0p0n0.bin; [ -s $f ] && python tools/race_dump_analyze.py $f 2 1406 12 > $out/analyze_$t.txt 2>&1; done
head -c 3000 $out/analyze_v58e11.txt
rm -f $out/*.bin
