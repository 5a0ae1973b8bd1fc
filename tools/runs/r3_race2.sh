#!/bin/bash
# round 3, call 2: where are the wrong outputs of the co-residency fault, and do idle cycles / store drains / defined values remove it?
set -u
out=$GRAFT_REPO_ROOT/gpurun_out/r3c3; mkdir -p $out
cd $GRAFT_REPO_ROOT
export KB_PROBE_DUMP=$out/dump
timeout 600 python tools/kernel_bench.py qkvprobe 2 1406 6 "58,1;62,1;63,1" 2>&1 | grep -E "qkvprobe|Error|error" > $out/probe_s2.log
cat $out/probe_s2.log
for t in v58e1 v62e1 v63e1; do f=$out/dump.${t}a0p0n0.bin; [ -s $f ] && python tools/race_dump_analyze.py $f 2 1406 24 > $out/analyze_$t.txt 2>&1; done
head -c 5000 $out/analyze_v58e1.txt; head -c 2500 $out/analyze_v62e1.txt; head -c 2500 $out/analyze_v63e1.txt
rm -f $out/*.bin
