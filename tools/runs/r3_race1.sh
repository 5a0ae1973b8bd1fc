#!/bin/bash
# round 3, call 1: does round 2's co-residency fault still reproduce, and which variation of the epilogue removes it?
set -u
out=$GRAFT_REPO_ROOT/gpurun_out/r3c1; mkdir -p $out
cd $GRAFT_REPO_ROOT
export KB_PROBE_DUMP=$out/dump
S="58,0;58,1;58,2;58,3;58,4;58,5;58,6;58,1,4;58,1,8;58,1,0,32768;62,0;62,1;63,0;63,1;61,0;61,1;57,1;57,1,0,0,1;57,1,0,0,2;58,0,0,0,1;58,0,0,0,2;58,1,0,32768,2"
timeout 600 python tools/kernel_bench.py qkvprobe 2 1406 6 "$S" 2>&1 | grep -E "qkvprobe|Error|error" > $out/probe_s2.log
KB_PROBE_DUMP=$out/dump8 timeout 600 python tools/kernel_bench.py qkvprobe 8 1406 4 "58,0;58,1;58,2;58,3;58,4;58,6;62,1;63,1;61,1" 2>&1 | grep -E "qkvprobe|Error|error" > $out/probe_s8.log
cat $out/probe_s2.log $out/probe_s8.log
for f in $out/dump.v58e1a0p0n0.bin; do [ -s $f ] && python tools/race_dump_analyze.py $f 2 1406 30 > $out/analyze_v58e1.txt 2>&1; done
ls -la $out | head -40
timeout 600 python bench.py > $out/bench_b1.json 2> $out/bench_b1.err; tail -c 600 $out/bench_b1.json
