#!/bin/bash
# round 3, call 9: sweep of the packed-fp32 operand selections with an MFMA + LDS-read partner on the SIMD, and the control without partners
set -u
out=$GRAFT_REPO_ROOT/gpurun_out/r3c9; mkdir -p $out
cd $GRAFT_REPO_ROOT/tools/probes
timeout 300 ./pk_opsel_sweep 4000 1 > $out/sweep_partners.log 2>&1
timeout 300 ./pk_opsel_sweep 4000 0 > $out/sweep_alone.log 2>&1
grep -E "FAILS|forms fail" $out/sweep_partners.log; tail -1 $out/sweep_alone.log
