#!/bin/bash
# round 3, call 6: the packed op_sel sequence alone (no GEMM): 1 vs 2 waves per SIMD, with / without an MFMA partner and streaming loads
set -u
out=$GRAFT_REPO_ROOT/gpurun_out/r3c6; mkdir -p $out
cd $GRAFT_REPO_ROOT/tools/probes
{ for m in 0 2 1 3; do timeout 120 ./pk_opsel_probe 2 $m 20000; done; timeout 120 ./pk_opsel_probe 1 0 20000; timeout 120 ./pk_opsel_probe 1 2 20000; } > $out/pk_probe.log 2>&1
cat $out/pk_probe.log
