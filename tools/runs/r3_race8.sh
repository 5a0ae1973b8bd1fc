#!/bin/bash
# round 3, call 8: the packed op_sel sequence with a partner wave on the same SIMD that runs MFMAs / LDS reads / global loads
set -u
out=$GRAFT_REPO_ROOT/gpurun_out/r3c8; mkdir -p $out
cd $GRAFT_REPO_ROOT/tools/probes
{ for m in 1 4 8 5 13 15 0; do timeout 120 ./pk_opsel_probe 2 $m 20000 | tail -1; done; } > $out/pk_probe.log 2>&1
cat $out/pk_probe.log
