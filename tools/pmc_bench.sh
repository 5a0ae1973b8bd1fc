#!/bin/bash
# HBM-traffic PMC passes over one bench.py invocation (run on the GPU box; separate passes, counters only — no tracing).
# NOTE: counter collection serialises every dispatch — the B = 1 NFE 16 line takes ~40 s per pass, but configs[2] (B = 32, NFE 32: 45 k large
# dispatches) did not finish ONE pass in 14 minutes (round 2 lost its last GPU minutes that way): use --nfe 2 there.
# usage: tools/pmc_bench.sh <tag> <bench.py args...>      -> gpurun_out/pmc_<tag>/{fetch,write}/..., gpurun_out/pmc_<tag>.json
set -u
tag=$1; shift
out=$GRAFT_REPO_ROOT/gpurun_out/pmc_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/fetch -o fetch -- python $GRAFT_REPO_ROOT/bench.py "$@" --steps 1 --warmup 0 --no-cpu-baseline --no-graph > $out/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/write -o write -- python $GRAFT_REPO_ROOT/bench.py "$@" --steps 1 --warmup 0 --no-cpu-baseline --no-graph > $out/write.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES --output-format csv -d $out/mfma -o mfma -- python $GRAFT_REPO_ROOT/bench.py "$@" --steps 1 --warmup 0 --no-cpu-baseline --no-graph > $out/mfma.log 2>&1
# durations from a pass WITHOUT counters (counter collection serialises and slows the dispatches): kernel trace only
rocprofv3 --kernel-trace --output-format csv -d $out/trace -o trace -- python $GRAFT_REPO_ROOT/bench.py "$@" --steps 1 --warmup 0 --no-cpu-baseline --no-graph > $out/trace.log 2>&1
python $GRAFT_REPO_ROOT/tools/pmc_summarize.py $out "$@" > $GRAFT_REPO_ROOT/gpurun_out/pmc_$tag.json
cat $GRAFT_REPO_ROOT/gpurun_out/pmc_$tag.json
