#!/bin/bash
# ragged corpus: utterances one by one against length-bucketed batches with a frame budget (f5-tts_amd/eval_batching.py)
set -u
out=$GRAFT_REPO_ROOT/gpurun_out/r2c27; mkdir -p $out
cd $GRAFT_REPO_ROOT
{
timeout 900 python tools/infer_batch.py --synthetic 48 --nfe 16 --out /tmp/o1 2>&1 | tail -2
for fpb in 3000 6000 12000 24000; do echo "== frames per batch $fpb"; timeout 900 python tools/infer_batch.py --synthetic 48 --nfe 16 --frames-per-batch $fpb --out /tmp/o2 2>&1 | tail -3; done
} > $out/ragged.log 2>&1
cat $out/ragged.log
