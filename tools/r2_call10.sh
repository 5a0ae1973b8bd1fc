#!/bin/bash
set -u
out=$GRAFT_REPO_ROOT/gpurun_out/r2c10; mkdir -p $out
cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests -q -m gpu > $out/gpu_tests.log 2>&1; tail -8 $out/gpu_tests.log
run() { tag=$1; shift; timeout 600 python bench.py --no-cpu-baseline "$@" > $out/bench_$tag.json 2> $out/bench_$tag.err; python - <<PY
import json
try:
    d=json.loads(open("$out/bench_$tag.json").read().strip().splitlines()[-1]); print("$tag", round(d["ms_per_step"],2), "rtf", round(d["rtf"],5), d["kernel_classes_ms"], round(d["roofline"]["frac"],4))
except Exception as e: print("$tag", "ERR", e, open("$out/bench_$tag.err").read()[-300:])
PY
}
run b1 --steps 10 --warmup 3
run b4 --steps 3 --warmup 1 --batch 4
run b32 --steps 2 --warmup 1 --batch 32 --nfe 32
run e2_b8_bigvgan --steps 2 --warmup 1 --model E2TTS_Base --batch 8 --vocoder bigvgan
python __graft_entry__.py smoke 2>&1 | tail -2
