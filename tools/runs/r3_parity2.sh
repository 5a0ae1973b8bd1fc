#!/bin/bash
# round 3, call 12: new tile table (two workgroups per CU at B = 2..16) + the new goldens (stress, real example, vocos head): GPU suite and bench lines
set -u
out=$GRAFT_REPO_ROOT/gpurun_out/r3c12; mkdir -p $out
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -q -m gpu -x -s 2>&1 | grep -E "stress|reference example|vocos_head|passed|failed|Error|assert" | head -40 > $out/gpu_tests.log; cat $out/gpu_tests.log
for cfg in "b1 --steps 10 --warmup 3" "b4_nfe32 --batch 4 --nfe 32 --steps 3 --warmup 1" "b8 --batch 8 --steps 3 --warmup 1" "b32_nfe32 --batch 32 --nfe 32 --steps 2 --warmup 1"; do set -- $cfg; tag=$1; shift
  timeout 900 python bench.py "$@" --no-cpu-baseline > $out/bench_$tag.json 2> $out/bench_$tag.err
  python - <<PY
import json
try:
    d=json.loads(open("$out/bench_$tag.json").read().strip().splitlines()[-1]); print("$tag", round(d["ms_per_step"],2), round(d["roofline"]["frac"],4), {k: round(v,1) for k,v in d["kernel_classes_ms"].items() if v > 1})
except Exception as e: print("$tag ERR", e, open("$out/bench_$tag.err").read()[-400:])
PY
done
timeout 600 python bench.py --model E2TTS_Base --batch 8 --vocoder bigvgan --steps 3 --warmup 1 --no-cpu-baseline > $out/bench_e2.json 2> $out/bench_e2.err; python - <<PY
import json
d=json.loads(open("$out/bench_e2.json").read().strip().splitlines()[-1]); print("e2_b8", round(d["ms_per_step"],2))
PY
