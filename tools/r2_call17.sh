#!/bin/bash
# k-split 96x128 adopted for the narrow outputs + LN (early parameter loads, paired stores): GPU tests, then B=1 with one chain / two chains
set -u
out=$GRAFT_REPO_ROOT/gpurun_out/r2c17; mkdir -p $out
cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests -q -m gpu -x 2>&1 | tail -3
for bs in -1 0 1 -1 0; do
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --batch 1 --nfe 16 --branch-streams $bs > $out/bench_b1_bs$bs.json 2>$out/bench_b1_bs$bs.err
python - <<PY
import json
d=json.loads(open("$out/bench_b1_bs$bs.json").read().strip().splitlines()[-1]); print("b1 bs$bs", round(d["ms_per_step"],2), d["kernel_classes_ms"], round(d["roofline"]["frac"],4))
PY
done
for b in "4 32 3 1" "32 32 2 1"; do set -- $b
timeout 600 python bench.py --steps $3 --warmup $4 --no-cpu-baseline --batch $1 --nfe $2 > $out/bench_b$1.json 2>$out/bench_b$1.err
python - <<PY
import json
d=json.loads(open("$out/bench_b$1.json").read().strip().splitlines()[-1]); print("b$1", round(d["ms_per_step"],2), d["kernel_classes_ms"], round(d["roofline"]["frac"],4))
PY
done
