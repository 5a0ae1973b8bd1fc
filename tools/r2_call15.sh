#!/bin/bash
set -u
out=$GRAFT_REPO_ROOT/gpurun_out/r2c15; mkdir -p $out
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -q -m gpu -x 2>&1 | tail -3
for b in "1 16 10 3" "32 32 2 1"; do set -- $b
timeout 600 python bench.py --steps $3 --warmup $4 --no-cpu-baseline --batch $1 --nfe $2 > $out/bench_b$1.json 2>$out/bench_b$1.err
python - <<PY
import json
d=json.loads(open("$out/bench_b$1.json").read().strip().splitlines()[-1]); print("b$1", round(d["ms_per_step"],2), d["kernel_classes_ms"], round(d["roofline"]["frac"],4))
PY
done
