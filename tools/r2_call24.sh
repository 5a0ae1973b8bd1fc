#!/bin/bash
# one chain against two chains at B = 8, 12, 16, 24, 32
set -u
out=$GRAFT_REPO_ROOT/gpurun_out/r2c24; mkdir -p $out
cd $GRAFT_REPO_ROOT
for b in 8 12 16 24 32; do for bs in 0 1 0 1; do
timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --batch $b --nfe 8 --branch-streams $bs > $out/b.json 2>$out/b.err
python - <<PY
import json
d=json.loads(open("$out/b.json").read().strip().splitlines()[-1]); print("B=$b nfe 8 branch_streams=$bs", round(d["ms_per_step"],2))
PY
done; done
