#!/bin/bash
# one chain against two chains at B = 1..4 with the round-2 tiles (the rule B*N <= 6144 dates from the round-1 kernels)
set -u
out=$GRAFT_REPO_ROOT/gpurun_out/r2c23; mkdir -p $out
cd $GRAFT_REPO_ROOT
for b in 1 2 3 4 6; do for bs in 0 1 0 1; do
timeout 600 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --batch $b --nfe 16 --branch-streams $bs > $out/b.json 2>$out/b.err
python - <<PY
import json
d=json.loads(open("$out/b.json").read().strip().splitlines()[-1]); print("B=$b branch_streams=$bs", round(d["ms_per_step"],2))
PY
done; done
