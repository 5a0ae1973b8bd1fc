#!/bin/bash
# k-step-split tiles adopted: GPU tests, then B=1 one chain / two chains / two chains with the 192x192 q|k|v tile, same box
set -u
out=$GRAFT_REPO_ROOT/gpurun_out/r2c22; mkdir -p $out
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -q -m gpu -x 2>&1 | tail -3
run() { tag=$1; shift
env "$@" timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --batch 1 --nfe 16 --branch-streams $BS > $out/b1.json 2>$out/b1.err
python - <<PY
import json
d=json.loads(open("$out/b1.json").read().strip().splitlines()[-1]); k=d["kernel_classes_ms"]; print("b1 $tag", round(d["ms_per_step"],2), k["gemm_block"], k["attention"], k["ln_modulate"])
PY
}
for rep in 1 2; do
BS=1 run "two chains (default picks)" X=1
BS=0 run "one chain (68/69)" X=1
BS=1 run "two chains, qkv 68" F5HIP_PP_VARIANT_N3072=68
BS=0 run "one chain, round-2d picks (56/55)" F5HIP_PP_VARIANT_N3072=56 F5HIP_PP_VARIANT_N2048=55
done
