#!/bin/bash
# round 3, call 14: lazy reference maximum in flash attention: kernel A/B, golden sweep with / without, bench lines
set -u
out=$GRAFT_REPO_ROOT/gpurun_out/r3c14; mkdir -p $out
cd $GRAFT_REPO_ROOT
{ echo "== exact running maximum (natural-log scores)"; timeout 300 python tools/kernel_bench.py attn 2>&1 | grep "^attn";
  echo "== lazy reference maximum (KB_ATTN_LOG2Q=1)"; KB_ATTN_LOG2Q=1 timeout 300 python tools/kernel_bench.py attn 2>&1 | grep "^attn";
  echo "== KB_ATTN_LOG2Q=1 F5HIP_ATTN_LAZY=0 (base-2 scores, exact maximum)"; KB_ATTN_LOG2Q=1 F5HIP_ATTN_LAZY=0 timeout 300 python tools/kernel_bench.py attn 2>&1 | grep "^attn"; } > $out/attn_ab.log 2>&1
cat $out/attn_ab.log
{ echo "== lazy"; timeout 900 python tools/attn_precision_check.py 2>&1 | grep -E "max"; echo "== F5HIP_ATTN_LAZY=0"; F5HIP_ATTN_LAZY=0 timeout 900 python tools/attn_precision_check.py base_v1_cfg1 base_v1_stress small_v1 e2_base_cfg5 tiny_v1_nfe16 tiny_mmdit_mask_ragged_b2 2>&1 | grep -E "max"; } > $out/attn_precision.log 2>&1
cat $out/attn_precision.log
timeout 1200 python -m pytest tests -q -m gpu -x 2>&1 | tail -2
for cfg in "b1 --steps 10 --warmup 3" "b32_nfe32 --batch 32 --nfe 32 --steps 2 --warmup 1"; do set -- $cfg; tag=$1; shift
  timeout 900 python bench.py "$@" --no-cpu-baseline > $out/bench_$tag.json 2> $out/bench_$tag.err
  python - <<PY
import json
d=json.loads(open("$out/bench_$tag.json").read().strip().splitlines()[-1]); print("$tag", round(d["ms_per_step"],2), {k: round(v,1) for k,v in d["kernel_classes_ms"].items() if v > 1})
PY
done
