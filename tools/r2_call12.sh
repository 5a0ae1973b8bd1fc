#!/bin/bash
set -u
out=$GRAFT_REPO_ROOT/gpurun_out/r2c12; mkdir -p $out
cd $GRAFT_REPO_ROOT
for sq in "1 1406" "2 1406" "8 1406" "64 1406"; do set -- $sq
  timeout 300 python tools/kernel_bench.py qkv fp16x3 $1 $2 50,51,52,55,59,53,54,56 5 2>&1 | grep -E "^qkv|QKV_CHECK"
done > $out/qkv.log 2>&1
grep "QKV_CHECK" $out/qkv.log | head -5
grep "^qkv" $out/qkv.log | awk '{k=$3" "$4; s[k]+=$(NF-4); n[k]++; d[k]+=$NF} END {for (k in s) printf "%s %.1f diff %d\n", k, s[k]/n[k], d[k]}' | sort -k2,2 -k3n
timeout 900 python -m pytest tests -q -m gpu -x 2>&1 | tail -3
for b in "1 16" "4 16"; do set -- $b
timeout 300 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --batch $1 --nfe $2 > $out/bench_b$1.json 2>$out/bench_b$1.err
python - <<PY
import json
d=json.loads(open("$out/bench_b$1.json").read().strip().splitlines()[-1]); print("b$1", round(d["ms_per_step"],2), d["kernel_classes_ms"]["gemm_block"], d["kernel_classes_ms"]["attention"], round(d["roofline"]["frac"],4))
PY
done
