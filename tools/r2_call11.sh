#!/bin/bash
set -u
out=$GRAFT_REPO_ROOT/gpurun_out/r2c11; mkdir -p $out
cd $GRAFT_REPO_ROOT
for sq in "2 1406" "1 1406" "8 1406" "8 1407" "64 1406"; do set -- $sq
  timeout 300 python tools/kernel_bench.py qkv fp16x3 $1 $2 50,51,52,55,57,59,53,54,56 5 2>&1 | grep -E "^qkv|QKV_CHECK"
done > $out/qkv.log 2>&1
for sq in "2 1406" "8 1406"; do set -- $sq
  F5HIP_PP_EXP=8 timeout 300 python tools/kernel_bench.py qkv fp16x3 $1 $2 50,51,55,59 5 2>&1 | grep -E "^qkv|QKV_CHECK" | sed 's/^/notr /'
done >> $out/qkv.log 2>&1
grep "QKV_CHECK" $out/qkv.log | head; grep "qkv" $out/qkv.log | awk '{print $1,$2,$3,$4,$5,$6,$7,$8,$9,$10,$11,$12}' | sort | uniq -c | awk '{print}' | sed 's/differing halves//' | sort -k4,4 -k3,3 | head -80
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -x 2>&1 | tail -3
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $out/bench_b1.json 2>$out/bench_b1.err
python - <<PY
import json
d=json.loads(open("$out/bench_b1.json").read().strip().splitlines()[-1]); print("b1", round(d["ms_per_step"],2), d["kernel_classes_ms"])
PY
