#!/bin/bash
# attention row sums on the matrix pipe (ones fragment) against the VALU sums (F5HIP_ATTN_VALU_SUM=1): tests, microbench, bench, golden errors
set -u
out=$GRAFT_REPO_ROOT/gpurun_out/r2c30; mkdir -p $out
cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests -q -m gpu 2>&1 | tail -3
{
for e in "X=1" "F5HIP_ATTN_VALU_SUM=1" "X=1" "F5HIP_ATTN_VALU_SUM=1"; do echo "== $e"; env $e timeout 200 python tools/kernel_bench.py attn 2>&1 | grep "^attn fp16 "; done
} > $out/attn.log 2>&1; cat $out/attn.log
for e in "X=1" "F5HIP_ATTN_VALU_SUM=1" "X=1" "F5HIP_ATTN_VALU_SUM=1"; do
env $e timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('b1 $e', round(d['ms_per_step'],2), d['kernel_classes_ms']['attention'])"
done
for e in "X=1" "F5HIP_ATTN_VALU_SUM=1"; do
env $e timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --batch 32 --nfe 8 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('b32 nfe8 $e', round(d['ms_per_step'],2), d['kernel_classes_ms']['attention'])"
done
timeout 600 python tools/attn_precision_check.py base_v1_cfg1 base_v1_cfg3_b4 e2_base_cfg5 small_v1 small_e2 tiny48_ragged_b2 tiny_unett_ragged_b2 2>&1 | grep -v amdgpu | tee $out/attn_precision.log
