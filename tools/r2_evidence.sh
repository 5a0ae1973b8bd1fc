#!/bin/bash
# Round 2 evidence run (one GPU call): the numbers DESIGN.md / profiles/ quote, all from ONE box and ONE source state.
#   gpu tests -> bench lines (B=1 default with cpu_baseline, B=4, B=32 NFE 32, E2-TTS + BigVGAN B=8) -> rocprofv3 kernel-trace summaries
#   -> PMC traffic / MFMA-busy passes (separate runs) -> microbenchmark tables.   Outputs: gpurun_out/r2ev/ (copied to profiles/r02*).
set -u
R=$GRAFT_REPO_ROOT; out=$R/gpurun_out/${EVDIR:-r2ev}; rm -rf $out; mkdir -p $out
cd $R
timeout 1500 python -m pytest tests -q -m gpu 2>&1 | tail -4 > $out/gpu_tests.log; cat $out/gpu_tests.log
timeout 900 python bench.py --steps 10 --warmup 3 > $out/bench_b1.json 2> $out/bench_b1.err; tail -c 300 $out/bench_b1.err
timeout 600 python bench.py --steps 10 --warmup 3 --precision fp16 --no-cpu-baseline > $out/bench_b1_fp16.json 2> $out/bench_b1_fp16.err
timeout 600 python bench.py --steps 3 --warmup 1 --batch 4 --nfe 32 --no-cpu-baseline > $out/bench_b4_nfe32.json 2> $out/bench_b4.err
timeout 600 python bench.py --steps 3 --warmup 1 --batch 8 --nfe 16 --no-cpu-baseline > $out/bench_b8.json 2> $out/bench_b8.err
timeout 600 python bench.py --steps 2 --warmup 1 --batch 32 --nfe 32 --no-cpu-baseline > $out/bench_b32_nfe32.json 2> $out/bench_b32.err
timeout 600 python bench.py --steps 2 --warmup 1 --batch 32 --nfe 32 --precision fp16 --no-cpu-baseline > $out/bench_b32_nfe32_fp16.json 2> $out/bench_b32_fp16.err
timeout 600 python bench.py --model E2TTS_Base --batch 8 --vocoder bigvgan --steps 3 --warmup 1 --no-cpu-baseline > $out/bench_e2_b8_bigvgan.json 2> $out/bench_e2.err
for f in $out/bench_*.json; do python - <<PY
import json
d=json.loads(open("$f").read().strip().splitlines()[-1]); print("$f".split("/")[-1], round(d["ms_per_step"],2), d["value"], d.get("roofline",{}).get("frac"))
PY
done
# kernel-trace summaries (rocprofv3 --kernel-trace --stats of the same commands)
cd /tmp && export TMPDIR=/tmp
for cfg in "b1 --steps 3 --warmup 1" "b32_nfe32 --steps 1 --warmup 1 --batch 32 --nfe 32"; do set -- $cfg; tag=$1; shift
  d=$out/trace_$tag; mkdir -p $d
  timeout 900 rocprofv3 --kernel-trace --stats -d $d -o trace -- python $R/bench.py "$@" --no-cpu-baseline > $d/bench.log 2>&1
  db=$(find $d -name "*.db" | head -1)
  [ -n "$db" ] && python $R/tools/rocpd_summary.py $db > $out/kernel_stats_$tag.md 2>&1
  find $d -name "*.db" -delete; find $d -name "*.csv" -size +2M -delete
  head -25 $out/kernel_stats_$tag.md
done
# PMC passes over the default bench line (separate runs; counters only)
cd $R
bash tools/pmc_bench.sh fp16x3_b1 --batch 1 --nfe 16 > /dev/null 2>&1; cp gpurun_out/pmc_fp16x3_b1.json $out/pmc_fp16x3_b1.json; cat $out/pmc_fp16x3_b1.json | head -40
rm -rf gpurun_out/pmc_fp16x3_b1
# microbenchmark tables: the engine's tile per shape next to the alternatives, the q|k|v projection, attention, hipBLASLt yardstick
B1="2812,3072,1024;2812,1024,1024;2812,2048,1024;2812,1024,2048;1406,3072,1024;1406,1024,1024;1406,2048,1024;1406,1024,2048"
BIG="11248,3072,1024;11248,1024,2048;22496,3072,1024;22496,1024,2048;89984,2048,1024;89984,3072,1024;89984,1024,1024;89984,1024,2048"
{
for epi in 1 2; do KB_SHAPES=$B1 KB_PRECS=fp16x3 KB_EPI=$epi KB_VARIANTS=-1,1,55,56,59,66,68,69,70 timeout 300 python tools/kernel_bench.py gemm 2>&1 | grep ^gemm | sed "s/^/epi$epi /"; done
for epi in 1 2; do KB_SHAPES=$BIG KB_PRECS=fp16x3,fp16 KB_EPI=$epi KB_VARIANTS=-1,2,50,51,52 timeout 400 python tools/kernel_bench.py gemm 2>&1 | grep ^gemm | sed "s/^/epi$epi /"; done
for sq in "1 1406" "2 1406" "8 1406" "64 1406"; do timeout 200 python tools/kernel_bench.py qkv fp16x3 $sq -1,1,50,51,55,56,68 20 2>&1 | grep -E "^qkv" | awk 'NR%3==0'; done
timeout 300 python tools/kernel_bench.py attn 2>&1 | grep ^attn
} > $out/kernel_bench.log 2>&1
tail -30 $out/kernel_bench.log
[ -x tools/probes/hipblaslt_ref ] && timeout 300 tools/probes/hipblaslt_ref > $out/hipblaslt_ref.log 2>&1
timeout 600 python tools/attn_precision_check.py > $out/attn_precision.log 2>&1; tail -12 $out/attn_precision.log
