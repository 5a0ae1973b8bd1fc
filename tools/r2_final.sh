#!/bin/bash
# final state of round 2: gpu tests, smoke, the three headline bench lines, kernel-trace summaries, PMC traffic (source hash) -> gpurun_out/r2fin
set -u
R=$GRAFT_REPO_ROOT; out=$R/gpurun_out/r2fin; rm -rf $out; mkdir -p $out
cd $R
timeout 1200 python -m pytest tests -q -m gpu 2>&1 | tail -3 > $out/gpu_tests.log; cat $out/gpu_tests.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 900 python bench.py --steps 10 --warmup 3 > $out/bench_b1.json 2> $out/bench_b1.err
timeout 600 python bench.py --steps 2 --warmup 1 --batch 32 --nfe 32 --no-cpu-baseline > $out/bench_b32_nfe32.json 2> $out/bench_b32.err
timeout 600 python bench.py --model E2TTS_Base --batch 8 --vocoder bigvgan --steps 3 --warmup 1 --no-cpu-baseline > $out/bench_e2_b8_bigvgan.json 2> $out/bench_e2.err
for f in $out/bench_*.json; do python - <<PY
import json
d=json.loads(open("$f").read().strip().splitlines()[-1]); print("$f".split("/")[-1], round(d["ms_per_step"],2), round(d["value"]), d["roofline"]["frac"], d["roofline"]["traffic"])
PY
done
cd /tmp && export TMPDIR=/tmp
for cfg in "b1 --steps 3 --warmup 1" "b32_nfe32 --steps 1 --warmup 1 --batch 32 --nfe 32"; do set -- $cfg; tag=$1; shift
  d=$out/trace_$tag; mkdir -p $d
  timeout 900 rocprofv3 --kernel-trace --stats -d $d -o trace -- python $R/bench.py "$@" --no-cpu-baseline > $d/bench.log 2>&1
  db=$(find $d -name "*.db" | head -1)
  [ -n "$db" ] && python $R/tools/rocpd_summary.py $db > $out/kernel_stats_$tag.md 2>&1
  rm -rf $d
  head -8 $out/kernel_stats_$tag.md | cut -c1-160
done
cd $R
bash tools/pmc_bench.sh fp16x3_b1 --batch 1 --nfe 16 > /dev/null 2>&1; cp gpurun_out/pmc_fp16x3_b1.json $out/pmc_fp16x3_b1.json; head -8 $out/pmc_fp16x3_b1.json
rm -rf gpurun_out/pmc_fp16x3_b1
