# end-of-round artefacts: default bench line (with cpu_baseline), B=32 NFE=32 line, rocprofv3 kernel trace of the packed schedule
set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/final; rm -rf $O; mkdir -p $O
python $R/bench.py --schedule default > $O/bench_b1_fp16x3.json 2> $O/bench_b1.err
tail -1 $O/bench_b1_fp16x3.json | cut -c1-300
python $R/bench.py --schedule default --batch 32 --nfe 32 --steps 2 --warmup 1 --no-cpu-baseline > $O/bench_b32_nfe32_fp16x3.json 2> $O/bench_b32.err
tail -1 $O/bench_b32_nfe32_fp16x3.json | cut -c1-300
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/prof -o trace -- python $R/bench.py --schedule default --steps 2 --warmup 1 --no-cpu-baseline --branch-streams 0 > $O/prof.log 2>&1 || true
db=$(find $O/prof -name "*.db" | head -1)
python $R/tools/rocpd_summary.py $db > $O/kernel_stats_b1_packed.md 2>&1 || true
find $O/prof -name "*.db" -delete
head -12 $O/kernel_stats_b1_packed.md | cut -c1-150
