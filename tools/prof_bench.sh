# rocprofv3 kernel trace of the default bench line (2 branch streams) and of the packed single-stream schedule; summaries -> gpurun_out/
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for mode in "-1" "0"; do
  out=$R/gpurun_out/prof_bs$mode
  rm -rf $out; mkdir -p $out
  rocprofv3 --kernel-trace --stats -d $out -o trace -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --branch-streams $mode > $out/bench.log 2>&1 || true
  db=$(find $out -name "*.db" | head -1)
  python $R/tools/rocpd_summary.py $db > $out/summary.md 2>&1 || true
  tail -2 $out/bench.log | cut -c1-600
  tail -8 $out/summary.md
  find $out -name "*.db" -delete
done
