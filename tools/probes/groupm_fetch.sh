#!/bin/bash
# HBM-side fetch of a block GEMM against the tile rasterisation (F5HIP_GEMM_GROUPM = row tiles per group; an XCD takes a contiguous run of
# the tile order, gemm_pp.h): one counter pass per setting (counters only, never with tracing) + an un-instrumented timing run.
# usage (GPU box, repo root): bash tools/probes/groupm_fetch.sh OUTDIR
out=${1:?outdir}; R=$PWD; mkdir -p $out
export KB_SHAPES="2812,3072,1024;2812,2048,1024;2812,1024,1024;2812,1024,2048" KB_PRECS=fp16m KB_EPI=1 KB_VARIANTS=-1
for gm in 1 2 3 5 8; do
  echo "== F5HIP_GEMM_GROUPM=$gm"
  F5HIP_GEMM_GROUPM=$gm timeout 300 python tools/kernel_bench.py gemm 2>&1 | grep ^gemm | cut -c1-120
  ( cd /tmp; export TMPDIR=/tmp; F5HIP_GEMM_GROUPM=$gm timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/$out/gm$gm -o f -- python $R/tools/kernel_bench.py gemm > $R/$out/gm$gm.log 2>&1 )
  python - $out/gm$gm <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: [0, 0.0])
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == "FETCH_SIZE" and "gemm_pp_kernel" in r["Kernel_Name"]:
            k = (r["Kernel_Name"][:70], r["Grid_Size"])
            acc[k][0] += 1; acc[k][1] += float(r["Counter_Value"])
for k, (n, v) in sorted(acc.items()):
    print(f"   {k[0]} grid {k[1]}: {n} launches, FETCH_SIZE/launch {v / n:.0f} KiB (x 2: the gfx950 correction of tools/pmc_summarize.py = {v / n * 2048 / 1e6:.1f} MB)")
PY
done
