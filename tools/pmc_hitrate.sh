#!/bin/bash
# L2 hit rate of one GEMM microbench config under two rasterisations.  usage: tools/pmc_hitrate.sh
set -u
out=$GRAFT_REPO_ROOT/gpurun_out/pmc_hit
rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
for G in 0 22; do
  for shape in "6 2816 1024 2048" "1 2816 3072 1024"; do
    set -- $shape
    d=$out/g${G}_n$3
    F5HIP_GEMM_GROUPM=$G rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --output-format csv -d $d -o p -- env KB_EPI=1 python $GRAFT_REPO_ROOT/tools/kernel_bench.py one fp16x3 $1 $2 $3 $4 10 > $d.log 2>&1
  done
done
python - <<PY
import csv, glob, collections
for f in sorted(glob.glob("$out/**/*counter_collection.csv", recursive=True)):
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"][:50]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(k, r["Counter_Name"])] += 1
    for k, d in agg.items():
        if "gemm" in k and "DF16_Li3" in k:
            n = cnt[(k, "TCC_HIT_sum")]
            h, m, q = d["TCC_HIT_sum"] / n, d["TCC_MISS_sum"] / n, d.get("TCC_REQ_sum", 0) / max(n, 1)
            print(f.split("pmc_hit/")[1].split("/")[0], k[:40], f"dispatches {n} hit {h:.4g} miss {m:.4g} req {q:.4g} hit-rate {h / max(h + m, 1):.3f} miss bytes(128B) {m * 128 / 1e6:.1f} MB")
PY
