#!/bin/bash
# B = 32 block GEMMs: HBM traffic and L2 hit rate of the 256x256 tile per rasterisation (counters first), then time per rasterisation;
# key-split attention time at B = 1 (for the record: the option was never timed)
set -u
R=$GRAFT_REPO_ROOT; out=$R/gpurun_out/r2c19; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
pmc() {  # tag env... -- kernel_bench one args
  tag=$1; shift; i=0
  for set in "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum"; do
    i=$((i+1))
    timeout 200 rocprofv3 --pmc $set --output-format csv -d $out/pmc_$tag/p$i -o p$i -- python $R/tools/kernel_bench.py one "$@" > $out/pmc_$tag.p$i.log 2>&1
  done
}
for gm in 1 4 16; do
  KB_EPI=1 F5HIP_GEMM_GROUPM=$gm pmc ff1_b32_gm$gm fp16x3 50 89984 2048 1024 3
  KB_EPI=2 F5HIP_GEMM_GROUPM=$gm pmc ff2_b32_gm$gm fp16x3 50 89984 1024 2048 3
done
python - <<PY > $out/pmc_summary.txt 2>&1
import csv, glob, collections
for f in sorted(glob.glob("$out/pmc_*/p*/**/*counter_collection.csv", recursive=True)):
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"][:60]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(k, r["Counter_Name"])] += 1
    for k, d in agg.items():
        if "gemm_pp" in k:
            for c, v in d.items():
                print(f.split("r2c19/")[1].split("/")[0], f"{c:22s} per-dispatch {v / cnt[(k, c)]:.5g}  dispatches {cnt[(k, c)]}")
PY
cat $out/pmc_summary.txt
cd $R
for gm in 1 2 4 8 16 32; do echo "== group_m $gm"
F5HIP_GEMM_GROUPM=$gm KB_SHAPES="89984,2048,1024;89984,3072,1024;89984,1024,2048" KB_PRECS=fp16x3 KB_EPI=1 KB_VARIANTS=50,51 timeout 300 python tools/kernel_bench.py gemm 2>&1 | grep ^gemm | cut -c1-200
done > $out/groupm.log 2>&1
cat $out/groupm.log
for kv in 1 2 3 1 2; do
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --attn-kv-split $kv > $out/b1_kv$kv.json 2>$out/b1_kv$kv.err
python - <<PY
import json
d=json.loads(open("$out/b1_kv$kv.json").read().strip().splitlines()[-1]); k=d["kernel_classes_ms"]; print("b1 kv_split $kv", round(d["ms_per_step"],2), k["gemm_block"], k["attention"])
PY
done
