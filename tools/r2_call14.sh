#!/bin/bash
set -u
out=$GRAFT_REPO_ROOT/gpurun_out/r2c14; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
pmc() {  # tag, then kernel_bench args; env KB_EPI from caller
  tag=$1; shift; i=0
  for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES" \
             "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
             "GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum" "SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC SQ_INSTS_SMEM"; do
    i=$((i+1))
    timeout 200 rocprofv3 --pmc $set --output-format csv -d $out/pmc_$tag/p$i -o p$i -- python $GRAFT_REPO_ROOT/tools/kernel_bench.py "$@" > $out/pmc_$tag.p$i.log 2>&1
  done
}
KB_EPI=1 pmc ff1_b32 one fp16x3 50 89984 2048 1024 5
KB_EPI=1 pmc ff1_b1 one fp16x3 55 2812 2048 1024 10
KB_EPI=2 pmc out_b1 one fp16x3 59 2812 1024 1024 10
pmc attn_b1 oneattn fp16x3 2 1406 10
pmc attn_b32 oneattn fp16x3 64 1406 5
python - <<PY > $out/pmc_summary.txt 2>&1
import csv, glob, collections
for f in sorted(glob.glob("$out/pmc_*/p*/**/*counter_collection.csv", recursive=True)):
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"][:60]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(k, r["Counter_Name"])] += 1
    for k, d in agg.items():
        if "gemm_pp" in k or "flash_attn" in k:
            print(f.split("r2c14/")[1].split("/")[0], k)
            for c, v in d.items():
                print(f"   {c:28s} per-dispatch {v / cnt[(k, c)]:.5g}  dispatches {cnt[(k, c)]}")
PY
cat $out/pmc_summary.txt
