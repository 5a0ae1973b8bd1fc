#!/bin/bash
# Round 2, GPU call 1: counters and yardsticks on the round-1 kernels BEFORE touching them (VERDICT "counters first").
set -u
out=$GRAFT_REPO_ROOT/gpurun_out/r2c1; mkdir -p $out
cd $GRAFT_REPO_ROOT
timeout 300 tools/probes/hipblaslt_ref > $out/hipblaslt_ref.log 2>&1
B1="2812,3072,1024;2812,1024,1024;2812,2048,1024;2812,1024,2048;1406,3072,1024;1406,1024,1024;1406,2048,1024;1406,1024,2048"
BIG="22496,3072,1024;22496,1024,2048;89984,2048,1024;89984,3072,1024;89984,1024,1024;89984,1024,2048"
for epi in 0 1 2; do
  KB_SHAPES=$B1 KB_PRECS=fp16x3,fp16 KB_EPI=$epi KB_VARIANTS=1,6,2,8,0 timeout 300 python tools/kernel_bench.py gemm 2>&1 | grep ^gemm | sed "s/^/epi$epi /"
done > $out/kb_b1.log 2>&1
KB_SHAPES=$B1 KB_PRECS=fp16x3 KB_EPI=1 KB_VARIANTS=1,17,18,19,20 timeout 300 python tools/kernel_bench.py gemm 2>&1 | grep ^gemm | sed "s/^/abl128x64 /" >> $out/kb_b1.log 2>&1
for epi in 0 1 2; do
  KB_SHAPES=$BIG KB_PRECS=fp16x3,fp16 KB_EPI=$epi KB_VARIANTS=2,31,21,5 timeout 300 python tools/kernel_bench.py gemm 2>&1 | grep ^gemm | sed "s/^/epi$epi /"
done > $out/kb_big.log 2>&1
KB_SHAPES=$BIG KB_PRECS=fp16x3,fp16 KB_EPI=1 KB_VARIANTS=2,9,10,11,12,15 timeout 300 python tools/kernel_bench.py gemm 2>&1 | grep ^gemm | sed "s/^/abl128x128 /" >> $out/kb_big.log 2>&1
# PMC passes (separate runs, no tracing domains)
cd /tmp && export TMPDIR=/tmp
pmc() {  # tag, then kernel_bench 'one' args; env KB_EPI from caller
  tag=$1; shift; i=0
  for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES" \
             "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
             "GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum"; do
    i=$((i+1))
    timeout 200 rocprofv3 --pmc $set --output-format csv -d $out/pmc_$tag/p$i -o p$i -- python $GRAFT_REPO_ROOT/tools/kernel_bench.py one "$@" > $out/pmc_$tag.p$i.log 2>&1
  done
}
KB_EPI=1 pmc qkvlike_b1 fp16x3 1 2812 3072 1024 10
KB_EPI=2 pmc out_b1 fp16x3 6 2812 1024 1024 10
KB_EPI=1 pmc ff1_b32_x3 fp16x3 21 89984 2048 1024 5
KB_EPI=1 pmc ff1_b32_f16 fp16 21 89984 2048 1024 5
python - <<PY > $out/pmc_summary.txt 2>&1
import csv, glob, collections
for f in sorted(glob.glob("$out/pmc_*/p*/**/*counter_collection.csv", recursive=True)):
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"][:70]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(k, r["Counter_Name"])] += 1
    for k, d in agg.items():
        if "gemm" in k:
            print(f.split("r2c1/")[1].split("/")[0], k)
            for c, v in d.items():
                print(f"   {c:28s} per-dispatch {v / cnt[(k, c)]:.5g}  dispatches {cnt[(k, c)]}")
PY
cd $GRAFT_REPO_ROOT
timeout 600 python bench.py --schedule default --model E2TTS_Base --batch 8 --vocoder bigvgan --steps 3 --warmup 1 --no-cpu-baseline > $out/bench_e2_bigvgan_b8.json 2> $out/bench_e2_bigvgan_b8.err
timeout 600 python bench.py --schedule default --batch 32 --nfe 32 --steps 2 --warmup 1 --no-cpu-baseline > $out/bench_b32.json 2> $out/bench_b32.err
timeout 300 python bench.py --schedule default --steps 10 --warmup 3 --no-cpu-baseline > $out/bench_b1.json 2> $out/bench_b1.err
cat $out/hipblaslt_ref.log $out/kb_b1.log $out/kb_big.log; cat $out/pmc_summary.txt | head -150; tail -c 600 $out/bench_e2_bigvgan_b8.json; tail -c 300 $out/bench_e2_bigvgan_b8.err
