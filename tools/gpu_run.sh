#!/bin/bash
# tools/gpu_run.sh — the ONE script behind every GPU measurement of this repo.  Run on the GPU box through gpurun:
#
#     gpurun --timeout 1800 -- 'bash tools/gpu_run.sh <recipe> [args...]'
#
# Every recipe writes under gpurun_out/<tag>/ (merged back by gpurun); what is kept as evidence is copied to profiles/ by hand, named per
# round (profiles/README.md says which recipe produced which file).  Rounds 1-2 used one throw-away script per call (tools/r2_call*.sh,
# removed in round 3: `git show 6c9c0d8:tools/r2_call7.sh` still shows any of them).
#
#   tests [pytest args]                 the GPU suite (default: tests -m gpu)
#   bench <tag> [bench.py args]         one bench.py line -> gpurun_out/<tag>/bench.json, summary on stdout
#   evidence <tag>                      the round-end set: GPU suite, smoke, headline bench lines, rocprofv3 kernel-trace summaries,
#                                       PMC passes (B = 1 at NFE 16, B = 32 at NFE 2), microbenchmark tables, golden precision sweep
#   counters <tag>                      the rocprofv3 half of `evidence` alone
#   pmc <tag> [bench.py args]           FETCH / WRITE / MFMA-busy counter passes + a kernel-trace pass of one bench.py command
#   tiles <tag> <M,N,K;..> <variants> [epilogue] [precisions]    tools/kernel_bench.py gemm over shapes x tile ids
#   qkv <tag> <variants> <seqs nseq>..  the fused q|k|v projection: time + value check against the generic kernel
#   attn <tag>                          flash attention microbenchmark: exact running maximum vs lazy reference maximum
#   race <tag> <seqs> <nseq> <reps> <specs>    csrc/race_probe.hip through tools/kernel_bench.py qkvprobe (+ tools/race_dump_analyze.py on
#                                       every non-empty dump); specs = "tile,expt[,abl[,lds_pad[,noise]]];..."
#   pkprobe <tag>                       tools/probes/pk_opsel_probe (the standalone reproducer of the packed-fp32 operand fault) and
#                                       tools/probes/pk_opsel_sweep (every operand selection), with and without partner waves
#   ragged <tag> [frame budgets]        48 ragged synthetic utterances: bucketed / list order padded / list order packed rows
#   ab <tag> [baseline]                 round 6: this build against tools/.ab/<baseline>/ on one box (GEMM tiles, q|k|v, bench.py B = 1 / B = 32)
#   sweep <tag>                         round 6: tools/sharpness_sweep.py (error of every operand mode against goldens with logits x 1 .. 4)
#   mx <tag>                            round 4: the fp16m mode (MX-fp6 correction lines) against fp16x3 in one call — GPU parity tests of the
#                                       mode, the GEMM tiles on the B = 1 / B = 8 / B = 32 shapes in both modes, bench.py lines in both modes
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
recipe=${1:?recipe}; shift
cd "$R"

line() {  # one-line summary of a bench.py JSON file
  python - "$1" "$2" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    r = d.get("roofline", {})
    print(sys.argv[2], "ms/step", round(d["ms_per_step"], 2), "value", round(d["value"]), "roofline.frac", r.get("frac") and round(r["frac"], 4), "traffic", r.get("traffic"),
          {k: round(v, 1) for k, v in d.get("kernel_classes_ms", {}).items() if v > 1})
except Exception as e:
    print(sys.argv[2], "ERR", e)
PY
}

trace() {  # rocprofv3 kernel-trace summary of a bench.py command -> $1/kernel_stats_$2.md
  local out=$1 tag=$2; shift 2
  local d=$R/$out/trace_$tag; mkdir -p $d
  (cd /tmp && TMPDIR=/tmp timeout 1200 rocprofv3 --kernel-trace --stats -d $d -o trace -- python $R/bench.py "$@" --no-cpu-baseline > $d/bench.log 2>&1)
  local db; db=$(find $d -name "*.db" | head -1)
  [ -n "$db" ] && python tools/rocpd_summary.py $db > $out/kernel_stats_$tag.md 2>&1
  rm -rf $d
  head -12 $out/kernel_stats_$tag.md | cut -c1-170
}

pmc() {  # counter passes (separate runs: counters only, never with tracing) + an un-instrumented kernel-trace pass for the durations
  local out=$1 tag=$2; shift 2
  local d=$R/$out/pmc_$tag; mkdir -p $d
  local common="--steps 1 --warmup 0 --no-cpu-baseline --no-graph"
  ( cd /tmp; export TMPDIR=/tmp
    timeout 1500 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $d/fetch -o fetch -- python $R/bench.py "$@" $common > $d/fetch.log 2>&1
    timeout 1500 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $d/write -o write -- python $R/bench.py "$@" $common > $d/write.log 2>&1
    timeout 1500 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES --output-format csv -d $d/mfma -o mfma -- python $R/bench.py "$@" $common > $d/mfma.log 2>&1
    timeout 1500 rocprofv3 --kernel-trace --output-format csv -d $d/trace -o trace -- python $R/bench.py "$@" $common > $d/trace.log 2>&1 )
  python tools/pmc_summarize.py $d "$@" > $out/pmc_$tag.json
  rm -rf $d
  head -c 1500 $out/pmc_$tag.json
}

case $recipe in
ab)
  # round 6: the CURRENT build against a baseline build of libf5hip.so / libf5hip_bench.so kept under tools/.ab/<name>/ (git-ignored; made in the
  # build container from `git archive <commit> f5-tts_amd/csrc include`), alternating on ONE box: GEMM tiles on the B = 1 and many-round shapes
  # (KB_EPI 1 = the FeedForward GELU epilogue, 2 = gate + residual), the q|k|v launch, bench.py steps at B = 1 and B = 32
  tag=${1:?tag}; base=${2:-base}; out=gpurun_out/$tag; rm -rf $out; mkdir -p $out
  BASE="F5HIP_LIB=$R/tools/.ab/$base/libf5hip.so F5HIP_BENCH_LIB=$R/tools/.ab/$base/libf5hip_bench.so"
  B1="2812,3072,1024;2812,1024,1024;2812,2048,1024;2812,1024,2048"
  MID="11248,2048,1024;22496,2048,1024;89984,2048,1024;89984,1024,2048;89984,1024,1024"
  { for which in base new base new; do
      pre=""; [ $which = base ] && pre="env $BASE"
      for epi in 1 2; do
        KB_SHAPES=$B1 KB_PRECS=fp16m,fp16 KB_EPI=$epi KB_VARIANTS=-1 timeout 300 $pre python tools/kernel_bench.py gemm 2>&1 | grep ^gemm | sed "s/^/$which epi$epi /" | cut -c1-200
        KB_SHAPES=$MID KB_PRECS=fp16m,fp16 KB_EPI=$epi KB_VARIANTS=-1 timeout 400 $pre python tools/kernel_bench.py gemm 2>&1 | grep ^gemm | sed "s/^/$which epi$epi /" | cut -c1-200
      done
      for sq in "2 1406" "64 1406"; do timeout 300 $pre python tools/kernel_bench.py qkv fp16m $sq -1 20 2>&1 | grep -E "^qkv" | awk 'NR%3==0' | sed "s/^/$which /"; done
    done; } > $out/kernel_bench_ab.log 2>&1
  cat $out/kernel_bench_ab.log | cut -c1-220
  Q="--no-cpu-baseline --no-other-configs"
  for i in 1 2 3; do
    env $BASE timeout 600 python bench.py --steps 10 --warmup 3 $Q > $out/b1_base_$i.json 2>> $out/bench.err; line $out/b1_base_$i.json b1_base_$i
    timeout 600 python bench.py --steps 10 --warmup 3 $Q > $out/b1_new_$i.json 2>> $out/bench.err; line $out/b1_new_$i.json b1_new_$i
  done
  for i in 1 2; do
    env $BASE timeout 900 python bench.py --steps 2 --warmup 1 --batch 32 --nfe 32 $Q > $out/b32_base_$i.json 2>> $out/bench.err; line $out/b32_base_$i.json b32_base_$i
    timeout 900 python bench.py --steps 2 --warmup 1 --batch 32 --nfe 32 $Q > $out/b32_new_$i.json 2>> $out/bench.err; line $out/b32_new_$i.json b32_new_$i
  done
  tail -3 $out/bench.err ;;
sweep)
  # round 6: the sharpness sweep (tools/sharpness_sweep.py) and the per-golden attention-score table in both half-precision modes
  tag=${1:?tag}; out=gpurun_out/$tag; mkdir -p $out
  timeout 1500 python tools/sharpness_sweep.py > $out/sharpness_sweep.md 2> $out/sharpness_sweep.err; cat $out/sharpness_sweep.md; tail -3 $out/sharpness_sweep.err ;;
p8)
  # round 5: the ping-pong 256x256 kernel (csrc/gemm_p8.h, tile id 80) against the lockstep 256x256 tile (50) and the heuristic's choice:
  # value checks on the GPU, times on the many-round shapes with ablations, hipBLASLt yardstick, counters, B = 32 / B = 8 step A/B
  tag=${1:?tag}; out=gpurun_out/$tag; rm -rf $out; mkdir -p $out
  { for epi in 0 1 2; do KB_CHECK=1 KB_SHAPES="5000,2048,1024;3000,1024,2048" KB_PRECS=fp16 KB_EPI=$epi KB_VARIANTS=80 timeout 300 python tools/kernel_bench.py gemm 2>&1 | grep -E "KB_CHECK|ERR" | head -8; done
    for epi in 1 2; do KB_CHECK=1 KB_SHAPES="5000,2048,1024;3000,1024,2048" KB_PRECS=fp16m KB_EPI=$epi KB_VARIANTS=80 timeout 300 python tools/kernel_bench.py gemm 2>&1 | grep -E "KB_CHECK|ERR" | head -8; done
    for prec in fp16 fp16m; do timeout 300 python tools/kernel_bench.py qkv $prec 8 1406 80 5 2>&1 | grep -E "^qkv|QKV_CHECK" | tail -2; done; } > $out/check.log 2>&1
  cat $out/check.log | cut -c1-250
  BIG="11248,2048,1024;22496,2048,1024;44992,2048,1024;89984,2048,1024;89984,1024,2048;89984,3072,1024;89984,1024,1024"
  { for prec in fp16 fp16m; do
      KB_SHAPES=$BIG KB_PRECS=$prec KB_EPI=1 KB_VARIANTS=-1,50,80,1080,4080,8080,9080 timeout 600 python tools/kernel_bench.py gemm 2>&1 | grep ^gemm | sed "s/^/epi1 /" | cut -c1-400
      KB_SHAPES=$BIG KB_PRECS=$prec KB_EPI=2 KB_VARIANTS=-1,50,80 timeout 600 python tools/kernel_bench.py gemm 2>&1 | grep ^gemm | sed "s/^/epi2 /" | cut -c1-400
    done
    for sq in "8 1406" "32 1406" "64 1406"; do for prec in fp16 fp16m; do timeout 300 python tools/kernel_bench.py qkv $prec $sq -1,50,80 10 2>&1 | grep -E "^qkv" | awk 'NR%3==0'; done; done; } > $out/kernel_bench.log 2>&1
  cat $out/kernel_bench.log | cut -c1-330
  [ -x tools/probes/hipblaslt_ref ] && timeout 300 tools/probes/hipblaslt_ref > $out/hipblaslt_ref.log 2>&1; tail -4 $out/hipblaslt_ref.log
  ( cd /tmp; export TMPDIR=/tmp
    for v in 50 80; do for prec in fp16 fp16m; do
      timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES --output-format csv -d $R/$out/pmc_a_${prec}_$v -o c -- python $R/tools/kernel_bench.py one $prec $v 89984 2048 1024 3 > /dev/null 2>&1
      timeout 300 rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES --output-format csv -d $R/$out/pmc_b_${prec}_$v -o c -- python $R/tools/kernel_bench.py one $prec $v 89984 2048 1024 3 > /dev/null 2>&1
    done; done )
  python - $out <<'PY'
import csv, glob, sys, collections
for d in sorted(glob.glob(sys.argv[1] + "/pmc_*")):
    acc = collections.defaultdict(lambda: [0.0, 0])
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "gemm_p" in r.get("Kernel_Name", ""):
                a = acc[r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
    print(d.split("/")[-1], {k: round(v[0] / max(v[1], 1)) for k, v in acc.items()})
PY
  rm -rf $out/pmc_*
  Q="--no-cpu-baseline --no-other-configs"
  for v in -1 80 -1 80; do
    F5HIP_PP_VARIANT=$v timeout 900 python bench.py --steps 2 --warmup 1 --batch 32 --nfe 32 $Q > $out/b32_v$v.json 2>> $out/bench.err; line $out/b32_v$v.json b32_nfe32_variant$v
  done
  for v in -1 80; do
    F5HIP_PP_VARIANT=$v timeout 900 python bench.py --steps 3 --warmup 1 --batch 8 $Q > $out/b8_v$v.json 2>> $out/bench.err; line $out/b8_v$v.json b8_variant$v
  done
  tail -3 $out/bench.err ;;
mx)
  tag=${1:?tag}; out=gpurun_out/$tag; rm -rf $out; mkdir -p $out
  timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "fp16m or full_size or stress_golden or reference_example or small_models or configs2 or configs4 or packed_rows or trained_like or qkv_epilogue_score_corrections or sharpness" -s 2>&1 | grep -E "max-abs|passed|failed|rror" | cut -c1-220 > $out/gpu_tests_fp16m.log; tail -25 $out/gpu_tests_fp16m.log
  B1="2812,3072,1024;2812,1024,1024;2812,2048,1024;2812,1024,2048;1406,2048,1024;1406,1024,2048"
  BIG="11248,2048,1024;22496,1024,2048;89984,2048,1024;89984,1024,2048"
  { for epi in 1 2; do
      KB_SHAPES=$B1 KB_PRECS=fp16x3 KB_EPI=$epi KB_VARIANTS=-1 timeout 300 python tools/kernel_bench.py gemm 2>&1 | grep ^gemm | sed "s/^/epi$epi /" | cut -c1-330
      KB_SHAPES=$B1 KB_PRECS=fp16m KB_EPI=$epi KB_VARIANTS=-1,55,56,59,66,68,69 timeout 300 python tools/kernel_bench.py gemm 2>&1 | grep ^gemm | sed "s/^/epi$epi /" | cut -c1-400
      KB_SHAPES=$BIG KB_PRECS=fp16x3 KB_EPI=$epi KB_VARIANTS=-1 timeout 300 python tools/kernel_bench.py gemm 2>&1 | grep ^gemm | sed "s/^/epi$epi /" | cut -c1-330
      KB_SHAPES=$BIG KB_PRECS=fp16m KB_EPI=$epi KB_VARIANTS=-1,50,61,62,63 timeout 300 python tools/kernel_bench.py gemm 2>&1 | grep ^gemm | sed "s/^/epi$epi /" | cut -c1-400
    done
    KB_CHECK=1 KB_SHAPES="2812,1024,1024" KB_PRECS=fp16m KB_EPI=2 KB_VARIANTS=56,66 timeout 200 python tools/kernel_bench.py gemm 2>&1 | grep -E "KB_CHECK" | head -4
    for sq in "2 1406" "8 1406" "64 1406"; do
      timeout 300 python tools/kernel_bench.py qkv fp16x3 $sq -1 20 2>&1 | grep -E "^qkv" | awk 'NR%3==0'
      timeout 300 python tools/kernel_bench.py qkv fp16m $sq -1,50,56,61,68 20 2>&1 | grep -E "^qkv|QKV_CHECK" | awk '/QKV_CHECK/ || ++n%3==0'
    done; } > $out/kernel_bench_mx.log 2>&1
  cat $out/kernel_bench_mx.log | cut -c1-260
  for prec in fp16x3 fp16m fp16x3 fp16m; do timeout 600 python bench.py --steps 10 --warmup 3 --precision $prec --no-cpu-baseline --no-other-configs >> $out/bench_b1_ab.jsonl 2> $out/bench.err; done
  python - $out/bench_b1_ab.jsonl <<'PY'
import json, sys
for l in open(sys.argv[1]):
    d = json.loads(l); r = d.get("roofline", {})
    print("B=1", d["dtype"][:7], "ms/step", round(d["ms_per_step"], 2), "roofline.frac", round(r.get("frac", 0), 4), {k: round(v, 1) for k, v in d.get("kernel_classes_ms", {}).items() if v > 1})
PY
  for prec in fp16x3 fp16m; do timeout 900 python bench.py --steps 2 --warmup 1 --batch 32 --nfe 32 --precision $prec --no-cpu-baseline > $out/bench_b32_nfe32_$prec.json 2>> $out/bench.err; line $out/bench_b32_nfe32_$prec.json b32_$prec; done
  timeout 600 python bench.py --steps 3 --warmup 1 --batch 8 --precision fp16m --no-cpu-baseline > $out/bench_b8_fp16m.json 2>> $out/bench.err; line $out/bench_b8_fp16m.json b8_fp16m
  for prec in fp16x3 fp16m; do timeout 600 python bench.py --model E2TTS_Base --batch 8 --vocoder bigvgan --steps 3 --warmup 1 --precision $prec --no-cpu-baseline > $out/bench_e2_b8_bigvgan_$prec.json 2>> $out/bench.err; line $out/bench_e2_b8_bigvgan_$prec.json e2_b8_$prec; done ;;
tests)
  timeout 1800 python -m pytest ${@:-tests -q -m gpu} 2>&1 | tail -5 ;;
bench)
  tag=${1:?tag}; shift; out=gpurun_out/$tag; mkdir -p $out
  timeout 1500 python bench.py "$@" > $out/bench.json 2> $out/bench.err; line $out/bench.json "$tag" ;;
pmc)
  tag=${1:?tag}; shift; out=gpurun_out/$tag; mkdir -p $out; pmc $out $tag "$@" ;;
evidence)
  tag=${1:?tag}; out=gpurun_out/$tag; rm -rf $out; mkdir -p $out
  timeout 1800 python -m pytest tests -q -m gpu 2>&1 | tail -4 > $out/gpu_tests.log; cat $out/gpu_tests.log
  timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee $out/smoke.log
  # the counter passes first, installed as profiles/<tag>_pmc_*.json in this copy of the tree: the bench lines of the SAME call then quote them as
  # roofline.traffic (bench.py takes a committed PMC json only if its kernel_source_hash is that of the sources it runs)
  pmc $out fp16m_b1 --batch 1 --nfe 16 --no-other-configs
  pmc $out fp16m_b32 --batch 32 --nfe 2 --no-other-configs
  for f in $out/pmc_*.json; do cp $f profiles/${tag}_$(basename $f); done
  # the headline line exactly as the driver runs it (default precision fp16m; carries cpu_baseline and other_configs = configs[2], configs[4])
  timeout 1500 python bench.py --steps 10 --warmup 3 > $out/bench_b1.json 2> $out/bench_b1.err
  Q="--no-cpu-baseline --no-other-configs"
  timeout 600 python bench.py --steps 10 --warmup 3 --precision fp16x3 $Q > $out/bench_b1_fp16x3.json 2> /dev/null
  timeout 600 python bench.py --steps 10 --warmup 3 --precision fp16 $Q > $out/bench_b1_fp16.json 2> /dev/null
  timeout 600 python bench.py --steps 10 --warmup 3 --attn-impl 3 $Q > $out/bench_b1_plain_scores.json 2> /dev/null  # the attention of rounds 2-4 (plain fp16 q, k): outside the tolerance on the trained-like golden
  timeout 600 python bench.py --steps 10 --warmup 3 --branch-streams 1 $Q > $out/bench_b1_two_chains.json 2> /dev/null  # the cond / uncond halves as two concurrent chains of 1406 rows
  timeout 600 python bench.py --steps 3 --warmup 1 --batch 4 --nfe 32 $Q > $out/bench_b4_nfe32.json 2> /dev/null
  timeout 600 python bench.py --steps 3 --warmup 1 --batch 8 $Q > $out/bench_b8.json 2> /dev/null
  timeout 900 python bench.py --steps 2 --warmup 1 --batch 32 --nfe 32 $Q > $out/bench_b32_nfe32.json 2> /dev/null
  timeout 900 python bench.py --steps 2 --warmup 1 --batch 32 --nfe 32 --attn-impl 3 $Q > $out/bench_b32_nfe32_plain_scores.json 2> /dev/null
  timeout 900 python bench.py --steps 2 --warmup 1 --batch 32 --nfe 32 --precision fp16x3 $Q > $out/bench_b32_nfe32_fp16x3.json 2> /dev/null
  timeout 600 python bench.py --model E2TTS_Base --batch 8 --vocoder bigvgan --steps 3 --warmup 1 $Q > $out/bench_e2_b8_bigvgan.json 2> /dev/null
  for f in $out/bench_*.json; do line $f $(basename $f .json); done
  trace $out b1 --steps 3 --warmup 1 --no-other-configs
  trace $out b32_nfe32 --steps 1 --warmup 1 --batch 32 --nfe 32 --no-other-configs
  B1="2812,3072,1024;2812,1024,1024;2812,2048,1024;2812,1024,2048"
  MID="11248,2048,1024;22496,1024,2048;89984,2048,1024;89984,1024,2048"
  { for epi in 1 2; do
      KB_SHAPES=$B1 KB_PRECS=fp16x3,fp16m,fp16 KB_EPI=$epi KB_VARIANTS=-1 timeout 300 python tools/kernel_bench.py gemm 2>&1 | grep ^gemm | sed "s/^/epi$epi /" | cut -c1-200
      KB_SHAPES=$MID KB_PRECS=fp16x3,fp16m,fp16 KB_EPI=$epi KB_VARIANTS=-1 timeout 400 python tools/kernel_bench.py gemm 2>&1 | grep ^gemm | sed "s/^/epi$epi /" | cut -c1-200
    done
    for sq in "2 1406" "8 1406" "64 1406"; do for prec in fp16x3 fp16m; do timeout 300 python tools/kernel_bench.py qkv $prec $sq -1 20 2>&1 | grep -E "^qkv" | awk 'NR%3==0'; done; done
    KB_ATTN_LOG2Q=1 timeout 300 python tools/kernel_bench.py attn 2>&1 | grep ^attn | sed "s/^/lazy /"; } > $out/kernel_bench.log 2>&1
  tail -12 $out/kernel_bench.log | cut -c1-200
  [ -x tools/probes/hipblaslt_ref ] && timeout 300 tools/probes/hipblaslt_ref > $out/hipblaslt_ref.log 2>&1
  timeout 900 python tools/attn_precision_check.py > $out/attn_precision.log 2>&1; tail -6 $out/attn_precision.log ;;
counters)  # the rocprofv3 half of `evidence` alone (kernel-trace summaries + PMC passes)
  tag=${1:?tag}; out=gpurun_out/$tag; mkdir -p $out
  trace $out b1 --steps 3 --warmup 1 --no-other-configs
  trace $out b32_nfe32 --steps 1 --warmup 1 --batch 32 --nfe 32 --no-other-configs
  pmc $out fp16m_b1 --batch 1 --nfe 16 --no-other-configs
  pmc $out fp16m_b32 --batch 32 --nfe 2 --no-other-configs ;;
tiles)
  tag=${1:?tag}; out=gpurun_out/$tag; mkdir -p $out
  KB_SHAPES=${2:?shapes} KB_VARIANTS=${3:?variants} KB_EPI=${4:-1} KB_PRECS=${5:-fp16x3} timeout 900 python tools/kernel_bench.py gemm 2>&1 | grep ^gemm | cut -c1-340 | tee $out/tiles.log ;;
qkv)
  tag=${1:?tag}; variants=${2:?variants}; shift 2; out=gpurun_out/$tag; mkdir -p $out
  for sq in "$@"; do timeout 600 python tools/kernel_bench.py qkv fp16x3 $sq $variants 20 2>&1 | grep -E "^qkv|QKV_CHECK"; done | tee $out/qkv.log ;;
attn)
  tag=${1:?tag}; out=gpurun_out/$tag; mkdir -p $out
  { echo "== exact running maximum"; timeout 300 python tools/kernel_bench.py attn 2>&1 | grep ^attn
    echo "== lazy reference maximum (KB_ATTN_LOG2Q=1)"; KB_ATTN_LOG2Q=1 timeout 300 python tools/kernel_bench.py attn 2>&1 | grep ^attn; } | tee $out/attn_ab.log ;;
race)
  tag=${1:?tag}; out=gpurun_out/$tag; mkdir -p $out
  KB_PROBE_DUMP=$R/$out/dump timeout 900 python tools/kernel_bench.py qkvprobe ${2:?seqs} ${3:?nseq} ${4:?reps} "${5:?specs}" 2>&1 | grep -E "qkvprobe|QKV_PROBE|rror" | tee $out/probe.log
  for f in $out/dump.*.bin; do [ -s $f ] && [ $(stat -c %s $f) -lt 4000000 ] && python tools/race_dump_analyze.py $f $2 $3 16 > ${f%.bin}.txt 2>&1; done
  rm -f $out/*.bin; head -c 2500 $(ls $out/dump.*.txt 2>/dev/null | head -1) 2>/dev/null ;;
pkprobe)
  tag=${1:?tag}; out=$R/gpurun_out/$tag; mkdir -p $out; cd tools/probes
  [ -f pk_opsel_sweep.hip ] || python gen_pk_opsel_sweep.py  # the sweep's source is generated (1 300 lines): only the generator is in the tree
  for b in pk_opsel_probe pk_opsel_sweep; do [ -x $b ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -o $b $b.hip 2> /dev/null; done
  { for m in 0 1 4 5 13; do timeout 120 ./pk_opsel_probe 2 $m 20000 | tail -1; done; } > $out/pk_probe.log 2>&1
  timeout 600 ./pk_opsel_sweep 4000 1 > $out/sweep_partners.log 2>&1; timeout 600 ./pk_opsel_sweep 4000 0 > $out/sweep_alone.log 2>&1
  cat $out/pk_probe.log; grep -E "FAILS|forms fail" $out/sweep_partners.log | cut -c1-160; tail -1 $out/sweep_alone.log ;;
ragged)
  tag=${1:?tag}; shift; out=gpurun_out/$tag; mkdir -p $out
  for fpb in ${@:-6000 24000}; do
    echo "== frame budget $fpb: 200 length classes (the reference's bucketing)"; timeout 900 python tools/infer_batch.py --synthetic 48 --frames-per-batch $fpb --attn-mask --out /tmp/o1 2>&1 | tail -2
    echo "== frame budget $fpb: one class (list order), padded layout"; timeout 900 python tools/infer_batch.py --synthetic 48 --frames-per-batch $fpb --attn-mask --num-buckets 1 --out /tmp/o2 2>&1 | tail -2
    echo "== frame budget $fpb: one class (list order), packed rows"; timeout 900 python tools/infer_batch.py --synthetic 48 --frames-per-batch $fpb --attn-mask --num-buckets 1 --packed --out /tmp/o3 2>&1 | tail -2
  done | tee $out/packed_ragged.log ;;
*)
  echo "unknown recipe $recipe" >&2; exit 2 ;;
esac
