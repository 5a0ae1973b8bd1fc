#!/bin/bash
# First GPU call of the next round: everything that was written without GPU minutes, most valuable first, each step bounded.
#   gpurun --timeout 1500 -- 'bash tools/r2_first_call.sh'        (outputs under gpurun_out/r2_first/)
set -u
out=gpurun_out/r2_first; mkdir -p $out
# 1. BigVGAN first light (tests/test_zz_gpu_bigvgan.py), strict
F5HIP_BIGVGAN_GPU=1 timeout 900 python -m pytest tests/test_zz_gpu_bigvgan.py -x -q -m gpu > $out/bigvgan_tests.log 2>&1; echo "bigvgan tests exit $?" | tee -a $out/summary.txt
F5HIP_STREAMK_GPU=1 timeout 900 python -m pytest tests/test_zz_gpu_streamk.py -x -q -m gpu > $out/streamk_tests.log 2>&1; echo "streamk tests exit $?" | tee -a $out/summary.txt
# 2. stream-K with the reduce-scattered epilogue (variants 42 / 43): correctness against the plain tiling (KB_CHECK: byte + numeric distance,
#    err word, flags left set), then time against the B = 1 defaults (6 for N <= 1024, 1 otherwise) and the old stream-K (40 / 41)
for shape in "2816 1024 1024" "2816 1024 2048" "2816 2048 1024" "2816 3072 1024"; do set -- $shape
  for v in 42 43; do KB_CHECK=1 KB_EPI=1 timeout 120 python tools/kernel_bench.py one fp16x3 $v $1 $2 $3 5 2>&1 | grep -E "^gemm|rep 2"; done
done > $out/skrs_check.log 2>&1
for epi in 1 2; do for shape in "2816 1024 1024" "2816 1024 2048" "2816 2048 1024" "2816 3072 1024"; do set -- $shape
  for v in 1 6 40 41 42 43; do KB_EPI=$epi timeout 120 python tools/kernel_bench.py one fp16x3 $v $1 $2 $3 20 2>&1 | grep -E "^gemm" | sed "s/^/epi$epi /"; done
done; done > $out/skrs_time.log 2>&1
for g in 128 192 256; do KB_SKGRID=$g KB_EPI=2 timeout 120 python tools/kernel_bench.py one fp16x3 42 2816 1024 2048 20 2>&1 | grep -E "^gemm" | sed "s/^/grid$g /"; done >> $out/skrs_time.log 2>&1
# 3. BASELINE configs[4] as named (E2-TTS Base + BigVGAN, batch 8), the three conv implementations are a context option: default (0) only here
timeout 600 python bench.py --schedule default --model E2TTS_Base --batch 8 --vocoder bigvgan --steps 3 --warmup 1 --no-cpu-baseline > $out/bench_e2_bigvgan_b8.json 2> $out/bench_e2_bigvgan_b8.err
# 3b. the headline with the stream-K block GEMMs (packed schedule) against the default two-chain schedule
for sk in 42 43; do F5HIP_BENCH_STREAMK=$sk timeout 600 python bench.py --branch-streams 0 --no-cpu-baseline > $out/bench_b1_sk$sk.json 2> $out/bench_b1_sk$sk.err; done
F5HIP_SK_GENERIC_EPI=1 F5HIP_BENCH_STREAMK=42 timeout 600 python bench.py --branch-streams 0 --no-cpu-baseline > $out/bench_b1_sk42_generic_epi.json 2> $out/bench_b1_sk42_generic_epi.err
F5HIP_BENCH_STREAMK=42 F5HIP_BENCH_STREAMK_SPLIT=1 timeout 600 python bench.py --branch-streams 1 --no-cpu-baseline > $out/bench_b1_sk42_split.json 2> $out/bench_b1_sk42_split.err
timeout 600 python bench.py --schedule default --branch-streams 0 --no-cpu-baseline > $out/bench_b1_packed.json 2> $out/bench_b1_packed.err
for kvs in 2 3; do F5HIP_BENCH_KVSPLIT=$kvs timeout 600 python bench.py --no-cpu-baseline > $out/bench_b1_kvsplit$kvs.json 2> $out/bench_b1_kvsplit$kvs.err; done
F5HIP_BENCH_KVSPLIT=2 F5HIP_BENCH_STREAMK=42 timeout 600 python bench.py --branch-streams 0 --no-cpu-baseline > $out/bench_b1_sk42_kvsplit2.json 2> $out/bench_b1_sk42_kvsplit2.err
# 3c. calibration: the vendor library on a PLAIN fp16 GEMM at the same shapes (what the part does vs what our fused k-loop loses)
timeout 300 tools/probes/hipblaslt_ref > $out/hipblaslt_ref.log 2>&1
# 3d. the bench as the driver runs it (--schedule auto: children probe the switches above, see config.schedule in the line) and the receiver-path first light
timeout 900 python bench.py > $out/bench_b1_auto.json 2> $out/bench_b1_auto.err
F5HIP_FIRST_LIGHT_GPU=1 timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k receiver > $out/first_light_tests.log 2>&1; echo "first-light tests exit $?" | tee -a $out/summary.txt
# 3e. what the static fixes of the end of round 1 bought: the QKV epilogue's general index path again (the conv-pos scratch fix has no switch:
#     compare bench_b1.json's kernel_classes.convpos with profiles/r01f_bench_b1_fp16x3.json)
F5HIP_QKV_EPI_GENERIC=1 timeout 600 python bench.py --schedule default --no-cpu-baseline > $out/bench_b1_qkv_generic.json 2> $out/bench_b1_qkv_generic.err
# 4. the headline, unchanged code path (regression check of the header refactors: F5_DYN_LDS macro, split headers)
timeout 600 python bench.py --schedule default > $out/bench_b1.json 2> $out/bench_b1.err
cat $out/hipblaslt_ref.log; tail -3 $out/bigvgan_tests.log; cat $out/skrs_check.log; cat $out/skrs_time.log; tail -3 $out/streamk_tests.log; cat $out/bench_e2_bigvgan_b8.json $out/bench_b1_sk42.json $out/bench_b1_sk43.json $out/bench_b1_sk42_generic_epi.json $out/bench_b1_sk42_split.json $out/bench_b1_kvsplit2.json $out/bench_b1_kvsplit3.json $out/bench_b1_sk42_kvsplit2.json $out/bench_b1_packed.json $out/bench_b1_auto.json $out/bench_b1_qkv_generic.json $out/bench_b1.json
