#!/bin/bash
set -u
out=$GRAFT_REPO_ROOT/gpurun_out/r2c5; mkdir -p $out
cd $GRAFT_REPO_ROOT
T="tests/test_gpu_parity.py::test_full_size_batch_rows_equal_single_utterance"
for v in 0 50 51 57 58 61; do
  echo "== F5HIP_PP_VARIANT=$v"; F5HIP_PP_VARIANT=$v timeout 300 python -m pytest $T -x -q -m gpu 2>&1 | grep -E "passed|failed|AssertionError: assert" | head -3
done > $out/bisect.log 2>&1
echo "== default, groupm 1" >> $out/bisect.log; F5HIP_GEMM_GROUPM=1 timeout 300 python -m pytest $T -x -q -m gpu 2>&1 | grep -E "passed|failed|AssertionError: assert" | head -3 >> $out/bisect.log
for n in 3072 2048 1024; do echo "== only N=$n through variant 58, rest old" >> $out/bisect.log
  a=0; b=0; c=0; [ $n = 3072 ] && a=58; [ $n = 2048 ] && b=58; [ $n = 1024 ] && c=58
  F5HIP_PP_VARIANT_N3072=$a F5HIP_PP_VARIANT_N2048=$b F5HIP_PP_VARIANT_N1024=$c timeout 300 python -m pytest $T -x -q -m gpu 2>&1 | grep -E "passed|failed|AssertionError: assert" | head -3 >> $out/bisect.log
done
for epi in 1 2; do KB_CHECK=1 KB_EPI=$epi timeout 120 python tools/kernel_bench.py one fp16x3 58 11248 1024 1024 3 2>&1 | grep -E "KB_CHECK|^gemm"; done >> $out/bisect.log 2>&1
cat $out/bisect.log
