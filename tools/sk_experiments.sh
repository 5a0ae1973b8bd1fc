#!/bin/bash
# stream-K GEMM experiments of round 1 (variants 40 / 41, gemm_sk.h), in the order they were run; results in DESIGN.md 4.  usage: bash tools/sk_experiments.sh [1-5]
sel=${1:-all}

if [ "$sel" = all ] || [ "$sel" = 1 ]; then  # ---- experiment 1
# stream-K GEMM: bit-exactness against the plain tiling (same summation order inside a k-tile; partial sums are added in a fixed order, so
# results may differ in the last bits where a tile was split: reported as differing bytes) and timing at the B=1 shapes
for shape in "2816 1024 1024" "2816 1024 2048" "2816 2048 1024" "2816 3072 1024" "1408 1024 1024" "1408 3072 1024"; do set -- $shape
  for v in 40 41; do KB_CHECK=1 KB_EPI=1 timeout 120 python tools/kernel_bench.py one fp16x3 $v $1 $2 $3 20 2>&1 | grep -E "^gemm|KB_CHECK"; done; done
fi

if [ "$sel" = all ] || [ "$sel" = 2 ]; then  # ---- experiment 2
for G in 88 176 256; do for v in 40; do echo -n "G=$G "; KB_SKGRID=$G KB_CHECK=1 KB_EPI=1 timeout 120 python tools/kernel_bench.py one fp16x3 $v 2816 1024 1024 20 2>&1 | grep -E "^gemm|rep 2" | tr '\n' ' '; echo; done; done
echo -n "v30 "; KB_EPI=1 python tools/kernel_bench.py one fp16x3 30 2816 1024 1024 20 2>&1 | grep "^gemm"
fi

if [ "$sel" = all ] || [ "$sel" = 3 ]; then  # ---- experiment 3
run() { echo -n "$1 G=$2 M=$3 N=$4 K=$5: "; KB_SKGRID=$2 KB_EPI=1 timeout 120 python tools/kernel_bench.py one fp16x3 $1 $3 $4 $5 20 2>&1 | grep -E "^gemm" | sed 's/.*st=0//'; }
run 40 256 2048 4096 1024   # 256 tiles, one per workgroup
run 30 256 2048 4096 1024
run 40 256 2048 2048 1024   # 128 tiles, two workgroups per tile
run 40 128 2048 2048 1024   # 128 tiles, one per workgroup
run 30 128 2048 2048 1024
run 40 256 2048 1024 1024   # 64 tiles, four workgroups per tile
run 40 192 2048 2048 1024   # 128 tiles, 1.5 workgroups per tile
fi

if [ "$sel" = all ] || [ "$sel" = 4 ]; then  # ---- experiment 4
export F5HIP_SK_DEBUG=1
KB_SKGRID=256 KB_EPI=1 timeout 120 python tools/kernel_bench.py one fp16x3 40 2048 2048 1024 5 2>&1 | grep -E "^gemm|sk wg"
KB_SKGRID=128 KB_EPI=1 timeout 120 python tools/kernel_bench.py one fp16x3 40 2048 2048 1024 5 2>&1 | grep -E "^gemm|sk wg"
fi

if [ "$sel" = all ] || [ "$sel" = 5 ]; then  # ---- experiment 5
export F5HIP_SK_DEBUG=1
KB_CHECK=1 KB_SKGRID=128 KB_EPI=1 timeout 120 python tools/kernel_bench.py one fp16x3 40 2048 2048 1024 5 2>&1 | grep -E "^gemm|sk wg   0|sk wg   8|rep 2"
KB_CHECK=1 KB_SKGRID=256 KB_EPI=1 timeout 120 python tools/kernel_bench.py one fp16x3 40 2048 2048 1024 5 2>&1 | grep -E "^gemm|sk wg   0|sk wg   8|rep 2"
unset F5HIP_SK_DEBUG
for shape in "2816 1024 1024" "2816 1024 2048" "2816 2048 1024" "2816 3072 1024"; do set -- $shape
  for v in 40 41; do KB_EPI=1 timeout 120 python tools/kernel_bench.py one fp16x3 $v $1 $2 $3 20 2>&1 | grep -E "^gemm"; done; done
KB_EPI=0 timeout 120 python tools/kernel_bench.py one fp16x3 40 2816 3072 1024 20 2>&1 | grep -E "^gemm"
KB_EPI=2 timeout 120 python tools/kernel_bench.py one fp16x3 40 2816 1024 1024 20 2>&1 | grep -E "^gemm"
KB_EPI=2 timeout 120 python tools/kernel_bench.py one fp16x3 6 2816 1024 1024 20 2>&1 | grep -E "^gemm"
fi
