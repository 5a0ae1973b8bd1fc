for M in 1408 2048 2816 4096 5632 6144 8192; do KB_EPI=1 python tools/kernel_bench.py one fp16x3 6 $M 1024 1024 30 2>&1 | grep "^gemm"; done
for M in 1408 2048 2688 2816 4096 8192; do KB_EPI=1 python tools/kernel_bench.py one fp16x3 1 $M 3072 1024 30 2>&1 | grep "^gemm"; done
for M in 2816 4096 8192; do KB_EPI=1 python tools/kernel_bench.py one fp16x3 6 $M 3072 1024 30 2>&1 | grep "^gemm"; done
