# does a power-of-two operand row stride (K = 1024 -> 4 KB rows) camp on L2 channels?  compare per-k-tile time with odd strides
for K in 992 1024 1056 1120 2016 2048 2080; do KB_EPI=1 python tools/kernel_bench.py one fp16x3 6 2816 1024 $K 30 2>&1 | grep "^gemm"; done
for K in 992 1024 1056; do KB_EPI=1 python tools/kernel_bench.py one fp16x3 1 2816 3072 $K 30 2>&1 | grep "^gemm"; done
