# fixed cost vs per-k-tile cost of a B=1 launch: K sweep
for E in 1 0; do for K in 64 256 512 1024 2048 4096; do echo -n "epi=$E "; KB_EPI=$E python tools/kernel_bench.py one fp16x3 6 2816 1024 $K 30 2>&1 | grep "^gemm"; done; done
for K in 64 256 1024 2048; do echo -n "M=4096 "; KB_EPI=1 python tools/kernel_bench.py one fp16x3 6 4096 1024 $K 30 2>&1 | grep "^gemm"; done
for K in 64 256 1024 2048; do echo -n "M=1408 "; KB_EPI=1 python tools/kernel_bench.py one fp16x3 6 1408 1024 $K 30 2>&1 | grep "^gemm"; done
