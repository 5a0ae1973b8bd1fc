export F5HIP_SK_DEBUG=1
KB_SKGRID=256 KB_EPI=1 timeout 120 python tools/kernel_bench.py one fp16x3 40 2048 2048 1024 5 2>&1 | grep -E "^gemm|sk wg"
KB_SKGRID=128 KB_EPI=1 timeout 120 python tools/kernel_bench.py one fp16x3 40 2048 2048 1024 5 2>&1 | grep -E "^gemm|sk wg"
