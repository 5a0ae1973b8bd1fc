export F5HIP_SK_DEBUG=1
KB_CHECK=1 KB_SKGRID=128 KB_EPI=1 timeout 120 python tools/kernel_bench.py one fp16x3 40 2048 2048 1024 5 2>&1 | grep -E "^gemm|sk wg   0|sk wg   8|rep 2"
KB_CHECK=1 KB_SKGRID=256 KB_EPI=1 timeout 120 python tools/kernel_bench.py one fp16x3 40 2048 2048 1024 5 2>&1 | grep -E "^gemm|sk wg   0|sk wg   8|rep 2"
unset F5HIP_SK_DEBUG
for shape in "2816 1024 1024" "2816 1024 2048" "2816 2048 1024" "2816 3072 1024"; do set -- $shape
  for v in 40 41; do KB_EPI=1 timeout 120 python tools/kernel_bench.py one fp16x3 $v $1 $2 $3 20 2>&1 | grep -E "^gemm"; done; done
KB_EPI=0 timeout 120 python tools/kernel_bench.py one fp16x3 40 2816 3072 1024 20 2>&1 | grep -E "^gemm"
KB_EPI=2 timeout 120 python tools/kernel_bench.py one fp16x3 40 2816 1024 1024 20 2>&1 | grep -E "^gemm"
KB_EPI=2 timeout 120 python tools/kernel_bench.py one fp16x3 6 2816 1024 1024 20 2>&1 | grep -E "^gemm"
