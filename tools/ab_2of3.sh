# upper bound on what cheaper correction products (MX-fp8) could buy: 2 of the 3 fp16x3 MFMAs, same bytes
for shape in "2816 1024 1024" "2816 1024 2048" "2816 3072 1024" "2816 2048 1024"; do set -- $shape
  for v in 6 24; do KB_EPI=1 python tools/kernel_bench.py one fp16x3 $v $1 $2 $3 30 2>&1 | grep "^gemm"; done; done
for shape in "22496 3072 1024" "89984 2048 1024" "89984 1024 2048"; do set -- $shape
  for v in 21 25; do KB_EPI=1 python tools/kernel_bench.py one fp16x3 $v $1 $2 $3 8 2>&1 | grep "^gemm"; done; done
