for shape in "2816 1024 1024" "2816 1024 2048" "2816 3072 1024" "2816 2048 1024" "1408 1024 1024" "1408 3072 1024"; do set -- $shape
  for v in 6 26 27 28 29 7; do KB_EPI=1 python tools/kernel_bench.py one fp16x3 $v $1 $2 $3 30 2>&1 | grep "^gemm"; done; done
