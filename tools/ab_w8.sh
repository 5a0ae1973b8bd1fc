for M in 1408 2816 5632 11264; do
for shape in "1024 1024" "1024 2048" "2048 1024" "3072 1024"; do set -- $shape
  for v in 1 6 30 31 21; do echo -n "M=$M "; KB_EPI=1 python tools/kernel_bench.py one fp16x3 $v $M $1 $2 20 2>&1 | grep "^gemm" | sed 's/gemm fp16x3 //'; done; done; done
