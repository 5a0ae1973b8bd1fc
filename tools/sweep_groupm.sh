# tile rasterisation sweep at B=1 shapes: F5HIP_GEMM_GROUPM = row-tiles per group (row-tile fastest inside a group)
for G in 0 4 6 11 22; do
  for shape in "6 2816 1024 1024" "6 2816 1024 2048" "1 2816 3072 1024" "1 2816 2048 1024" "6 1408 1024 1024" "1 1408 3072 1024"; do
    set -- $shape
    echo -n "groupm=$G "; F5HIP_GEMM_GROUPM=$G KB_EPI=1 python tools/kernel_bench.py one fp16x3 $1 $2 $3 $4 30 2>&1 | grep "^gemm"
  done
done
