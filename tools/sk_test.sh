# stream-K GEMM: bit-exactness against the plain tiling (same summation order inside a k-tile; partial sums are added in a fixed order, so
# results may differ in the last bits where a tile was split: reported as differing bytes) and timing at the B=1 shapes
for shape in "2816 1024 1024" "2816 1024 2048" "2816 2048 1024" "2816 3072 1024" "1408 1024 1024" "1408 3072 1024"; do set -- $shape
  for v in 40 41; do KB_CHECK=1 KB_EPI=1 timeout 120 python tools/kernel_bench.py one fp16x3 $v $1 $2 $3 20 2>&1 | grep -E "^gemm|KB_CHECK"; done; done
