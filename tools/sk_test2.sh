for G in 88 176 256; do for v in 40; do echo -n "G=$G "; KB_SKGRID=$G KB_CHECK=1 KB_EPI=1 timeout 120 python tools/kernel_bench.py one fp16x3 $v 2816 1024 1024 20 2>&1 | grep -E "^gemm|rep 2" | tr '\n' ' '; echo; done; done
echo -n "v30 "; KB_EPI=1 python tools/kernel_bench.py one fp16x3 30 2816 1024 1024 20 2>&1 | grep "^gemm"
