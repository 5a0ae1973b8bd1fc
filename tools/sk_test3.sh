run() { echo -n "$1 G=$2 M=$3 N=$4 K=$5: "; KB_SKGRID=$2 KB_EPI=1 timeout 120 python tools/kernel_bench.py one fp16x3 $1 $3 $4 $5 20 2>&1 | grep -E "^gemm" | sed 's/.*st=0//'; }
run 40 256 2048 4096 1024   # 256 tiles, one per workgroup
run 30 256 2048 4096 1024
run 40 256 2048 2048 1024   # 128 tiles, two workgroups per tile
run 40 128 2048 2048 1024   # 128 tiles, one per workgroup
run 30 128 2048 2048 1024
run 40 256 2048 1024 1024   # 64 tiles, four workgroups per tile
run 40 192 2048 2048 1024   # 128 tiles, 1.5 workgroups per tile
