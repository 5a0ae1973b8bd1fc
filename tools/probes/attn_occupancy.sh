export KB_ATTN_LOG2Q=1
for cfg in "2 1024" "4 1024" "8 1024" "2 1406" "2 1536"; do
  for pipe in 1 0; do
    echo -n "waves4 pipe$pipe: "; F5HIP_ATTN_WAVES=4 F5HIP_ATTN_PIPE=$pipe timeout 120 python tools/kernel_bench.py oneattn fp16 $cfg 20 2>&1 | grep "^attn"
  done
  echo -n "waves6 pipe1: "; F5HIP_ATTN_WAVES=6 F5HIP_ATTN_PIPE=1 timeout 120 python tools/kernel_bench.py oneattn fp16 $cfg 20 2>&1 | grep "^attn"
done
