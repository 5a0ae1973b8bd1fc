for w in 4 6; do for p in fp16x3 fp16; do echo -n "waves=$w "; F5HIP_ATTN_WAVES=$w python tools/kernel_bench.py --schedule default oneattn $p 2 1406 30 2>&1 | grep "^attn"; done; done
for w in 4 6; do echo -n "waves=$w B'=1 "; F5HIP_ATTN_WAVES=$w python tools/kernel_bench.py --schedule default oneattn fp16x3 1 1406 30 2>&1 | grep "^attn"; done
python -m pytest tests -m gpu -x -q -k "flash or key_padding or golden" 2>&1 | tail -2
F5HIP_ATTN_WAVES=6 python -m pytest tests -m gpu -x -q -k "flash or key_padding or mmdit or ragged" 2>&1 | tail -2
for i in 1 2; do python bench.py --schedule default --steps 5 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d[\"ms_per_step\"], d[\"value\"])"; done
