run() { env "$@" timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-other-configs 2>>gpurun_out/r04v/err | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$*', round(d['ms_per_step'],2), {k: round(v,1) for k,v in d['kernel_classes_ms'].items() if v>1})" | tee -a gpurun_out/r04v/b1_knobs.log; }
mkdir -p gpurun_out/r04v
run X=0
run F5HIP_ATTN_WAVES=4
run F5HIP_ATTN_PIPE=0
run F5HIP_ATTN_VALU_SUM=0
run F5HIP_LN_EARLY=1
run F5HIP_LN_LATE=1
run X=0
run F5HIP_PP_VARIANT_N2048=55
run F5HIP_PP_VARIANT_N3072=56
run F5HIP_PP_VARIANT_N1024=63
run F5HIP_PP_VARIANT_N1024=66
run X=0
