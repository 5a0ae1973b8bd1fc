run() { python bench.py --schedule default --steps 4 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],2))"; }
echo "base $(run)"
for v in 2 10 6; do echo "N3072=$v $(F5HIP_GEMM_VARIANT_N3072=$v run)"; done
for v in 2 6 10; do echo "N2048=$v $(F5HIP_GEMM_VARIANT_N2048=$v run)"; done
for v in 1 2 8; do echo "N1024=$v $(F5HIP_GEMM_VARIANT_N1024=$v run)"; done
echo "base $(run)"
