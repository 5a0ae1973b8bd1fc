#!/bin/bash
# round 3, call 13: per-output-channel weight conditioning — the stress / standard / real-example goldens with and without it, B = 1 bench
set -u
out=$GRAFT_REPO_ROOT/gpurun_out/r3c13; mkdir -p $out
cd $GRAFT_REPO_ROOT
T="tests/test_gpu_parity.py::test_dynamic_range_stress_golden_full_size tests/test_gpu_parity.py::test_full_size_base_config1_golden tests/test_gpu_parity.py::test_reference_example_prompt_and_text_golden tests/test_gpu_parity.py::test_full_size_e2_unett_golden"
echo "== conditioned" > $out/cond.log
timeout 900 python -m pytest $T -q -m gpu -s 2>&1 | grep -E "max-abs|passed|failed" >> $out/cond.log
echo "== F5HIP_NO_WEIGHT_CONDITIONING=1" >> $out/cond.log
F5HIP_NO_WEIGHT_CONDITIONING=1 timeout 900 python -m pytest $T -q -m gpu -s 2>&1 | grep -E "max-abs|passed|failed" >> $out/cond.log
cat $out/cond.log
timeout 600 python bench.py --no-cpu-baseline > $out/bench_b1.json 2> $out/bench_b1.err; python - <<PY
import json
d=json.loads(open("$out/bench_b1.json").read().strip().splitlines()[-1]); print("B=1", round(d["ms_per_step"],2), {k: round(v,1) for k,v in d["kernel_classes_ms"].items() if v > 1})
PY
timeout 1200 python -m pytest tests -q -m gpu -x 2>&1 | tail -2
