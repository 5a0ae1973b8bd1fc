#!/bin/bash
# A/B within one box: k-split 66 vs 59 for the narrow outputs, LN early/paired vs late; and the batch-rows test's error with each
set -u
out=$GRAFT_REPO_ROOT/gpurun_out/r2c18; mkdir -p $out
cd $GRAFT_REPO_ROOT
cat > /tmp/rows.py <<'PY'
import sys, torch
sys.path.insert(0, ".")
from f5_tts_amd import synth
from f5_tts_amd.engine import F5HipCFM, F5HipEngine
from oracle import make_golden as MG
c = MG.FULL_CASES["base_v1_cfg1"]
cfg, wav, text, duration, lens = MG.case_inputs(c)
eng = F5HipEngine(cfg, None, device=0)
eng.load_state_dict(synth.synth_dit_state_dict(cfg, seed=c["wseed"]))
model = F5HipCFM(eng, precision="fp16x3")
kw = dict(c["kw"], steps=4)
for rep in range(3):
    one, _ = model.sample(wav.cuda(), text, duration, **kw)
    many, _ = model.sample(wav.repeat(4, 1).cuda(), text.repeat(4, 1), duration, **kw)
    print("rows", [float((many[b].cpu() - one[0].cpu()).abs().max()) for b in range(4)], bool(torch.equal(many[0], many[3])), flush=True)
    if rep == 0: first = one.clone()
    else: print("B=1 repeat equal", bool(torch.equal(first, one)))
eng.close()
PY
for e in "X=1" "F5HIP_PP_VARIANT_N1024=59"; do echo "== $e"; env $e timeout 300 python /tmp/rows.py 2>&1 | grep -E "rows|repeat|Error" ; done
for e in "X=1" "F5HIP_PP_VARIANT_N1024=59" "F5HIP_LN_LATE=1" "X=1" "F5HIP_PP_VARIANT_N1024=59" "F5HIP_LN_LATE=1"; do
env $e timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --batch 1 --nfe 16 > $out/b1.json 2>$out/b1.err
python - <<PY
import json
d=json.loads(open("$out/b1.json").read().strip().splitlines()[-1]); k=d["kernel_classes_ms"]; print("b1 $e", round(d["ms_per_step"],2), k["gemm_block"], k["attention"], k["ln_modulate"])
PY
done
for e in "X=1" "F5HIP_LN_LATE=1"; do
env $e timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --batch 32 --nfe 32 > $out/b32.json 2>$out/b32.err
python - <<PY
import json
d=json.loads(open("$out/b32.json").read().strip().splitlines()[-1]); k=d["kernel_classes_ms"]; print("b32 $e", round(d["ms_per_step"],2), k["gemm_block"], k["attention"], k["ln_modulate"])
PY
done
