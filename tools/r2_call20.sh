#!/bin/bash
# BigVGAN conv implementations at full size: agreement with the default, then time inside configs[4]
set -u
R=$GRAFT_REPO_ROOT; out=$R/gpurun_out/r2c20; mkdir -p $out
cd $R
python - <<'PY' 2>&1 | tail -8
import torch, sys
sys.path.insert(0, ".")
from f5_tts_amd import config, synth
from f5_tts_amd.bigvgan import F5HipBigVGAN
bcfg = config.BIGVGAN_V2_24K_100B_256X
for prec in ("fp16x3", "fp32"):
    voc = F5HipBigVGAN(bcfg, device=0, precision=prec).load_state_dict(synth.synth_bigvgan_state_dict(bcfg, seed=0))
    mel = torch.randn(2, 100, 300, generator=torch.Generator().manual_seed(3)).cuda()
    ref = voc(mel).cpu()
    for impl in (1, 2):
        voc.set_option("conv_impl", impl)
        got = voc(mel).cpu()
        again = voc(mel).cpu()
        print(prec, "conv_impl", impl, "max |diff| vs impl 0:", float((got - ref).abs().max()), "of", float(ref.abs().max()), "repeatable", bool(torch.equal(got, again)))
PY
for impl in 0 1 2 0 1 2; do
timeout 600 python bench.py --model E2TTS_Base --batch 8 --vocoder bigvgan --steps 3 --warmup 1 --no-cpu-baseline --bigvgan-conv-impl $impl > $out/e2_impl$impl.json 2> $out/e2_impl$impl.err
python - <<PY
import json
d=json.loads(open("$out/e2_impl$impl.json").read().strip().splitlines()[-1]); print("conv_impl $impl", round(d["ms_per_step"],1), {k:(v["ms"]) for k,v in d["vocoder_classes"].items()})
PY
done
