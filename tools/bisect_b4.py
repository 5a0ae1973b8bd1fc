import os, sys, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import f5_tts_amd
from f5_tts_amd import config, synth
from f5_tts_amd.engine import F5HipCFM, F5HipEngine
from oracle import make_golden as MG
c = MG.FULL_CASES["base_v1_cfg1"]
cfg, wav, text, duration, lens = MG.case_inputs(c)
eng = F5HipEngine(cfg, None, device=0)
eng.load_state_dict(synth.synth_dit_state_dict(cfg, seed=c["wseed"]))
model = F5HipCFM(eng, precision="fp16x3")
kw = dict(c["kw"], steps=2)
one, _ = model.sample(wav.cuda(), text, duration, **kw)
for bs in (0, 1):
    eng.set_option("branch_streams", bs)
    for rep in range(3):
        many, _ = model.sample(wav.repeat(4, 1).cuda(), text.repeat(4, 1), duration, **kw)
        print("branch_streams", bs, "rep", rep, "err", [round(float((many[b] - one[0]).abs().max()), 5) for b in range(4)], flush=True)
