"""How does the CPU oracle scale with OpenMP threads on this host?  (bench.py's cpu_baseline picks its thread count from this.)"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import f5_tts_amd  # noqa: E402,F401
from f5_tts_amd import config, synth  # noqa: E402
from oracle import f5_oracle as O  # noqa: E402

print("cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
try:
    print("cpu.max", open("/sys/fs/cgroup/cpu.max").read().strip())
except Exception as e:
    print("cpu.max n/a", e)
os.system("lscpu | grep -E 'Model name|^CPU\\(s\\)|Thread|Socket|NUMA node\\(s\\)'")
cfg = config.F5TTS_V1_BASE
sd = synth.synth_dit_state_dict(cfg, seed=0)
wav = synth.synth_wave(120000, seed=0)
text = synth.synth_text_ids(1, 220, cfg.text_num_embeds, seed=0)
kw = dict(cfg_strength=2.0, sway_sampling_coef=-1.0, seed=0, use_epss=False)
for nt in [int(a) for a in sys.argv[1:]] or [8, 16, 32, 64]:
    torch.set_num_threads(nt)
    t0 = time.perf_counter()
    O.cfm_sample(sd, cfg, wav, text, 1406, steps=1, **kw)
    t1 = time.perf_counter()
    O.cfm_sample(sd, cfg, wav, text, 1406, steps=2, **kw)
    t2 = time.perf_counter()
    print(f"threads {nt}: steps=1 {t1 - t0:.2f}s  steps=2 {t2 - t1:.2f}s  per-step {(t2 - t1) - (t1 - t0):.2f}s", flush=True)
