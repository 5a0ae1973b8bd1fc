"""The q|k|v epilogue's MX-fp6 score-correction words and the flash kernel that reads them, on the GPU, for every tile family and three
shapes in both half-precision modes (f5hip_bench_qkv check = 2: P words decoded on the host against the generic kernel's hi + lo values,
attention on the MX planes against the split-q,k kernel): python tools/mxqk_check.py  ->  one `status 0 diff 0` line per case
(profiles/r05h_mxqk_check_scale_overlap.log: before the register fix of DESIGN.md section 4.6; r05j_mxqk_check.log: after)."""
import ctypes as C, sys, os
sys.path.insert(0, os.getcwd())
import torch
import f5_tts_amd
from f5_tts_amd import binding, config
from f5_tts_amd.engine import F5HipEngine
eng = F5HipEngine(config.PRESETS["F5TTS_v1_Base"] if "F5TTS_v1_Base" in config.PRESETS else config.DIT_TINY, None, device=0)
for prec, variants in (("fp16m", (50, 57, 59, 61, 80, -1)), ("fp16x3", (50, 57, -1))):
    for v in variants:
        for seqs, nseq in ((2, 150), (1, 1406), (8, 1024)):
            ms, diff = C.c_double(), C.c_int64()
            st = eng.bench_lib.f5hip_bench_qkv(eng._ctx, binding.PRECISIONS[prec], v, seqs, nseq, 1024, 1, 2, C.byref(ms), C.byref(diff))
            print(prec, "variant", v, seqs, nseq, "status", st, "diff", diff.value, flush=True)
