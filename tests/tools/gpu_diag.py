"""GPU diagnostic sweep (prints errors stage by stage; used during bring-up through gpurun)."""
import os
import sys
import time
import traceback

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import f5_tts_amd  # noqa: E402,F401
from f5_tts_amd import config, synth  # noqa: E402
from f5_tts_amd.engine import F5HipCFM, F5HipEngine, F5HipVocos  # noqa: E402
from oracle import f5_oracle as O  # noqa: E402


def err(a, b):
    a, b = a.detach().cpu().float(), b.detach().cpu().float()
    return (a - b).abs().max().item(), b.abs().max().item()


def section(name, fn):
    print(f"=== {name}", flush=True)
    try:
        fn()
    except Exception:
        traceback.print_exc()
    sys.stdout.flush()


def main():
    print(torch.__version__, torch.cuda.get_device_name(0))
    cfg, vcfg = config.DIT_TINY, config.VOCOS_TINY
    sd, vsd = synth.synth_dit_state_dict(cfg, seed=1), synth.synth_vocos_state_dict(vcfg, seed=1)
    eng = F5HipEngine(cfg, vcfg, device=0)
    eng.load_state_dict({**sd, **vsd})
    wav = synth.synth_wave(256 * 60, seed=3)
    text = synth.synth_text_ids(1, 40, cfg.text_num_embeds, seed=2)
    kw = dict(steps=16, cfg_strength=2.0, sway_sampling_coef=-1.0, seed=7)

    def t_mel():
        m = eng.mel(wav.cuda(), frame_major=False)
        print("mel", err(m, O.vocos_mel(wav)))

    def t_sample(prec):
        def f():
            model = F5HipCFM(eng, precision=prec)
            out, traj = model.sample(wav.cuda(), text, 200, **kw)
            ref, rtraj, aux = O.cfm_sample(sd, cfg, wav, text, 200, return_steps=True, **kw)
            n = 200
            tc = eng.debug_tensor(0, (1, n, cfg.text_dim))
            tu = eng.debug_tensor(1, (1, n, cfg.text_dim))
            print(prec, "text_cond", err(tc, aux["text_cond"]), "text_uncond", err(tu, aux["text_uncond"]))
            for s in (1, 2, 8, 16):
                print(prec, f"traj[{s}]", err(traj[s], rtraj[s]))
            print(prec, "out", err(out, ref))
        return f

    def t_vocos():
        mel = O.vocos_mel(synth.synth_wave(256 * 80, seed=5))
        w = F5HipVocos(eng).decode(mel.cuda())
        print("vocos", err(w, O.vocos_decode(vsd, mel, vcfg.num_layers)))

    def t_ragged():
        model = F5HipCFM(eng, precision="fp32")
        wav2 = synth.synth_wave(256 * 60, seed=3, batch=2)
        text2 = synth.synth_text_ids(2, 40, cfg.text_num_embeds, seed=2)
        text2[1, 30:] = -1
        dur, lens = torch.tensor([200, 170]), torch.tensor([61, 50])
        k2 = dict(kw, steps=8)
        out, traj = model.sample(wav2.cuda(), text2, dur, lens=lens, **k2)
        ref, rtraj = O.cfm_sample(sd, cfg, wav2, text2, dur, lens=lens, **k2)
        print("ragged out", err(out, ref), "traj1", err(traj[1], rtraj[1]))

    section("mel", t_mel)
    section("sample fp32", t_sample("fp32"))
    section("vocos", t_vocos)
    section("ragged", t_ragged)
    section("sample fp16x3", t_sample("fp16x3"))
    section("sample fp16", t_sample("fp16"))

    # v0 config (rope head 0 only, no text mask padding)
    cfg0 = config.DIT_TINY_V0
    sd0 = synth.synth_dit_state_dict(cfg0, seed=2)
    eng0 = F5HipEngine(cfg0, None, device=0)
    eng0.load_state_dict(sd0)

    def t_v0():
        out, traj = F5HipCFM(eng0).sample(wav.cuda(), text, 200, **kw)
        ref, rtraj = O.cfm_sample(sd0, cfg0, wav, text, 200, **kw)
        print("v0 out", err(out, ref))

    section("v0", t_v0)

    # full size, 2 steps, timing
    def t_full():
        fcfg = config.F5TTS_V1_BASE
        fsd = synth.synth_dit_state_dict(fcfg, seed=0)
        fe = F5HipEngine(fcfg, config.VOCOS_MEL_24K, device=0)
        fe.load_state_dict({**fsd, **synth.synth_vocos_state_dict(config.VOCOS_MEL_24K, seed=0)})
        w = synth.synth_wave(120000, seed=0)
        tx = synth.synth_text_ids(1, 220, fcfg.text_num_embeds, seed=0)
        k = dict(steps=2, cfg_strength=2.0, sway_sampling_coef=-1.0, seed=0, use_epss=False)
        t0 = time.time()
        ref, rtraj = O.cfm_sample(fsd, fcfg, w, tx, 1406, **k)
        print("oracle 2 steps full size: %.1fs" % (time.time() - t0))
        for prec in ("fp32", "fp16x3", "fp16"):
            m = F5HipCFM(fe, precision=prec)
            out, traj = m.sample(w.cuda(), tx, 1406, **k)
            torch.cuda.synchronize()
            t0 = time.time()
            out, traj = m.sample(w.cuda(), tx, 1406, **k)
            torch.cuda.synchronize()
            print(prec, "full out", err(out, ref), "traj1", err(traj[1], rtraj[1]), "time %.3fs" % (time.time() - t0))

    section("full", t_full)


if __name__ == "__main__":
    main()
