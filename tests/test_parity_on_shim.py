"""The GPU parity tests' own assertions (tests/test_gpu_parity.py: reference-minted goldens through the product's host classes and the
C ABI), executed on the CPU: libf5hip's engine — api.cpp and every kernel translation unit — is built for the host through
tests/hipemu/hipemu.h and driven by the unmodified F5HipEngine / F5HipCFM.  A cross-section of the goldens that finishes in a minute:
every backbone (DiT, UNetT, MMDiT), ragged batches with row / key masks, qk RMSNorm, long skip, average upsampling, the midpoint
solver — in the exact-fp32 and the fp16x3 operand modes.  The whole matrix (all goldens, all modes, full size) is the `-m gpu` suite."""
import contextlib
import os
import sys
import types

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_parity as G  # noqa: E402
from test_hipemu import CLANG, engine_emu_lib, host_alias  # noqa: E402,F401  (the fixture that builds the emulated library)

pytestmark = pytest.mark.skipif(not os.path.exists(CLANG), reason="needs the ROCm host clang++")

ALL_CASES = os.environ.get("F5HIP_SHIM_ALL_CASES") == "1"  # every tiny golden in both modes (~10 min): run once after touching a kernel
CASES = sorted(G.MG.CASES) if ALL_CASES else ["tiny_qknorm", "tiny_longskip", "tiny_avgup", "tiny_v1_midpoint", "tiny_unett_add_ragged_b2", "tiny_mmdit_mask_ragged_b2", "tiny_v1_nocfg_b2"]


@pytest.fixture(scope="module")
def shim_engines(engine_emu_lib):  # noqa: F811
    """What test_gpu_parity's `engines` fixture returns, over the emulated library ("device" tensors are CPU tensors)."""
    from f5_tts_amd import config, synth
    from f5_tts_amd import engine as E

    mp = pytest.MonkeyPatch()
    mp.setattr(E, "load_library", lambda *a, **k: engine_emu_lib)
    mp.setattr(E, "_as_tensor", host_alias)
    mp.setattr(torch.cuda, "device", lambda *_a, **_k: contextlib.nullcontext())
    mp.setattr(torch.cuda, "current_stream", lambda *_a, **_k: types.SimpleNamespace(cuda_stream=0))
    mp.setattr(torch.Tensor, "cuda", lambda self, *a, **k: self)
    orig_init = E.F5HipEngine.__init__

    def init_on_cpu(self, dit_cfg, vocos_cfg=None, device=0):  # tests that build their own engine: a cuda descriptor in, CPU tensors after
        orig_init(self, dit_cfg, vocos_cfg, device="cuda:0" if not isinstance(device, int) else device)
        self.device = torch.device("cpu")

    mp.setattr(E.F5HipEngine, "__init__", init_on_cpu)
    cache = {}

    def get(preset, wseed, vocos=False, stress=False, trained=False, sharp=None):
        key = (preset, wseed, vocos, stress, trained, sharp)
        if key not in cache:
            cfg = config.PRESETS[preset]
            sd = synth.synth_dit_state_dict(cfg, seed=wseed)
            if stress:
                sd = synth.stress_dit_state_dict(sd, cfg, seed=wseed)
            if trained:
                sd = synth.trained_like_dit_state_dict(sd, cfg, seed=wseed)
            if sharp:
                sd = synth.sharpen_attention_state_dict(sd, sharp)
            vcfg = config.VOCOS_TINY if vocos else None
            eng = E.F5HipEngine(cfg, vcfg, device=0)  # a descriptor only (init_on_cpu)
            if vocos:
                sd = {**sd, **synth.synth_vocos_state_dict(vcfg, seed=1)}
            eng.load_state_dict(sd)
            cache[key] = eng
        return cache[key]

    yield get
    for e in cache.values():
        e.close()
    mp.undo()


# 80-100 s each on the shim (dim 48 with ragged rows through the generic tiles; dim 1024 / 16 heads through the tiles of the real model): part of
# the GPU suite, on the shim only with F5HIP_SHIM_FULL=1 — the CPU suite keeps to a few minutes
HEAVY_MX_CASES = ("tiny48_ragged_b2", "tiny1024_ragged_b2") if os.environ.get("F5HIP_SHIM_FULL") == "1" else ()


@pytest.mark.parametrize("name,prec,tol", [(n, "fp32", G.TIGHT) for n in CASES] +
                         [(n, "fp16x3", G.X3TOL) for n in (CASES if ALL_CASES else ("tiny_qknorm", "tiny_unett_add_ragged_b2", "tiny_mmdit_mask_ragged_b2"))] +
                         # fp16m: MX lines in the DiT block GEMMs (LayerNorm / flash / GELU producers, the MX k-loop); the other backbones run it as fp16x3
                         [(n, "fp16m", G.MXTOL) for n in (CASES if ALL_CASES else ("tiny_v1_ragged_b2", "tiny_mask_ragged_b3", "tiny_inner512", "tiny_unett_noskip", "tiny_v1_trained_like", "tiny_unett_trained_like",
                                                                                "tiny_qknorm_trained_like", "tiny_mmdit_trained_like") + HEAVY_MX_CASES)])
def test_reference_golden_on_the_shim(shim_engines, name, prec, tol):
    G.test_sample_matches_reference_golden(shim_engines, name, prec, tol)


def test_mel_front_ends_on_the_shim(shim_engines):
    G.test_mel_matches_reference_golden(shim_engines)


# the rest of the GPU suite's tiny-model tests, verbatim (each is one call into tests/test_gpu_parity.py)
ENGINE_TESTS = ["test_bigvgan_mel_matches_reference_golden", "test_mel_too_short_raises", "test_text_longer_than_frames_and_unknown_ids",
                "test_edit_mask_and_no_ref_audio", "test_vocos_decode_matches_oracle_golden", "test_vocos_batch_and_min_frames",
                "test_flash_attention_equals_materialised_attention", "test_invalid_arguments_raise", "test_speech_edit_matches_oracle",
                "test_all_padding_text_and_single_frame_prompt", "test_weight_blob_receiver_equals_the_rank_that_loaded",
                "test_fp16m_runs_as_fp16x3_where_the_mx_tiles_do_not_apply", "test_hip_istft_against_the_reference_conv_istft_fixture",
                "test_attention_stats_follow_the_sharpness"]
if os.environ.get("F5HIP_SHIM_FULL") == "1":  # 15-80 s each on the shim; pass as well (the CPU suite keeps to a few minutes without them)
    # test_graph_replay_equals_eager: stream capture is emulated by recording closures (tests/hipemu/hipemu.h GraphRec); the default suite
    # covers the captured path through tests/test_bench_on_shim.py
    ENGINE_TESTS += ["test_graph_replay_equals_eager", "test_bigvgan_type_sampler_and_glue", "test_text_embedding_and_velocity_taps", "test_determinism_and_batch_consistency"]


@pytest.mark.parametrize("fn", ENGINE_TESTS)
def test_gpu_suite_function_on_the_shim(shim_engines, fn):
    getattr(G, fn)(shim_engines)


def test_packed_rows_on_the_shim(shim_engines):
    # two CFG chains (the row tables are shared by both); one chain and the fp16 mode as well on the GPU
    G.test_packed_rows_equal_the_padded_layout_on_the_valid_rows(shim_engines, 1, precisions=("fp16x3", "fp16m") if os.environ.get("F5HIP_SHIM_FULL") == "1" else ("fp16m",))


@pytest.mark.parametrize("nw", [513, 256 * 20 + 255])
def test_mel_edge_lengths_on_the_shim(shim_engines, nw):
    G.test_mel_edge_lengths(shim_engines, nw)


@pytest.mark.parametrize("knob", ["F5HIP_PP_VARIANT=0", "F5HIP_PP_VARIANT=57", "F5HIP_PP_VARIANT=80"])
def test_fp16m_under_the_tuning_knobs(knob):
    """The knobs INTEGRATION.md lists as 'no change of the arithmetic contract' are read once per process, so each runs in a child: a forced
    tile that has no MX instantiation (57) and 'never the pipelined kernel' (0) used to fail EVERY fp16m call
    (ADVICE r04); now the call runs in fp16x3.  A forced tile that IS instantiated for MX lines (80, the ping-pong kernel) stays in fp16m."""
    import subprocess

    k, v = knob.split("=")
    r = subprocess.run([sys.executable, "-m", "pytest", "-q", "-x", os.path.join(ROOT, "tests", "test_parity_on_shim.py"), "-k",
                        "test_reference_golden_on_the_shim and tiny_v1_ragged_b2 and fp16m"], capture_output=True, text=True,
                       env={**os.environ, k: v}, cwd=ROOT, timeout=1800)
    assert r.returncode == 0 and " passed" in r.stdout and "no tests ran" not in r.stdout, (knob, r.stdout[-1500:], r.stderr[-500:])
