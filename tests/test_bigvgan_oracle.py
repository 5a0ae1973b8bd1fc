"""CPU tests of the BigVGAN row (BASELINE.json configs[4] vocoder; reference call sites utils_infer.py:130-144,512-513).

The generator's source is absent from /root/reference (empty, un-vendored submodule), so parity is UNPINNED: these tests check
  (1) the oracle restatement (oracle/bigvgan_oracle.py) for internal consistency against independent torch formulations
      (nn.ConvTranspose1d / weight_norm modules, scipy's Kaiser window), its documented shapes and edge cases;
  (2) the channels-last / tap-gathered-GEMM / closed-form-resampler formulation the HIP kernels implement (tests/bigvgan_model.py)
      against the oracle;
  (3) the host-only C++ of the library (filter, tap ranges, GEMM weight matrices: csrc/bigvgan_host.h, compiled with g++) against (2);
  (4) the host mirror's checkpoint handling (weight-norm folding, config.json mapping, state-dict contract).
The GPU parity tests proper are in tests/test_gpu_bigvgan.py."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import f5_tts_amd  # noqa: E402,F401
import bigvgan_model as M  # noqa: E402
from f5_tts_amd import config, synth  # noqa: E402
from f5_tts_amd import bigvgan as host  # noqa: E402
from oracle import bigvgan_oracle as BO  # noqa: E402

CFGS = {"tiny": config.BIGVGAN_TINY, "tiny2": config.BIGVGAN_TINY2}


# ---- (1) the oracle against independent formulations -------------------------------------------------------------------------------
def test_filter_matches_scipy_kaiser_and_is_a_unit_gain_lowpass():
    from scipy.signal.windows import kaiser

    f = BO.aa_filter()
    assert f.shape == (12,) and abs(float(f.sum()) - 1.0) < 1e-6 and torch.allclose(f, f.flip(0), atol=1e-7)
    half_width, half = 0.3, 6
    A = 2.285 * (half - 1) * np.pi * 4 * half_width + 7.95
    beta = 0.1102 * (A - 8.7) if A > 50 else (0.5842 * (A - 21) ** 0.4 + 0.07886 * (A - 21) if A >= 21 else 0.0)
    t = np.arange(-half, half) + 0.5
    g = 0.5 * kaiser(12, beta, sym=True) * np.sinc(0.5 * t)
    assert np.allclose(f.numpy(), g / g.sum(), atol=1e-6)
    # DC passes, Nyquist of the upsampled rate is rejected
    assert abs(float((f * torch.tensor([(-1.0) ** i for i in range(12)])).sum())) < 1e-3


def test_activation1d_keeps_length_and_is_identity_like_for_a_dc_signal_without_snake():
    f = BO.aa_filter()
    x = torch.full((1, 3, 50), 0.7)
    assert torch.allclose(BO.downsample1d(BO.upsample1d(x, f), f), x, atol=1e-5)  # replicate padding: DC in = DC out
    for L in (1, 2, 7, 64):
        y = BO.downsample1d(BO.upsample1d(torch.randn(2, 3, L), f), f)
        assert y.shape == (2, 3, L)


@pytest.mark.parametrize("name", list(CFGS))
def test_oracle_equals_a_module_built_from_torch_layers(name):
    """The functional oracle against the same network assembled from nn.Conv1d / nn.ConvTranspose1d modules with weight_norm applied and
    removed (torch's own remove_weight_norm) — checks padding / stride / layout conventions and fold_weight_norm in one go."""
    import torch.nn as nn
    from torch.nn.utils import remove_weight_norm, weight_norm

    cfg = CFGS[name]
    raw = synth.synth_bigvgan_state_dict(cfg, seed=3, raw_weight_norm=True)
    sd = BO.fold_weight_norm(raw)
    nk, c0 = len(cfg.resblock_kernel_sizes), cfg.upsample_initial_channel

    def load(mod, key):
        mod = weight_norm(mod)
        mod.weight_g.data.copy_(raw[key + ".weight_g"])
        mod.weight_v.data.copy_(raw[key + ".weight_v"])
        if mod.bias is not None:
            mod.bias.data.copy_(raw[key + ".bias"])
        remove_weight_norm(mod)
        assert torch.allclose(mod.weight, sd[key + ".weight"], atol=1e-6), key
        return mod

    f = BO.aa_filter()
    mel = torch.randn(2, cfg.num_mels, 11)
    with torch.no_grad():
        x = load(nn.Conv1d(cfg.num_mels, c0, 7, padding=3), "conv_pre")(mel)
        for i, (u, k) in enumerate(zip(cfg.upsample_rates, cfg.upsample_kernel_sizes)):
            ch = c0 // 2 ** (i + 1)
            x = load(nn.ConvTranspose1d(c0 // 2 ** i, ch, k, u, padding=(k - u) // 2), f"ups.{i}.0")(x)
            xs = 0
            for j in range(nk):
                p, kk = f"resblocks.{i * nk + j}", cfg.resblock_kernel_sizes[j]
                r = x
                for m, d in enumerate(cfg.resblock_dilation_sizes[j]):
                    if cfg.resblock == "1":
                        t = load(nn.Conv1d(ch, ch, kk, dilation=d, padding=(kk * d - d) // 2), f"{p}.convs1.{m}")(BO.activation1d(sd, f"{p}.activations.{2 * m}.", r, cfg, f))
                        r = load(nn.Conv1d(ch, ch, kk, padding=(kk - 1) // 2), f"{p}.convs2.{m}")(BO.activation1d(sd, f"{p}.activations.{2 * m + 1}.", t, cfg, f)) + r
                    else:
                        r = load(nn.Conv1d(ch, ch, kk, dilation=d, padding=(kk * d - d) // 2), f"{p}.convs.{m}")(BO.activation1d(sd, f"{p}.activations.{m}.", r, cfg, f)) + r
                xs = xs + r
            x = xs / nk
        x = BO.activation1d(sd, "activation_post.", x, cfg, f)
        x = load(nn.Conv1d(x.shape[1], 1, 7, padding=3, bias=cfg.use_bias_at_final), "conv_post")(x)
        want = torch.tanh(x) if cfg.use_tanh_at_final else x.clamp(-1, 1)
    got = BO.bigvgan_forward(sd, cfg, mel)
    assert got.shape == (2, 1, 11 * cfg.hop)
    assert torch.allclose(got, want, atol=2e-6)


def test_published_checkpoint_shape_contract():
    """Sizes of nvidia/bigvgan_v2_24khz_100band_256x as the config describes them: hop 256, 112 M parameters, channels 1536 -> 24."""
    cfg = config.BIGVGAN_V2_24K_100B_256X
    assert cfg.hop == 256 == config.HOP_LENGTH and cfg.num_mels == config.N_MEL_CHANNELS
    assert [cfg.channels(i) for i in range(6)] == [768, 384, 192, 96, 48, 24]
    n = 100 * 1536 * 7 + 1536
    for i, (u, k) in enumerate(zip(cfg.upsample_rates, cfg.upsample_kernel_sizes)):
        ch = cfg.channels(i)
        n += 2 * ch * ch * k + ch
        for kk, dil in zip(cfg.resblock_kernel_sizes, cfg.resblock_dilation_sizes):
            n += len(dil) * 2 * (ch * ch * kk + ch) + 2 * len(dil) * 2 * ch
    n += 2 * 24 + 24 * 7
    assert 111e6 < n < 113e6  # the model card's "112M"


# ---- (2) the kernels' formulation against the oracle ---------------------------------------------------------------------------------
@pytest.mark.parametrize("L", [1, 2, 3, 5, 6, 11, 12, 13, 33, 64, 97])
@pytest.mark.parametrize("logscale", [True, False])
def test_closed_form_activation_equals_the_resampler_chain(L, logscale):
    g = torch.Generator().manual_seed(L)
    f = BO.aa_filter()
    x = torch.randn(2, 5, L, generator=g) * 2
    al = 0.4 * torch.randn(5, generator=g) if logscale else 0.5 + torch.rand(5, generator=g)
    be = 0.4 * torch.randn(5, generator=g) if logscale else 0.5 + torch.rand(5, generator=g)
    want = BO.downsample1d(BO.snake(BO.upsample1d(x, f), al, be, logscale), f)
    got = M.aa_snake(x.transpose(1, 2), al, be, f, logscale).transpose(1, 2)
    assert torch.allclose(got, want, atol=3e-6)


@pytest.mark.parametrize("cin,cout,k,d,L", [(20, 64, 7, 1, 9), (32, 32, 3, 5, 40), (24, 24, 11, 3, 17), (12, 12, 11, 5, 4), (100, 48, 7, 1, 1)])
def test_tap_gathered_gemm_equals_conv1d(cin, cout, k, d, L):
    g = torch.Generator().manual_seed(k * 100 + d)
    x, w, b = torch.randn(2, cin, L, generator=g), torch.randn(cout, cin, k, generator=g), torch.randn(cout, generator=g)
    want = torch.nn.functional.conv1d(x, w, b, dilation=d, padding=(k * d - d) // 2)
    got = M.conv_cl(x.transpose(1, 2), w, b, d).transpose(1, 2)
    assert torch.allclose(got, want, atol=1e-4)


@pytest.mark.parametrize("cin,cout,k,u,L", [(64, 32, 8, 4, 7), (32, 16, 4, 2, 10), (48, 24, 4, 2, 1), (16, 8, 16, 8, 5), (8, 8, 7, 3, 6), (8, 4, 2, 2, 3)])
def test_phase_stacked_gemm_equals_conv_transpose1d(cin, cout, k, u, L):
    g = torch.Generator().manual_seed(k * 10 + u)
    x, w, b = torch.randn(2, cin, L, generator=g), torch.randn(cin, cout, k, generator=g), torch.randn(cout, generator=g)
    want = torch.nn.functional.conv_transpose1d(x, w, b, stride=u, padding=(k - u) // 2)
    assert want.shape[-1] == L * u
    got = M.convt_cl(x.transpose(1, 2), w, b, u).transpose(1, 2)
    assert torch.allclose(got, want, atol=1e-4)


def test_implicit_gemm_form_equals_the_tap_gathered_form():
    g = torch.Generator().manual_seed(0)
    y = torch.randn(2, 19, 24, generator=g)
    w = torch.randn(40, 24, 11, generator=g)
    wm = M.conv_weight_matrix(w, 32)
    a = M.im2col(y, 11, -5 * 3, 3, 32) @ wm.t()
    assert torch.allclose(M.conv_implicit_cl(y, wm, 11, -15, 3, 32), a, atol=1e-5)
    wt = torch.randn(24, 12, 8, generator=g)
    s0, nt = M.convt_taps(8, 4)
    wmt = M.convt_weight_matrix(wt, 4, 32)
    assert torch.allclose(M.conv_implicit_cl(y, wmt, nt, s0, 1, 32), M.im2col(y, nt, s0, 1, 32) @ wmt.t(), atol=1e-5)


@pytest.mark.parametrize("name", list(CFGS))
@pytest.mark.parametrize("T", [1, 13])
def test_whole_generator_in_the_kernels_formulation(name, T):
    cfg = CFGS[name]
    sd = synth.synth_bigvgan_state_dict(cfg, seed=1)
    mel = torch.randn(2, cfg.num_mels, T, generator=torch.Generator().manual_seed(T))
    want = BO.bigvgan_forward(sd, cfg, mel)[:, 0]
    got = M.bigvgan_forward_cl(sd, cfg, mel, BO.aa_filter())
    assert got.shape == (2, T * cfg.hop) and torch.allclose(got, want, atol=2e-5)
    assert float(want.abs().max()) > 0.2 and float((want.abs() >= 1.0).float().mean()) < 0.2  # neither silent nor clamped flat


# ---- (3) the library's host-only C++ -------------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def host_exe():
    out_dir = os.path.join(ROOT, "tests", "c_abi", "_build")
    os.makedirs(out_dir, exist_ok=True)
    exe = os.path.join(out_dir, "bigvgan_host_test")
    r = subprocess.run(["g++", "-std=c++17", "-O1", "-Wall", "-I", os.path.join(ROOT, "f5-tts_amd", "csrc"),
                        os.path.join(ROOT, "tests", "c_abi", "bigvgan_host_test.cpp"), "-o", exe], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    return exe


def run_host(exe, args, data=None):
    r = subprocess.run([exe] + [str(a) for a in args], input=data.numpy().astype("<f4").tobytes() if data is not None else None, capture_output=True)
    assert r.returncode == 0, r.stderr
    return r.stdout


def test_library_filter_equals_the_oracle_filter(host_exe):
    f = np.frombuffer(run_host(host_exe, ["filter"]), dtype="<f4")
    assert np.allclose(f, BO.aa_filter().numpy(), atol=2e-7)


@pytest.mark.parametrize("k,u", [(8, 4), (4, 2), (16, 8), (7, 3), (2, 2), (6, 2), (12, 4)])
def test_library_convt_taps(host_exe, k, u):
    s0, nt = map(int, run_host(host_exe, ["taps", k, u]).split())
    assert (s0, nt) == M.convt_taps(k, u)


@pytest.mark.parametrize("cout,cin,k,cpad", [(64, 20, 7, 32), (24, 24, 11, 32), (16, 100, 7, 128), (5, 33, 3, 64)])
def test_library_conv_matrix(host_exe, cout, cin, k, cpad):
    w = torch.randn(cout, cin, k, generator=torch.Generator().manual_seed(cin))
    m = np.frombuffer(run_host(host_exe, ["conv", cout, cin, k, cpad], w), dtype="<f4").reshape(cout, k * cpad)
    assert np.array_equal(m, M.conv_weight_matrix(w, cpad).numpy())


@pytest.mark.parametrize("cin,cout,k,u,cpad", [(64, 32, 8, 4, 64), (48, 24, 4, 2, 64), (8, 8, 7, 3, 32), (16, 8, 16, 8, 32)])
def test_library_convt_matrix(host_exe, cin, cout, k, u, cpad):
    w = torch.randn(cin, cout, k, generator=torch.Generator().manual_seed(cout))
    s0, nt = M.convt_taps(k, u)
    m = np.frombuffer(run_host(host_exe, ["convt", cin, cout, k, u, cpad], w), dtype="<f4").reshape(u * cout, nt * cpad)
    assert np.array_equal(m, M.convt_weight_matrix(w, u, cpad).numpy())


# ---- (4) host mirror: checkpoint handling --------------------------------------------------------------------------------------------
def test_fold_weight_norm_both_spellings_and_filter_buffers():
    cfg = config.BIGVGAN_TINY
    raw = synth.synth_bigvgan_state_dict(cfg, seed=5, raw_weight_norm=True)
    raw["resblocks.0.activations.0.upsample.filter"] = torch.zeros(1, 1, 12)
    raw["resblocks.0.activations.0.downsample.lowpass.filter"] = torch.zeros(1, 1, 12)
    a = host.fold_weight_norm(raw)
    b = BO.fold_weight_norm(raw)
    assert sorted(a) == sorted(b) and all(torch.equal(a[k], b[k]) for k in a)
    assert not any("filter" in k or k.endswith("_g") or k.endswith("_v") for k in a)
    par = {}
    for k, v in raw.items():
        par[k.replace(".weight_g", ".parametrizations.weight.original0").replace(".weight_v", ".parametrizations.weight.original1")] = v
    c = host.fold_weight_norm(par)
    assert sorted(c) == sorted(a) and all(torch.equal(a[k], c[k]) for k in a)
    # ConvTranspose1d: the norm runs over (out, k) for each INPUT channel (weight_norm dim 0)
    v, g = raw["ups.0.0.weight_v"], raw["ups.0.0.weight_g"]
    assert torch.allclose(a["ups.0.0.weight"], g * v / v.reshape(v.shape[0], -1).norm(dim=1).view(-1, 1, 1))
    # already-folded dicts pass through
    d = host.fold_weight_norm(a)
    assert sorted(d) == sorted(a) and all(torch.equal(a[k], d[k]) for k in a)


def test_config_json_mapping_and_struct():
    h = dict(num_mels=100, upsample_rates=[4, 4, 2, 2, 2, 2], upsample_kernel_sizes=[8, 8, 4, 4, 4, 4], upsample_initial_channel=1536,
             resblock="1", resblock_kernel_sizes=[3, 7, 11], resblock_dilation_sizes=[[1, 3, 5]] * 3, activation="snakebeta", snake_logscale=True,
             use_tanh_at_final=False, use_bias_at_final=False, sampling_rate=24000, hop_size=256)
    assert host.config_from_json(h) == config.BIGVGAN_V2_24K_100B_256X
    c = host._config_c(config.BIGVGAN_TINY2)
    assert (c.num_upsamples, c.num_kernels, c.resblock, c.activation, c.use_tanh_at_final, c.use_bias_at_final) == (3, 2, 2, 0, 1, 1)
    assert list(c.upsample_rates)[:3] == [2, 2, 2] and list(c.resblock_num_dilations)[:2] == [2, 3] and list(c.resblock_dilation_sizes[1])[:3] == [1, 3, 5]
    with pytest.raises(NotImplementedError):
        host._config_c(config.BigVGANConfig(activation="relu"))


@pytest.mark.skipif(torch.cuda.is_available(), reason="only meaningful without a GPU")
def test_no_cpu_path():
    from f5_tts_amd import binding

    with pytest.raises((binding.F5HipError, RuntimeError, ValueError)):
        host.F5HipBigVGAN(config.BIGVGAN_TINY, device="cpu")
    with pytest.raises((binding.F5HipError, RuntimeError, ValueError)):
        host.F5HipBigVGAN(config.BIGVGAN_TINY, device=0)  # f5hip_bigvgan_create: no HIP device here
    lib = binding.load_library()
    import ctypes as C

    bad = host._config_c(config.BIGVGAN_TINY)
    bad.num_kernels = 9
    ctx = C.c_void_p()
    assert lib.f5hip_bigvgan_create(C.byref(bad), 0, C.byref(ctx)) == 1 and not ctx.value  # rejected before any device is touched
    assert b"num_kernels" in lib.f5hip_bigvgan_last_error(None)
