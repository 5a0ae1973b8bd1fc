"""The HIP kernels of the BigVGAN path executed ON THE CPU, thread for thread, through tests/hipemu/hipemu.h (host clang++, one
std::thread per HIP thread, the amdgcn builtins emulated with their documented semantics) and checked against tests/bigvgan_model.py.

Why: those kernels were written in a session without GPU minutes.  This runs the very source hipcc compiles (csrc/bigvgan_kernels.h,
csrc/conv_gemm.h) — index arithmetic, LDS image + swizzle, MFMA fragment layout, buffer-descriptor bounds, epilogue — so what is left
for the GPU to reveal is hardware behaviour (timing, caches), not logic.  The shim itself is validated first: the GPU-proven
gemm_kernel (gemm.h, the DiT path's GEMM) must produce a correct product through it in all three operand layouts."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import bigvgan_model as M  # noqa: E402

CLANG = "/opt/rocm/lib/llvm/bin/clang++"
OP_F32, OP_F16, OP_F16X3 = 0, 1, 2
pytestmark = pytest.mark.skipif(not os.path.exists(CLANG), reason="needs the ROCm host clang++")


@pytest.fixture(scope="module")
def exe():
    out_dir = os.path.join(ROOT, "tests", "c_abi", "_build")
    os.makedirs(out_dir, exist_ok=True)
    path = os.path.join(out_dir, "hipemu_run_kernels")
    src = os.path.join(ROOT, "tests", "hipemu", "run_kernels.cpp")
    deps = [src, os.path.join(ROOT, "tests", "hipemu", "hipemu.h")] + [os.path.join(ROOT, "f5-tts_amd", "csrc", h) for h in
                                                                         ("common.h", "gemm.h", "kernels.h", "conv_gemm.h", "bigvgan_kernels.h")]
    if not os.path.exists(path) or any(os.path.getmtime(d) > os.path.getmtime(path) for d in deps):
        r = subprocess.run([CLANG, "-std=c++20", "-O1", "-pthread", "-DF5_HIPEMU", "-I", os.path.join(ROOT, "tests", "hipemu"), "-I",
                            os.path.join(ROOT, "f5-tts_amd", "csrc"), "-Wno-unknown-pragmas", "-Wno-pass-failed", "-Wno-psabi", src, "-o", path],
                           capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-3000:]
    return path


def run(exe, d, mode, *args, **files):
    for name, arr in files.items():
        with open(os.path.join(d, name + ".bin"), "wb") as f:
            f.write(arr if isinstance(arr, bytes) else np.ascontiguousarray(arr).tobytes())
    r = subprocess.run([exe, str(d), mode] + [str(int(a)) for a in args], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr
    return open(os.path.join(d, "out.bin"), "rb").read()


def operand_bytes(x: np.ndarray, op: int) -> bytes:
    """[rows, K] fp32 -> the GEMM operand layout (kernels.h OP_*): fp32 rows, fp16 rows, packed [K/32][32 hi | 32 lo] rows."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    if op == OP_F32:
        return x.tobytes()
    hi = x.astype(np.float16)
    if op == OP_F16:
        return hi.tobytes()
    lo = (x - hi.astype(np.float32)).astype(np.float16)
    rows, K = x.shape
    assert K % 32 == 0
    return np.concatenate((hi.reshape(rows, K // 32, 1, 32), lo.reshape(rows, K // 32, 1, 32)), axis=2).tobytes()


def operand_values(x: np.ndarray, op: int) -> np.ndarray:
    """what the operand layout represents, as float64"""
    x = np.asarray(x, dtype=np.float32)
    if op == OP_F32:
        return x.astype(np.float64)
    hi = x.astype(np.float16)
    if op == OP_F16:
        return hi.astype(np.float64)
    lo = (x - hi.astype(np.float32)).astype(np.float16)
    return hi.astype(np.float64) + lo.astype(np.float64)


def decode_operand(buf: bytes, rows: int, K: int, op: int) -> np.ndarray:
    if op == OP_F32:
        return np.frombuffer(buf, dtype="<f4").reshape(rows, K).astype(np.float64)
    h = np.frombuffer(buf, dtype="<f2")
    if op == OP_F16:
        return h.reshape(rows, K).astype(np.float64)
    h = h.reshape(rows, K // 32, 2, 32).astype(np.float64)
    return (h[:, :, 0] + h[:, :, 1]).reshape(rows, K)


TOL = {OP_F32: 2e-5, OP_F16: 2e-5, OP_F16X3: 2e-5}  # against the product of the values the operands REPRESENT; x3 drops lo*lo (~1e-7)


# ---- the shim against a GPU-proven kernel --------------------------------------------------------------------------------------------
@pytest.mark.parametrize("op", [OP_F32, OP_F16, OP_F16X3])
@pytest.mark.parametrize("tile", [1, 2])
def test_shim_runs_the_gpu_proven_gemm_kernel_correctly(exe, tmp_path, op, tile):
    rng = np.random.default_rng(op * 10 + tile)
    batch, Mr, N, K = 2, 150, 72, 96  # ragged in M and N: partial tiles, rows / channels past the end
    A = rng.standard_normal((batch * Mr, K)).astype(np.float32)
    W = rng.standard_normal((N, K)).astype(np.float32)
    bias = rng.standard_normal(N).astype(np.float32)
    res = rng.standard_normal((batch, Mr, N)).astype(np.float32)
    out = run(exe, tmp_path, "gemm", op, Mr, N, K, batch, tile, 1, A=operand_bytes(A, op), W=operand_bytes(W, op), bias=bias, res=res)
    got = np.frombuffer(out, dtype="<f4").reshape(batch, Mr, N)
    want = (operand_values(A, op) @ operand_values(W, op).T).reshape(batch, Mr, N) + bias + res
    assert np.abs(got - want).max() < TOL[op] * np.abs(want).max()


# ---- the new kernels -----------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("op", [OP_F32, OP_F16, OP_F16X3])
@pytest.mark.parametrize("ntaps,shift0,dstep,cpad,L,N,C", [(7, -9, 3, 32, 70, 24, 24), (3, -1, 1, 64, 150, 96, 48), (11, -25, 5, 64, 33, 72, 40),
                                                          (1, 0, 1, 128, 129, 16, 100)])
def test_implicit_gemm_conv_kernel(exe, tmp_path, op, ntaps, shift0, dstep, cpad, L, N, C):
    if op == OP_F16 and cpad % 64:
        pytest.skip("plain fp16: a tap segment must be a whole number of 128-byte k-tiles (launch_conv_gemm rejects it; bigvgan_api.cpp "
                    "routes such layers through the tap-gathered path)")
    rng = np.random.default_rng(ntaps * 7 + op)
    batch, K = 2, ntaps * cpad
    y = np.zeros((batch, L, cpad), dtype=np.float32)
    y[:, :, :C] = rng.standard_normal((batch, L, C))
    W = rng.standard_normal((N, K)).astype(np.float32)
    bias = rng.standard_normal(N).astype(np.float32)
    res = rng.standard_normal((batch, L, N)).astype(np.float32)
    out = run(exe, tmp_path, "conv", op, L, N, K, batch, 1 if N <= 64 else 2, 1, ntaps, shift0, dstep, cpad,
              A=operand_bytes(y.reshape(batch * L, cpad), op), W=operand_bytes(W, op), bias=bias, res=res)
    got = np.frombuffer(out, dtype="<f4").reshape(batch, L, N)
    yv = torch.from_numpy(operand_values(y.reshape(batch * L, cpad), op).reshape(batch, L, cpad))
    want = M.conv_implicit_cl(yv.float(), torch.from_numpy(operand_values(W, op)).float(), ntaps, shift0, dstep, cpad).numpy() + bias + res
    assert np.abs(got - want).max() < 5e-5 * max(1.0, np.abs(want).max())


@pytest.mark.parametrize("L,C", [(1, 4), (5, 24), (33, 70), (70, 12), (129, 64)])
@pytest.mark.parametrize("logscale", [0, 1])
def test_activation1d_kernel_fp32(exe, tmp_path, L, C, logscale):
    from oracle import bigvgan_oracle as BO

    g = torch.Generator().manual_seed(L * 100 + C)
    B = 2
    x = torch.randn(B, L, C, generator=g) * 1.5
    al = 0.4 * torch.randn(C, generator=g) if logscale else 0.5 + torch.rand(C, generator=g)
    be = 0.4 * torch.randn(C, generator=g) if logscale else 0.5 + torch.rand(C, generator=g)
    f = BO.aa_filter()
    out = run(exe, tmp_path, "aa", B, L, C, logscale, -1, 0, x=x.numpy(), alpha=al.numpy(), beta=be.numpy(), f=f.numpy())
    got = torch.from_numpy(np.frombuffer(out, dtype="<f4").reshape(B, L, C).copy())
    assert torch.allclose(got, M.aa_snake(x, al, be, f, bool(logscale)), atol=3e-6)
    # and against the oracle's resampler chain directly
    want = BO.downsample1d(BO.snake(BO.upsample1d(x.transpose(1, 2), f), al, be, bool(logscale)), f).transpose(1, 2)
    assert torch.allclose(got, want, atol=5e-6)


@pytest.mark.parametrize("op", [OP_F32, OP_F16, OP_F16X3])
def test_activation1d_kernel_fused_operand_emission(exe, tmp_path, op):
    from oracle import bigvgan_oracle as BO

    g = torch.Generator().manual_seed(op)
    B, L, C, cpad = 2, 41, 40, 64
    x, al, be, f = torch.randn(B, L, C, generator=g), 0.3 * torch.randn(C, generator=g), 0.3 * torch.randn(C, generator=g), BO.aa_filter()
    out = run(exe, tmp_path, "aa", B, L, C, 1, op, cpad, x=x.numpy(), alpha=al.numpy(), beta=be.numpy(), f=f.numpy())
    got = decode_operand(out, B * L, cpad, op).reshape(B, L, cpad)
    z = np.zeros((B, L, cpad), dtype=np.float32)
    z[:, :, :C] = M.aa_snake(x, al, be, f, True).numpy()
    assert np.array_equal(got[:, :, C:], np.zeros((B, L, cpad - C)))  # the pad channels are written, and zero
    assert np.abs(got - z).max() < (2e-3 if op == OP_F16 else 3e-6)


@pytest.mark.parametrize("op", [OP_F32, OP_F16, OP_F16X3])
@pytest.mark.parametrize("channel_major", [0, 1])
def test_im2col_kernel(exe, tmp_path, op, channel_major):
    g = torch.Generator().manual_seed(op + 10 * channel_major)
    B, L, C, ntaps, shift0, dstep, cpad = 2, 37, 20, 7, -3, 1, 32
    y = torch.randn(B, L, C, generator=g)
    src = y.transpose(1, 2).contiguous() if channel_major else y
    sb, sl, sc = (C * L, 1, L) if channel_major else (L * C, C, 1)
    out = run(exe, tmp_path, "im2col", B, L, C, ntaps, shift0, dstep, cpad, op, sb, sl, sc, src=src.numpy())
    got = decode_operand(out, B * L, ntaps * cpad, op).reshape(B, L, ntaps * cpad)
    want = operand_values(M.im2col(y, ntaps, shift0, dstep, cpad).numpy().reshape(B * L, -1), op).reshape(B, L, -1)
    assert np.array_equal(got, want)
    # dilated taps
    out = run(exe, tmp_path, "im2col", B, L, C, 11, -25, 5, cpad, op, L * C, C, 1, src=y.numpy())
    got = decode_operand(out, B * L, 11 * cpad, op).reshape(B, L, 11 * cpad)
    assert np.array_equal(got, operand_values(M.im2col(y, 11, -25, 5, cpad).numpy().reshape(B * L, -1), op).reshape(B, L, -1))


@pytest.mark.parametrize("nk", [1, 2, 3, 4])
def test_mean_kernel(exe, tmp_path, nk):
    g = torch.Generator().manual_seed(nk)
    rs = [torch.randn(1000, generator=g) for _ in range(nk)]
    out = run(exe, tmp_path, "mean", nk, 1000, **{f"r{j}": r.numpy() for j, r in enumerate(rs)})
    acc = rs[0].clone()
    for r in rs[1:]:
        acc = acc + r
    assert torch.equal(torch.from_numpy(np.frombuffer(out, dtype="<f4").copy()), acc / nk)


@pytest.mark.parametrize("use_tanh,has_bias,L,C", [(0, 0, 300, 24), (1, 1, 5, 12), (0, 1, 257, 16)])
def test_conv_post_kernel(exe, tmp_path, use_tanh, has_bias, L, C):
    g = torch.Generator().manual_seed(L)
    B = 2
    y, w, b = torch.randn(B, L, C, generator=g), 0.3 * torch.randn(1, C, 7, generator=g), torch.randn(1, generator=g)
    out = run(exe, tmp_path, "post", B, L, C, use_tanh, has_bias, y=y.numpy(), w7=w[0].t().contiguous().numpy(), bias=b.numpy())
    got = torch.from_numpy(np.frombuffer(out, dtype="<f4").reshape(B, L).copy())
    want = torch.nn.functional.conv1d(y.transpose(1, 2), w, b if has_bias else None, padding=3)[:, 0]
    want = torch.tanh(want) if use_tanh else want.clamp(-1, 1)
    assert torch.allclose(got, want, atol=2e-5)


# ---- the whole BigVGAN path (context, weight layouts, enqueue logic + every kernel) on the CPU ------------------------------------------
@pytest.fixture(scope="module")
def emu_lib():
    import ctypes as C

    import f5_tts_amd  # noqa: F401
    from f5_tts_amd import binding

    out_dir = os.path.join(ROOT, "tests", "c_abi", "_build")
    os.makedirs(out_dir, exist_ok=True)
    path = os.path.join(out_dir, "libf5hip_bigvgan_emu.so")
    src = os.path.join(ROOT, "tests", "hipemu", "emu_bigvgan_lib.cpp")
    csrc = os.path.join(ROOT, "f5-tts_amd", "csrc")
    deps = [src, os.path.join(ROOT, "tests", "hipemu", "hipemu.h")] + [os.path.join(csrc, h) for h in os.listdir(csrc) if h.endswith((".h", ".hip", ".cpp"))]
    if not os.path.exists(path) or any(os.path.getmtime(d) > os.path.getmtime(path) for d in deps):
        r = subprocess.run([CLANG, "-std=c++20", "-O1", "-pthread", "-shared", "-fPIC", "-Wl,-Bsymbolic", "-DF5_HIPEMU", "-I", os.path.join(ROOT, "tests", "hipemu"),
                            "-I", csrc, "-Wno-unknown-pragmas", "-Wno-pass-failed", "-Wno-psabi", src, "-o", path], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-3000:]
    # -Bsymbolic + RTLD_LOCAL: other tests load the real libf5hip.so with RTLD_GLOBAL; the emulated library must keep calling its own
    # launch_* functions, not the GPU ones of the same name
    lib = C.CDLL(path, mode=C.RTLD_LOCAL)
    for name, (res, args) in binding.SYMBOLS.items():
        if name.startswith("f5hip_bigvgan_"):
            fn = getattr(lib, name)
            fn.restype, fn.argtypes = res, args
    return lib


class EmuBigVGAN:
    """The C ABI driven directly with host pointers (the emulated library's "device" memory is host memory)."""

    def __init__(self, lib, cfg, sd):
        import ctypes as C

        from f5_tts_amd.bigvgan import _config_c

        self.lib, self.cfg, self.C = lib, cfg, C
        self.ctx = C.c_void_p()
        c = _config_c(cfg)
        assert lib.f5hip_bigvgan_create(C.byref(c), 0, C.byref(self.ctx)) == 0, lib.f5hip_bigvgan_last_error(None)
        name, numel = C.c_char_p(), C.c_int64()
        for i in range(lib.f5hip_bigvgan_num_tensors(self.ctx)):
            assert lib.f5hip_bigvgan_tensor_info(self.ctx, i, C.byref(name), C.byref(numel)) == 0
            t = sd[name.value.decode()].detach().float().contiguous()
            assert t.numel() == numel.value, name.value
            assert lib.f5hip_bigvgan_load_tensor(self.ctx, name.value, C.c_void_p(t.data_ptr()), t.numel()) == 0
        assert lib.f5hip_bigvgan_finalize(self.ctx) == 0, lib.f5hip_bigvgan_last_error(self.ctx)

    def option(self, key, value):
        assert self.lib.f5hip_bigvgan_set_option(self.ctx, key.encode(), value) == 0, self.lib.f5hip_bigvgan_last_error(self.ctx)

    def forward(self, mel, precision=0, channel_major=True, out_shape=None):
        C = self.C
        b, t = mel.shape[0], (mel.shape[2] if channel_major else mel.shape[1])
        mel = mel.contiguous().float()
        out = torch.full(out_shape or (b, t * self.cfg.hop), float("nan"))
        st = self.lib.f5hip_bigvgan_forward(self.ctx, C.c_void_p(mel.data_ptr()), b, t, int(channel_major), precision, C.c_void_p(out.data_ptr()), None)
        assert st == 0, self.lib.f5hip_bigvgan_last_error(self.ctx)
        return out

    def close(self):
        self.lib.f5hip_bigvgan_destroy(self.ctx)


@pytest.mark.parametrize("name", ["BIGVGAN_TINY", "BIGVGAN_TINY2"])
def test_whole_generator_through_the_product_orchestration_code(emu_lib, name):
    """bigvgan_api.cpp end to end on the CPU: weight matrices, buffer ping-pong, strides, every launch — stage by stage against the
    oracle, in fp32, then the waveform in the two fp16 operand modes, all three conv implementations, both input layouts."""
    from f5_tts_amd import config, synth
    from oracle import bigvgan_oracle as BO

    cfg = getattr(config, name)
    sd = synth.synth_bigvgan_state_dict(cfg, seed=2)
    voc = EmuBigVGAN(emu_lib, cfg, sd)
    T = 7
    mel = torch.randn(2, cfg.num_mels, T, generator=torch.Generator().manual_seed(11))
    want, stages = BO.bigvgan_forward(sd, cfg, mel, return_stages=True)
    for k, ref in enumerate(stages):
        voc.option("stop_after_stage", k)
        got = voc.forward(mel, out_shape=(2, ref.shape[2], ref.shape[1])).transpose(1, 2)
        assert (got - ref).abs().max().item() < 2e-5 * max(1.0, ref.abs().max().item()), f"stage {k}"
    voc.option("stop_after_stage", -1)
    voc.option("conv_impl", 0)  # the launch counts below are those of the materialised-operand path (the library default is 2)
    voc.option("profile", 1)
    base = voc.forward(mel)
    voc.option("profile", 0)
    import ctypes as C

    name, calls, ms, fl, by = C.c_char_p(), C.c_int64(), C.c_double(), C.c_double(), C.c_double()
    stats = {}
    for i in range(emu_lib.f5hip_bigvgan_num_kernel_stats(voc.ctx)):
        assert emu_lib.f5hip_bigvgan_kernel_stat(voc.ctx, i, C.byref(name), C.byref(calls), C.byref(ms), C.byref(fl), C.byref(by)) == 0
        stats[name.value.decode()] = (calls.value, fl.value, by.value)
    nconv = 1 + len(cfg.upsample_rates) + sum((2 if cfg.resblock == "1" else 1) * len(d) for d in cfg.resblock_dilation_sizes) * len(cfg.upsample_rates)
    assert stats["conv_gemm"][0] == nconv == stats["operand"][0] and stats["conv_gemm"][1] > 0  # one GEMM + one operand emission per conv
    assert stats["activation1d"][0] == nconv - 1 - len(cfg.upsample_rates) + 1 and stats["other"][0] == len(cfg.upsample_rates) + 1
    assert (base - want[:, 0]).abs().max().item() < 3e-5
    assert (voc.forward(mel.transpose(1, 2).contiguous(), channel_major=False) - base).abs().max().item() == 0.0
    for impl in (1, 2):
        voc.option("conv_impl", impl)
        assert (voc.forward(mel) - base).abs().max().item() < 1e-5, impl
    for prec, tol in ((1, 2e-4), (2, 3e-2)):  # F5HIP_PREC_FP16X3, F5HIP_PREC_FP16
        for impl in (0, 1, 2):
            voc.option("conv_impl", impl)
            assert (voc.forward(mel, precision=prec) - want[:, 0]).abs().max().item() < tol, (prec, impl)
    voc.close()


def test_emulated_library_rejects_what_the_real_one_rejects(emu_lib):
    import ctypes as C

    from f5_tts_amd import config, synth
    from f5_tts_amd.bigvgan import _config_c

    cfg = config.BIGVGAN_TINY
    ctx = C.c_void_p()
    assert emu_lib.f5hip_bigvgan_create(C.byref(_config_c(cfg)), 0, C.byref(ctx)) == 0
    assert emu_lib.f5hip_bigvgan_finalize(ctx) == 3 and b"never loaded" in emu_lib.f5hip_bigvgan_last_error(ctx)  # F5HIP_ERR_STATE
    x = torch.zeros(4)
    assert emu_lib.f5hip_bigvgan_load_tensor(ctx, b"conv_pre.bias", C.c_void_p(x.data_ptr()), 4) == 1  # wrong size
    assert emu_lib.f5hip_bigvgan_load_tensor(ctx, b"nope", C.c_void_p(x.data_ptr()), 4) == 1
    assert emu_lib.f5hip_bigvgan_forward(ctx, C.c_void_p(x.data_ptr()), 1, 1, 1, 0, C.c_void_p(x.data_ptr()), None) == 3  # not finalised
    assert emu_lib.f5hip_bigvgan_set_option(ctx, b"conv_impl", 7) == 1
    emu_lib.f5hip_bigvgan_destroy(ctx)


# ---- flash attention (csrc/attention_kernel.h): the GPU-proven kernel through the shim, then its key-split variant ---------------------
def _attn_case(rng, Bp, heads, n, nsplit, kvlen=None, log2q=False, qgain=1.0):
    bh = Bp * heads
    q = (rng.standard_normal((bh, n, 64)) * 0.5 * qgain / 8.0).astype(np.float32)  # pre-scaled by 1/sqrt(64), as the QKV epilogue leaves it
    if log2q:
        q *= np.float32(np.log2(np.e))  # ... and by log2(e): the kernel's scores are base-2 logarithms
    k = (rng.standard_normal((bh, n, 64)) * 1.5).astype(np.float32)
    v = rng.standard_normal((bh, n, 64)).astype(np.float32)
    ldv = (n + 7) & ~7
    vt = np.zeros((bh, 64, ldv), dtype=np.float32)
    vt[:, :, :n] = v.transpose(0, 2, 1)
    hi = lambda x: x.astype(np.float16)  # noqa: E731
    lo = lambda x: (x - x.astype(np.float16).astype(np.float32)).astype(np.float16)  # noqa: E731
    files = dict(q=hi(q), k=hi(k), vt=hi(vt))
    if nsplit >= 2:
        files.update(q_lo=lo(q), k_lo=lo(k))
    if nsplit == 3:
        files.update(vt_lo=lo(vt))
    if kvlen is not None:
        files["kvlen"] = np.asarray(kvlen, dtype=np.int32)
    val = (lambda x: hi(x).astype(np.float64) + lo(x).astype(np.float64)) if nsplit >= 2 else (lambda x: hi(x).astype(np.float64))
    vval = (lambda x: hi(x).astype(np.float64) + lo(x).astype(np.float64)) if nsplit == 3 else (lambda x: hi(x).astype(np.float64))
    s = val(q) @ val(k).transpose(0, 2, 1) * (np.log(2.0) if log2q else 1.0)
    if kvlen is not None:
        for b in range(Bp):
            s[b * heads:(b + 1) * heads, :, kvlen[b]:] = -np.inf
    p = np.exp(s - s.max(-1, keepdims=True))
    want = (p / p.sum(-1, keepdims=True)) @ vval(v)  # [bh, n, 64]
    want = want.reshape(Bp, heads, n, 64).transpose(0, 2, 1, 3).reshape(Bp, n, heads * 64)
    return files, want


@pytest.mark.parametrize("nsplit,kvs,n,kvlen", [(2, 1, 200, None), (1, 1, 70, None), (3, 1, 130, [130, 77]),
                                                (2, 2, 200, None), (2, 3, 333, None), (1, 4, 70, None), (3, 2, 130, [130, 77]), (2, 8, 64, None)])
def test_flash_attention_kernel_and_its_key_split_variant(exe, tmp_path, nsplit, kvs, n, kvlen):
    """kvs == 1: the kernel the DiT parity suite has proven on the GPU, run through the shim (validates the shim on __shfl_xor, V^T
    tiles, online softmax).  kvs > 1: the key-split variant + merge kernel written without a GPU, against the same softmax."""
    rng = np.random.default_rng(n + kvs)
    Bp, heads = 2, 2
    files, want = _attn_case(rng, Bp, heads, n, nsplit, kvlen)
    run(exe, tmp_path, "attn", nsplit, Bp, heads, n, kvs, 1, int(kvlen is not None), **files)
    got = decode_operand(open(os.path.join(tmp_path, "out.bin"), "rb").read(), Bp * n, heads * 64, OP_F16X3).reshape(Bp, n, heads * 64)
    tol = 3e-3 if nsplit < 3 else 2e-5  # P (and V) rounded to fp16 in the PV product unless everything is split
    assert np.abs(got - want).max() < tol * max(1.0, np.abs(want).max())


@pytest.mark.parametrize("form,kvs,n,kvlen", [(0, 1, 200, None), (0, 1, 130, [130, 77]), (0, 3, 333, None), (40, 1, 450, [450, 301]), (60, 1, 450, None), (60, 1, 130, [1, 130])])
def test_flash_attention_with_mx_corrected_scores(exe, tmp_path, form, kvs, n, kvlen):
    """flash_attn_kernel<2, 1> (round 5's default in the parity modes): scores = fp16 hi . hi + both correction products as one MX-fp6 MFMA
    per 32 head channels, from P words packed the way the q|k|v epilogue packs them.  Logits of tens, so a slip in the P-word layout (or in
    which half-wave reads which word) shows: the output must follow the softmax of the SPLIT values where plain fp16 q, k are ~10x off.
    Forms: the exact-maximum kernel, its key-split variant + merge, and the lazy 4- / 6-wave forms the engine launches."""
    rng = np.random.default_rng(n + kvs + form)
    Bp, heads = 2, 2
    files, want = _attn_case(rng, Bp, heads, n, 2, kvlen, log2q=form != 0, qgain=8.0)
    run(exe, tmp_path, "attn", 4, Bp, heads, n, kvs, 1, int(kvlen is not None), form, **files)
    got = decode_operand(open(os.path.join(tmp_path, "out.bin"), "rb").read(), Bp * n, heads * 64, OP_F16X3).reshape(Bp, n, heads * 64)
    err_mx = np.abs(got - want).mean()
    plain_form = {0: (), 40: (4,), 60: (16,)}[form]  # plain fp16 scores in the same units: the exact-maximum kernel / the pipelined lazy forms (log2q)
    run(exe, tmp_path, "attn", 1, Bp, heads, n, kvs, 1, int(kvlen is not None), *plain_form, **{k: v for k, v in files.items() if not k.endswith("_lo")})
    plain = decode_operand(open(os.path.join(tmp_path, "out.bin"), "rb").read(), Bp * n, heads * 64, OP_F16X3).reshape(Bp, n, heads * 64)
    err_plain = np.abs(plain - want).mean()
    assert np.abs(got - want).max() < 3e-3 * max(1.0, np.abs(want).max())
    print(f"mean |error| against the split-value softmax: MX-corrected {err_mx:.2e}, plain fp16 scores {err_plain:.2e}")
    assert err_mx < 0.5 * err_plain, (err_mx, err_plain)


@pytest.mark.parametrize("form,n,kvlen,qgain", [(42, 200, None, 8.0), (42, 130, [130, 77], 8.0), (43, 333, None, 8.0), (62, 450, [450, 301], 8.0),
                                                (42, 333, None, 0.25), (62, 450, [450, 301], 0.25)])
def test_flash_attention_with_v_as_hi_lo_halves(exe, tmp_path, form, n, kvlen, qgain):
    """Round 6, flash_attn_kernel<2, 2> / <2, 3> (engine option attn_impl 6 / 7): the MX-corrected scores with V read as fp16 hi + lo halves
    (O = V_hi P + V_lo P), P as well in the <2, 3> form.  SHARP rows (q gain 8: softmax rows close to one-hot) are where the rounding of V is
    not averaged away: the output must follow the split-value softmax closer than the plain P . V form does — by the full factor where P is
    split too.  FLAT rows (q gain 0.25: hundreds of comparable weights) are where the lazy <2, 2> forms SKIP the V_lo product on every tile
    after the first (VADAPT, attention_kernel.h): the output then equals the plain form's to the averaged-away rounding."""
    rng = np.random.default_rng(n + form)
    Bp, heads = 2, 2
    files, want = _attn_case(rng, Bp, heads, n, 3, kvlen, log2q=True, qgain=qgain)  # nsplit 3: V's remainder plane and a reference on the split values
    run(exe, tmp_path, "attn", 4, Bp, heads, n, 1, 1, int(kvlen is not None), form, **files)
    got = decode_operand(open(os.path.join(tmp_path, "out.bin"), "rb").read(), Bp * n, heads * 64, OP_F16X3).reshape(Bp, n, heads * 64)
    run(exe, tmp_path, "attn", 4, Bp, heads, n, 1, 1, int(kvlen is not None), 60 if form == 62 else 40, **{k: v for k, v in files.items() if k != "vt_lo"})
    base = decode_operand(open(os.path.join(tmp_path, "out.bin"), "rb").read(), Bp * n, heads * 64, OP_F16X3).reshape(Bp, n, heads * 64)
    e, e0 = np.abs(got - want).mean(), np.abs(base - want).mean()
    print(f"form {form}: mean |error| against the split-value softmax {e:.2e}; plain fp16 P . V {e0:.2e}")
    assert np.abs(got - want).max() < 3e-3 * max(1.0, np.abs(want).max())
    if qgain < 1.0:  # skipped remainder products: nothing gained, nothing lost
        assert e < 1.05 * e0 and np.abs(got - base).max() < 2e-4 * max(1.0, np.abs(want).max()), (e, e0)
    else:
        assert e < (0.25 if form == 43 else 0.8) * e0, (e, e0)


@pytest.mark.parametrize("pipe,n,kvlen", [(4, 200, None), (14, 70, None), (6, 450, [450, 301]), (16, 130, [130, 77]), (4, 64, None), (4, 333, [1, 333])])
def test_software_pipelined_flash_attention(exe, tmp_path, pipe, n, kvlen):
    """flash_pipe_kernel (scores of tile t + 1 issued inside the softmax of tile t; skewed K / V^T ring): 4- and 6-wave blocks, row sums on
    either pipe, one tile, odd and even tile counts, masked tails down to a single valid key."""
    rng = np.random.default_rng(n + pipe)
    Bp, heads = 2, 2
    files, want = _attn_case(rng, Bp, heads, n, 1, kvlen, log2q=True)
    run(exe, tmp_path, "attn", 1, Bp, heads, n, 1, 1, int(kvlen is not None), pipe, **files)
    got = decode_operand(open(os.path.join(tmp_path, "out.bin"), "rb").read(), Bp * n, heads * 64, OP_F16X3).reshape(Bp, n, heads * 64)
    assert np.abs(got - want).max() < 3e-3 * max(1.0, np.abs(want).max())


# ---- the DiT engine itself (api.cpp + every kernel translation unit) on the CPU: the product's own host classes over the shim ------------
@pytest.fixture(scope="module")
def engine_emu_lib():
    import ctypes as C

    import f5_tts_amd  # noqa: F401
    from f5_tts_amd import binding

    import hipemu_build  # tests/hipemu_build.py: conftest.py starts the stale translation units in the background at collection; this waits and links

    path = hipemu_build.ensure()
    lib = C.CDLL(path, mode=C.RTLD_LOCAL)
    for name, (res, args) in {**binding.SYMBOLS, **binding.BENCH_SYMBOLS}.items():  # (one library here: engine + microbench.cpp)
        if hasattr(lib, name):  # the BigVGAN / fault-reproducer entry points are not part of this build
            fn = getattr(lib, name)
            fn.restype, fn.argtypes = res, args
    return lib


def host_alias(ptr, numel, _device=None):
    """engine._as_tensor for the shim: its "device" pointers are host pointers, so the alias is a torch view of that memory."""
    import ctypes as C

    return torch.frombuffer((C.c_float * numel).from_address(ptr), dtype=torch.float32)


@pytest.fixture()
def emu_engine(engine_emu_lib, monkeypatch):
    """F5HipEngine / F5HipCFM (the product's host classes, unmodified) over the emulated library: "device" tensors are CPU tensors."""
    import contextlib
    import types

    from f5_tts_amd import engine as E

    monkeypatch.setattr(E, "load_library", lambda *a, **k: engine_emu_lib)
    monkeypatch.setattr(E, "load_bench_library", lambda *a, **k: engine_emu_lib)
    monkeypatch.setattr(E, "_as_tensor", host_alias)
    monkeypatch.setattr(torch.cuda, "device", lambda *_a, **_k: contextlib.nullcontext())
    monkeypatch.setattr(torch.cuda, "current_stream", lambda *_a, **_k: types.SimpleNamespace(cuda_stream=0))
    made = []

    def make(cfg, vcfg=None):
        eng = E.F5HipEngine(cfg, vcfg, device="cuda:0")  # a descriptor only; no GPU is touched
        eng.device = torch.device("cpu")
        made.append(eng)
        return eng

    yield make
    for eng in made:
        eng.close()


def test_engine_on_the_shim_matches_the_oracle_and_the_schedule_switches_change_nothing(emu_engine):
    """The tiny DiT through api.cpp and every kernel translation unit on the CPU: fp32 against the oracle (what smoke() checks on the
    GPU), then fp16x3 with each schedule switch (one chain / two chains, key-split attention) against the default path."""
    from f5_tts_amd import config, synth
    from f5_tts_amd.engine import F5HipCFM
    from oracle import f5_oracle as O

    cfg, vcfg = config.DIT_TINY, config.VOCOS_TINY
    sd, vsd = synth.synth_dit_state_dict(cfg, seed=1), synth.synth_vocos_state_dict(vcfg, seed=1)
    eng = emu_engine(cfg, vcfg)
    eng.load_state_dict({**sd, **vsd})
    B, n = 2, 200
    wav = synth.synth_wave(256 * 40, seed=3, batch=B)
    text = synth.synth_text_ids(B, 30, cfg.text_num_embeds, seed=2)
    dur = torch.tensor([n, n - 37])  # ragged: row masks, padded keys
    kw = dict(steps=1, cfg_strength=2.0, sway_sampling_coef=-1.0, seed=7)
    ref, _ = O.cfm_sample(sd, cfg, wav, text, dur, **kw)
    out, _ = F5HipCFM(eng, precision="fp32").sample(wav, text, dur, **kw)
    assert (out - ref).abs().max().item() < 1e-4
    gen = out[:1, 40:, :].permute(0, 2, 1)
    assert (eng.vocos_decode(gen.contiguous()) - O.vocos_decode(vsd, gen, vcfg.num_layers)).abs().max().item() < 1e-3
    model = F5HipCFM(eng, precision="fp16x3")
    base, _ = model.sample(wav, text, dur, **kw)
    assert (base - ref).abs().max().item() < 5e-4
    for opts in (dict(branch_streams=0), dict(attn_kv_split=3, branch_streams=0), dict(attn_kv_split=2, branch_streams=1)):
        for k, v in opts.items():
            eng.set_option(k, v)
        got, _ = model.sample(wav, text, dur, **kw)
        assert (got - base).abs().max().item() < 2e-4, opts
        for k in opts:
            eng.set_option(k, {"branch_streams": -1, "attn_kv_split": 1}[k])


def test_qkv_epilogue_fast_index_path_equals_the_general_path(tmp_path):
    """csrc/gemm.h EpiQKVFast (no integer divisions, 32-bit offsets: the default since the end of round 1) against EpiQKV (the path every GPU
    parity run of round 1 used): tests/hipemu/qkv_index_check.cpp pushes every (row, 4-channel unit) of full-size QKV outputs through both
    and compares the slabs byte for byte; plus the invariant-multiplier division exhaustively and the refusal cases."""
    exe = str(tmp_path / "qkv_index_check")
    csrc, emu = os.path.join(ROOT, "f5-tts_amd", "csrc"), os.path.join(ROOT, "tests", "hipemu")
    r = subprocess.run([CLANG, "-x", "c++", "-std=c++20", "-O2", "-pthread", "-DF5_HIPEMU", "-I", emu, "-I", csrc, "-Wno-unknown-pragmas", "-Wno-pass-failed",
                        "-Wno-psabi", os.path.join(emu, "qkv_index_check.cpp"), "-o", exe], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    r = subprocess.run([exe] + (["big"] if os.environ.get("F5HIP_SHIM_FULL") == "1" else []), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "all ok" in r.stdout and "FAIL" not in r.stdout, r.stdout[-3000:]
    assert r.stdout.count("byte-identical") >= 8
