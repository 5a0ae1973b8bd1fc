"""Build of tests/c_abi/_build/engine_emu/libf5hip_engine_emu.so — the whole engine (api.cpp + every kernel translation unit) compiled for
the HOST over tests/hipemu/hipemu.h.  gemm.hip alone takes ~2 minutes of one core, so conftest.py starts the stale translation units in the
background as soon as the collection shows that a test will need the library, and orders those tests last; the fixture waits here."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLANG = "/opt/rocm/lib/llvm/bin/clang++"
OUT = os.path.join(ROOT, "tests", "c_abi", "_build", "engine_emu")
LIB = os.path.join(OUT, "libf5hip_engine_emu.so")
SOURCES = ("gemm.hip", "gemm_p8.hip", "elementwise.hip", "convpos.hip", "attention.hip", "audio.hip", "api.cpp", "microbench.cpp")
_running = []  # (Popen, object path) started by start_background()


def _objects():
    return [os.path.join(OUT, src + ".o") for src in SOURCES]


def _stale():
    csrc, emu = os.path.join(ROOT, "f5-tts_amd", "csrc"), os.path.join(ROOT, "tests", "hipemu")
    deps = [os.path.join(emu, "hipemu.h"), os.path.join(ROOT, "include", "f5hip.h")] + [os.path.join(csrc, h) for h in os.listdir(csrc) if h.endswith(".h")]
    out = []
    for src, obj in zip(SOURCES, _objects()):
        sp = os.path.join(csrc, src)
        if not os.path.exists(obj) or any(os.path.getmtime(d) > os.path.getmtime(obj) for d in deps + [sp]):
            out.append(([CLANG, "-x", "c++", "-std=c++20", "-O1", "-pthread", "-fPIC", "-DF5_HIPEMU", "-I", emu, "-I", csrc, "-Wno-unknown-pragmas",
                         "-Wno-pass-failed", "-Wno-psabi", "-c", sp, "-o", obj + ".part"], obj))
    return out


def start_background():
    """Start the compilers for every stale translation unit and return at once (idempotent)."""
    if _running or not os.path.exists(CLANG):
        return
    os.makedirs(OUT, exist_ok=True)
    for cmd, obj in _stale():
        _running.append((subprocess.Popen(cmd, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, text=True), obj))


def ensure():
    """The up-to-date library's path (compiles / waits for the background compilers / links as needed)."""
    os.makedirs(OUT, exist_ok=True)
    start_background()
    while _running:
        proc, obj = _running.pop()
        _, err = proc.communicate()
        assert proc.returncode == 0, err[-3000:]
        if os.path.exists(obj + ".part"):
            os.replace(obj + ".part", obj)  # an interrupted compile never leaves a fresh-looking object behind
        else:  # another process building the same tree got there first (both compiled the same sources)
            assert os.path.exists(obj), obj
    objs = _objects()
    if not os.path.exists(LIB) or any(os.path.getmtime(o) > os.path.getmtime(LIB) for o in objs):
        r = subprocess.run([CLANG, "-shared", "-fPIC", "-pthread", "-Wl,-Bsymbolic", "-Wl,--no-undefined", "-o", LIB] + objs, capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-3000:]
    return LIB
