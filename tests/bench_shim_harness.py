"""Run the repo's bench.py — unmodified main(), real self-launch and rank protocol — with libf5hip built for the host
(tests/hipemu) instead of a GPU: `python tests/bench_shim_harness.py --tiny ...` (F5HIP_EMU_LIB = the emulated library).  Test infrastructure:
what it prints is not a measurement of anything."""
import contextlib
import ctypes as C
import os
import sys
import types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import torch  # noqa: E402

import f5_tts_amd  # noqa: E402,F401
from f5_tts_amd import binding  # noqa: E402
from f5_tts_amd import engine as E  # noqa: E402


def host_alias(ptr, numel, _device=None):
    return torch.frombuffer((C.c_float * numel).from_address(ptr), dtype=torch.float32)


def install():
    lib = C.CDLL(os.environ["F5HIP_EMU_LIB"], mode=C.RTLD_LOCAL)
    for name, (res, args) in binding.SYMBOLS.items():
        if hasattr(lib, name):
            fn = getattr(lib, name)
            fn.restype, fn.argtypes = res, args
    E.load_library = lambda *a, **k: lib
    E._as_tensor = host_alias
    torch.cuda.device = lambda *_a, **_k: contextlib.nullcontext()
    torch.cuda.current_stream = lambda *_a, **_k: types.SimpleNamespace(cuda_stream=0)
    torch.cuda.set_device = lambda *_a, **_k: None
    torch.cuda.synchronize = lambda *_a, **_k: None
    torch.Tensor.cuda = lambda self, *a, **k: self
    orig_init = E.F5HipEngine.__init__

    def init_on_cpu(self, dit_cfg, vocos_cfg=None, device=0):
        orig_init(self, dit_cfg, vocos_cfg, device="cuda:0")  # a descriptor only: the emulated library never touches a GPU
        self.device = torch.device("cpu")

    E.F5HipEngine.__init__ = init_on_cpu
    if os.environ.get("SHIM_RANKS_SHARE_ONE_DEVICE"):  # the refuse case of bench.py's device census: every rank reports the first rank's device
        from f5_tts_amd import dist as fdist

        real = fdist.device_identity
        fdist.device_identity = lambda local, device_type="cuda": dict(real(local, device_type), device_index=0, pci_bus_id="0000:05:00.0")
    import bench

    bench.DEVICE_TYPE = "cpu"
    bench.LAUNCH_CMD = [sys.executable, os.path.abspath(__file__)]  # --gpus N > 1 re-executes THIS harness per rank
    return bench


if __name__ == "__main__":
    install().main()
