import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import f5_tts_amd  # noqa: E402,F401  (registers the hyphenated package dir under an importable name)


def pytest_configure(config):
    try:  # the CPU oracle is small-tensor work: a few threads beat an oversubscribed pool on shared CI hosts
        import torch

        torch.set_num_threads(min(4, os.cpu_count() or 1))
    except Exception:  # pragma: no cover
        pass
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `pytest -m gpu` through gpurun)")


def _needs_host_engine(item):
    return "engine_emu_lib" in getattr(item, "fixturenames", ())


def pytest_collection_modifyitems(config, items):
    # tests over the host build of the engine go last: its compile (tests/hipemu_build.py, ~2 min of one core) then runs beside the others
    items.sort(key=_needs_host_engine)  # stable
    try:
        import torch

        has_gpu = torch.cuda.is_available()
    except Exception:  # pragma: no cover
        has_gpu = False
    if has_gpu:
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


def pytest_collection_finish(session):
    try:
        import torch

        if torch.cuda.is_available():
            return
    except Exception:  # pragma: no cover
        pass
    if any(_needs_host_engine(it) and "gpu" not in it.keywords for it in session.items):
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import hipemu_build

        hipemu_build.start_background()
