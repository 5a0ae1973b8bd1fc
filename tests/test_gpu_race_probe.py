"""GPU regression test around the fault that round 2 called a "co-residency race" (csrc/race_probe.hip).

Root cause (round 3): on gfx950 a packed-fp32 instruction whose op_sel takes src0's low half from the low register and src1's low half
from the HIGH register (``v_pk_mul_f32 .. op_sel:[0,1]`` and the same selection on v_pk_add_f32 / v_pk_fma_f32) reads src1 as ZERO in lanes
48-63 while another wave on the same SIMD issues MFMAs and LDS reads.  The library is built without packed fp32 (tests/test_isa_hazards.py
scans for the form); here the reproducer runs the tiles that put two workgroups on a CU:

* the production epilogue form (EXPT 0), the asm-pinned SAFE operand order (EXPT 21) and the single-lane rotation (EXPT 14) must give ZERO wrong
  outputs on every launch — these are the forms the engine relies on now that those tiles are back in its table;
* the asm-pinned FAILING form (EXPT 18) is run and its count printed: it reproduces on the MI355X boxes of this project (thousands of wrong
  outputs per launch).  It is not asserted — a firmware that fixes the fault must not turn the suite red.
"""
import ctypes as C
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import f5_tts_amd  # noqa: E402,F401
from f5_tts_amd import config  # noqa: E402

pytestmark = pytest.mark.gpu
LAUNCHES = 4


@pytest.fixture(scope="module")
def probe():
    from f5_tts_amd.engine import F5HipEngine

    eng = F5HipEngine(config.DIT_TINY, None, device=0)

    def run(tile, expt, lds_pad=0):
        bad = (C.c_int64 * LAUNCHES)()
        st = eng.bench_lib.f5hip_bench_qkv_probe(eng._ctx, tile, expt, 0, lds_pad, 0, 2, 1406, LAUNCHES, bad, None)
        assert st == 0, st
        return list(bad)

    yield run
    eng.close()


@pytest.mark.parametrize("tile", [58, 62, 63])
@pytest.mark.parametrize("expt", [0, 21, 14])
def test_two_workgroups_per_cu_tiles_are_exact_in_the_safe_forms(probe, tile, expt):
    assert probe(tile, expt) == [0] * LAUNCHES


def test_the_failing_operand_selection_is_reported(probe):
    two_per_cu = probe(58, 18)
    one_per_cu = probe(58, 18, lds_pad=32768)  # the same code with one workgroup per CU: no partner wave, no fault
    print(f"v_pk_mul_f32 op_sel:[0,1] in the q|k|v epilogue, tile 58: wrong outputs per launch {two_per_cu} with two workgroups per CU, "
          f"{one_per_cu} with one")
    assert one_per_cu == [0] * LAUNCHES
