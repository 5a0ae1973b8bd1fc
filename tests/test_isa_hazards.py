"""Static checks on the gfx950 code objects inside the built libf5hip.so — things neither a CPU run of the source (tests/hipemu) nor the
compiler can vouch for:

* gemm_sk.h / gemm_skrs.h issue `global_load_dwordx4 ... sc1` from inline asm WITHOUT a wait (all partners' partial sums in flight
  together) and wait later (`settle`).  The compiler does not know those registers are in flight: if it ever placed a copy, a spill or any
  other use of a destination register between the load and the `s_waitcnt vmcnt(0)`, the kernel would read stale registers on the GPU and
  nowhere else.  The disassembly is scanned for exactly that.
* the kernels of the default schedule and of the probed ones (stream-K reduce-scatter, key-split attention) must not spill to scratch.
"""
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "f5-tts_amd", "csrc", "libf5hip.so")
LL = "/opt/rocm/lib/llvm/bin"
TOOLS = [os.path.join(LL, t) for t in ("llvm-objcopy", "clang-offload-bundler", "llvm-objdump", "llvm-readelf")]
pytestmark = pytest.mark.skipif(not os.path.exists(LIB) or not all(os.path.exists(t) for t in TOOLS), reason="needs the built library and the ROCm llvm tools")
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


@pytest.fixture(scope="module")
def code_objects(tmp_path_factory):
    """Every gfx950 code object of the fat binary (one bundle per translation unit)."""
    d = tmp_path_factory.mktemp("isa")
    fb = str(d / "fatbin")
    subprocess.run([TOOLS[0], f"--dump-section=.hip_fatbin={fb}", LIB], check=True, capture_output=True)
    blob = open(fb, "rb").read()
    starts = [m.start() for m in re.finditer(re.escape(MAGIC), blob)]
    assert starts, "no offload bundles in .hip_fatbin"
    out = []
    for k, s in enumerate(starts):
        piece, co = str(d / f"bundle{k}"), str(d / f"bundle{k}.co")
        open(piece, "wb").write(blob[s:starts[k + 1] if k + 1 < len(starts) else len(blob)])
        r = subprocess.run([TOOLS[1], "--unbundle", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--input={piece}", f"--output={co}"],
                           capture_output=True, text=True)
        if r.returncode == 0 and os.path.exists(co) and os.path.getsize(co) > 0:
            out.append(co)
    assert out, "no gfx950 code object could be extracted"
    return out


def vregs(text):
    regs = set()
    for m in re.finditer(r"\bv\[(\d+):(\d+)\]", text):
        regs |= set(range(int(m.group(1)), int(m.group(2)) + 1))
    for m in re.finditer(r"\bv(\d+)\b", text):
        regs.add(int(m.group(1)))
    return regs


def test_no_instruction_touches_a_register_with_an_inline_asm_load_in_flight(code_objects):
    checked, kernels, bad = 0, 0, []
    for co in code_objects:
        dis = subprocess.run([TOOLS[2], "-d", "--no-show-raw-insn", co], capture_output=True, text=True, check=True).stdout
        name, pending, seen = None, {}, False
        for ln in dis.splitlines():
            m = re.match(r"^[0-9a-f]+ <(.+)>:$", ln)
            if m:
                name, pending = m.group(1), {}
                seen = False
                continue
            if name is None or not ("gemm_sk" in name):  # the only kernels with un-waited inline-asm loads
                continue
            ins = ln.split("//")[0].strip()
            if not ins or ins.endswith(":"):
                continue
            if ins.startswith("global_load_dwordx4") and " sc1" in ins:
                dst, rest = ins.split(",", 1)
                if vregs(rest) & set(pending):
                    bad.append((name, ins, "address register still in flight"))
                for r in vregs(dst):
                    pending[r] = ins
                checked += 1
                if not seen:
                    kernels, seen = kernels + 1, True
                continue
            if ins.startswith("s_waitcnt") and re.search(r"vmcnt\(0\)", ins):
                pending = {}
                continue
            if pending:
                hit = vregs(ins) & set(pending)
                if hit:
                    bad.append((name, ins, f"touches v{sorted(hit)[0]} while {pending[sorted(hit)[0]]} is in flight"))
    assert kernels >= 10 and checked >= 100, (kernels, checked)  # the stream-K kernels are in the build and were scanned
    assert not bad, bad[:5]


def kernel_metadata(co):
    txt = subprocess.run([TOOLS[3], "--notes", co], capture_output=True, text=True, check=True).stdout
    out, cur = [], {}
    for ln in txt.splitlines():
        for key in ("name", "private_segment_fixed_size", "vgpr_count", "vgpr_spill_count"):
            m = re.match(rf"\s+\.{key}:\s+(\S+)", ln)
            if m:
                if key == "name" and "name" in cur and "vgpr_count" in cur:
                    out.append(cur)
                    cur = {}
                cur[key] = m.group(1)
    if "name" in cur:
        out.append(cur)
    return out


def test_default_and_probed_kernels_do_not_spill(code_objects):
    """Scratch is tolerated only in microbenchmark-only experiments (the 16-wave 256x256 tile capped at 128 VGPRs, the first stream-K
    version gemm_sk.h); everything the engine can launch — the default dispatch and the schedules bench.py probes — must be spill-free."""
    experiments = re.compile(r"gemm_sk_kernel|gemm_kernel\w*Li4ELi4ELi0E|gemm_kernel<[^>]*, 4, 4, 0>")
    spilled, n = [], 0
    for co in code_objects:
        for k in kernel_metadata(co):
            n += 1
            if int(k.get("private_segment_fixed_size", "0")) > 0 and not experiments.search(k["name"]):
                spilled.append((k["name"], k["private_segment_fixed_size"]))
    assert n > 150, n
    assert not spilled, spilled
