#!/usr/bin/env python3
"""Instruction mix of the EPILOGUE (everything behind the last MFMA) of the gfx950 kernels in an object of f5-tts_amd/csrc (no GPU needed).

    python tools/epilogue_mix.py f5-tts_amd/csrc/gemm_p8.o [name pattern ...]

Per kernel: static VALU / transcendental / store / load / scratch counts and the issue cycles they cost one wave (4 per full-rate VALU
instruction, 16 per quarter-rate transcendental; packed fp32 is off in this build).  A many-round GEMM launch spends its epilogues with the
matrix pipe idle, so these cycles x 2 waves per SIMD are wall time per tile (round 6: the tanh-GELU epilogue was 90 cycles per output)."""
import collections
import re
import subprocess
import sys
import tempfile

LL = "/opt/rocm/lib/llvm/bin"
TRANS = re.compile(r"v_(exp|rcp|rsq|log|sqrt|sin|cos)_")


def disassemble(obj):
    with tempfile.TemporaryDirectory() as d:
        subprocess.run([f"{LL}/llvm-objcopy", f"--dump-section=.hip_fatbin={d}/fb", obj], check=True, capture_output=True)
        subprocess.run([f"{LL}/clang-offload-bundler", "--unbundle", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--input={d}/fb", f"--output={d}/co"],
                       check=True, capture_output=True)
        return subprocess.run([f"{LL}/llvm-objdump", "-d", "--no-show-raw-insn", f"{d}/co"], capture_output=True, text=True, check=True).stdout


def main():
    pats = [re.compile(p) for p in sys.argv[2:]]
    kernels, cur = {}, None
    for ln in disassemble(sys.argv[1]).splitlines():
        m = re.match(r"^[0-9a-f]+ <(.+)>:$", ln)
        if m:
            cur = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
            cur = re.sub(r"\(.*$", "", cur).replace("void ", "")
            kernels[cur] = []
            continue
        t = ln.split("//")[0].strip()
        if cur and t and not t.endswith(":") and t != "...":
            kernels[cur].append(t.split()[0])
    print("| kernel | instructions | epilogue | VALU | transcendental | issue cycles / wave | stores | loads | scratch |")
    print("|---|---:|---:|---:|---:|---:|---:|---:|---:|")
    for name, ops in kernels.items():
        if pats and not any(p.search(name) for p in pats):
            continue
        last = max((i for i, o in enumerate(ops) if o.startswith("v_mfma")), default=-1)
        epi = collections.Counter(ops[last + 1:])
        tr = sum(n for o, n in epi.items() if TRANS.match(o))
        va = sum(n for o, n in epi.items() if o.startswith("v_")) - tr
        print(f"| `{name}` | {len(ops)} | {len(ops) - last - 1} | {va} | {tr} | {4 * va + 16 * tr} | {sum(n for o, n in epi.items() if 'store' in o and 'scratch' not in o)} | "
              f"{sum(n for o, n in epi.items() if 'load' in o and 'scratch' not in o)} | {sum(n for o, n in epi.items() if 'scratch' in o)} |")
        if len(pats) == 1 and len([k for k in kernels if pats[0].search(k)]) == 1:
            print("\n" + ", ".join(f"{o} {n}" for o, n in epi.most_common(40)))


if __name__ == "__main__":
    main()
