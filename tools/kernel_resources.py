#!/usr/bin/env python3
"""Per-kernel register / scratch / occupancy table of the gfx950 code objects inside the built libf5hip.so (no GPU needed).

    python tools/kernel_resources.py [pattern ...] > profiles/rNN_kernel_resources.md

vgpr_count is the unified count (arch + acc VGPRs, granule 8); a SIMD has 512 per lane, so waves/SIMD = min(8, 512 // vgprs) before the
LDS and workgroup-size limits.  Dynamic LDS is a launch parameter and is quoted from the launchers in DESIGN.md, not from here."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "f5-tts_amd", "csrc", "libf5hip.so")
LL = "/opt/rocm/lib/llvm/bin"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def code_objects(workdir):
    fb = os.path.join(workdir, "fatbin")
    subprocess.run([f"{LL}/llvm-objcopy", f"--dump-section=.hip_fatbin={fb}", LIB], check=True, capture_output=True)
    blob = open(fb, "rb").read()
    starts = [m.start() for m in re.finditer(re.escape(MAGIC), blob)]
    for k, s in enumerate(starts):
        piece, co = os.path.join(workdir, f"b{k}"), os.path.join(workdir, f"b{k}.co")
        open(piece, "wb").write(blob[s:starts[k + 1] if k + 1 < len(starts) else len(blob)])
        r = subprocess.run([f"{LL}/clang-offload-bundler", "--unbundle", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--input={piece}", f"--output={co}"],
                           capture_output=True)
        if r.returncode == 0 and os.path.exists(co) and os.path.getsize(co):
            yield co


def kernels(co):
    txt = subprocess.run([f"{LL}/llvm-readelf", "--notes", co], capture_output=True, text=True, check=True).stdout
    for block in txt.split("  - .agpr_count:")[1:]:
        k = {"agpr_count": block.split("\n", 1)[0].strip()}
        for key in ("name", "private_segment_fixed_size", "group_segment_fixed_size", "sgpr_count", "vgpr_count", "vgpr_spill_count", "max_flat_workgroup_size"):
            m = re.search(rf"^\s+\.{key}:\s+(\S+)", block, re.M)
            if m:
                k[key] = m.group(1)
        yield k


def main():
    pats = [re.compile(p) for p in sys.argv[1:]]
    rows = []
    with tempfile.TemporaryDirectory() as d:
        for co in code_objects(d):
            for k in kernels(co):
                # binutils' c++filt predates the _Float16 mangling (DF16_): demangle it as the old half type and name it back
                name = subprocess.run(["c++filt", k["name"].replace("DF16_", "Dh")], capture_output=True, text=True).stdout.strip()
                name = name.replace("(anonymous namespace)::", "").replace("half", "_Float16")
                name = re.sub(r"\(.*$", "", name).replace("void ", "")
                if pats and not any(p.search(name) for p in pats):
                    continue
                v = int(k["vgpr_count"])
                rows.append((name, int(k["max_flat_workgroup_size"]), v, int(k["agpr_count"]), int(k["sgpr_count"]), int(k["group_segment_fixed_size"]),
                             int(k["private_segment_fixed_size"]), min(8, 512 // max(v, 1))))
    print("| kernel | threads | VGPRs (arch+acc) | of which acc | SGPRs | static LDS B | scratch B | waves/SIMD by registers |")
    print("|---|---:|---:|---:|---:|---:|---:|---:|")
    for r in sorted(rows):
        print("| `" + r[0] + "` | " + " | ".join(str(x) for x in r[1:]) + " |")


if __name__ == "__main__":
    main()
