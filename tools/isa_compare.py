#!/usr/bin/env python3
"""Compare the gfx950 kernels of two builds instruction by instruction (no GPU needed).

    python tools/isa_compare.py OLD_DIR NEW_DIR [file.o ...]      (directories holding the objects of f5-tts_amd/csrc/Makefile)

For every object: kernels whose disassembly (addresses and padding stripped) is identical, kernels that differ, kernels that were renamed
(template parameters added) but are instruction-identical to an old one, and new kernels.  Used at the end of round 1 to show that the
header refactors made without GPU minutes left the default path's kernels exactly as the last GPU-verified build had them:
    git archive <last GPU-verified commit> f5-tts_amd/csrc include | tar x -C /tmp/old && make -C /tmp/old/f5-tts_amd/csrc
    python tools/isa_compare.py /tmp/old/f5-tts_amd/csrc f5-tts_amd/csrc
"""
import hashlib
import os
import re
import subprocess
import sys
import tempfile

LL = "/opt/rocm/lib/llvm/bin"


def kernels(obj, workdir):
    fb, co = os.path.join(workdir, "fb"), os.path.join(workdir, "co")
    subprocess.run([f"{LL}/llvm-objcopy", f"--dump-section=.hip_fatbin={fb}", obj], check=True, capture_output=True)
    subprocess.run([f"{LL}/clang-offload-bundler", "--unbundle", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--input={fb}", f"--output={co}"],
                   check=True, capture_output=True)
    dis = subprocess.run([f"{LL}/llvm-objdump", "-d", "--no-show-raw-insn", co], capture_output=True, text=True, check=True).stdout
    out, cur = {}, None
    for ln in dis.splitlines():
        m = re.match(r"^[0-9a-f]+ <(.+)>:$", ln)
        if m:
            cur = m.group(1)
            out[cur] = []
            continue
        t = re.sub(r"<[^>]*>", "<L>", re.sub(r"^\s*[0-9a-f]+:\s*", "", ln.split("//")[0]).strip())
        if cur and t and t != "...":  # "..." = alignment padding between functions
            out[cur].append(t)
    return {k: hashlib.md5("\n".join(v).encode()).hexdigest() + f":{len(v)}" for k, v in out.items()}


def demangle(name):
    try:
        return subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()[:180]
    except OSError:
        return name[:180]


def main():
    old_dir, new_dir = sys.argv[1], sys.argv[2]
    files = sys.argv[3:] or sorted(f for f in os.listdir(new_dir) if f.endswith(".o") and os.path.exists(os.path.join(old_dir, f)))
    rc = 0
    for f in files:
        with tempfile.TemporaryDirectory() as d1, tempfile.TemporaryDirectory() as d2:
            try:
                a, b = kernels(os.path.join(old_dir, f), d1), kernels(os.path.join(new_dir, f), d2)
            except subprocess.CalledProcessError:
                print(f"{f}: no gfx950 code object (host-only translation unit)")
                continue
        by_hash = {}
        for k, v in b.items():
            by_hash.setdefault(v, []).append(k)
        same = [k for k in a if b.get(k) == a[k]]
        differ = [k for k in a if k in b and b[k] != a[k]]
        renamed = [k for k in a if k not in b and a[k] in by_hash]
        gone = [k for k in a if k not in b and a[k] not in by_hash]
        new = [k for k in b if k not in a and not any(a[o] == b[k] for o in renamed)]
        print(f"{f}: {len(same)} identical, {len(renamed)} renamed but identical, {len(differ)} differ, {len(gone)} removed, {len(new)} new")
        for k in differ:
            print("   DIFFERS", demangle(k), a[k].split(":")[1], "->", b[k].split(":")[1], "instructions")
        for k in gone:
            print("   REMOVED", demangle(k))
        rc |= bool(differ or gone)
    return rc


if __name__ == "__main__":
    sys.exit(main())
