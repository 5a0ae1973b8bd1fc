"""Summarise a rocprofv3 rocpd (.db) kernel trace into a per-kernel stats table (the `--stats` view):
calls, total/avg/min/max duration, share of GPU time.  Usage: python tools/rocpd_summary.py results.db > profiles/x.md"""
import re
import sqlite3
import subprocess
import sys


def demangle(names):
    try:
        out = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-cxxfilt"], input="\n".join(names), capture_output=True, text=True, check=True).stdout.split("\n")
        return dict(zip(names, out))
    except Exception:
        return {n: n for n in names}


def demangle_local(name: str) -> str:
    """llvm-cxxfilt is not in the image and binutils' c++filt does not know _Float16 (DF16_): decode our own kernels."""
    m = re.match(r"_Z\d+(gemm_kernel|gemm_glds_kernel)I(DF16_|f)Li(\d)ELi(\d)ELi(\d)E\d+(Epi[A-Za-z]+)Li(\d)ELi(\d)ELi(\d)E", name)
    if m:
        t = "_Float16" if m.group(2) == "DF16_" else "float"
        return f"{m.group(1)}<{t}, {m.group(3)}, {m.group(4)}, {m.group(5)}, {m.group(6)}, {m.group(7)}, {m.group(8)}, {m.group(9)}>("
    m = re.match(r"_ZN12_GLOBAL__N_1(\d+)(\w+?)I(.*?)EEv", name)
    if m:
        n = int(m.group(1))
        ident = (m.group(2) + "I" + m.group(3))[:n]
        targs = (m.group(2) + "I" + m.group(3))[n + 1:]
        targs = targs.replace("DF16_", "_Float16,").replace("Li", "").replace("E", ",").strip(",")
        return f"{ident}<{targs}>("
    return name


def short(name: str) -> str:
    name = demangle_local(name)
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"\s*\[clone .*\]$", "", name)
    m = re.match(r"(?:void\s+)?([\w:]+)(<.*>)?\(", name)
    if m:
        t = m.group(2) or ""
        if len(t) > 60:
            t = t[:57] + "...>"
        return m.group(1) + t
    return name[:90]


def main(path):
    c = sqlite3.connect(path)
    cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else "kernel_name"
    rows = c.execute(f"select {name_col}, start, end from kernels").fetchall()
    agg = {}
    dm = demangle(sorted({r[0] for r in rows}))
    for n, s, e in rows:
        a = agg.setdefault(short(dm[n]), [0, 0, 1 << 62, 0])
        d = e - s
        a[0] += 1; a[1] += d; a[2] = min(a[2], d); a[3] = max(a[3], d)
    tot = sum(a[1] for a in agg.values())
    print("| kernel | calls | total ms | avg us | min us | max us | % GPU time |")
    print("|---|---:|---:|---:|---:|---:|---:|")
    for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print(f"| `{k}` | {a[0]} | {a[1] / 1e6:.3f} | {a[1] / a[0] / 1e3:.2f} | {a[2] / 1e3:.2f} | {a[3] / 1e3:.2f} | {100 * a[1] / tot:.2f} |")
    print(f"\ntotal kernel time {tot / 1e6:.3f} ms over {len(rows)} dispatches")
    timeline(rows)


def timeline(rows):
    """GPU occupancy over the densest part of the trace (the NFE loops): time with 0 / 1 / >= 2 kernels resident, the gap between
    consecutive kernels, and the share of wall time that is not covered by any kernel (launch latency, drain, cache write-back)."""
    ev = sorted((s, e) for _, s, e in rows)
    if len(ev) < 100:
        return
    # the trace has long idle stretches (python, weight upload): keep the dispatches whose predecessor ended < 200 us earlier
    segs, cur = [], [ev[0]]
    last_end = ev[0][1]
    for s, e in ev[1:]:
        if s - last_end > 200_000:
            segs.append(cur)
            cur = []
        cur.append((s, e))
        last_end = max(last_end, e)
    segs.append(cur)
    segs = [g for g in segs if len(g) >= 500]
    if not segs:
        return
    print("\n### timeline (bursts of >= 500 dispatches with no idle stretch > 200 us)\n")
    print("| burst | dispatches | wall ms | no kernel | 1 kernel | >= 2 kernels | median gap us (serial part) |")
    print("|---:|---:|---:|---:|---:|---:|---:|")
    for bi, g in enumerate(segs):
        pts = []
        for s, e in g:
            pts.append((s, 1)); pts.append((e, -1))
        pts.sort()
        depth, t_prev, occ = 0, pts[0][0], [0, 0, 0]
        for t, d in pts:
            occ[min(depth, 2)] += t - t_prev
            depth += d
            t_prev = t
        wall = pts[-1][0] - pts[0][0]
        gaps = sorted(max(0, g[i + 1][0] - g[i][1]) for i in range(len(g) - 1))
        print(f"| {bi} | {len(g)} | {wall / 1e6:.2f} | {100 * occ[0] / wall:.1f} % | {100 * occ[1] / wall:.1f} % | {100 * occ[2] / wall:.1f} % | {gaps[len(gaps) // 2] / 1e3:.2f} |")


if __name__ == "__main__":
    main(sys.argv[1])
