"""Decode the wrong q|k|v outputs dumped by `tools/kernel_bench.py qkvprobe` (KB_PROBE_DUMP=prefix): where they are (sequence, head,
token, channel; which lanes and registers of the wave tile that is), and what the wrong value IS — the rope input reconstructed from the
reference pair is put through the candidates (no rotation, sign of the sine term lost, one term missing, (cos, sin) of another token).
usage: python tools/race_dump_analyze.py <dump.bin> <seqs> <nseq> [max_lines]"""
import math
import struct
import sys
from collections import Counter, defaultdict


def main():
    path, seqs, nseq = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
    max_lines = int(sys.argv[4]) if len(sys.argv) > 4 else 40
    H, dh = 16, 64
    ldv = (nseq + 7) & ~7
    rec = struct.Struct("<iiqffff")
    data = open(path, "rb").read()
    rows = [rec.unpack_from(data, o) for o in range(0, len(data) - rec.size + 1, rec.size)]
    print(f"{path}: {len(rows)} wrong outputs")
    print("per launch:", dict(sorted(Counter(r[0] for r in rows).items())))
    print("per plane (0 q, 1 k, 2 v^T):", dict(Counter(r[1] for r in rows)))
    where = Counter()
    kinds = Counter()
    deltas = Counter()
    lines = []
    for rep, plane, idx, ref, got, pref, pgot in rows:
        if plane >= 2:
            continue
        d = idx % dh
        tok = (idx // dh) % nseq
        bh = idx // (dh * nseq)
        m = (bh // H) * nseq + tok  # global row of the GEMM
        where[(f"row%32={'16-31' if m % 32 >= 16 else '0-15'}", f"row%64={'32-63' if m % 64 >= 32 else '0-31'}", f"d%8={d % 8}")] += 1
        sc = 0.125 if plane == 0 else 1.0
        ang = tok / (10000.0 ** ((d & ~1) / dh))
        c, s = math.cos(ang), math.sin(ang)
        # reference pair (a0, a1) = rot(x0, x1): invert
        a0, a1 = (ref, pref) if d % 2 == 0 else (pref, ref)
        x0, x1 = (a0 * c + a1 * s) / sc, (-a0 * s + a1 * c) / sc
        tol = 3e-3 * max(abs(ref), abs(pref), 1e-2)
        cand = {}
        if d % 2 == 0:
            cand = {"unrotated x0": x0, "sign of sin term lost": x0 * c + x1 * s, "cos term only": x0 * c, "sin term only": -x1 * s, "zero": 0.0,
                    "x1 unrotated": x1, "a1 (partner's value)": a1 / sc}
        else:
            cand = {"unrotated x1": x1, "sign of sin term lost": x1 * c - x0 * s, "cos term only": x1 * c, "sin term only": x0 * s, "zero": 0.0,
                    "x0 unrotated": x0, "a0 (partner's value)": a0 / sc}
        hit = [k for k, v in cand.items() if abs(v * sc - got) < tol]
        if not hit:
            # (cos, sin) of another token p': the same formula with p' in place of tok
            for p2 in range(nseq):
                a2 = p2 / (10000.0 ** ((d & ~1) / dh))
                v = (x0 * math.cos(a2) - x1 * math.sin(a2)) if d % 2 == 0 else (x1 * math.cos(a2) + x0 * math.sin(a2))
                if abs(v * sc - got) < tol:
                    deltas[p2 - tok] += 1
                    hit = ["(cos, sin) of token %+d" % (p2 - tok)]
                    break
        kinds[hit[0] if hit else "none of the candidates"] += 1
        if len(lines) < max_lines:
            lines.append(f"   launch {rep} plane {plane} seq {bh // H} head {bh % H} token {tok} (row {m}, %64 = {m % 64}) d {d}: ref {ref:+.5f} got {got:+.5f} "
                         f"partner ref {pref:+.5f} got {pgot:+.5f} x=({x0 * sc:+.5f},{x1 * sc:+.5f}) -> {hit}")
    print("where (GEMM row mod 32 / mod 64, channel mod 8):")
    for k, v in sorted(where.items()):
        print("   ", k, v)
    print("what the wrong value is:", dict(kinds.most_common()))
    print("token offsets of 'another token' matches:", dict(deltas.most_common(10)))
    print("\n".join(lines))


if __name__ == "__main__":
    main()
