"""CPU model of the stream-K work partition of f5-tts_amd/csrc/gemm_sk.h (variants 40/41 of the GEMM microbenchmark): the same
integer arithmetic as the kernel, checked for the invariants the kernel's synchronisation relies on."""
import random

import pytest


def schedule(tiles, KT, G):
    """-> per workgroup b: list of segments (tile, k0, k1, role, waits) exactly as gemm_sk_kernel walks them."""
    assert G % 8 == 0
    gx = G // 8
    out = {}
    for b in range(G):
        xcd, wi = b & 7, b >> 3
        tlo, thi = xcd * tiles // 8, (xcd + 1) * tiles // 8
        ix = (thi - tlo) * KT
        sb = lambda i: ix * i // gx  # noqa: E731  share_begin
        itb, ite = sb(wi), sb(wi + 1)
        segs = []
        it = itb
        while it < ite:
            lt, k0 = divmod(it, KT)
            n = min(KT - k0, ite - it)
            k1 = k0 + n
            if k0 > 0:
                role, waits = "producer", []
            else:
                role, waits = "finisher", []
                if k1 < KT:
                    tile_end = (lt + 1) * KT
                    for p in range(wi + 1, gx):
                        pbeg, pend = sb(p), sb(p + 1)
                        if pbeg >= tile_end:
                            break
                        if pbeg == pend:
                            continue
                        waits.append(xcd + 8 * p)
                        if pend >= tile_end:
                            break
            segs.append((tlo + lt, k0, k1, role, waits))
            it += n
        out[b] = segs
    return out


@pytest.mark.parametrize("tiles,KT,G", [(88, 32, 256), (88, 64, 256), (176, 32, 256), (264, 32, 256), (264, 32, 128), (5, 7, 8), (3, 100, 64),
                                        (1000, 3, 256), (7, 1, 256)] + [(random.Random(i).randint(1, 400), random.Random(i + 99).randint(1, 70),
                                                                         8 * random.Random(i + 7).randint(1, 40)) for i in range(40)])
def test_partition_invariants(tiles, KT, G):
    sch = schedule(tiles, KT, G)
    cover = {}
    producers = {}
    finishers = {}
    for b, segs in sch.items():
        for si, (t, k0, k1, role, waits) in enumerate(segs):
            assert 0 <= k0 < k1 <= KT
            for k in range(k0, k1):
                assert (t, k) not in cover, "an iteration is computed twice"
                cover[(t, k)] = b
            if role == "producer":
                assert si == 0, "a partial is only ever the FIRST segment of a share (published early)"
                producers.setdefault(t, []).append(b)
            else:
                assert t not in finishers, "two workgroups finish the same tile"
                finishers[t] = (b, waits)
    assert len(cover) == tiles * KT, "every (tile, k-tile) is computed exactly once"
    assert set(finishers) == set(range(tiles)), "every tile has a finisher"
    # balance: shares inside an XCD class differ by at most one iteration
    for x in range(8):
        sizes = [sum(k1 - k0 for _, k0, k1, _, _ in sch[b]) for b in range(x, G, 8)]
        assert max(sizes) - min(sizes) <= 1
    for t, (b, waits) in finishers.items():
        assert sorted(waits) == sorted(producers.get(t, [])), "the finisher waits for exactly the workgroups that published a part of its tile"
        assert all(w > b and (w & 7) == (b & 7) for w in waits), "same XCD class, higher-numbered"
    # every producer slot is consumed by exactly one finisher (flags return to 0)
    consumed = sorted(w for _, waits in finishers.values() for w in waits)
    assert consumed == sorted(b for bs in producers.values() for b in bs) and len(set(consumed)) == len(consumed)
