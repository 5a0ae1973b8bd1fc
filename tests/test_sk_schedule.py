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


# ---------------------------------------------------------------------------------------------------------------------------------
# Protocol model of f5-tts_amd/csrc/gemm_skrs.h (variants 42 / 43): stream-K with the epilogue reduce-scattered over a tile's
# contributors.  Each contributor publishes its partial sums and finishes the accumulator units it owns (unit u of every lane belongs to
# contributor u % c) at the END of its share — the head owner right after its head (its last segment), the others after their whole
# share (their partners' sums only exist then).  tests/test_hipemu.py runs the kernel source itself on the CPU; this model
# checks, under random speeds and limited residency with in-order dispatch, that the protocol never deadlocks, finishes every unit
# exactly once from all c partial sums, and leaves every flag at zero.
# ---------------------------------------------------------------------------------------------------------------------------------
def contributors(sb, gx, KT, lt):
    """workgroup indices (within the class) whose share intersects local tile lt, in k order (all shares non-empty)."""
    lo, hi = lt * KT, (lt + 1) * KT
    return [i for i in range(gx) if sb(i) < hi and sb(i + 1) > lo]


def simulate_reduce_scatter(tiles, KT, gx, resident, seed):
    rng = random.Random(seed)
    ix = tiles * KT
    assert ix >= gx
    sb = lambda i: ix * i // gx  # noqa: E731
    UNITS = 16
    ready = {}         # (wg, slot) -> True once published
    readers_left = {}  # (wg, slot) -> external readers still to come
    finished = {}      # (tile, unit) -> list of contributor partials summed
    # build each workgroup's program: a list of steps
    progs = []
    for i in range(gx):
        steps, deferred = [], None
        it, ite = sb(i), sb(i + 1)
        while it < ite:
            lt, k0 = divmod(it, KT)
            n = min(KT - k0, ite - it)
            k1 = k0 + n
            cs = contributors(sb, gx, KT, lt)
            c, r = len(cs), cs.index(i)
            steps.append(("compute", n))
            if c > 1:
                slot = "B" if r == 0 else "A"
                n_owners = min(c, UNITS)
                steps.append(("publish", slot, n_owners - (1 if r < n_owners else 0)))
                mine = [q for q in range(UNITS) if q % c == r]
                job = ("finish", lt, mine, [(p, "B" if cs.index(p) == 0 else "A") for p in cs if p != i], cs)
                if r == 0:
                    steps.append(job)
                elif mine:
                    assert deferred is None
                    deferred = job
            else:
                steps.append(("finish", lt, list(range(UNITS)), [], cs))
            it += n
        if deferred:
            steps.append(deferred)
        progs.append(steps)
    # discrete-event run: in-order dispatch, at most `resident` workgroups alive
    pc = [0] * gx
    busy_until = [0.0] * gx
    alive, next_dispatch, done = [], 0, 0
    t = 0.0
    for _ in range(200000):
        while len(alive) < resident and next_dispatch < gx:
            alive.append(next_dispatch)
            next_dispatch += 1
        progressed = False
        for i in list(alive):
            if busy_until[i] > t:
                continue
            if pc[i] == len(progs[i]):
                alive.remove(i)
                done += 1
                progressed = True
                continue
            st = progs[i][pc[i]]
            if st[0] == "compute":
                busy_until[i] = t + st[1] * rng.uniform(0.5, 1.5)
                pc[i] += 1
                progressed = True
            elif st[0] == "publish":
                ready[(i, st[1])] = True
                readers_left[(i, st[1])] = st[2]
                pc[i] += 1
                progressed = True
            else:
                _, lt, mine, others, cs = st
                if all(ready.get(o, False) for o in others):
                    for q in mine:
                        assert (lt, q) not in finished
                        finished[(lt, q)] = sorted(cs)
                    if mine:
                        for o in others:
                            readers_left[o] -= 1
                            assert readers_left[o] >= 0
                            if readers_left[o] == 0:
                                ready[o] = False
                    busy_until[i] = t + rng.uniform(0.1, 1.0)
                    pc[i] += 1
                    progressed = True
        if done == gx:
            break
        if not progressed:
            pending = [b for b in busy_until if b > t]
            assert pending or next_dispatch < gx or False, "deadlock"
            if not pending:
                raise AssertionError(f"deadlock at t={t}: alive={alive} pcs={[pc[i] for i in alive]}")
            t = min(pending)
    assert done == gx
    assert len(finished) == tiles * UNITS and all(len(v) >= 1 for v in finished.values())
    for (lt, q), cs in finished.items():
        assert cs == sorted(contributors(sb, gx, KT, lt))
    assert not any(ready.values()), "a ready flag is left set"
    # owners that own nothing never read: their expected-reader counts must still reach zero
    assert all(v == 0 for v in readers_left.values())


@pytest.mark.parametrize("seed", range(30))
def test_reduce_scatter_protocol_is_deadlock_free_and_complete(seed):
    rng = random.Random(1000 + seed)
    gx = rng.randint(2, 32)
    KT = rng.randint(2, 64)
    tiles = rng.randint(max(1, (gx + KT - 1) // KT), 40)
    # Residency requirement of any scheme whose finisher waits for HIGHER-numbered workgroups (the head of a tile is computed last,
    # its tails first): the c contributors of a tile plus one must be able to be alive together, otherwise the workgroups that
    # are alive can all be waiting for one that cannot be dispatched (the model finds that deadlock at once).  The launcher therefore
    # sizes the grid to the CU count (every workgroup resident); other streams' kernels only delay dispatch, they retire on their own.
    ix = tiles * KT
    sb = lambda i: ix * i // gx  # noqa: E731
    c_max = max(len(contributors(sb, gx, KT, lt)) for lt in range(tiles))
    resident = rng.randint(min(gx, c_max + 1), gx)  # fewer slots than workgroups: dispatch in order as slots free up
    simulate_reduce_scatter(tiles, KT, gx, resident, seed)


def test_too_little_residency_deadlocks_in_the_model():
    """The hazard is real: 3 contributors, 2 resident workgroups."""
    with pytest.raises(AssertionError, match="deadlock"):
        simulate_reduce_scatter(tiles=1, KT=30, gx=3, resident=2, seed=0)
