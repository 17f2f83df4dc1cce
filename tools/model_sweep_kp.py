#!/usr/bin/env python3
"""Executable model of sweep_tile_kp + sweep_tile_kn (csrc/swg_sweep.hip): the 2 <= k tile algorithm -- the k best spanning
carry-ins (stars) as a threshold, candidates, end points that act only when their interval was a member, overlap passes
only where the member set changes -- in plain Python, checked against the oracle's plane_sweep_query on random multi-segment
inputs.  Tiles are tiny (TB = 8) and the candidate list small (CCAP = 5) so that every path (threshold / no threshold,
list overflow -> plain evaluation of the tile, k above KSTAR_MAX) is hit.
Development tool: run it after touching the kernels' logic.   python tools/model_sweep_kp.py [cases]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import gen, orc  # noqa: E402

POS_BITS = 20


def overlap_exceeds(a_s, a_e, b_s, b_e, thr):
    os_, oe = max(a_s, b_s), min(a_e, b_e)
    ol = float(oe - os_) if oe > os_ else 0.0
    ml = float(min(a_e - a_s, b_e - b_s))
    if not ml > 0.0:
        return False
    return ol / ml > thr


def sweep_axis_model(seg, start, end, key, k, thr, TB=8, CCAP=5, KSTAR_MAX=4, stats=None):
    n = len(start)
    X = ((seg.astype(np.int64) + 1) << POS_BITS) | start.astype(np.int64)
    order = np.argsort(X, kind="stable")
    S = X[order]
    I = order
    E = ((seg[order].astype(np.int64) + 1) << POS_BITS) | end[order].astype(np.int64)
    K = key[order]
    ntiles = (n + TB - 1) // TB
    tile_x = [int(S[b * TB]) for b in range(ntiles)]
    single = np.zeros(n, bool)
    sg = S >> POS_BITS
    for p in range(n):
        prev_same = p > 0 and sg[p - 1] == sg[p]
        next_same = p + 1 < n and sg[p + 1] == sg[p]
        if not prev_same and not next_same:
            single[I[p]] = True
    carry = [[] for _ in range(ntiles)]
    for p in range(n):
        s, e = int(S[p]), int(E[p])
        if e > s:
            b = p // TB + 1
            while b < ntiles and tile_x[b] < e:
                carry[b].append(p)
                b += 1
    top = np.zeros(n, bool)
    ovl = np.zeros(n, bool)
    BIG = 1 << 62

    def prio(p):
        return (int(K[p]), int(S[p]), int(I[p]))

    def active(p, px):
        return int(S[p]) <= px < int(E[p])

    def plain_tile(own, cin, x_next):
        """sweep_tile_kn: every start coordinate (last begin of its run, unless the run continues) and every end coordinate
        inside the tile's range; T(x) = the k best actives; non-members against every member when more than k are active."""
        pts = []
        for j, p in enumerate(own):
            xs = int(S[p])
            if xs != 0 and (j == len(own) - 1 or int(S[own[j + 1]]) != xs) and xs != x_next:
                pts.append(xs)
        for p in own:
            if int(S[p]) != 0 and int(E[p]) > int(S[p]) and int(E[p]) < x_next:
                pts.append(int(E[p]))
        for p in cin:
            if int(E[p]) < x_next:
                pts.append(int(E[p]))
        for px in pts:
            act = sorted((p for p in own + cin if active(p, px)), key=prio)
            members = act[:k]
            for m in members:
                top[I[m]] = True
            if len(act) > k and thr < 1.0:
                for y in act[k:]:
                    for m in members:
                        if overlap_exceeds(int(S[y]), int(E[y]), int(S[m]), int(E[m]), thr):
                            ovl[I[y]] = True

    for b in range(ntiles):
        own = [p for p in range(b * TB, min(n, (b + 1) * TB))]
        x_b = tile_x[b]
        x_next = tile_x[b + 1] if b + 1 < ntiles else BIG
        cin = carry[b]
        if stats is not None:
            stats["tiles"] += 1
        if k > KSTAR_MAX:
            plain_tile(own, cin, x_next)
            stats and stats.__setitem__("plain", stats["plain"] + 1)
            continue
        span = sorted((p for p in cin if int(E[p]) >= x_next), key=prio)
        stars = span[:k]
        have_thr = len(stars) == k
        thr_p = stars[-1] if have_thr else None

        def is_cand(p):
            return thr_p is None or prio(p) < prio(thr_p)
        cc = [p for p in cin if int(E[p]) < x_next and is_cand(p)]
        if len(cc) > CCAP:  # more candidates than the LDS list holds: the plain kernel takes the tile
            plain_tile(own, cin, x_next)
            stats and stats.__setitem__("overflow", stats["overflow"] + 1)
            continue
        if stats is not None:
            stats["thr"] += have_thr
        co = [p for p in own if int(S[p]) != 0 and is_cand(p)]
        pts = []
        for j, p in enumerate(own):
            xs = int(S[p])
            if xs != 0 and (j == len(own) - 1 or int(S[own[j + 1]]) != xs) and xs != x_next:
                pts.append((xs, None))
        for p in co:
            if int(E[p]) > int(S[p]) and int(E[p]) < x_next:
                pts.append((int(E[p]), p))
        for p in cc:
            pts.append((int(E[p]), p))
        for px, ender in pts:
            if ender is not None:
                # the ending interval was a member just before x iff fewer than k stars / candidates with s < x <= e rank above it
                above = sum(prio(s_) < prio(ender) for s_ in stars)
                above += sum(int(S[p]) < px <= int(E[p]) and prio(p) < prio(ender) for p in co + cc)
                if above >= k:
                    continue
            act = sorted(set(stars) | {p for p in co + cc if active(p, px)}, key=prio)
            members = act[:k]
            for m in members:
                top[I[m]] = True
            if stats is not None:
                stats["points"] += 1
            if thr >= 1.0 or len(members) < k:
                continue
            tau = members[-1]
            full = ender is not None or any(int(S[m]) == px for m in members) or px == x_b
            if full:
                targets = [p for p in own + cin if active(p, px) and prio(tau) < prio(p)]
                if stats is not None:
                    stats["full"] += 1
            else:
                targets = [p for p in own if int(S[p]) == px and int(E[p]) > px and prio(tau) < prio(p)]
            for y in targets:
                for m in members:
                    if overlap_exceeds(int(S[y]), int(E[y]), int(S[m]), int(E[m]), thr):
                        ovl[I[y]] = True
    return single | (top & ~ovl)


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 400
    rng = np.random.default_rng(11)
    stats = dict(tiles=0, thr=0, overflow=0, plain=0, full=0, points=0)
    bad = 0
    for c in range(cases):
        nseg = int(rng.integers(1, 4))
        n = int(rng.integers(2, 140))
        span = int(rng.choice([300, 3000, 30000]))
        max_len = int(rng.choice([40, 400, 4000]))
        levels = [0.8, 0.9, 0.95] if rng.random() < 0.5 else None
        qs, qe, ts, te, ident = gen.random_segment(rng, n, span=span, max_len=max_len, ident_levels=levels)
        seg = rng.integers(0, nseg, n)
        thr = float(rng.choice([0.0, 0.5, 0.95, 1.0]))
        k = int(rng.choice([2, 2, 3, 4, 5]))
        scoring = int(rng.integers(0, 5))
        key = np.array([-orc.score(int(a), int(b), float(i), scoring) for a, b, i in zip(qs, qe, ident)])
        ikey = np.searchsorted(np.unique(key), key)
        got = sweep_axis_model(seg, qs, qe, ikey, k, thr, stats=stats)
        want = np.zeros(n, bool)
        for s in range(nseg):
            idx = np.nonzero(seg == s)[0]
            if len(idx) == 0:
                continue
            kept = orc.plane_sweep(0, qs[idx], qe[idx], ts[idx], te[idx], ident[idx], k_q=k, thr=thr, scoring=scoring)
            want[idx[kept]] = True
        if not np.array_equal(got, want):
            bad += 1
            print("MISMATCH case", c, "n", n, "nseg", nseg, "k", k, "thr", thr, "diff", np.nonzero(got != want)[0][:10])
    print("cases", cases, "mismatches", bad, stats)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
