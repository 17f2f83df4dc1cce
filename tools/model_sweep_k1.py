#!/usr/bin/env python3
"""Executable model of sweep_tile_k1 (csrc/swg_sweep.hip): the k = 1 tile algorithm with the spanning-best (S*) pruning
and the "top changed" rule, in plain Python, checked against the oracle's plane_sweep_query on random multi-segment inputs.
Tiles are tiny (TB = 8) and the candidate list small (CCAP = 4) so that every path (pruned, unpruned, overflow) is hit.
Development tool: run it after touching the kernel's logic.   python tools/model_sweep_k1.py [cases]"""
import sys
import os
import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import orc, gen  # noqa: E402

POS_BITS = 20


def overlap_exceeds(a_s, a_e, b_s, b_e, thr):
    os_, oe = max(a_s, b_s), min(a_e, b_e)
    ol = float(oe - os_) if oe > os_ else 0.0
    ml = float(min(a_e - a_s, b_e - b_s))
    if not ml > 0.0:
        return False
    return ol / ml > thr


def sweep_axis_model(seg, start, end, key, thr, TB=8, CCAP=6, STAR_MIN=3, stats=None):
    n = len(start)
    X = ((seg.astype(np.int64) + 1) << POS_BITS) | start.astype(np.int64)
    order = np.argsort(X, kind="stable")
    S = X[order]
    I = order
    E = ((seg[order].astype(np.int64) + 1) << POS_BITS) | end[order].astype(np.int64)
    K = key[order]
    ntiles = (n + TB - 1) // TB
    tile_x = [int(S[b * TB]) for b in range(ntiles)]
    # single-in-segment flags
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
        tb = p // TB
        if e > s:
            b = tb + 1
            while b < ntiles and tile_x[b] < e:
                carry[b].append(p)
                b += 1
    top = np.zeros(n, bool)
    ovl = np.zeros(n, bool)
    BIG = 1 << 62

    def prio(p):
        return (int(K[p]), int(S[p]), int(I[p]))

    for b in range(ntiles):
        own = list(range(b * TB, min(n, (b + 1) * TB)))
        x_b = tile_x[b]
        x_next = tile_x[b + 1] if b + 1 < ntiles else BIG
        cin = carry[b]
        # step A: best carry-in spanning the whole range
        span = [p for p in cin if int(E[p]) >= x_next]
        use_star = len(cin) >= STAR_MIN  # short lists are not pruned (lists longer than CCAP always are: CCAP >= STAR_MIN)
        star = min(span, key=prio) if (span and use_star) else None
        # step B: candidates
        def is_cand(p):
            return star is None or prio(p) < prio(star)
        co = [p for p in own if is_cand(p)]
        if star is None:
            cc_all = list(cin)  # no pruning: every carry-in, the spanning ones included
        else:
            cc_all = [p for p in cin if int(E[p]) < x_next and is_cand(p)]
        if len(cin) > CCAP and star is None:  # the long-list path of the kernel lists only the entries that end inside
            cc_all = [p for p in cin if int(E[p]) < x_next]
        cc_in_lds = len(cc_all) <= CCAP
        assert not (len(cin) > CCAP and star is None and span), "without S* no carry-in spans the tile"
        cc_complete = cc_in_lds and star is None   # then the list holds every carry-in
        pass1_carry = cc_all if cc_in_lds else cin  # overflow: scan every carry-in (a superset is harmless)
        if stats is not None:
            stats["tiles"] += 1
            stats["pruned"] += star is not None
            stats["overflow"] += not cc_in_lds
        # step C: points = (coordinate, the interval whose END it is or None for a start coordinate)
        pts = []
        for j, p in enumerate(own):
            xs = int(S[p])
            if xs != 0 and (j == len(own) - 1 or int(S[own[j + 1]]) != xs) and xs != x_next:
                pts.append((xs, None))
        for p in co:
            if int(S[p]) != 0 and int(E[p]) > int(S[p]) and int(E[p]) < x_next:
                pts.append((int(E[p]), p))
        for p in (cc_all if cc_in_lds else cin):
            if int(E[p]) < x_next:
                pts.append((int(E[p]), p))
        cands = pass1_carry + co   # S* is handled apart: it ranks below every candidate and never ends in the tile
        for px, ender in pts:
            if ender is not None:
                # an end coordinate acts only if its interval was the top just before x
                was_top = not any(int(S[p]) < px <= int(E[p]) and prio(p) < prio(ender) for p in cands)
                if not was_top:
                    continue
            T = star
            for p in cands:
                if int(S[p]) <= px < int(E[p]) and (T is None or prio(p) < prio(T)):
                    T = p
            if T is None:
                continue
            top[I[T]] = True
            if stats is not None:
                stats["points"] += 1
            if thr >= 1.0:
                continue
            need_full = ender is not None or int(S[T]) == px or px == x_b
            if need_full:
                targets = [p for p in own + cin if int(S[p]) <= px < int(E[p])]
                if stats is not None:
                    stats["full"] += 1
            else:
                targets = [p for p in own if int(S[p]) == px and int(E[p]) > px]
            for p in targets:
                if p != T and overlap_exceeds(int(S[p]), int(E[p]), int(S[T]), int(E[T]), thr):
                    ovl[I[p]] = True
    return single | (top & ~ovl)


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    rng = np.random.default_rng(5)
    stats = dict(tiles=0, pruned=0, overflow=0, full=0, points=0)
    bad = 0
    for c in range(cases):
        nseg = int(rng.integers(1, 4))
        n = int(rng.integers(2, 120))
        span = int(rng.choice([300, 3000, 30000]))
        max_len = int(rng.choice([40, 400, 4000]))
        levels = [0.8, 0.9, 0.95] if rng.random() < 0.5 else None
        qs, qe, ts, te, ident = gen.random_segment(rng, n, span=span, max_len=max_len, ident_levels=levels)
        seg = rng.integers(0, nseg, n)
        thr = float(rng.choice([0.0, 0.5, 0.95, 1.0]))
        scoring = int(rng.integers(0, 5))
        key = np.array([-orc.score(int(a), int(b), float(i), scoring) for a, b, i in zip(qs, qe, ident)])
        # order-preserving integer key: rank of -score (ties share a rank)
        uniq = np.unique(key)
        ikey = np.searchsorted(uniq, key)
        got = sweep_axis_model(seg, qs, qe, ikey, thr, stats=stats)
        want = np.zeros(n, bool)
        for s in range(nseg):
            idx = np.nonzero(seg == s)[0]
            if len(idx) == 0:
                continue
            kept = orc.plane_sweep(0, qs[idx], qe[idx], ts[idx], te[idx], ident[idx], k_q=1, thr=thr, scoring=scoring)
            want[idx[kept]] = True
        if not np.array_equal(got, want):
            bad += 1
            print("MISMATCH case", c, "n", n, "nseg", nseg, "thr", thr, "diff", np.nonzero(got != want)[0][:10])
    print("cases", cases, "mismatches", bad, stats)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
