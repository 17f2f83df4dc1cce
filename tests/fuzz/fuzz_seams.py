#!/usr/bin/env python3
"""Differential fuzz of the lower public seams on the GPU against the oracle:
  swg_plane_sweep (query / target / both; k from 1 to huge and infinity; any threshold and scoring; deep single
  segments with ties, duplicates, zero-length and zero-identity records), swg_plane_sweep_scaffolds and
  swg_merge_chains (dense groups, tiny to huge gaps, both strands).
    python tests/fuzz/fuzz_seams.py --minutes 5"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

import sweepga_amd as sw  # noqa: E402
from tests import gen, orc  # noqa: E402


def sweep_case(rng, ctx):
    n = int(rng.choice([1, 2, 3, 17, 64, 255, 256, 257, 600, 1500, 4000, 12000]))
    span = int(rng.choice([50, 500, 5_000, 100_000, 4_000_000_000]))
    levels = [0.9, 0.95] if rng.random() < 0.4 else None
    qs, qe, ts, te, ident = gen.random_segment(rng, n, span=span, max_len=max(2, min(span // 4, 2_000_000)), ident_levels=levels,
                                               zero_frac=float(rng.choice([0.0, 0.03, 0.3])), dup_frac=float(rng.choice([0.0, 0.1, 0.5])))
    if rng.random() < 0.3:
        g = max(1, span // 40)
        for a in (qs, qe, ts, te):
            a[:] = a // g * g
    bad = []
    for _ in range(6):
        ks = [1, 1, 2, 3, 7, 50, 1000, orc.K_INF]
        k = ks[int(rng.integers(0, len(ks)))]
        kt = [k, 1, 4, orc.K_INF][int(rng.integers(0, 4))]
        thr = float(rng.choice([0.0, 0.3, 0.5, 0.95, 0.999, 1.0]))
        scoring = int(rng.integers(0, 5))
        axis = int(rng.integers(0, 3))
        want = np.zeros(n, dtype=np.uint8)
        want[orc.plane_sweep(axis, qs, qe, ts, te, ident, k_q=k, k_t=kt, thr=thr, scoring=scoring)] = 1
        got = np.zeros(n, dtype=np.uint8)
        ctx.check(ctx.lib.swg_plane_sweep(ctx.handle, axis, n, *(a.ctypes.data_as(C.c_void_p) for a in (qs, qe, ts, te, ident)),
                                          k, kt, thr, scoring, got.ctypes.data_as(C.c_void_p)))
        if not np.array_equal(got, want):
            bad.append(dict(kind="sweep", n=n, k=min(k, 10**9), kt=min(kt, 10**9), thr=thr, scoring=scoring, axis=axis,
                            nbad=int((got != want).sum())))
    return bad


def chains_case(rng):
    n = int(rng.choice([2, 50, 800, 5000, 20000]))
    span = int(rng.choice([2_000, 100_000, 3_000_000]))
    rec = gen.random_records(rng, n, n_genomes=int(rng.integers(2, 4)), chrs_per_genome=int(rng.integers(1, 3)), span=span,
                             max_len=int(rng.choice([200, 5000])), self_frac=0.0, syntenic_frac=float(rng.choice([0.5, 0.95])),
                             minus_frac=float(rng.choice([0.0, 0.5])), zero_frac=float(rng.choice([0.0, 0.05])))
    if rng.random() < 0.4:
        g = int(rng.choice([50, 500]))
        for a in (rec.qs, rec.qe, rec.ts, rec.te):
            a[:] = a // g * g
    gap = int(rng.choice([0, 1, 100, 5_000, 50_000, 10_000_000]))
    if gap == 0:
        gap = 1
    got_of, got = sw.merge_mappings_into_chains(gen.records_to_meta(rec), gap)
    want_of, want_cols, want_wid = orc.merge_chains(rec, gap)
    ok = np.array_equal(got_of, want_of) and len(got["query_start"]) == len(want_cols[0])
    ok = ok and all(np.array_equal(got[k], w) for k, w in zip(("query_start", "query_end", "target_start", "target_end"), want_cols[:4]))
    ok = ok and np.array_equal(got["weighted_identity"].view(np.uint64), want_wid.view(np.uint64))
    bad = [] if ok else [dict(kind="chains", n=len(rec), gap=gap, ndiff=int((got_of != want_of).sum()))]
    # scaffolds seam on the chains the oracle built
    nc = len(want_cols[0])
    if nc:
        head = {}
        for i, c in enumerate(want_of):
            head.setdefault(int(c), i)
        chains = [(rec.qname[head[c]], rec.tname[head[c]], int(want_cols[0][c]), int(want_cols[1][c]), int(want_cols[2][c]),
                   int(want_cols[3][c]), float(want_wid[c])) for c in range(nc)]
        mode = int(rng.integers(0, 3))
        mq = None if rng.random() < 0.5 else int(rng.integers(1, 4))
        mt = None if rng.random() < 0.5 else int(rng.integers(1, 4))
        thr = float(rng.choice([0.0, 0.5, 0.95, 1.0]))
        scoring = int(rng.integers(0, 5))
        got_k = sw.plane_sweep_scaffolds(chains, sw.FilterMode(mode), mq, mt, thr, sw.ScoringFunction(scoring))
        want_k = orc.plane_sweep_scaffolds(chains, mode, mq or 0, mt or 0, thr, scoring)
        if list(got_k) != list(want_k):
            bad.append(dict(kind="scaffolds", nc=nc, mode=mode, mq=mq, mt=mt, thr=thr, scoring=scoring))
    return bad


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--minutes", type=float, default=5.0)
    ap.add_argument("--seed", type=int, default=0)
    args = ap.parse_args()
    ctx = sw.default_context(0)
    t0, seed, cases, fails = time.time(), args.seed, 0, 0
    while time.time() - t0 < args.minutes * 60:
        rng = np.random.default_rng(seed)
        bad = sweep_case(rng, ctx) if seed % 2 == 0 else chains_case(rng)
        cases += 1
        if bad:
            fails += 1
            os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
            with open(os.path.join(ROOT, "gpurun_out", f"fuzz_seam_fail_{seed}.json"), "w") as fh:
                json.dump(bad, fh)
            print("FAIL seed", seed, bad[:2], flush=True)
        seed += 1
    print(json.dumps(dict(cases=cases, failures=fails, next_seed=seed)))
    return 1 if fails else 0


if __name__ == "__main__":
    sys.exit(main())
