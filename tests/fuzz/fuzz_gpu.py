#!/usr/bin/env python3
"""Differential fuzz of the whole filter on the GPU against the CPU oracle, beyond the fixed seeds of tests/:
random record sets (sizes 1 .. 60k, 1-6 genomes, sparse to very dense, ties and duplicates, zero-length and
zero-identity records, PanSN and plain names, coordinates against the 32-bit limit or moved beyond it per sequence) x random configurations (filter modes and k, overlap thresholds,
scorings, gaps, masses, deviations, identity / length cut-offs, self, scaffolds-only).  Exact equality of status
and chain numbers is required.  A failing case is written to gpurun_out/fuzz_fail_<seed>.json for replay.

    python tests/fuzz/fuzz_gpu.py --minutes 5 [--seed 0]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

import sweepga_amd as sw  # noqa: E402
from tests import gen, orc  # noqa: E402


def random_case(rng, extras=True):
    n = int(rng.choice([1, 2, 3, 17, 255, 256, 257, 1000, 4095, 4096, 4097, 9000, 20000, 60000],
                       p=[.03, .03, .03, .05, .05, .05, .05, .15, .05, .05, .05, .15, .16, .10]))
    span = int(rng.choice([2_000, 50_000, 400_000, 3_000_000]))
    max_len = int(rng.choice([300, 3_000, 20_000]))
    rec = gen.random_records(rng, n, n_genomes=int(rng.integers(1, 7)), chrs_per_genome=int(rng.integers(1, 4)), span=span,
                             max_len=min(max_len, max(span // 2, 60)), pansn=bool(rng.integers(0, 2)),
                             self_frac=float(rng.choice([0.0, 0.02, 0.3])), minus_frac=float(rng.choice([0.0, 0.2, 0.5, 1.0])),
                             syntenic_frac=float(rng.choice([0.0, 0.7, 0.98])), zero_frac=float(rng.choice([0.0, 0.01, 0.1])))
    if rng.random() < 0.3:  # heavy ties: few identity levels, coordinates on a coarse grid
        g = int(rng.choice([50, 500]))
        for a in (rec.qs, rec.qe, rec.ts, rec.te):
            a[:] = a // g * g
        rec.identity[:] = rng.choice([0.8, 0.9, 0.95], len(rec))
        rec.matches[:] = np.floor(rec.identity * rec.block_length).astype(np.uint64)
    if extras and rng.random() < 0.1:  # coordinates up against the 32-bit limit of the device layout
        off = np.uint64(2**32 - 1 - int(max(rec.qe.max(), rec.te.max())))
        for a in (rec.qs, rec.qe, rec.ts, rec.te):
            a += off
    if extras and rng.random() < 0.15:  # identity outliers (dv:f: can produce values outside [0, 1])
        k = rng.random(len(rec)) < 0.05
        rec.identity[k] = rng.choice([-0.5, 0.0, 1.0, 1.5], int(k.sum()))
    modes = [sw.FilterMode.OneToOne, sw.FilterMode.OneToMany, sw.FilterMode.ManyToMany]
    kq = rng.choice([None, 1, 2, 5])
    kt = rng.choice([None, 1, 3])
    kw = dict(
        mapping_filter_mode=modes[int(rng.integers(0, 3))],
        mapping_max_per_query=None if kq is None else int(kq), mapping_max_per_target=None if kt is None else int(kt),
        scaffold_filter_mode=modes[int(rng.integers(0, 3))],
        scaffold_max_per_query=None if rng.random() < 0.5 else int(rng.integers(1, 4)),
        scaffold_max_per_target=None if rng.random() < 0.5 else int(rng.integers(1, 4)),
        overlap_threshold=float(rng.choice([0.0, 0.3, 0.95, 1.0])),
        scaffold_gap=int(rng.choice([0, 1, 500, 10_000, 50_000, 10_000_000])),
        min_scaffold_length=int(rng.choice([0, 1_000, 10_000])),
        scaffold_overlap_threshold=float(rng.choice([0.0, 0.5, 1.0])),
        scaffold_max_deviation=int(rng.choice([0, 1, 2_000, 100_000])),
        scoring_function=sw.ScoringFunction(int(rng.integers(0, 5))),
        min_identity=float(rng.choice([0.0, 0.8, 0.97])),
        min_scaffold_identity=float(rng.choice([0.0, 0.85])),
        min_block_length=int(rng.choice([0, 100, 2_000])),
    )
    keep_self, scaffolds_only = bool(rng.random() < 0.3), bool(rng.random() < 0.15)
    # (drawn last, so that the cases of earlier campaigns keep their seeds) every sequence moved by a constant of its own, up to
    # 2^62: RecordMeta's u64 coordinates through swg_filter64 / swg_filter_multi64 and the per-sequence rebasing
    if extras and rng.random() < 0.12:
        rec, _ = gen.shifted(rec, rng)
    # --wide-gaps (drawn after everything else, off by default: earlier campaigns keep their cases): gap limits at the borders of
    # the deep candidate kernel's three loops (<= 46340: 32-bit distances; < 2^31: 64-bit; beyond: the generic loop, where a
    # limit of u64::MAX wraps like release Rust) -- never drawn by the list above
    if WIDE_GAPS and kw["scaffold_gap"] != 0 and rng.random() < 0.5:
        wide = [46_340, 46_341, 2**31 - 1, 2**31, 2**32 + 5, 2**63, 2**64 - 1, 2**22, 2**22 + 1]
        kw["scaffold_gap"] = wide[int(rng.integers(0, len(wide)))]  # (by index: rng.choice would round the values to f64)
    return rec, kw, keep_self, scaffolds_only


_CTXS = []
WIDE_GAPS = False


def contexts(k):
    while len(_CTXS) < k:
        _CTXS.append(sw.Context(0))
    return _CTXS[:k]


GROUPED = False   # --grouped: every case's records sorted by query genome (what an aligner writes), so that swg_filter /
                  # swg_filter_multi stream them in ranges when SWG_STREAM_CHUNK asks for small ones; and every third case
                  # hands the filter NO identity column where the records allow it (identity = matches / max(block, 1))


def group_by_query_genome(rec):
    import copy
    g = np.array([q.split("#")[0] if "#" in q else q for q in rec.qname])
    order = np.argsort(g, kind="stable")
    out = copy.copy(rec)
    for k in ("qs", "qe", "ts", "te", "block_length", "identity", "matches", "strand"):
        setattr(out, k, np.ascontiguousarray(getattr(rec, k)[order]))
    out.qname = [rec.qname[i] for i in order]
    out.tname = [rec.tname[i] for i in order]
    out.rank = np.arange(len(order), dtype=np.uint64)
    return out


def run_case(seed, extras=True):
    """extras=False reproduces the generator of the first campaign (the one that found the candidate-list tie bug)."""
    rng = np.random.default_rng(seed)
    rec, kw, keep_self, scaffolds_only = random_case(rng, extras)
    if GROUPED:
        rec = group_by_query_genome(rec)
    cfg = sw.FilterConfig(**kw)
    ocfg = orc.Config(**{k: (int(v) if hasattr(v, "value") else (0 if v is None else v)) for k, v in kw.items()})
    ocfg.keep_self, ocfg.scaffolds_only = keep_self, scaffolds_only
    f = sw.PafFilter(cfg).with_keep_self(keep_self).with_scaffolds_only(scaffolds_only)
    packed = sw.pack_records(gen.records_to_meta(rec))
    if GROUPED and seed % 3 == 0 and not packed.wide and np.array_equal(
            packed.cols["identity"], packed.cols["matches"] / np.maximum(packed.cols["block_len"], 1)):
        packed.cols["identity"] = None   # derived on the device
    status, chain = f.filter_columns(packed)
    ost, och = orc.apply_filters(ocfg, rec)
    ok = np.array_equal(status, ost) and np.array_equal(chain, och)
    if ok and seed % 8 == 0:  # the sharded entry point must give the same answer
        st2, ch2 = f.filter_columns_multi(packed, contexts(2 + seed % 3))
        ok = np.array_equal(st2, ost) and np.array_equal(ch2, och)
        status, chain = st2, ch2
    return ok, len(rec), kw, keep_self, scaffolds_only, int((status != ost).sum()), int((chain != och).sum())


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--minutes", type=float, default=5.0)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--grouped", action="store_true", help="records sorted by query genome; with SWG_STREAM_CHUNK=<small>: the streamed path")
    ap.add_argument("--wide-gaps", action="store_true", help="half of the scaffolding cases with a gap limit at 46340 / 46341 / 2^31 -+ 1 / "
                                                                "beyond 2^32 / u64::MAX (combine with SWG_CHAIN_DEEP=1)")
    args = ap.parse_args()
    global GROUPED, WIDE_GAPS
    GROUPED = args.grouped
    WIDE_GAPS = args.wide_gaps
    t0 = time.time()
    seed, cases, records, fails = args.seed, 0, 0, 0
    while time.time() - t0 < args.minutes * 60:
        ok, n, kw, keep_self, scaffolds_only, bs, bc = run_case(seed)
        cases += 1
        records += n
        if not ok:
            fails += 1
            os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
            with open(os.path.join(ROOT, "gpurun_out", f"fuzz_fail_{seed}.json"), "w") as fh:
                json.dump(dict(seed=seed, n=n, keep_self=keep_self, scaffolds_only=scaffolds_only, bad_status=bs, bad_chain=bc,
                               cfg={k: (int(v) if hasattr(v, "value") else v) for k, v in kw.items()}), fh)
            print("FAIL seed", seed, "n", n, "bad status", bs, "bad chain", bc, flush=True)
        seed += 1
    print(json.dumps(dict(cases=cases, records=records, failures=fails, first_seed=args.seed, next_seed=seed,
                          minutes=(time.time() - t0) / 60)))
    return 1 if fails else 0


if __name__ == "__main__":
    sys.exit(main())
