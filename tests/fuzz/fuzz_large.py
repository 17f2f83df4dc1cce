#!/usr/bin/env python3
"""Million-record shapes the bench workload does not have, GPU vs oracle (exact status and chain numbers):
single giant chromosome pair at high depth, many chromosomes per genome (segments spanning several target sequences),
thousands of tiny genome pairs, heavy coordinate ties, minus-strand only.  Oracle runs are spread over host threads.
--pair-major: the same shapes with the records grouped by (query, target) pair, pairs in random order -- what an aligner writes:
inputs of this size then take the pair-resident scaffold stage through the runs of the input (csrc/swg_pair.hip; without the
flag the records come in random order, which is the global-sort stage's case), and the launch table must show that it ran.
    python tests/fuzz/fuzz_large.py [--records 1500000] [--pair-major]"""
import argparse
import json
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

import sweepga_amd as sw  # noqa: E402
from tests import gen, orc  # noqa: E402

SHAPES = [
    dict(name="giant_pair_deep", n_genomes=2, chrs_per_genome=1, span=3_000_000, max_len=20_000, syntenic_frac=0.9, scale=0.4),
    dict(name="many_chromosomes", n_genomes=6, chrs_per_genome=12, span=2_000_000, max_len=8_000, syntenic_frac=0.7, scale=1.0),
    dict(name="tiny_pairs", n_genomes=120, chrs_per_genome=1, span=300_000, max_len=5_000, syntenic_frac=0.8, scale=1.0),
    dict(name="ties_grid", n_genomes=4, chrs_per_genome=3, span=400_000, max_len=6_000, syntenic_frac=0.7, scale=0.6, grid=500),
    dict(name="minus_only", n_genomes=5, chrs_per_genome=2, span=5_000_000, max_len=10_000, syntenic_frac=0.9, scale=1.0, minus_frac=1.0),
    dict(name="non_pansn_names", n_genomes=8, chrs_per_genome=3, span=1_000_000, max_len=6_000, syntenic_frac=0.6, scale=1.0, pansn=False),
    # ~39,800 pairs of ~25 records: far more pairs than the pair-resident path takes (its rule: pair_plan, swg_pair.hip)
    dict(name="tiny_pairs_40k", n_genomes=200, chrs_per_genome=1, span=200_000, max_len=4_000, syntenic_frac=0.8, scale=0.66),
]
CONFIGS = [
    ("sweep", dict(mapping_filter_mode="OneToOne", scaffold_gap=0)),
    ("full", dict(mapping_filter_mode="OneToOne", scaffold_filter_mode="OneToOne", scaffold_max_deviation=20_000)),
    ("default", dict()),
    ("k2_gap5k", dict(mapping_filter_mode="OneToMany", mapping_max_per_query=2, scaffold_gap=5_000, min_scaffold_length=2_000,
                      scaffold_max_deviation=3_000, overlap_threshold=0.5)),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--records", type=int, default=1_500_000)
    ap.add_argument("--seed", type=int, default=7)
    ap.add_argument("--pair-major", action="store_true")
    args = ap.parse_args()
    results, lock = [], threading.Lock()
    jobs = []
    t_all = time.time()
    for si, shape in enumerate(SHAPES):
        rng = np.random.default_rng(args.seed + si)
        n = int(args.records * shape["scale"])
        rec = gen.random_records(rng, n, n_genomes=shape["n_genomes"], chrs_per_genome=shape["chrs_per_genome"], span=shape["span"],
                                 max_len=shape["max_len"], syntenic_frac=shape["syntenic_frac"], pansn=shape.get("pansn", True),
                                 minus_frac=shape.get("minus_frac", 0.2))
        if "grid" in shape:
            for a in (rec.qs, rec.qe, rec.ts, rec.te):
                a[:] = a // shape["grid"] * shape["grid"]
        if args.pair_major:
            from tests.test_gpu_pairs import pair_major
            rec = pair_major(rec, rng)
        packed = sw.pack_records(gen.records_to_meta(rec))
        # (a context of its own per shape: a context remembers what the device handed back on calls of about this size -- deep pairs,
        # heavy ties -- and the next shape is of the same size)
        ctx = sw.Context(0)
        for cname, kw in CONFIGS:
            kwg = {k: (getattr(sw.FilterMode, v) if isinstance(v, str) else v) for k, v in kw.items()}
            f = sw.PafFilter(sw.FilterConfig(**kwg), ctx=ctx)
            t0 = time.perf_counter()
            st, ch = f.filter_columns(packed)
            gpu_s = time.perf_counter() - t0
            if args.pair_major and kw.get("scaffold_gap", 1) != 0 and os.environ.get("SWG_GROUP_FUSED", "1") != "0":
                ctx.profile_reset()
                ctx.profile(True)
                f.filter_columns(packed)
                ctx.profile(False)
                table = ctx.profile_table()
                took = "pair_renumber" in table and "chain_cuts" not in table and "cuts_from_scan" not in table
                print("   path:", shape["name"], cname, "pair-resident" if took else "global-sort (a pair beyond the largest size class, or a condition met on the device)", flush=True)
                # (left to the global-sort stage by rule or on the device: deep pairs, heavy ties, thousands of tiny pairs)
                assert took or shape["name"] in ("giant_pair_deep", "ties_grid", "tiny_pairs", "tiny_pairs_40k"), (shape["name"], cname, sorted(table))
            okw = {k: (int(getattr(sw.FilterMode, v)) if isinstance(v, str) else v) for k, v in kw.items()}
            jobs.append((shape["name"], cname, rec, orc.Config(**okw), st.copy(), ch.copy(), gpu_s))
        ctx.close()
        print("generated + filtered", shape["name"], n, flush=True)

    def worker():
        while True:
            with lock:
                if not jobs:
                    return
                sname, cname, rec, ocfg, st, ch, gpu_s = jobs.pop()
            t0 = time.perf_counter()
            ost, och = orc.apply_filters(ocfg, rec)
            cpu_s = time.perf_counter() - t0
            with lock:
                results.append(dict(shape=sname, config=cname, n=len(rec), kept=int((ost != 0).sum()), status_equal=bool(np.array_equal(st, ost)),
                                    chain_equal=bool(np.array_equal(ch, och)), gpu_s=round(gpu_s, 3), oracle_s=round(cpu_s, 1)))
                print(results[-1], flush=True)

    th = [threading.Thread(target=worker) for _ in range(min(len(jobs), max(1, (os.cpu_count() or 1) // 2), 24))]
    for t in th:
        t.start()
    for t in th:
        t.join()
    bad = [r for r in results if not (r["status_equal"] and r["chain_equal"])]
    print(json.dumps(dict(cases=len(results), failures=len(bad), minutes=(time.time() - t_all) / 60)))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
