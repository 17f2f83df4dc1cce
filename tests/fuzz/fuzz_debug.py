#!/usr/bin/env python3
"""Replays failing seeds of tests/fuzz/fuzz_gpu.py seam by seam: mapping sweep alone, merge_mappings_into_chains on the
sweep survivors, plane_sweep_scaffolds on the filtered chains, then the whole pipeline.
    python tests/fuzz/fuzz_debug.py 477 581 ..."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "fuzz"))

import sweepga_amd as sw  # noqa: E402
from fuzz_gpu import random_case  # noqa: E402
from tests import gen, orc  # noqa: E402


def sub_records(rec, keep):
    return orc.Records([rec.qname[i] for i in keep], [rec.tname[i] for i in keep],
                       *(np.ascontiguousarray(getattr(rec, f)[keep]) for f in ("qs", "qe", "ts", "te", "block_length", "identity", "matches",
                                                                                "strand")), np.arange(len(keep), dtype=np.uint64))


def main():
    for seed in map(int, sys.argv[1:]):
        rng = np.random.default_rng(seed)
        rec, kw, keep_self, scaffolds_only = random_case(rng, extras=False)
        okw = {k: (int(v) if hasattr(v, "value") else (0 if v is None else v)) for k, v in kw.items()}
        print("== seed", seed, "n", len(rec), "keep_self", keep_self, "scaffolds_only", scaffolds_only, okw)
        # 1. mapping sweep alone
        kw0, okw0 = dict(kw, scaffold_gap=0), dict(okw, scaffold_gap=0)
        f0 = sw.PafFilter(sw.FilterConfig(**kw0)).with_keep_self(keep_self)
        st0, _ = f0.filter_columns(sw.pack_records(gen.records_to_meta(rec)))
        oc0 = orc.Config(**okw0)
        oc0.keep_self = keep_self
        ost0, _ = orc.apply_filters(oc0, rec)
        print("  1. mapping sweep: differing records", int((st0 != ost0).sum()), "survivors", int((ost0 != 0).sum()))
        keep = np.nonzero(ost0)[0]
        if len(keep) == 0:
            continue
        sub = sub_records(rec, keep)
        # 2. chaining seam
        gap = okw["scaffold_gap"]
        got_of, got = sw.merge_mappings_into_chains(gen.records_to_meta(sub), gap)
        want_of, want_cols, want_wid = orc.merge_chains(sub, gap)
        d_of = int((got_of != want_of).sum())
        same_cols = all(np.array_equal(got[k], w) for k, w in zip(("query_start", "query_end", "target_start", "target_end"), want_cols[:4])) \
            if len(got["query_start"]) == len(want_cols[0]) else False
        same_wid = len(got["weighted_identity"]) == len(want_wid) and np.array_equal(got["weighted_identity"].view(np.uint64), want_wid.view(np.uint64))
        print("  2. merge_chains: chain_of diffs", d_of, "chains", len(want_cols[0]), "vs", len(got["query_start"]), "cols equal", same_cols,
              "identity bits equal", same_wid)
        if d_of:
            bad = np.nonzero(got_of != want_of)[0][:5]
            for b in bad:
                print("     rec", int(b), sub.qname[b], sub.tname[b], chr(sub.strand[b]), int(sub.qs[b]), int(sub.qe[b]), int(sub.ts[b]), int(sub.te[b]),
                      "got chain", int(got_of[b]), "want", int(want_of[b]))
        # 3. whole pipeline
        f = sw.PafFilter(sw.FilterConfig(**kw)).with_keep_self(keep_self).with_scaffolds_only(scaffolds_only)
        st, ch = f.filter_columns(sw.pack_records(gen.records_to_meta(rec)))
        oc = orc.Config(**okw)
        oc.keep_self, oc.scaffolds_only = keep_self, scaffolds_only
        ost, och = orc.apply_filters(oc, rec)
        bs, bc = np.nonzero(st != ost)[0], np.nonzero(ch != och)[0]
        print("  3. pipeline: bad status", len(bs), "bad chain", len(bc), "stats chains", f.last_stats.n_chains, "kept", f.last_stats.n_chains_kept)
        for b in bs[:5]:
            print("     rec", int(b), rec.qname[b], rec.tname[b], chr(rec.strand[b]), int(rec.qs[b]), int(rec.qe[b]), int(rec.ts[b]), int(rec.te[b]),
                  "status got", int(st[b]), "want", int(ost[b]), "chain got", int(ch[b]), "want", int(och[b]))
        if len(bc) and not len(bs):
            print("     chain partition equal:", orc.same_chain_partition(ch, och))


if __name__ == "__main__":
    main()
