#!/usr/bin/env python3
"""Shrinks a failing chaining case: isolates the (q,t,strand) group of the first differing record, then greedily
drops records while the GPU/oracle chain assignments still differ.  python tests/fuzz/fuzz_debug2.py <seed>"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "fuzz"))

import sweepga_amd as sw  # noqa: E402
from fuzz_gpu import random_case  # noqa: E402
from tests import gen, orc  # noqa: E402


def differs(sub, gap):
    got_of, _ = sw.merge_mappings_into_chains(gen.records_to_meta(sub), gap)
    want_of, _, _ = orc.merge_chains(sub, gap)
    return not np.array_equal(got_of, want_of), got_of, want_of


def take(rec, keep):
    return orc.Records([rec.qname[i] for i in keep], [rec.tname[i] for i in keep],
                       *(np.ascontiguousarray(getattr(rec, f)[keep]) for f in ("qs", "qe", "ts", "te", "block_length", "identity", "matches",
                                                                                "strand")), np.arange(len(keep), dtype=np.uint64))


seed = int(sys.argv[1])
rng = np.random.default_rng(seed)
rec, kw, keep_self, _ = random_case(rng, extras=False)
okw = {k: (int(v) if hasattr(v, "value") else (0 if v is None else v)) for k, v in kw.items()}
oc0 = orc.Config(**dict(okw, scaffold_gap=0))
oc0.keep_self = keep_self
ost0, _ = orc.apply_filters(oc0, rec)
sub = take(rec, np.nonzero(ost0)[0])
gap = okw["scaffold_gap"]
d, got_of, want_of = differs(sub, gap)
print("seed", seed, "survivors", len(sub), "gap", gap, "differs", d)
b = int(np.nonzero(got_of != want_of)[0][0])
grp = [i for i in range(len(sub)) if (sub.qname[i], sub.tname[i], sub.strand[i]) == (sub.qname[b], sub.tname[b], sub.strand[b])]
cur = take(sub, grp)
d, _, _ = differs(cur, gap)
print("group of first bad record:", len(cur), "records, differs alone:", d)
if d:
    idx = list(range(len(cur)))
    chunk = max(len(idx) // 2, 1)
    while chunk >= 1:
        i = 0
        while i < len(idx):
            trial = idx[:i] + idx[i + chunk:]
            if trial and differs(take(cur, trial), gap)[0]:
                idx = trial
            else:
                i += chunk
        chunk //= 2
    small = take(cur, idx)
    _, g, w = differs(small, gap)
    print("minimal:", len(small), "records; gap", gap)
    order = np.argsort(small.qs, kind="stable")
    for k in order:
        print("  idx", int(k), "q", int(small.qs[k]), int(small.qe[k]), "t", int(small.ts[k]), int(small.te[k]), chr(small.strand[k]), "got", int(g[k]), "want", int(w[k]))
