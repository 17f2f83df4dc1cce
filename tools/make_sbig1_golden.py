#!/usr/bin/env python3
"""Oracle answers for BASELINE.json configs[2] at FULL size (10^7 mappings in one pair), computed on the CPU and committed
as fingerprints: tests/golden/sbig1_full_size.json.  No GPU involved; ~25 min of one core per flag set (the two run side by
side).  The reference's inversion capture (src/paf_filter.rs:535-597) loops over kept '+' chains x '-' mappings of the pair
(1.4 * 10^6 x 10^6 here, hours), so the oracle evaluates that one step through its bucket index (tests/test_oracle_fast_cpu.py
holds it against the literal loop); everything else is the literal restatement.

    python3 tools/make_sbig1_golden.py [n] [out.json]
"""
import json
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests import orc, sbig1_numpy  # noqa: E402

FLAGS = {
    "default": dict(),
    "full": dict(mapping_filter_mode=orc.ONE_TO_ONE, scaffold_filter_mode=orc.ONE_TO_ONE, scaffold_gap=50_000, min_scaffold_length=10_000,
                 scaffold_max_deviation=20_000),
    "sweep": dict(mapping_filter_mode=orc.ONE_TO_ONE, scaffold_gap=0),
}


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
    out = sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "tests", "golden", "sbig1_full_size.json")
    cols = sbig1_numpy.gen(n)
    orc.set_fast_inversion(True)
    res, secs = {}, {}

    def run(name):
        st, ch = np.zeros(n, np.uint8), np.zeros(n, np.uint32)
        t0 = time.time()
        orc.apply_filters_ids(orc.Config(**FLAGS[name]), cols, sbig1_numpy.NAMES, 0, n, st, ch)
        secs[name] = round(time.time() - t0, 1)
        res[name] = sbig1_numpy.fingerprint(st, ch)

    th = [threading.Thread(target=run, args=(k,)) for k in FLAGS]
    for t in th:
        t.start()
    for t in th:
        t.join()
    doc = {"workload": f"S-big1 (tests/sbig1_numpy.py, PCG64 seed 1234): {n} mappings, one pair, {sbig1_numpy.CHR_LEN} bp",
           "numpy": np.__version__, "oracle": "oracle/liboracle.so, apply_filters with the indexed inversion capture",
           "oracle_seconds": secs, "flags": {k: {a: int(b) for a, b in v.items()} for k, v in FLAGS.items()}, "expected": res}
    with open(out, "w") as f:
        json.dump(doc, f, indent=1)
    print(json.dumps(doc))


if __name__ == "__main__":
    main()
