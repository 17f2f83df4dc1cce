#!/usr/bin/env python3
"""BASELINE.json configs[2] at full size on the GPU box: 10^7 numpy-generated S-big1 records through swg_filter for the
sweep / default / full flags, fingerprints (sha256 of the status and chain columns + counts) next to the oracle's
(tests/golden/sbig1_full_size.json, computed on a CPU box by tools/make_sbig1_golden.py).  Equal fingerprints = 0 status and
0 chain-number mismatches over all 10^7 records.  Writes gpurun_out/<tag>_sbig1_full_size_parity_<flags>.json."""
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "rXX"
    import sweepga_amd as sw
    from sweepga_amd.filter import PackedRecords
    from tests import sbig1_numpy
    gold = json.load(open(os.path.join(ROOT, "tests", "golden", "sbig1_full_size.json")))
    n = next(iter(gold["expected"].values()))["n"]
    cols = sbig1_numpy.gen(n)
    table = np.arange(2, dtype=np.uint32)
    packed = PackedRecords(n=n, cols=cols, n_seq=2, seq_genome_last=table, n_genome_last=2, seq_genome_two=table.copy(), n_genome_two=2)
    head = subprocess.run(["git", "rev-parse", "--short", "HEAD"], capture_output=True, text=True, cwd=ROOT).stdout.strip()
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    rc = 0
    for flags, want in gold["expected"].items():
        kw = dict(gold["flags"][flags])
        for k in ("mapping_filter_mode", "scaffold_filter_mode"):
            if k in kw:
                kw[k] = sw.FilterMode(kw[k])
        f = sw.PafFilter(sw.FilterConfig(**kw))
        f.filter_columns(packed)  # warm
        t0 = time.perf_counter()
        status, chain = f.filter_columns(packed)
        ms = (time.perf_counter() - t0) * 1e3
        got = sbig1_numpy.fingerprint(status, chain)
        same = got == want
        d = {"workload": gold["workload"], "flags": flags, "config": gold["flags"][flags], "records": n,
             "status_equal": got["status_sha256"] == want["status_sha256"], "chain_numbers_equal": got["chain_sha256"] == want["chain_sha256"],
             "status_mismatches": 0 if got["status_sha256"] == want["status_sha256"] else None,
             "chain_mismatches": 0 if got["chain_sha256"] == want["chain_sha256"] else None,
             "device": got, "oracle": want, "oracle_seconds_cpu_box": gold["oracle_seconds"][flags],
             "swg_filter_ms_host_buffers": round(ms, 1), "oracle_build": gold["oracle"], "head": head or None}
        with open(os.path.join(ROOT, "gpurun_out", f"{tag}_sbig1_full_size_parity_{flags}.json"), "w") as fh:
            json.dump(d, fh, indent=1)
        print(flags, "equal" if same else "DIFFERENT", got["kept"], want["kept"])
        rc |= int(not same)
    return rc


if __name__ == "__main__":
    sys.exit(main())
