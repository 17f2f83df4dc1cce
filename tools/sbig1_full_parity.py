#!/usr/bin/env python3
"""BASELINE.json configs[2] (S-big1) at FULL size for the scaffold flag sets: 10^7 mappings in one chromosome pair through
swg_filter_device, status AND chain numbers of every record against the oracle's apply_filters (src/paf_filter.rs:379-747).
The reference's inversion capture (:535-597) loops over kept '+' chains x '-' mappings of the pair -- 1.4 * 10^6 x 10^6 here,
hours -- so this run (and only this run) switches the oracle to its indexed evaluation of that step, which
tests/test_oracle_fast_cpu.py holds against the literal loop; everything else is the literal restatement (the k = inf
sweep walks the whole active set per event, the chaining scan is O(n x window): ~10-15 min of one host thread for the
default flags, about a minute for the full flags; the two runs go on their own host threads at once).  Outside the pytest
budget; run through gpurun:

    python3 tools/sbig1_full_parity.py <tag>     ->  gpurun_out/<tag>_sbig1_full_size_parity_{default,full}.json
"""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N, CHR = 10_000_000, 248_956_422


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "rXX"
    n = int(sys.argv[2]) if len(sys.argv) > 2 else N
    cases = {"default": ("default", n, CHR), "full": ("full", n, CHR)}
    t0 = time.time()
    r = subprocess.run([sys.executable, "-m", "tests.sbig1_check", json.dumps(cases)], capture_output=True, text=True, cwd=ROOT,
                       env={**os.environ, "SBIG1_FAST_INVERSION": "1"})
    if r.returncode != 0:
        sys.stderr.write(r.stderr[-4000:])
        return r.returncode
    res = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    head = subprocess.run(["git", "rev-parse", "--short", "HEAD"], capture_output=True, text=True, cwd=ROOT).stdout.strip()
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    rc = 0
    for p in cases:
        d = dict(res[p], workload=f"BASELINE.json configs[2] (S-big1): {n} mappings, one pair, {CHR} bp, seed 1234", pipeline=p,
                 checker="oracle apply_filters (step 4b through its bucket index, tests/test_oracle_fast_cpu.py), one host thread per flag set", wall_s_both=round(time.time() - t0, 1), head=head or None)
        with open(os.path.join(ROOT, "gpurun_out", f"{tag}_sbig1_full_size_parity_{p}.json"), "w") as f:
            json.dump(d, f, indent=1)
        print(json.dumps(d))
        rc |= int(d["status_mismatches"] != 0 or d["chain_mismatches"] != 0)
    return rc


if __name__ == "__main__":
    sys.exit(main())
