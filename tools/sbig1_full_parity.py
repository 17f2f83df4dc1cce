#!/usr/bin/env python3
"""BASELINE.json configs[2] (S-big1) at FULL size for the scaffold flag sets: 10^7 mappings in one chromosome pair through
swg_filter_device, status AND chain numbers of every record against the oracle's apply_filters (src/paf_filter.rs:379-747;
the chaining scan :784-851 is O(n x window): ~8 min of one host thread for the default flags, ~80 s for the full flags; the
two oracle runs go on their own host threads at once).  Outside the pytest budget; run through gpurun:

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
    r = subprocess.run([sys.executable, "-m", "tests.sbig1_check", json.dumps(cases)], capture_output=True, text=True, cwd=ROOT)
    if r.returncode != 0:
        sys.stderr.write(r.stderr[-4000:])
        return r.returncode
    res = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    head = subprocess.run(["git", "rev-parse", "--short", "HEAD"], capture_output=True, text=True, cwd=ROOT).stdout.strip()
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    rc = 0
    for p in cases:
        d = dict(res[p], workload=f"BASELINE.json configs[2] (S-big1): {n} mappings, one pair, {CHR} bp, seed 1234", pipeline=p,
                 checker="oracle apply_filters, one host thread per flag set", wall_s_both=round(time.time() - t0, 1), head=head or None)
        with open(os.path.join(ROOT, "gpurun_out", f"{tag}_sbig1_full_size_parity_{p}.json"), "w") as f:
            json.dump(d, f, indent=1)
        print(json.dumps(d))
        rc |= int(d["status_mismatches"] != 0 or d["chain_mismatches"] != 0)
    return rc


if __name__ == "__main__":
    sys.exit(main())
