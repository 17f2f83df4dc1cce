#!/usr/bin/env python3
"""Differential fuzz of bin/alnstats against oracle/alnstats-ref (CPU only): random PAFs with hostile numeric columns, odd
names, CRLF, missing final newlines; one-file (-d) and two-file modes; stdout and exit status must match.

    python tests/fuzz/fuzz_alnstats.py --minutes 2 [--seed 0]
"""
import argparse
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

NUMS = ["0", "1", "7", "100", "4294967295", "4294967296", "+5", "-3", "", " 4", "1e3", "007", "18446744073709551615",
        "18446744073709551616", "abc"]
NAMES = ["g1#1#a", "g1#1#b", "g2#1#a", "g2#2#a", "plain", "x#y", "#", "a##", "café#1#z", "g1#1#a "]


def random_paf(rng):
    n = int(rng.choice([0, 1, 3, 40, 600]))
    hostile = rng.random() < 0.25
    out = []
    for _ in range(n):
        q, t = NAMES[rng.integers(0, len(NAMES))], NAMES[rng.integers(0, len(NAMES))]
        ql = int(rng.choice([0, 1000, 50_000]))
        qs = int(rng.integers(0, 40_000))
        qe = qs + int(rng.integers(0, 3000))
        f = [q, str(ql), str(qs), str(qe), "+-"[rng.integers(0, 2)], t, str(int(rng.choice([0, 1000, 80_000]))), "0", "10",
             str(int(rng.integers(0, 3000))), str(int(rng.integers(1, 3000))), "60"]
        if hostile and rng.random() < 0.1:
            f[int(rng.choice([1, 2, 3, 6, 9, 10]))] = NUMS[rng.integers(0, len(NUMS))]
        if rng.random() < 0.05:
            f = f[:int(rng.integers(0, 11))]
        if rng.random() < 0.3:
            f.append("tp:A:P")
        out.append("\t".join(f))
    nl = "\r\n" if rng.random() < 0.2 else "\n"
    return nl.join(out) + (nl if rng.random() < 0.8 else "")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--minutes", type=float, default=2.0)
    ap.add_argument("--seed", type=int, default=0)
    args = ap.parse_args()
    from sweepga_amd import build
    exe = build.build_alnstats()
    ref = os.path.join(ROOT, "oracle", "alnstats-ref")
    tmp = tempfile.mkdtemp(prefix="fuzz_alnstats_")
    a, b = os.path.join(tmp, "a.paf"), os.path.join(tmp, "b.paf")
    t0, seed, cases, fails = time.time(), args.seed, 0, 0
    while time.time() - t0 < args.minutes * 60:
        rng = np.random.default_rng(seed)
        for path in (a, b):
            with open(path, "w", newline="", encoding="utf-8") as fh:
                fh.write(random_paf(rng))
        for argv in ([a], [a, "-d"], [a, b]):
            r1 = subprocess.run([exe, *argv], capture_output=True)
            r2 = subprocess.run([ref, *argv], capture_output=True)
            cases += 1
            if (r1.returncode, r1.stdout) != (r2.returncode, r2.stdout):
                fails += 1
                keep = os.path.join(ROOT, "gpurun_out", f"fuzz_alnstats_fail_{seed}.paf")
                os.makedirs(os.path.dirname(keep), exist_ok=True)
                with open(a, "rb") as src, open(keep, "wb") as dst:
                    dst.write(src.read())
                print("FAIL seed", seed, argv[1:], r1.returncode, r2.returncode, flush=True)
        seed += 1
    print({"cases": cases, "failures": fails, "next_seed": seed})
    return 1 if fails else 0


if __name__ == "__main__":
    sys.exit(main())
