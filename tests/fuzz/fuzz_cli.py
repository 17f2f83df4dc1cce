#!/usr/bin/env python3
"""End-to-end differential fuzz of the two command lines: sweepga-gpu (native ingest, GPU filter, ANI pre-pass,
native egress) against oracle/sweepga-ref on random PAF texts and random flag sets (including aniN thresholds,
--ani-method, --self, --scaffolds-only, --devices).  Outputs must be byte-identical.
    python tests/fuzz/fuzz_cli.py --minutes 5"""
import argparse
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from sweepga_amd import build  # noqa: E402
from tests import gen  # noqa: E402

REF = os.path.join(ROOT, "oracle", "sweepga-ref")


def random_flags(rng):
    fl = []
    if rng.random() < 0.7:
        fl += ["--num-mappings", str(rng.choice(["1:1", "1", "many:many", "2:3", "1:many", "many:1", "5", "3:1"]))]
    if rng.random() < 0.5:
        fl += ["--scaffold-filter", str(rng.choice(["1:1", "many:many", "1:many", "2:2"]))]
    if rng.random() < 0.5:
        fl += ["--overlap", str(rng.choice(["0", "0.5", "0.95", "1"]))]
    if rng.random() < 0.5:
        fl += ["--scoring", str(rng.choice(["ani", "length", "length-ani", "matches", "log-length-ani", "bogus"]))]
    fl += ["--scaffold-jump", str(rng.choice(["0", "2k", "20000", "50k", "1m"]))]
    if rng.random() < 0.6:
        fl += ["--scaffold-mass", str(rng.choice(["0", "1k", "10000"]))]
    if rng.random() < 0.5:
        fl += ["--scaffold-dist", str(rng.choice(["0", "5k", "100000"]))]
    if rng.random() < 0.4:
        fl += ["--scaffold-overlap", str(rng.choice(["0", "0.5", "1"]))]
    if rng.random() < 0.3:
        fl += ["--min-aln-length", str(rng.choice(["100", "1k", "0"]))]
    r = rng.random()
    if r < 0.35:
        fl += ["--min-aln-identity", str(rng.choice(["ani", "ani50", "ani50-5", "ANI50+1", "ani25-20", "ani50-0.5"]))]
    elif r < 0.6:
        fl += ["--min-aln-identity", str(rng.choice(["0.8", "90", "0", "0.97"]))]
    if rng.random() < 0.3:
        fl += ["--min-scaffold-identity", str(rng.choice(["ani50-10", "0.85", "ani", "80"]))]
    if rng.random() < 0.6:
        fl += ["--ani-method", str(rng.choice(["all", "orthogonal", "1:1", "n50", "n100", "n90-length", "n20-score", "n5-identity", "zzz"]))]
    if rng.random() < 0.2:   # tree sparsification of the input before the filter (src/main.rs:3640-3688)
        fl += ["--sparsify", str(rng.choice(["tree:1", "tree:2:1", "tree:1:1:0.3", "knn:3", "tree:0:2", "tree:1:0:0.9", "none", "random:0.5"]))]
    if rng.random() < 0.25:
        fl += ["--self"]
    if rng.random() < 0.1:
        fl += ["--scaffolds-only"]
    return fl


def random_paf(rng, rng_wide=None):
    """rng_wide (a second generator, so that the cases of earlier campaigns keep their seeds): one file in ten has every
    sequence's coordinates moved by a constant of its own, up to 2^62 -- the ingest's 64-bit pass and per-sequence rebasing."""
    n = int(rng.choice([1, 5, 300, 3000, 15000]))
    chr_len = int(rng.choice([20_000, 300_000, 2_000_000]))
    rec = gen.random_records(rng, n, n_genomes=int(rng.integers(1, 6)), chrs_per_genome=int(rng.integers(1, 4)), span=chr_len - 5000,
                             max_len=min(8000, chr_len // 3), pansn=bool(rng.random() < 0.8), minus_frac=float(rng.choice([0.0, 0.3])),
                             self_frac=float(rng.choice([0.0, 0.1])))
    if rng_wide is not None and rng_wide.random() < 0.1:
        rec, _ = gen.shifted(rec, rng_wide)
    out = []
    for ln in gen.records_to_paf(rng, rec).split("\n"):
        f = ln.split("\t")
        if len(f) >= 11:
            f[1] = f[6] = str(chr_len)
        out.append("\t".join(f))
    text = "\n".join(out)
    if rng.random() < 0.1:
        text = text.replace("\n", "\r\n")
    return text


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--minutes", type=float, default=5.0)
    ap.add_argument("--seed", type=int, default=0)
    args = ap.parse_args()
    tmp = tempfile.mkdtemp(prefix="fuzz_cli_")
    paf, o1, o2 = (os.path.join(tmp, x) for x in ("in.paf", "gpu.paf", "ref.paf"))
    t0, seed, cases, fails, skipped = time.time(), args.seed, 0, 0, 0
    while time.time() - t0 < args.minutes * 60:
        rng = np.random.default_rng(seed)
        with open(paf, "w", newline="") as fh:
            fh.write(random_paf(rng, np.random.default_rng(seed + 10**9)))
        flags = random_flags(rng)
        dev = ["--devices", "0,0"] if seed % 5 == 0 else []
        r1 = subprocess.run([build.CLI, paf, "--output-file", o1, "--quiet", *flags, *dev], capture_output=True, text=True)
        r2 = subprocess.run([REF, paf, "--output-file", o2, *flags], capture_output=True, text=True)
        cases += 1
        if r2.returncode != 0:  # the oracle refuses (e.g. NaN in the ANI pass): the GPU side must refuse too
            ok = r1.returncode != 0
            skipped += 1
        else:
            ok = r1.returncode == 0 and open(o1, "rb").read() == open(o2, "rb").read()
        if not ok:
            fails += 1
            keep = os.path.join(ROOT, "gpurun_out", f"fuzz_cli_fail_{seed}")
            os.makedirs(os.path.dirname(keep), exist_ok=True)
            with open(keep + ".paf", "wb") as fh:
                fh.write(open(paf, "rb").read())
            with open(keep + ".txt", "w") as fh:
                fh.write(" ".join(flags + dev) + "\n" + r1.stderr[-500:] + "\n" + r2.stderr[-500:])
            print("FAIL seed", seed, flags, dev, r1.returncode, r2.returncode, flush=True)
        seed += 1
    print({"cases": cases, "failures": fails, "oracle_refusals": skipped, "next_seed": seed})
    return 1 if fails else 0


if __name__ == "__main__":
    sys.exit(main())
