#!/usr/bin/env python3
"""CPU-only differential fuzz of the native PAF ingest (paf_io.cpp) against the oracle's extract_metadata:
lines assembled from hostile field values (empty, signed, huge, fractional, hex, spaces), random tag soups
(dv:f: / cg:Z: in any order and number, malformed ones), short lines, comment lines, CRLF, missing final newline.
    python tests/fuzz/fuzz_paf_io.py --minutes 2"""
import argparse
import ctypes as C
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from sweepga_amd import PafFile, SwgError  # noqa: E402
from tests import orc  # noqa: E402

NUMS = ["0", "1", "7", "100", "4294967295", "+5", "-3", "", " 4", "4 ", "1e3", "0x10", "12.5", "007", "99999999999999999999", "abc",
        "18446744073709551615", "4294967296"]
NAMES = ["a", "b", "g1#1#c1", "g1#1#c2", "g2#1#c1", "g2#2#c1", "#", "a#", "#a", "a#b#c#d", "", "x y", "chr1"]
TAGS = ["dv:f:0.1", "dv:f:1e-3", "dv:f:", "dv:f:x", "dv:f:inf", "dv:f:-0.5", "dv:f:+0.25", "dv:f:.5", "dv:f:5.", "dv:f:1e400",
        "cg:Z:10=", "cg:Z:10=2X3=", "cg:Z:5M", "cg:Z:", "cg:Z:=", "cg:Z:10", "cg:Z:3=4", "cg:Z:99999999999999999999=", "cg:Z:0=",
        "tp:A:P", "NM:i:3", "", "dv:f", "cg:Z", "xdv:f:0.1", "DV:F:0.1", "cg:Z:1=1=1="]


def random_text(rng):
    lines = []
    for _ in range(int(rng.integers(0, 60))):
        kind = rng.random()
        if kind < 0.08:
            lines.append(rng.choice(["", "#comment", "a\tb", "\t" * int(rng.integers(0, 14)), "# a\t1\t2\t3\t+\tb\t4\t5\t6\t7\t8\t9"]))
            continue
        f = [rng.choice(NAMES), rng.choice(NUMS), rng.choice(NUMS), rng.choice(NUMS), rng.choice(["+", "-", "", "*", "++"]), rng.choice(NAMES),
             rng.choice(NUMS), rng.choice(NUMS), rng.choice(NUMS), rng.choice(NUMS), rng.choice(NUMS)]
        if kind < 0.15:
            f = f[: int(rng.integers(1, 11))]
        f += [rng.choice(TAGS) for _ in range(int(rng.integers(0, 5)))]
        lines.append("\t".join(f))
    nl = "\r\n" if rng.random() < 0.2 else "\n"
    text = nl.join(lines)
    if lines and rng.random() < 0.8:
        text += nl
    return text


def oracle_columns(path, cap):
    cols = {k: np.zeros(cap, dtype=np.uint64) for k in ("rank", "qs", "qe", "ts", "te", "block", "matches")}
    ident = np.zeros(cap, dtype=np.float64)
    strand = np.zeros(cap, dtype=np.uint8)
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    f = orc.lib().orc_extract_metadata
    f.restype = C.c_int64
    n = f(str(path).encode(), C.c_uint64(cap), p(cols["rank"]), p(cols["qs"]), p(cols["qe"]), p(cols["ts"]), p(cols["te"]),
          p(cols["block"]), p(ident), p(cols["matches"]), p(strand))
    return n, cols, ident, strand


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--minutes", type=float, default=2.0)
    ap.add_argument("--seed", type=int, default=0)
    args = ap.parse_args()
    t0, seed, cases, fails = time.time(), args.seed, 0, 0
    tmp = tempfile.mkdtemp(prefix="fuzz_paf_")
    path = os.path.join(tmp, "in.paf")
    while time.time() - t0 < args.minutes * 60:
        rng = np.random.default_rng(seed)
        text = random_text(rng)
        with open(path, "w", newline="") as fh:
            fh.write(text)
        n, cols, ident, strand = oracle_columns(path, text.count("\n") + 2)
        ok = True
        try:
            with PafFile(path, threads=int(rng.integers(1, 4))) as pf:
                ok = pf.n == n and (pf.ranks == cols["rank"][:n]).all()
                # a file with values >= 2^32: coordinates relative to the sequence's smallest one, or (a sequence touched over 2^32
                # bases or more) to one constant per sweep segment -- absolute() adds back whichever it was
                off = pf.seq_offsets if pf.seq_offsets is not None else pf.record_offsets(0)
                for name, key, ids in (("q_start", "qs", "q_id"), ("q_end", "qe", "q_id"), ("t_start", "ts", "t_id"),
                                       ("t_end", "te", "t_id"), ("block_len", "block", None), ("matches", "matches", None)):
                    col = pf.absolute(name) if ids is not None else pf.column(name).astype(np.uint64)
                    ok = ok and (col == cols[key][:n]).all()
                if off is not None:
                    ok = ok and any(int(v) > 0xffffffff for k in ("qs", "qe", "ts", "te", "block", "matches") for v in cols[k][:n])
                ok = ok and (pf.column("identity").view(np.uint64) == ident[:n].view(np.uint64)).all()
                ok = ok and (pf.column("strand") == (strand[:n] != ord("+"))).all()
                rec = orc.parse_paf_text(text.replace("\r\n", "\n"))
                names = pf.names
                ok = ok and [names[i] for i in pf.column("q_id")] == rec.qname and [names[i] for i in pf.column("t_id")] == rec.tname
        except SwgError as e:  # only the documented u32 range limit may refuse an input the oracle accepts
            big = any(int(v) > 0xffffffff for k in ("qs", "qe", "ts", "te", "block", "matches") for v in cols[k][:max(n, 0)])
            ok = "2^32" in str(e) and big
        cases += 1
        if not ok:
            fails += 1
            keep = os.path.join(ROOT, "gpurun_out", f"fuzz_paf_fail_{seed}.paf")
            os.makedirs(os.path.dirname(keep), exist_ok=True)
            with open(keep, "w", newline="") as fh:
                fh.write(text)
            print("FAIL seed", seed, "->", keep, flush=True)
        seed += 1
    print({"cases": cases, "failures": fails, "next_seed": seed})
    return 1 if fails else 0


if __name__ == "__main__":
    sys.exit(main())
