#!/usr/bin/env python3
"""S-yeast: the stand-in for BASELINE.json configs[0]/[1] ("scerevisiae8 all-vs-all PAF").

The real PAF is not in the reference tree (tests/golden_data holds checksums only and the FASTA is a missing
large blob), so a yeast-SHAPED PAF is generated from the sequence names and lengths of
`data/scerevisiae8.fa.gz.fai` (8 genomes x 17 chromosomes) exactly as SURVEY.md §8(d) prescribes: for each
ordered genome pair every homologous chromosome pair is tiled with alignments (lognormal lengths, exponential
gaps, Beta identities, 2 % inversions, 5 % off-diagonal repeats), seed 42.

Run in the authoring container only (it reads /root/reference).  Writes
  tests/golden/syeast.paf.gz                  the input (committed; data, not reference source)
  tests/golden/syeast_expected.json           sha256 + kept-line counts of the CPU oracle's output for the
                                              flag sets of configs[0] and configs[1]
"""
import gzip
import hashlib
import json
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FAI = "/root/reference/data/scerevisiae8.fa.gz.fai"
SCALE = float(sys.argv[1]) if len(sys.argv) > 1 else 0.25  # fraction of each chromosome that gets tiled

FLAG_SETS = {
    "c1_scaffold_jump_0": ["--scaffold-jump", "0"],
    "c2_defaults": [],
    "c2_num_mappings_1to1": ["--num-mappings", "1:1"],
    "c2_full_1to1_rescue": ["--num-mappings", "1:1", "--scaffold-filter", "1:1", "--scaffold-dist", "20000"],
    # identity thresholds from the ANI pre-pass (main.rs:3571-3595)
    "c2_ani50_minus2": ["--min-aln-identity", "ani50-2"],
    "c2_orthogonal_ani_scaffolds": ["--num-mappings", "1:1", "--min-scaffold-identity", "ani50-1", "--ani-method", "orthogonal"],
}


def main():
    rng = np.random.default_rng(42)
    seqs = [(l.split("\t")[0], int(l.split("\t")[1])) for l in open(FAI)]
    genomes = {}
    for name, ln in seqs:
        genomes.setdefault(name.split("#")[0], []).append((name, ln))
    gnames = list(genomes)
    lines = []
    for gq in gnames:
        for gt in gnames:
            for ci, (qn, ql) in enumerate(genomes[gq]):
                tn, tl = genomes[gt][ci]
                if qn == tn:
                    continue  # identical names are self mappings
                span = int(min(ql, tl) * SCALE)
                pos = int(rng.integers(0, 2000))
                while pos < span - 300:
                    ln = int(np.clip(np.exp(rng.normal(np.log(8000), 1.0)), 200, 200_000))
                    ln = min(ln, span - pos)
                    ident = 0.90 + 0.099 * rng.beta(5, 2)
                    strand = "-" if rng.random() < 0.02 else "+"
                    ts = max(0, min(tl - ln, pos + int(rng.normal(0, 300))))
                    m = int(round(ident * ln))
                    tags = "\tNM:i:%d\tcg:Z:%d=%dX" % (ln - m, m, ln - m) if rng.random() < 0.5 else ""
                    lines.append(f"{qn}\t{ql}\t{pos}\t{pos + ln}\t{strand}\t{tn}\t{tl}\t{ts}\t{ts + ln}\t{m}\t{ln}\t60{tags}")
                    if rng.random() < 0.05:  # off-diagonal repeat
                        rq, rql = genomes[gq][int(rng.integers(0, 17))]
                        rt, rtl = genomes[gt][int(rng.integers(0, 17))]
                        if rq != rt:
                            rl = int(rng.integers(300, 3000))
                            a, b = int(rng.integers(0, rql - rl)), int(rng.integers(0, rtl - rl))
                            rid = rng.uniform(0.80, 0.95)
                            lines.append(f"{rq}\t{rql}\t{a}\t{a + rl}\t+\t{rt}\t{rtl}\t{b}\t{b + rl}\t{int(rid * rl)}\t{rl}\t60")
                    pos += ln + int(rng.exponential(500))
    text = "\n".join(lines) + "\n"
    gold = os.path.join(ROOT, "tests", "golden")
    os.makedirs(gold, exist_ok=True)
    with gzip.GzipFile(os.path.join(gold, "syeast.paf.gz"), "wb", mtime=0) as f:
        f.write(text.encode())
    expected = {"n_lines": len(lines), "input_sha256": hashlib.sha256(text.encode()).hexdigest(), "flag_sets": {}}
    ref = os.path.join(ROOT, "oracle", "sweepga-ref")
    with tempfile.TemporaryDirectory() as d:
        inp = os.path.join(d, "in.paf")
        open(inp, "w").write(text)
        for name, flags in FLAG_SETS.items():
            out = os.path.join(d, name + ".paf")
            subprocess.check_call([ref, inp, "--output-file", out, *flags])
            data = open(out, "rb").read()
            expected["flag_sets"][name] = {"flags": flags, "kept": data.count(b"\n"), "sha256": hashlib.sha256(data).hexdigest()}
    json.dump(expected, open(os.path.join(gold, "syeast_expected.json"), "w"), indent=1)
    print(json.dumps(expected, indent=1))


if __name__ == "__main__":
    main()
