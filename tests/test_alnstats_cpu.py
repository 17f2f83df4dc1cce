"""alnstats (src/bin/alnstats.rs): hand-computed vectors for the oracle's restatement (oracle/alnstats_ref.cpp -- the
reference holds no test or golden output for this binary, so parity is pinned by these), and the product's
swg_alnstats_* / bin/alnstats against the oracle on random and hostile inputs."""
import gzip
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.path.join(ROOT, "oracle", "alnstats-ref")


@pytest.fixture(scope="module")
def bins():
    if not os.path.exists(REF) or os.path.getmtime(REF) < os.path.getmtime(os.path.join(ROOT, "oracle", "alnstats_ref.cpp")):
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "alnstats-ref"])
    from sweepga_amd import build
    return build.build_alnstats(), REF


def run(exe, *args):
    r = subprocess.run([exe, *map(str, args)], capture_output=True)
    return r.returncode, r.stdout


KAT = ("A#1#c1\t1000\t0\t500\t+\tB#1#c1\t2000\t0\t500\t450\t500\t60\n"
       "A#1#c1\t1000\t500\t1000\t+\tB#1#c1\t2000\t600\t1100\t500\t500\t60\ttp:A:P\n"
       "A#1#c2\t3000\t0\t300\t-\tA#1#c1\t1000\t0\t300\t300\t300\t60\n"        # same genome, other chromosome
       "A#1#c1\t1000\t0\t100\t+\tA#1#c1\t1000\t0\t100\t100\t100\t60\r\n"      # self mapping (CRLF)
       "too\tfew\tfields\n"
       "B#1#c1\t2000\t0\t1900\t+\tA#1#c2\t3000\t0\t1900\t950\t1900\t60")      # no final newline


def kat_text(label, detailed):
    """print_stats (:166-228) written out with the reference's format strings.  bases = 500+500+300+100+1900, matches =
    450+500+300+100+950; genome sizes A#1# = 1000 + 3000, B#1# = 2000; A->B: 1000 bases / 4000 = 25 %, identity 950/1000;
    B->A: 1900 / 2000 = 95 % (not > 95), identity 950/1900."""
    o = [f"\nStatistics for {label}:", "=" * 60,
         f"Total mappings:        {'5':>12}", f"Total bases:           {'3,300':>12}",
         f"Average identity:      {2300 / 3300 * 100.0:>11.1f}%", f"Self mappings:         {'1':>12}",
         f"Inter-chromosomal:     {'1':>12}", f"Inter-genome:          {'3':>12}", f"Chromosome pairs:      {'4':>12}",
         f"Genome pairs:          {2:>12}", f"Average coverage:      {60.0:>11.1f}%", f"Pairs >95% coverage:   {'0/2':>12}"]
    if detailed:
        o += ["\nPer-genome-pair statistics:", "-" * 60,
              f"{'B#1':20} -> {'A#1':20} {95.0:6.1f}% cov, {50.0:6.1f}% id, {'1,900':>10} bp",
              f"{'A#1':20} -> {'B#1':20} {25.0:6.1f}% cov, {95.0:6.1f}% id, {'1,000':>10} bp"]
    return ("\n".join(o) + "\n").encode()


def test_oracle_hand_computed(bins, tmp_path):
    _, ref = bins
    p = tmp_path / "x.paf"
    p.write_bytes(KAT.encode())
    assert run(ref, p) == (0, kat_text(str(p), False))
    assert run(ref, p, "-d") == (0, kat_text(str(p), True))
    assert b"69.7%" in kat_text("x", False)


def test_oracle_compare_hand_computed(bins, tmp_path):
    """compare_stats (:230-284): file 2 = the first two lines of file 1 (1000 bases, 950 matches, one genome pair at
    1000 / 1000 = 100 % since only A#1#c1 is seen)."""
    _, ref = bins
    a, b = tmp_path / "a.paf", tmp_path / "b.paf"
    a.write_bytes(KAT.encode())
    b.write_bytes("".join(KAT.splitlines(keepends=True)[:2]).encode())
    fa, fb = str(a), str(b)
    o = [f"\nComparison: {fa} vs {fb}", "=" * 60,
         "\nMappings:", f"  {'Before':30} {'5':>12}", f"  {'After':30} {'2':>12}", f"  {'Change':30} {'-3':>12} ({-60.0:+.1f}%)",
         "\nTotal bases:", f"  {'Before':30} {'3,300':>12}", f"  {'After':30} {'1,000':>12}",
         f"  {'Change':30} {'-2,300':>12} ({100.0 * -2300 / 3300:+.1f}%)",
         "\nAverage identity:", f"  {fa:30} {2300 / 3300 * 100:>11.1f}%", f"  {fb:30} {95.0:>11.1f}%",
         f"  {'Change':30} {(0.95 - 2300 / 3300) * 100:>+10.1f}%",
         "\nInter-chromosomal:", f"  {'Before':30} {'1':>12}", f"  {'After':30} {'0':>12}", f"  {'Change':30} {'-1':>12} ({-100.0:+.1f}%)",
         "\nChromosome pairs:", f"  {'Before':30} {'4':>12}", f"  {'After':30} {'1':>12}", f"  {'Change':30} {'-3':>12} ({-75.0:+.1f}%)",
         "\nAverage genome pair coverage:", f"  {fa:30} {60.0:>11.1f}%", f"  {fb:30} {100.0:>11.1f}%", f"  {'Change':30} {40.0:>+10.1f}%",
         "\nGenome pairs with >95% coverage:", f"  {fa:30} {'0/2':>12}", f"  {fb:30} {'1/1':>12}"]
    assert run(ref, a, b) == (0, ("\n".join(o) + "\n").encode())


def test_product_hand_computed(bins, tmp_path):
    exe, _ = bins
    from sweepga_amd import AlnStats
    p = tmp_path / "x.paf"
    p.write_bytes(KAT.encode())
    assert run(exe, p) == (0, kat_text(str(p), False))
    assert run(exe, "--detailed", p) == (0, kat_text(str(p), True))
    for threads in (1, 3):
        with AlnStats(text=KAT, threads=threads) as s:
            m = s.summary
            assert (m.total_mappings, m.total_bases, m.total_matches, m.self_mappings, m.inter_chromosomal, m.inter_genome,
                    m.chr_pair_count, m.genome_pairs, m.above_95_pct) == (5, 3300, 2300, 1, 1, 3, 4, 2, 0)
            assert m.avg_identity == 2300 / 3300 and m.avg_coverage == 60.0
            assert s.pairs == [("A#1#", "B#1#", 25.0, 1000, 950), ("B#1#", "A#1#", 95.0, 1900, 950)]
            assert s.report("lbl", True) == kat_text("lbl", True)


def random_paf(rng, n, crlf=False):
    names = [f"g{g}#{h}#c{c}" for g in range(3) for h in (1, 2) for c in range(2)] + ["plain", "x#y", "café#1#z", "#", "a##"]
    lens = {nm: int(rng.integers(5_000, 50_000)) for nm in names}
    out = []
    for i in range(n):
        q, t = names[rng.integers(0, len(names))], names[rng.integers(0, len(names))]
        if rng.random() < 0.05:
            t = q
        qs = int(rng.integers(0, lens[q] - 1000))
        ql = int(rng.integers(0, 1000))
        m = int(rng.integers(0, ql + 1))
        ln = lens[q] if rng.random() < 0.98 else int(rng.integers(1, 9)) * 1000      # sizes: last writer wins
        out.append(f"{q}\t{ln}\t{qs}\t{'+' if rng.random() < 0.1 else ''}{qs + ql}\t-\t{t}\t{lens[t]}\t0\t{ql}\t{m}\t{ql}\t60\tcg:Z:5=")
        if rng.random() < 0.02:
            out.append(rng.choice(["", "# c", "a\tb", "q\t1\t2\t3\t+\tt\t4\t5\t6\t7"]))
    nl = "\r\n" if crlf else "\n"
    return nl.join(out) + (nl if rng.random() < 0.7 else "")


@pytest.mark.parametrize("seed", range(6))
def test_product_matches_oracle_on_random_files(bins, tmp_path, seed):
    exe, ref = bins
    from sweepga_amd import AlnStats
    rng = np.random.default_rng(seed)
    a, b = tmp_path / "a.paf", tmp_path / "b.paf"
    ta = random_paf(rng, int(rng.choice([0, 1, 40, 3000, 30000])), crlf=seed % 2 == 1)
    tb = random_paf(rng, int(rng.choice([1, 500, 8000])))
    a.write_bytes(ta.encode())
    b.write_bytes(tb.encode())
    for args in ((a,), (a, "-d"), (b, "-d"), (a, b), (b, a, "-d")):
        assert run(exe, *args) == run(ref, *args), args
    # every thread count folds to the same statistics (slices merged in file order)
    want = run(ref, a, "-d")[1]
    for threads in (1, 2, 5, 16):
        with AlnStats(text=ta, threads=threads) as s:
            assert s.report(str(a), True) == want, threads
    gz = tmp_path / "a.paf.gz"
    gz.write_bytes(gzip.compress(ta.encode()))
    assert run(exe, gz, "-d")[1].replace(str(gz).encode(), str(a).encode()) == want


def test_errors_and_edge_values(bins, tmp_path):
    exe, ref = bins
    good = "q#1#a\t100\t0\t50\t+\tt#1#b\t100\t0\t50\t40\t50\t60\n"
    cases = {
        "bad_qlen": good + "q#1#a\tx\t0\t50\t+\tt#1#b\t100\t0\t50\t40\t50\t60\n",
        "bad_block": good * 3 + "q#1#a\t100\t0\t50\t+\tt#1#b\t100\t0\t50\t40\t-5\t60\n" + good,
        "bad_start_neg": "q#1#a\t100\t-0\t50\t+\tt#1#b\t100\t0\t50\t40\t50\t60\n",
        "overflow": "q#1#a\t100\t0\t18446744073709551616\t+\tt#1#b\t100\t0\t50\t40\t50\t60\n",
        "wrapping_len": "q#1#a\t100\t60\t50\t+\tt#1#b\t100\t0\t50\t40\t50\t60\n",            # end < start: wraps (release build)
        "inf_coverage": "q#1#a\t0\t0\t50\t+\tt#1#b\t100\t0\t50\t40\t50\t60\n",               # genome of size 0
        "nan_coverage_single": "q#1#a\t0\t5\t5\t+\tt#1#b\t100\t0\t50\t40\t50\t60\n",         # 0 / 0, one pair: sort never compares
        "nan_coverage_two": "q#1#a\t0\t5\t5\t+\tt#1#b\t100\t0\t50\t40\t50\t60\n" + good.replace("q#1#a", "z#1#a"),
        "ties": "".join(f"g{i}#1#a\t100\t0\t50\t+\th#1#b\t100\t0\t50\t40\t50\t60\n" for i in (3, 1, 2)),
        "empty": "",
        "only_short": "a\tb\n\n",
    }
    for name, text in cases.items():
        p = tmp_path / f"{name}.paf"
        p.write_bytes(text.encode())
        for flags in ((), ("-d",)):
            a, b = run(exe, p, *flags), run(ref, p, *flags)
            assert a == b, (name, flags, a, b)
    assert run(exe, tmp_path / "bad_qlen.paf")[0] == 1 and run(exe, tmp_path / "nan_coverage_two.paf", "-d")[0] == 101
    assert b"inf%" in run(exe, tmp_path / "inf_coverage.paf")[1] and b"NaN%" in run(exe, tmp_path / "nan_coverage_single.paf", "-d")[1]
    assert run(exe, tmp_path / "missing.paf")[0] == 1 == run(ref, tmp_path / "missing.paf")[0]
    from sweepga_amd import AlnStats, SwgError
    with pytest.raises(SwgError, match="Invalid block length \\(line 4\\)"):
        AlnStats(text=cases["bad_block"], threads=1)


def test_fuzz_slice_against_oracle(bins):
    """A slice of tests/fuzz/fuzz_alnstats.py."""
    import sys
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "fuzz", "fuzz_alnstats.py"), "--minutes", "0.1", "--seed", "77000"],
                       capture_output=True, text=True)
    assert r.returncode == 0 and "'failures': 0" in r.stdout, r.stdout[-500:] + r.stderr[-500:]
