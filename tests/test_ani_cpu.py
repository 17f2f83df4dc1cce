"""ANI pre-pass, CPU side: the oracle's restatement (main.rs:296-688, cli.rs:76-130) against hand-computed
known answers (the reference has no test of its own for this pass -> these pin the formula, not the binary),
and the native host helpers (parse_ani_method / parse_identity_value / the ANI view of the lines) against
the oracle.  No GPU needed."""
import ctypes as C

import numpy as np
import pytest

from sweepga_amd import PafFile, parse_ani_method, parse_identity_value
from sweepga_amd._lib import SwgAniInput, load
from tests import orc

L = "\t".join


def line(q, t, m, b, qlen=1000, tlen=2000, tags=()):
    return L([q, str(qlen), "0", "100", "+", t, str(tlen), "0", "100", str(m), str(b), "60", *tags])


PAF = "\n".join([
    line("A#1#c1", "B#1#c1", 90, 100),                       # pair (A#1#,B#1#): 90/100
    line("B#1#c1", "A#1#c2", 50, 100, qlen=2000, tlen=500),  # same unordered pair, reversed: + 50/100 -> 140/200 = 0.7
    line("A#1#c1", "A#1#c2", 10, 100),                       # same genome: skipped
    line("A#1#c1", "C#1#c1", 80, 100, tlen=3000, tags=("dv:f:bad", "dv:f:0.25", "dv:f:0.5")),  # first valid dv: 0.75*100
    line("C#1#c1", "B#1#c1", 99, 100, qlen=7, tlen=9),      # first-seen lengths stand (3000, 2000), not 7 / 9
    "#" + line("A#1#c1", "B#1#c1", 1, 100),                  # comment-led: skipped by the ANI pass
    "short\tline",
    line("D", "E", "x", "y"),                                # no '#': whole name is the genome; parse failures -> 0 / 1.0
]) + "\n"


def test_oracle_known_answers(tmp_path):
    p = tmp_path / "a.paf"
    p.write_text(PAF)
    # pairs: (A,B) 140/200 = 0.7 ; (A,C) 75/100 = 0.75 ; (B,C) 0.99 ; (D,E) 0/1 = 0.0  -> sorted 0, .7, .75, .99
    assert orc.calculate_ani_stats(p, orc.ANI_ALL) == (0.7 + 0.75) / 2.0
    # sizes: A#1#c1 1000, B#1#c1 2000, A#1#c2 500, C#1#c1 3000, D 1000, E 2000 = 9500
    # by identity desc: .99(BC) .9(AB) .75(AC) .5(BA) 0(DE); blocks 100,100,100,100,1
    #   n2 -> threshold 190: takes .99, .9 -> pairs (B,C) .99, (A,B) .9 -> median (.9+.99)/2
    assert orc.calculate_ani_stats(p, orc.ANI_NPERCENTILE, 2.0, orc.NSORT_IDENTITY) == (0.9 + 0.99) / 2.0
    #   n1 -> threshold 95: the first line alone crosses it
    assert orc.calculate_ani_stats(p, orc.ANI_NPERCENTILE, 1.0, orc.NSORT_IDENTITY) == 0.99
    #   n100 -> 9500 never reached: everything, same as ALL
    assert orc.calculate_ani_stats(p, orc.ANI_NPERCENTILE, 100.0, orc.NSORT_IDENTITY) == (0.7 + 0.75) / 2.0
    # by length desc (stable): the four block-100 lines in file order, then D/E; n2 (190) -> first two lines -> one pair
    assert orc.calculate_ani_stats(p, orc.ANI_NPERCENTILE, 2.0, orc.NSORT_LENGTH) == 140.0 / 200.0
    # by score = identity * max(ln 100, 1): same order as identity here
    assert orc.calculate_ani_stats(p, orc.ANI_NPERCENTILE, 2.0, orc.NSORT_SCORE) == (0.9 + 0.99) / 2.0
    empty = tmp_path / "e.paf"
    empty.write_text(line("A#1#c1", "A#1#c2", 10, 100) + "\n")
    assert orc.calculate_ani_stats(empty, orc.ANI_ALL) == 0.0
    assert orc.calculate_ani_stats(empty, orc.ANI_NPERCENTILE, 50.0) == 0.0


def test_oracle_orthogonal_method(tmp_path):
    # two overlapping mappings of the same query region onto B: the fixed 1:1 filter (>= 1 kb, scored by matches)
    # keeps the one with more matches; the 500 bp line is dropped by min_block_length
    rows = [L(["A#1#c", "9000", "0", "2000", "+", "B#1#c", "9000", "0", "2000", "1900", "2000", "60"]),
            L(["A#1#c", "9000", "0", "2000", "+", "B#1#c", "9000", "5000", "7000", "1000", "2000", "60"]),
            L(["A#1#c", "9000", "3000", "3500", "+", "B#1#c", "9000", "3000", "3500", "100", "500", "60"])]
    p = tmp_path / "o.paf"
    p.write_text("\n".join(rows) + "\n")
    assert orc.calculate_ani_stats(p, orc.ANI_ALL) == 3000.0 / 4500.0
    assert orc.calculate_ani_stats(p, orc.ANI_ORTHOGONAL) == 1900.0 / 2000.0


METHODS = ["all", "ALL", "orthogonal", "1:1", "n50", "N90-identity", "n100-score", "n12.5-length", "n0", "n101", "n50-foo",
           "n", "nx", "", "median", "n50-length-extra", "n+5", "n1e1"]


@pytest.mark.parametrize("s", METHODS)
def test_parse_ani_method_matches_oracle(s):  # main.rs:296-330
    got = parse_ani_method(s)
    want = orc.parse_ani_method(s)
    if want is None:
        assert got is None
    else:
        assert got is not None and (int(got.kind), int(got.sort)) == (want[0], want[2])
        if want[0] == orc.ANI_NPERCENTILE:
            assert got.percentile == want[1]


VALUES = ["0", "0.9", "90", "1", "1.0", "1.5", "100", "ani", "ani50", "ANI50", "ani50-2", "ani50+3", "ani75-0.5", "ani50+200",
          "ani50-200", "ani+1", "ani-", "ani50+x", "abc", "", "aniseed", "9e-1", "inf", "+0.5", "ani50-1-1"]


@pytest.mark.parametrize("ani", [None, 0.0, 0.8734, 0.999])
@pytest.mark.parametrize("v", VALUES)
def test_parse_identity_value_matches_oracle(v, ani):  # cli.rs:76-130
    want = orc.parse_identity_value(v, -1.0 if ani is None else ani)
    if want is None:
        with pytest.raises(ValueError):
            parse_identity_value(v, ani)
    else:
        got = parse_identity_value(v, ani)
        assert np.float64(got).tobytes() == np.float64(want).tobytes()


def test_native_ani_view_of_lines(tmp_path):  # main.rs:405-446, 531-586
    p = tmp_path / "a.paf"
    p.write_text(PAF)
    lib = load()
    for threads in (1, 4):
        with PafFile(p, threads=threads) as pf:
            a = SwgAniInput()
            assert lib.swg_paf_ani_input(pf.handle, threads, C.byref(a)) == 0
            n = pf.n
            assert n == 7 and a.n == n  # the '#'-led line has >= 11 fields: a record for the filter, not for the ANI pass
            view = lambda addr, dt: np.frombuffer((C.c_char * (n * np.dtype(dt).itemsize)).from_address(addr), dtype=dt).copy()
            assert view(a.eligible, np.uint8).tolist() == [1, 1, 0, 1, 1, 0, 1]
            assert view(a.matches, np.float64)[[0, 1, 3, 4, 6]].tolist() == [90.0, 50.0, (1.0 - 0.25) * 100.0, 99.0, 0.0]
            assert view(a.block_len, np.float64)[[0, 1, 3, 4, 6]].tolist() == [100.0, 100.0, 100.0, 100.0, 1.0]
            pair = view(a.pair, np.uint32)
            assert pair[0] == pair[1] and len({pair[0], pair[3], pair[4], pair[6]}) == 4
            assert a.total_genome_size == 9500.0


def test_native_ani_total_size_any_thread_count(tmp_path):
    rng = np.random.default_rng(5)
    rows, first = [], {}
    for i in range(20000):
        q = "g%d#1#c%d" % (rng.integers(0, 4), rng.integers(0, 3))
        t = "g%d#1#c%d" % (rng.integers(0, 4), rng.integers(0, 3))
        ql, tl = int(rng.integers(1, 10**6)), int(rng.integers(1, 10**6))
        rows.append(line(q, t, 5, 10, qlen=ql, tlen=tl))
        if q.split("#")[0] != t.split("#")[0]:
            first.setdefault(q, ql)
            first.setdefault(t, tl)
    p = tmp_path / "b.paf"
    p.write_text("\n".join(rows) + "\n")
    lib = load()
    for threads in (1, 3, 8):
        with PafFile(p, threads=threads) as pf:
            a = SwgAniInput()
            assert lib.swg_paf_ani_input(pf.handle, threads, C.byref(a)) == 0
            assert a.total_genome_size == float(sum(first.values()))
