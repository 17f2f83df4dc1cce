"""Pins the CPU oracle (oracle/) to the reference's own known-answer tests.

Every case cites the reference test it transcribes (paths relative to pangenome/sweepga).
Inputs and expected outputs are data from those tests; where a reference assert is weaker
than an exact answer, the assert is transcribed as written.  Reference tests that
contradict current reference code are not encoded.
"""
import math
import os
import subprocess
import tempfile

import pytest

from tests import orc
from tests.orc import (IDENTITY, K_INF, LENGTH, LENGTH_IDENTITY, LOG_LENGTH_IDENTITY, MANY_TO_MANY, MATCHES,
                       ONE_TO_MANY, ONE_TO_ONE)

LLI = LOG_LENGTH_IDENTITY
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def mk(qs, qe, ts, te, ident=1.0):
    return (qs, qe, ts, te, ident)


# ----------------------------------------------------------------------------------------
# src/plane_sweep_exact.rs:621-827 (in-crate unit tests)
# ----------------------------------------------------------------------------------------
def test_exact_empty_input():  # :626-630
    assert orc.plane_sweep_query([], 1, 0.95) == []


def test_exact_single_mapping():  # :633-646
    assert orc.plane_sweep_query([mk(100, 200, 300, 400, 0.95)], 1, 0.95) == [0]


def test_exact_non_overlapping():  # :649-674
    m = [mk(100, 200, 300, 400, 0.95), mk(300, 400, 500, 600, 0.90)]
    assert len(orc.plane_sweep_query(m, 1, 0.95)) == 2


def test_exact_overlapping():  # :677-703
    m = [mk(100, 200, 300, 400, 0.95), mk(150, 250, 350, 450, 0.90)]
    assert len(orc.plane_sweep_query(m, 1, 0.95)) == 2


def test_exact_secondaries():  # :706-744
    m = [mk(100, 200, 300, 400, 0.95), mk(100, 200, 500, 600, 0.90), mk(100, 200, 700, 800, 0.85)]
    kept = orc.plane_sweep_query(m, 2, 0.95)
    assert len(kept) == 2 and 0 in kept and 1 in kept


def test_exact_overlap_threshold():  # :747-801
    m = [mk(100, 200, 300, 400, 0.95), mk(100, 200, 500, 600, 0.90), mk(100, 200, 700, 800, 0.85)]
    assert len(orc.plane_sweep_query(m, 1, 1.0)) == 1
    assert len(orc.plane_sweep_query(m, 2, 1.0)) == 2
    assert len(orc.plane_sweep_query(m, 2, 0.5)) == 2


def test_exact_chromosome_boundaries():  # :804-826 (u64::MAX coordinates)
    U = 2**64 - 1
    m = [mk(0, 100, 0, 100, 0.95), mk(U - 100, U, 1000, 1100, 0.90)]
    assert len(orc.plane_sweep_query(m, 1, 0.95)) == 2


# ----------------------------------------------------------------------------------------
# tests/test_plane_sweep.rs
# ----------------------------------------------------------------------------------------
def test_ps_empty_and_single():  # :26-47
    assert orc.plane_sweep_query([], 1, 0.95) == []
    assert orc.plane_sweep_query([mk(100, 200, 300, 400)], 1, 0.95) == [0]


def test_ps_non_overlapping():  # :50-70
    kept = orc.plane_sweep_query([mk(100, 200, 300, 400), mk(300, 400, 500, 600)], 1, 0.95)
    assert sorted(kept) == [0, 1]


def test_ps_overlapping_keep_best():  # :73-94
    assert len(orc.plane_sweep_query([mk(100, 250, 300, 450), mk(150, 350, 400, 600)], 1, 0.95)) == 2


def test_ps_identical_mappings():  # :97-140
    m = [mk(100, 200, 300, 400), mk(100, 200, 500, 600), mk(100, 200, 700, 800)]
    assert len(orc.plane_sweep_query(m, 1, 0.95)) == 1
    assert len(orc.plane_sweep_query(m, 2, 0.95)) == 2
    assert len(orc.plane_sweep_query(m, K_INF, 0.95)) == 3


def test_ps_contained():  # :143-171
    m = [mk(100, 300, 400, 600), mk(150, 180, 500, 530)]
    assert orc.plane_sweep_query(m, 1, 0.95) == [0]
    assert len(orc.plane_sweep_query(m, 2, 0.95)) == 2


def test_ps_overlap_threshold():  # :174-197
    m = [mk(100, 300, 400, 600), mk(100, 300, 700, 900), mk(100, 300, 1000, 1200), mk(100, 300, 1300, 1500)]
    assert len(orc.plane_sweep_query(m, 2, 0.5)) == 2


def test_ps_complex_overlaps():  # :200-222
    m = [mk(0, 100, 0, 100), mk(50, 150, 200, 300), mk(120, 220, 400, 500), mk(200, 300, 600, 700),
         mk(280, 380, 800, 900)]
    assert len(orc.plane_sweep_query(m, 1, 0.95)) >= 3


def test_ps_target_axis():  # :225-242
    m = [mk(100, 200, 300, 400), mk(300, 400, 350, 450), mk(500, 600, 600, 700)]
    assert 2 in orc.plane_sweep_target(m, 1, 0.95)


def test_ps_both_axes():  # :245-271
    m = [mk(100, 200, 300, 400), mk(100, 200, 500, 600), mk(300, 400, 300, 400), mk(500, 600, 700, 800)]
    assert 3 in orc.plane_sweep_both(m, 1, 1, 0.95)


def test_ps_score_calculation():  # :274-300
    assert orc.score(100, 200, 1.0, LLI) > orc.score(100, 110, 1.0, LLI)
    ratio = orc.score(0, 1000, 1.0, LLI) / orc.score(0, 100, 1.0, LLI)
    assert abs(ratio - math.log(1000) / math.log(100)) < 0.001


def test_ps_secondary_count():  # :303-347
    m = [mk(100, 200, 300, 400), mk(100, 190, 500, 590), mk(100, 180, 700, 780), mk(100, 170, 900, 970),
         mk(100, 160, 1100, 1160)]
    assert orc.plane_sweep_query(m, 1, 1.0) == [0]
    assert len(orc.plane_sweep_query(m, 3, 1.0)) == 3
    assert len(orc.plane_sweep_query(m, K_INF, 1.0)) == 5


def test_ps_strand_independence():  # :350-366
    assert len(orc.plane_sweep_query([mk(100, 200, 300, 400), mk(150, 250, 500, 600)], 1, 0.95)) == 2


def test_ps_event_ordering():  # :369-385 (zero-length never kept)
    m = [mk(100, 100, 300, 300), mk(100, 200, 400, 500), mk(100, 300, 600, 800)]
    assert 0 not in orc.plane_sweep_query(m, 1, 0.95)


def test_ps_real_world():  # :388-431
    m = [mk(1000, 2000, 5000, 6000), mk(1500, 2500, 7000, 8000), mk(3000, 4000, 9000, 10000),
         mk(3200, 3800, 11000, 11600), mk(5000, 5500, 15000, 15500), mk(5000, 5500, 16000, 16500),
         mk(5000, 5500, 17000, 17500), mk(5000, 5500, 18000, 18500), mk(8000, 12000, 20000, 24000)]
    kept = orc.plane_sweep_query(m, 1, 0.95)
    assert 8 in kept and len(kept) >= 4
    assert len(orc.plane_sweep_query(m, 2, 0.95)) > len(kept)


# ----------------------------------------------------------------------------------------
# tests/test_scoring_ranking.rs
# ----------------------------------------------------------------------------------------
def test_sr_identity_prefers_high_identity():  # :26-41 -- pins the sticky `overlapped` rule
    m = [mk(100, 500, 1000, 1400, 0.70), mk(100, 200, 2000, 2100, 0.99), mk(100, 300, 3000, 3200, 0.85)]
    assert orc.plane_sweep_query(m, 1, 0.95, IDENTITY) == [1]


def test_sr_length_prefers_long():  # :44-59
    m = [mk(100, 200, 1000, 1100, 0.99), mk(100, 600, 2000, 2500, 0.50), mk(100, 350, 3000, 3250, 0.75)]
    assert orc.plane_sweep_query(m, 1, 0.95, LENGTH) == [1]


def test_sr_length_identity():  # :62-77
    m = [mk(100, 200, 1000, 1100, 0.95), mk(100, 400, 2000, 2300, 0.60), mk(100, 300, 3000, 3200, 0.80)]
    assert orc.plane_sweep_query(m, 1, 0.95, LENGTH_IDENTITY) == [1]


def test_sr_log_length_identity():  # :80-95
    m = [mk(100, 200, 1000, 1100, 0.95), mk(100, 1100, 2000, 3000, 0.60), mk(100, 600, 3000, 3500, 0.75)]
    assert orc.plane_sweep_query(m, 1, 0.95, LLI) == [2]


def test_sr_identical_scores():  # :98-114
    m = [mk(100, 300, 1000, 1200, 0.90), mk(100, 280, 2000, 2180, 1.00), mk(100, 460, 3000, 3360, 0.50)]
    assert len(orc.plane_sweep_query(m, 1, 0.95, LENGTH_IDENTITY)) == 1


def test_sr_non_overlapping_preserved():  # :117-128
    m = [mk(100, 200, 1000, 1100, 0.50), mk(300, 500, 2000, 2200, 0.99), mk(600, 700, 3000, 3100, 0.30)]
    assert len(orc.plane_sweep_query(m, 1, 0.95, IDENTITY)) == 3


def test_sr_overlapping_best_survives():  # :131-149
    m = [mk(100, 300, 1000, 1200, 0.85), mk(150, 350, 2000, 2200, 0.90), mk(200, 400, 3000, 3200, 0.95)]
    assert 2 in orc.plane_sweep_query(m, 1, 0.95, IDENTITY)


def test_sr_contained():  # :152-184
    m = [mk(100, 500, 1000, 1400, 0.80), mk(200, 300, 2000, 2100, 0.99)]
    assert 1 in orc.plane_sweep_query(m, 1, 0.95, IDENTITY)
    assert 0 in orc.plane_sweep_query(m, 1, 0.95, LENGTH)
    assert 0 in orc.plane_sweep_query(m, 1, 0.95, LLI)


def test_sr_ranking_order():  # :187-207
    m = [mk(100, 200, 1000, 1100, 0.70), mk(100, 250, 2000, 2150, 0.80), mk(100, 300, 3000, 3200, 0.90),
         mk(100, 180, 4000, 4080, 0.99), mk(100, 220, 5000, 5120, 0.60)]
    assert sorted(orc.plane_sweep_query(m, 2, 0.95, LENGTH_IDENTITY)) == [1, 2]


def test_sr_extreme_values():  # :210-241
    m = [mk(100, 101, 1000, 1001, 1.00), mk(100, 100100, 2000, 102000, 0.01), mk(100, 1100, 3000, 4000, 0.50)]
    assert orc.plane_sweep_query(m, 1, 0.95, LENGTH)[0] == 1
    assert orc.plane_sweep_query(m, 1, 0.95, IDENTITY)[0] == 0
    assert orc.plane_sweep_query(m, 1, 0.95, LLI)[0] == 2


# ----------------------------------------------------------------------------------------
# src/plane_sweep_scaffold.rs:292-371
# ----------------------------------------------------------------------------------------
def test_scaffold_no_overlap():  # :292-328
    c = [("chr1", "chr1", 0, 1000, 0, 1000, 0.95), ("chr1", "chr1", 2000, 3000, 2000, 3000, 0.95)]
    assert len(orc.plane_sweep_scaffolds(c, ONE_TO_ONE, 1, 1, 0.5)) == 2


def test_scaffold_overlapping_keeps_best():  # :331-371
    c = [("chr1", "chr1", 0, 1000, 0, 1000, 0.90), ("chr1", "chr1", 900, 1900, 900, 1900, 0.98)]
    kept = orc.plane_sweep_scaffolds(c, ONE_TO_ONE, 1, 1, 0.95)
    assert 1 <= len(kept) <= 2
    if len(kept) == 1:
        assert kept[0] == 1


# ----------------------------------------------------------------------------------------
# tests/test_plane_sweep_symmetry.rs (plane_sweep_core)
# ----------------------------------------------------------------------------------------
def _iv(pairs):
    return [(b, e, float(e - b)) for b, e in pairs]


def test_core_symmetry_simple():  # :16-58
    maps = [(100, 200, 300, 400), (150, 250, 350, 450), (300, 400, 100, 200)]
    q = orc.plane_sweep_core(_iv([(a, b) for a, b, _, _ in maps]), 1, 0.95)
    t = orc.plane_sweep_core(_iv([(c, d) for _, _, c, d in maps]), 1, 0.95)
    assert len(q) == 2 and len(t) == 2


def test_core_symmetry_transposed():  # :61-104
    original = [(100, 500, 1000, 1400), (200, 400, 1100, 1300), (600, 900, 1500, 1800)]
    q = orc.plane_sweep_core(_iv([(a, b) for a, b, _, _ in original]), 1, 0.95)
    t = orc.plane_sweep_core(_iv([(c, d) for _, _, c, d in original]), 1, 0.95)
    assert sorted(q) == sorted(t)


# ----------------------------------------------------------------------------------------
# src/union_find.rs ordering (tests/test_binary_search_optimization.rs:222-226 expectation)
# ----------------------------------------------------------------------------------------
def test_union_find_get_sets_order():
    assert orc.union_find_sets(5, [(0, 1), (1, 2), (3, 4)]) == [[0, 1, 2], [3, 4]]
    assert orc.union_find_sets(3, []) == [[0], [1], [2]]  # :279-327 no merging
    assert orc.union_find_sets(4, [(0, 1), (1, 2), (2, 3)]) == [[0, 1, 2, 3]]  # :330-384 all merged


# ----------------------------------------------------------------------------------------
# src/pansn.rs tests (round_nice / clamp_scaffold_params) :306-342
# ----------------------------------------------------------------------------------------
def test_round_nice_steps():
    L = orc.lib()
    assert [L.orc_round_nice(v) for v in (0, 120, 480, 950, 2900, 7200)] == [0, 100, 500, 1000, 3000, 7000]


def test_clamp_scaffold_params():
    assert orc.clamp_scaffold_params(50_000, 10_000, 1000, False) == (50_000, 10_000)
    assert orc.clamp_scaffold_params(50_000, 10_000, None, True) == (50_000, 10_000)
    assert orc.clamp_scaffold_params(50_000, 10_000, 1000, True) == (10_000, 600)
    assert orc.clamp_scaffold_params(5_000, 3_000, 1_000_000, True) == (5_000, 3_000)


# ----------------------------------------------------------------------------------------
# src/main.rs:244-293 parse_filter_mode; src/cli.rs:26-61, 76-130
# ----------------------------------------------------------------------------------------
def test_parse_filter_mode():
    assert orc.parse_filter_mode("1:1") == (ONE_TO_ONE, 1, 1)
    for s in ("1", "1:∞", "1:infinity", "1:many"):
        assert orc.parse_filter_mode(s) == (ONE_TO_MANY, 1, None)
    for s in ("∞:1", "many:1"):
        assert orc.parse_filter_mode(s) == (MANY_TO_MANY, None, 1)
    for s in ("many:many", "∞:∞", "many", "∞", "-1", "-1:-1", "MANY:MANY"):
        assert orc.parse_filter_mode(s) == (MANY_TO_MANY, None, None)
    assert orc.parse_filter_mode("10:5") == (MANY_TO_MANY, 10, 5)
    assert orc.parse_filter_mode("2:many") == (MANY_TO_MANY, 2, None)
    assert orc.parse_filter_mode("0:3") == (MANY_TO_MANY, None, 3)  # 0 rejected -> None
    assert orc.parse_filter_mode("N:N") == (MANY_TO_MANY, None, None)  # both sides unparsable
    assert orc.parse_filter_mode("1:2:3") == (ONE_TO_ONE, 1, 1)
    assert orc.parse_filter_mode("5") == (ONE_TO_MANY, 5, None)
    assert orc.parse_filter_mode("0") is None  # reference exits the process
    assert orc.parse_filter_mode("none") == (ONE_TO_ONE, 1, 1)  # garbage -> 1:1 fallback


def test_parse_metric_number():
    assert orc.parse_metric_number("50k") == 50_000
    assert orc.parse_metric_number("10K") == 10_000
    assert orc.parse_metric_number("1.5m") == 1_500_000
    assert orc.parse_metric_number("2G") == 2_000_000_000
    assert orc.parse_metric_number("123") == 123
    assert orc.parse_metric_number("") is None
    assert orc.parse_metric_number("5x") is None


def test_parse_identity_value():
    assert orc.parse_identity_value("0.9") == 0.9
    assert orc.parse_identity_value("90") == 0.9
    assert orc.parse_identity_value("0") == 0.0
    assert orc.parse_identity_value("ani50") is None  # needs the ANI pre-pass (out of scope)
    assert orc.parse_identity_value("abc") is None


# ----------------------------------------------------------------------------------------
# CLI-level fixtures: inline PAFs of the reference's binary-invoking tests, replayed through
# oracle/sweepga-ref with the flags the tests pass.
# ----------------------------------------------------------------------------------------
def run_ref(paf_text, *flags):
    exe = os.path.join(ROOT, "oracle", "sweepga-ref")
    with tempfile.TemporaryDirectory() as d:
        inp = os.path.join(d, "in.paf")
        out = os.path.join(d, "out.paf")
        with open(inp, "w") as f:
            f.write(paf_text)
        subprocess.check_call([exe, inp, "--output-file", out, *flags])
        with open(out) as f:
            return f.read()


CG = "\t60\tNM:i:0\tcg:Z:"


def test_cli_mapping_plane_sweep_across_targets():  # tests/test_mapping_plane_sweep.rs:8-58
    paf = ("genome1#chrA\t100000\t10000\t20000\t+\tgenome2#chrA\t100000\t10000\t20000\t9500\t10000\t60\tNM:i:500\tcg:Z:9500=500X\n"
           "genome1#chrA\t100000\t12000\t18000\t+\tgenome2#chrB\t100000\t12000\t18000\t5400\t6000\t60\tNM:i:600\tcg:Z:5400=600X\n")
    out = run_ref(paf, "--num-mappings", "1:1", "--scaffold-jump", "0", "--min-aln-identity", "0", "--overlap", "0.5")
    assert "genome2#chrA" in out and "genome2#chrB" not in out


def test_cli_mapping_plane_sweep_target_axis():  # tests/test_mapping_plane_sweep.rs:61-102
    paf = ("genome1#chrA\t100000\t10000\t20000\t+\tgenome2#chrX\t100000\t10000\t20000\t9500\t10000\t60\tNM:i:500\tcg:Z:9500=500X\n"
           "genome1#chrB\t100000\t10000\t20000\t+\tgenome2#chrX\t100000\t12000\t22000\t9800\t10000\t60\tNM:i:200\tcg:Z:9800=200X\n")
    out = run_ref(paf, "--num-mappings", "1:1", "--scaffold-jump", "0", "--min-aln-identity", "0", "--overlap", "0.5")
    assert "genome1#chrB" in out and "genome1#chrA" not in out


def _l(q, qs, qe, t, ts, te, m, b):
    return f"{q}\t100000\t{qs}\t{qe}\t+\t{t}\t100000\t{ts}\t{te}\t{m}\t{b}\t60\tNM:i:{b - m}\tcg:Z:{m}={b - m}X\n"


SCAF_FLAGS = ("--scaffold-mass", "1000", "--scaffold-jump", "10000", "--min-aln-identity", "0", "--scaffold-filter", "1:1")


def test_cli_overlapping_scaffolds_same_pair():  # tests/test_scaffold_plane_sweep_filtering.rs:7-56
    paf = (_l("chr1", 10000, 15000, "target_chr1", 10000, 15000, 4750, 5000)
           + _l("chr1", 15000, 20000, "target_chr1", 15000, 20000, 4750, 5000)
           + _l("chr1", 12000, 17000, "target_chr1", 30000, 35000, 4900, 5000)
           + _l("chr1", 17000, 22000, "target_chr1", 35000, 40000, 4900, 5000))
    out = run_ref(paf, *SCAF_FLAGS, "--scaffold-dist", "0")
    assert "12000\t17000" in out or "17000\t22000" in out
    assert "10000\t15000" not in out and "15000\t20000" not in out


def test_cli_overlapping_scaffolds_different_targets():  # :58-118
    paf = (_l("chr1", 10000, 15000, "target_chr1", 10000, 15000, 4750, 5000)
           + _l("chr1", 15000, 20000, "target_chr1", 15000, 20000, 4750, 5000)
           + _l("chr1", 10000, 15000, "target_chr2", 10000, 15000, 4900, 5000)
           + _l("chr1", 15000, 20000, "target_chr2", 15000, 20000, 4900, 5000))
    out = run_ref(paf, *SCAF_FLAGS)
    assert "target_chr1" in out and "target_chr2" in out


def test_cli_contained_scaffold():  # :120-169
    paf = (_l("chr1", 15000, 18000, "target_chr1", 15000, 18000, 2940, 3000)
           + _l("chr1", 10000, 17500, "target_chr1", 10000, 17500, 7125, 7500)
           + _l("chr1", 17500, 25000, "target_chr1", 17500, 25000, 7125, 7500))
    out = run_ref(paf, *SCAF_FLAGS, "--scaffold-dist", "0")
    assert ("10000\t17500" in out or "17500\t25000" in out) and "15000\t18000" not in out


def test_cli_scaffolds_different_query_chromosomes():  # :171-224
    paf = (_l("query_chr1", 10000, 15000, "target_chr1", 10000, 15000, 4750, 5000)
           + _l("query_chr1", 15000, 20000, "target_chr1", 15000, 20000, 4750, 5000)
           + _l("query_chr2", 10000, 15000, "target_chr1", 10000, 15000, 4900, 5000)
           + _l("query_chr2", 15000, 20000, "target_chr1", 15000, 20000, 4900, 5000))
    out = run_ref(paf, *SCAF_FLAGS)
    assert "query_chr1" in out and "query_chr2" in out


def test_cli_scaffold_length_filtering():  # tests/test_scaffold_length_filter.rs:6-77
    paf = ""
    for i in range(10):
        s = 10000 + i * 2000
        paf += f"query1\t100000\t{s}\t{s + 1000}\t+\ttarget\t100000\t{s}\t{s + 1000}\t950\t1000\t60\tNM:i:50\tcg:Z:950=50X\n"
    for i in range(5):
        s = 50000 + i * 2000
        paf += f"query2\t100000\t{s}\t{s + 1000}\t+\ttarget\t100000\t{s}\t{s + 1000}\t950\t1000\t60\tNM:i:50\tcg:Z:950=50X\n"
    out = run_ref(paf, "--scaffold-mass", "10000", "--scaffold-jump", "10000", "--min-aln-identity", "0")
    lines = [x for x in out.split("\n") if x]
    assert len(lines) == 10 and all(x.startswith("query1") for x in lines)


def test_cli_scaffold_aligned_mass_filtering():  # tests/test_scaffold_length_filter.rs:79-126
    paf = ("query\t150000\t0\t1000\t+\ttarget\t150000\t0\t1000\t950\t1000\t60\tNM:i:50\tcg:Z:950=50X\n"
           "query\t150000\t99000\t100000\t+\ttarget\t150000\t99000\t100000\t950\t1000\t60\tNM:i:50\tcg:Z:950=50X\n")
    out = run_ref(paf, "--scaffold-mass", "50000", "--scaffold-jump", "100000", "--min-aln-identity", "0")
    assert len([x for x in out.split("\n") if x]) == 2


GP1 = ("A#1#chr1\t1000\t0\t500\t+\tB#1#chr1\t1000\t0\t500\t450\t500\t60\tcg:Z:500M\n"
       "A#1#chr1\t1000\t0\t500\t+\tC#1#chr1\t1000\t0\t500\t400\t500\t60\tcg:Z:500M\n"
       "A#1#chr1\t1000\t0\t500\t+\tD#1#chr1\t1000\t0\t500\t350\t500\t60\tcg:Z:500M\n")
GP2 = ("A#1#chr1\t1000\t0\t500\t+\tB#1#chr1\t1000\t0\t500\t450\t500\t60\tcg:Z:500M\n"
       "A#1#chr1\t1000\t0\t500\t+\tB#1#chr2\t1000\t0\t500\t400\t500\t60\tcg:Z:500M\n"
       "A#1#chr2\t1000\t0\t500\t+\tB#1#chr1\t1000\t0\t500\t350\t500\t60\tcg:Z:500M\n")


def test_cli_genome_pairs_preserved():  # tests/test_genome_pair_grouping.rs:13-60
    out = run_ref(GP1, "--scaffold-jump", "0")
    assert len([x for x in out.split("\n") if x]) == 3
    # also with an explicit 1:1: one per genome pair survives
    out = run_ref(GP1, "--scaffold-jump", "0", "--num-mappings", "1:1")
    assert len([x for x in out.split("\n") if x]) == 3


def test_cli_within_genome_pair():  # tests/test_genome_pair_grouping.rs:62-113 (needs explicit 1:1, App. C)
    out = run_ref(GP2, "--scaffold-jump", "0", "--num-mappings", "1:1")
    lines = [x for x in out.split("\n") if x]
    assert len(lines) == 1 and lines[0].startswith("A#1#chr1\t1000\t0\t500\t+\tB#1#chr1")
    assert lines[0].endswith("\tst:Z:unassigned")


def test_cli_grouping_bug_inputs():  # tests/test_grouping_bug.rs:10-16,102-107 with current flags
    paf = ("chrI_query\t10000\t1000\t2000\t+\tchrI_target1\t10000\t1000\t2000\t1000\t1000\t60\tcg:Z:1000M\n"
           "chrI_query\t10000\t1000\t2000\t+\tchrII_target2\t10000\t2000\t3000\t1000\t1000\t60\tcg:Z:1000M\n"
           "chrI_query\t10000\t1000\t2000\t+\tchrIII_target3\t10000\t3000\t4000\t1000\t1000\t60\tcg:Z:1000M\n"
           "chrII_query\t15000\t2000\t3000\t+\tchrI_target1\t10000\t2000\t3000\t1000\t1000\t60\tcg:Z:1000M\n"
           "chrII_query\t15000\t2000\t3000\t+\tchrII_target2\t10000\t4000\t5000\t1000\t1000\t60\tcg:Z:1000M\n")
    out = run_ref(paf, "--num-mappings", "1", "--scaffold-jump", "0")
    assert len([x for x in out.split("\n") if x]) == 5
    paf2 = ("query1\t5000\t1000\t2000\t+\ttarget_A\t10000\t3000\t4000\t1000\t1000\t60\tcg:Z:1000M\n"
            "query1\t5000\t1000\t2000\t+\ttarget_B\t10000\t5000\t6000\t1000\t1000\t60\tcg:Z:1000M\n"
            "query1\t5000\t1000\t2000\t+\ttarget_C\t10000\t7000\t8000\t1000\t1000\t60\tcg:Z:1000M\n"
            "query1\t5000\t1000\t2000\t+\ttarget_D\t10000\t1000\t2000\t1000\t1000\t60\tcg:Z:1000M\n")
    out = run_ref(paf2, "--num-mappings", "1", "--scaffold-jump", "0")
    assert len([x for x in out.split("\n") if x]) == 4


# tests/test_chaining_stability.rs:147-350 (PafFilter::filter_paf with an explicit FilterConfig)
CHAIN_CFG = orc.Config(min_block_length=0, mapping_filter_mode=MANY_TO_MANY, scaffold_filter_mode=MANY_TO_MANY,
                       overlap_threshold=0.0, scaffold_gap=10_000, min_scaffold_length=0,
                       scaffold_overlap_threshold=0.0, scaffold_max_deviation=20_000,
                       scoring_function=LLI, min_identity=0.0, min_scaffold_identity=0.0)


def _chains_of(out):
    chains = {}
    for line in out.split("\n"):
        f = line.split("\t")
        if len(f) < 13:
            continue
        cid = next((x[5:] for x in f[12:] if x.startswith("ch:Z:")), None)
        if cid:
            chains.setdefault(cid, []).append(f"{f[0]}:{f[2]}-{f[3]}")
    return chains


def _filter_text(cfg, text):
    with tempfile.TemporaryDirectory() as d:
        inp, out = os.path.join(d, "i.paf"), os.path.join(d, "o.paf")
        with open(inp, "w") as f:
            f.write(text)
        orc.filter_paf(cfg, inp, out)
        with open(out) as f:
            return f.read()


def test_nearest_neighbor_chaining():  # :147-240
    paf = ("querySeq\t10000\t0\t1000\t+\ttargetSeq\t10000\t0\t1000\t950\t1000\t60\n"
           "querySeq\t10000\t1100\t2100\t+\ttargetSeq\t10000\t1100\t2100\t950\t1000\t60\n"
           "querySeq\t10000\t5000\t6000\t+\ttargetSeq\t10000\t5000\t6000\t950\t1000\t60\n")
    chains = _chains_of(_filter_text(CHAIN_CFG, paf))
    assert len(chains) == 1
    (members,) = chains.values()
    assert len(members) == 3


def test_overlap_penalty():  # :243-350
    paf = ("querySeq\t10000\t0\t1000\t+\ttargetSeq\t10000\t0\t1000\t950\t1000\t60\n"
           "querySeq\t10000\t900\t1900\t+\ttargetSeq\t10000\t900\t1900\t950\t1000\t60\n"
           "querySeq\t10000\t1100\t2100\t+\ttargetSeq\t10000\t1100\t2100\t950\t1000\t60\n")
    chains = _chains_of(_filter_text(CHAIN_CFG, paf))
    assert chains
    a = next((c for c, m in chains.items() if any("0-1000" in x for x in m)), None)
    c_ = next((c for c, m in chains.items() if any("1100-2100" in x for x in m)), None)
    if a and c_:
        assert a == c_


def test_output_annotation_format():  # paf_filter.rs:1708-1718
    paf = ("q\t10000\t0\t6000\t+\tt\t10000\t0\t6000\t5900\t6000\t60\n"
           "q\t10000\t6100\t12100\t+\tt\t20000\t6100\t12100\t5900\t6000\t60\n"
           "short\t9\n")
    out = _filter_text(orc.Config(), paf)
    lines = out.split("\n")
    assert lines[0].endswith("\t60\tch:Z:chain_1\tst:Z:scaffold")
    assert lines[1].endswith("\t60\tch:Z:chain_1\tst:Z:scaffold")
    assert lines[2] == ""
