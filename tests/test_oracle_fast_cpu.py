"""The oracle's indexed evaluation of step 4b (inversion capture, src/paf_filter.rs:535-597) and of the rescue loop
(:686-718) against its literal chains x reverse-mappings / mappings x anchors loops.  The indexed form exists only so that BASELINE.json configs[2] (10^7 mappings in one
chromosome pair: 1.4 * 10^6 kept '+' chains x 10^6 '-' mappings) can be checked at full size (tools/sbig1_full_parity.py);
every other parity test runs the literal loop."""
import numpy as np
import pytest

from tests import gen, orc


@pytest.fixture(autouse=True)
def _literal_afterwards():
    yield
    orc.set_fast_inversion(False)


def _both(cfg, rec):
    orc.set_fast_inversion(False)
    a = orc.apply_filters(cfg, rec)
    orc.set_fast_inversion(True)
    b = orc.apply_filters(cfg, rec)
    orc.set_fast_inversion(False)
    return a, b


@pytest.mark.parametrize("seed", range(6))
def test_indexed_inversion_capture_equals_the_literal_loop(seed):
    rng = np.random.default_rng(1000 + seed)
    captured = 0
    for case in range(120):
        n = int(rng.choice([30, 300, 3000]))
        span = int(rng.choice([20_000, 200_000, 3_000_000]))
        rec = gen.random_records(rng, n, n_genomes=int(rng.integers(1, 4)), chrs_per_genome=int(rng.integers(1, 3)), span=span,
                                 max_len=int(rng.choice([300, 3000, 20000])), minus_frac=float(rng.choice([0.2, 0.5, 0.8])),
                                 syntenic_frac=float(rng.choice([0.7, 0.98])))
        cfg = orc.Config(scaffold_gap=int(rng.choice([500, 10_000, 50_000, 10_000_000])),
                         min_scaffold_length=int(rng.choice([0, 1000, 10_000])),
                         scaffold_max_deviation=int(rng.choice([0, 1, 300, 2000, 20_000, 5_000_000])),
                         mapping_filter_mode=int(rng.choice([orc.ONE_TO_ONE, orc.MANY_TO_MANY])),
                         scaffold_filter_mode=int(rng.choice([orc.ONE_TO_ONE, orc.MANY_TO_MANY])))
        (st_a, ch_a), (st_b, ch_b) = _both(cfg, rec)
        assert np.array_equal(st_a, st_b) and np.array_equal(ch_a, ch_b), (seed, case)
        minus = rec.strand == ord("-")
        captured += int(((st_a == orc.SCAFFOLD) & minus).sum())
    assert captured > 0   # the step was exercised ('-' records that ended as scaffold members or captured inversions)


def test_indexed_inversion_capture_on_a_deep_pair():
    """One dense chromosome pair (the shape the indexed form is for), 40,000 mappings at depth ~40."""
    rng = np.random.default_rng(7)
    rec = gen.random_records(rng, 40_000, n_genomes=2, chrs_per_genome=1, span=2_000_000, max_len=5000, minus_frac=0.1,
                             syntenic_frac=0.7, self_frac=0.0)
    (st_a, ch_a), (st_b, ch_b) = _both(orc.Config(), rec)
    assert np.array_equal(st_a, st_b) and np.array_equal(ch_a, ch_b)
    assert int((st_a != 0).sum()) > 1000
    # the full flag set of BASELINE.json configs[4]: 1:1 sweeps and rescue (the indexed rescue loop)
    cfg = orc.Config(mapping_filter_mode=orc.ONE_TO_ONE, scaffold_filter_mode=orc.ONE_TO_ONE, scaffold_max_deviation=20_000)
    (st_a, ch_a), (st_b, ch_b) = _both(cfg, rec)
    assert np.array_equal(st_a, st_b) and np.array_equal(ch_a, ch_b)
    assert int((st_a == orc.RESCUED).sum()) > 100
