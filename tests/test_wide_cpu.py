"""The argument behind swg_filter64 / the rebasing front ends (sweepga_amd/csrc/host/rebase.h), checked on the oracle:
apply_filters gives the same status and chain numbers when every sequence's coordinates are moved by a constant of its
own -- so records whose coordinates exceed 2^32 can be filtered in the 32-bit device layout after subtracting each
sequence's smallest coordinate.  Also: the host rebasing of the .1aln front end."""
import numpy as np
import pytest

from tests import gen, orc
from tests.test_gpu_scaffold import SCAFFOLD_CFGS


def _ocfg(cfg_i):
    return orc.Config(**SCAFFOLD_CFGS[cfg_i])


@pytest.mark.parametrize("cfg_i", range(len(SCAFFOLD_CFGS)))
@pytest.mark.parametrize("seed", range(3))
def test_oracle_is_invariant_under_per_sequence_shifts(seed, cfg_i):
    rng = np.random.default_rng(7000 + 10 * cfg_i + seed)
    n = int(rng.choice([50, 1_000, 6_000]))
    # small spans: many records sit within the rescue / chaining / inversion distances of coordinate 0, where the
    # reference's saturating window starts clamp (src/paf_filter.rs:574-592)
    rec0 = gen.random_records(rng, n, n_genomes=int(rng.integers(1, 4)), chrs_per_genome=int(rng.integers(1, 3)),
                              span=int(rng.choice([20_000, 300_000])), minus_frac=0.3)
    rec, off = gen.shifted(rec0, rng)
    ocfg = _ocfg(cfg_i)
    for keep_self, scaffolds_only in ((False, False), (True, True)):
        ocfg.keep_self, ocfg.scaffolds_only = keep_self, scaffolds_only
        st0, ch0 = orc.apply_filters(ocfg, rec0)
        st1, ch1 = orc.apply_filters(ocfg, rec)
        assert np.array_equal(st0, st1) and np.array_equal(ch0, ch1), (max(off.values()), int((st0 != st1).sum()))


@pytest.mark.parametrize("cfg_i", range(len(SCAFFOLD_CFGS)))
@pytest.mark.parametrize("seed", range(3))
def test_oracle_is_invariant_under_shifts_per_sweep_segment(seed, cfg_i):
    """The finer partition (rebase.h, columns_by_axis): query coordinates moved by a constant per (query sequence, genome of the
    target), target coordinates per (target sequence, genome of the query).  No step of apply_filters compares coordinates
    across those segments -- the fallback for a sequence that is touched over 2^32 bases or more."""
    rng = np.random.default_rng(7300 + 10 * cfg_i + seed)
    n = int(rng.choice([50, 1_000, 6_000]))
    rec0 = gen.random_records(rng, n, n_genomes=int(rng.integers(2, 5)), chrs_per_genome=int(rng.integers(1, 3)),
                              span=int(rng.choice([20_000, 300_000])), minus_frac=0.3)
    rec = gen.shifted_by_axis(rec0, rng)
    ocfg = _ocfg(cfg_i)
    for keep_self, scaffolds_only in ((False, False), (True, True)):
        ocfg.keep_self, ocfg.scaffolds_only = keep_self, scaffolds_only
        st0, ch0 = orc.apply_filters(ocfg, rec0)
        st1, ch1 = orc.apply_filters(ocfg, rec)
        assert np.array_equal(st0, st1) and np.array_equal(ch0, ch1), int((st0 != st1).sum())


def test_aln_front_end_rebases_wide_coordinates():
    from sweepga_amd import AlnRecords, SwgError
    qn = ["a x", "a", "b", "b"]
    tn = ["b", "c", "a", "c"]
    qs = np.array([2**33 + 100, 2**33 + 5, 2**40, 2**40 + 9], dtype=np.uint64)
    ql = np.array([1000, 10, 7, 0], dtype=np.uint64)
    ts = np.array([2**40 + 50, 3, 2**33 + 77, 2**32 - 10], dtype=np.uint64)
    tl = np.array([900, 20, 7, 5], dtype=np.uint64)
    with AlnRecords(qn, tn, qs, qs + ql, ts, ts + tl, [950, 9, 7, 0], "+-++") as a:
        names = a.names
        assert names == ["a", "b", "c"]
        off = a.seq_offsets
        assert [int(x) for x in off] == [2**33 + 5, 2**40, 3]           # smallest coordinate of a, b, c anywhere
        assert [int(x) for x in a.column("q_start")] == [95, 0, 0, 9]
        assert [int(x) for x in a.column("q_end")] == [1095, 10, 7, 9]
        assert [int(x) for x in a.column("t_start")] == [50, 0, 72, 2**32 - 13]
        assert [int(x) for x in a.column("t_end")] == [950, 20, 79, 2**32 - 8]
        assert [int(x) for x in a.column("block_len")] == [1900, 30, 14, 5]
        assert list(a.column("identity")) == [950 / 1000, 9 / 10, 1.0, 0.0]
    with pytest.raises(SwgError, match="sequence c spans 2\\^32 bases or more"):
        AlnRecords(qn, tn, qs, qs + ql, ts, ts + tl + np.array([0, 0, 0, 2**32], dtype=np.uint64), [950, 9, 7, 0], "+-++")
    with AlnRecords(["a"], ["b"], [5], [9], [1], [2], [3], "+") as a:
        assert a.seq_offsets is None
