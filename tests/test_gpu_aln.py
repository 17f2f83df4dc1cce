"""unified_filter::filter_file's .1aln branch (src/unified_filter.rs:310-317) on the GPU: records derived by swg_aln_open
from decoded alignments, filtered by swg_filter, against the oracle's records_from_1aln + apply_filters."""
import numpy as np
import pytest

from tests import orc
from tests.test_aln_cpu import oracle_records

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("seed", [1, 2])
def test_1aln_branch_matches_oracle(seed):
    import sweepga_amd as sw
    rng = np.random.default_rng(seed)
    n = 30_000
    heads = [f"g{g}#1#chr{c}" + str(rng.choice(["", " len=12345 circular", "\tdesc"])) for g in range(4) for c in range(3)]
    qi = rng.integers(0, len(heads), n)
    ti = np.where(rng.random(n) < 0.8, (qi + 3) % len(heads), rng.integers(0, len(heads), n))
    qn = [heads[i] for i in qi]
    tn = [heads[i] for i in ti]
    qs = rng.integers(0, 400_000, n)
    ql = np.minimum(np.exp(rng.normal(7.5, 1.0, n)).astype(np.int64) + 50, 20_000)
    ts = np.clip(qs + rng.normal(0, 3000, n).astype(np.int64), 0, 400_000)
    tl = np.maximum(ql + rng.integers(-30, 30, n), 1)
    matches = (ql * rng.uniform(0.75, 1.0, n)).astype(np.int64)
    strand = "".join(rng.choice(["+", "-"], n, p=[0.8, 0.2]))
    want_q, want_t, block, ident = oracle_records(qn, tn, qs, qs + ql, ts, ts + tl, matches, strand)
    u = lambda a: np.ascontiguousarray(np.asarray(a, dtype=np.uint64))
    rec = orc.Records(want_q, want_t, u(qs), u(qs + ql), u(ts), u(ts + tl), u(block), np.ascontiguousarray(ident), u(matches),
                      np.array([ord(c) for c in strand], dtype=np.uint8), u(np.arange(n)))
    with sw.AlnRecords(qn, tn, qs, qs + ql, ts, ts + tl, matches, strand) as a:
        packed = a.packed()
    for kw, okw in ((dict(), dict()),
                    (dict(mapping_filter_mode=sw.FilterMode.OneToOne, scaffold_gap=0), dict(mapping_filter_mode=orc.ONE_TO_ONE, scaffold_gap=0)),
                    (dict(mapping_filter_mode=sw.FilterMode.OneToOne, scaffold_filter_mode=sw.FilterMode.OneToOne, scaffold_gap=20_000,
                          min_scaffold_length=5_000, scaffold_max_deviation=10_000),
                     dict(mapping_filter_mode=orc.ONE_TO_ONE, scaffold_filter_mode=orc.ONE_TO_ONE, scaffold_gap=20_000,
                          min_scaffold_length=5_000, scaffold_max_deviation=10_000))):
        st, ch = sw.PafFilter(sw.FilterConfig(**kw)).filter_columns(packed)
        ost, och = orc.apply_filters(orc.Config(**okw), rec)
        assert np.array_equal(st, ost) and np.array_equal(ch, och)
