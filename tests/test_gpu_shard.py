"""Sharded filtering on the GPU: the two shards of a 2-way plan filtered one after the other on the one test
GPU (as two ranks would on two GPUs) and merged must equal the unsharded call, chain numbers included."""
import numpy as np
import pytest

from tests import gen

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_sharded_equals_unsharded(seed):
    import sweepga_amd as sw
    from sweepga_amd import shard
    rng = np.random.default_rng(seed)
    rec = gen.random_records(rng, 30_000, n_genomes=5, chrs_per_genome=2, span=500_000)
    packed = sw.pack_records(gen.records_to_meta(rec))
    cfg = sw.FilterConfig(mapping_filter_mode=sw.FilterMode.OneToOne, scaffold_gap=20_000, min_scaffold_length=3_000,
                          scaffold_filter_mode=sw.FilterMode.OneToOne, scaffold_max_deviation=15_000)
    f = sw.PafFilter(cfg)
    want_st, want_ch = f.filter_columns(packed)
    world = 2
    pl = shard.plan(packed, world)
    assert pl.sharded
    parts = []
    for r in range(world):
        idx = np.nonzero(pl.shard_of_record == r)[0]
        st, ch = f.filter_columns(shard.subset(packed, idx))
        parts.append((idx, st.copy(), ch.copy()))
    st, ch = shard.merge(packed, pl, parts, shard.retained_mask(packed, 0, 0.0, False))
    assert np.array_equal(st, want_st)
    assert np.array_equal(ch, want_ch)
