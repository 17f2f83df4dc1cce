"""N > 1 path on the CPU: two gloo processes shard one record set by genome pair, each filters its shard
(with the CPU oracle standing in for the per-rank GPU call -- the sharding / renumbering / gather logic is
what is under test), results are all-gathered and must equal the unsharded filter, chain numbers included."""
import os
import socket

import numpy as np
import pytest

from tests import gen, orc


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _oracle_filter_fn(ocfg):
    def fn(packed):
        names = packed.index.names
        c = packed.cols
        u = lambda a: np.ascontiguousarray(a.astype(np.uint64))
        rec = orc.Records([names[i] for i in c["q_id"]], [names[i] for i in c["t_id"]], u(c["q_start"]), u(c["q_end"]),
                          u(c["t_start"]), u(c["t_end"]), u(c["block_len"]), np.ascontiguousarray(c["identity"]),
                          u(c["matches"]), np.where(c["strand"] == 0, ord("+"), ord("-")).astype(np.uint8),
                          u(np.arange(packed.n)))
        return orc.apply_filters(ocfg, rec)
    return fn


CFGS = [dict(mapping_filter_mode=orc.ONE_TO_ONE, scaffold_gap=0),
        dict(scaffold_gap=20_000, min_scaffold_length=3_000, scaffold_filter_mode=orc.ONE_TO_ONE, scaffold_max_deviation=15_000),
        dict(mapping_filter_mode=orc.ONE_TO_ONE, scaffold_gap=10_000, min_scaffold_length=1_000, min_block_length=100,
             min_identity=0.75)]


def _worker(rank, world, port, seed, out_dir, n_genomes=4):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import sweepga_amd as sw
        from sweepga_amd import shard
        rng = np.random.default_rng(seed)
        rec = gen.random_records(rng, 6000, n_genomes=n_genomes, chrs_per_genome=2, span=300_000)
        packed = sw.pack_records(gen.records_to_meta(rec))

        def all_gather(obj):
            outs = [None] * world
            dist.all_gather_object(outs, obj)
            return outs

        for ci, kw in enumerate(CFGS):
            ocfg = orc.Config(**kw)
            st, ch = shard.filter_sharded(packed, _oracle_filter_fn(ocfg), rank, world, ocfg.min_block_length,
                                          ocfg.min_identity, ocfg.keep_self, all_gather)
            pl = shard.plan(packed, world)
            np.savez(os.path.join(out_dir, f"r{rank}_c{ci}.npz"), st=st, ch=ch, n_mine=int((pl.shard_of_record == rank).sum()),
                     sharded=pl.sharded)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("seed,world,n_genomes", [(11, 2, 4), (12, 2, 4), (13, 8, 6)])
def test_gloo_sharding_equals_unsharded(tmp_path, seed, world, n_genomes):
    import torch.multiprocessing as mp
    mp.spawn(_worker, args=(world, _free_port(), seed, str(tmp_path), n_genomes), nprocs=world, join=True)
    rng = np.random.default_rng(seed)
    rec = gen.random_records(rng, 6000, n_genomes=n_genomes, chrs_per_genome=2, span=300_000)
    for ci, kw in enumerate(CFGS):
        want_st, want_ch = orc.apply_filters(orc.Config(**kw), rec)
        parts = [np.load(tmp_path / f"r{r}_c{ci}.npz") for r in range(world)]
        assert all(bool(p["sharded"]) for p in parts)
        assert all(int(p["n_mine"]) > 0 for p in parts)             # both ranks really had work
        assert sum(int(p["n_mine"]) for p in parts) == len(rec)
        for p in parts:                                             # every rank ends with the global result
            assert np.array_equal(p["st"], want_st)
            assert np.array_equal(p["ch"], want_ch), (ci, int((p["ch"] != want_ch).sum()))


def test_plan_is_balanced_and_refuses_nonconforming_names():
    import sweepga_amd as sw
    from sweepga_amd import shard
    rng = np.random.default_rng(5)
    rec = gen.random_records(rng, 20_000, n_genomes=6, chrs_per_genome=2)
    packed = sw.pack_records(gen.records_to_meta(rec))
    pl = shard.plan(packed, 4)
    loads = np.bincount(pl.shard_of_record, minlength=4)
    assert pl.sharded and loads.min() > 0 and loads.max() < 1.5 * loads.mean()
    # a genome pair never straddles shards
    for p in range(pl.n_pairs):
        assert len(set(pl.shard_of_record[pl.pair_of_record == p].tolist())) == 1
    # names with 2 or >= 4 '#'-parts: the two prefix rules disagree -> no sharding (everything on rank 0)
    meta = gen.records_to_meta(rec)
    meta[0].query_name = "x#y#z#w"
    pl2 = shard.plan(sw.pack_records(meta), 4)
    assert not pl2.sharded and set(pl2.shard_of_record.tolist()) == {0}


def test_lpt_balance_on_the_span_size_distribution():
    """S-pan: 9,900 genome pairs, lognormal(0.5) sizes summing to 10^8 -> LPT imbalance far below 5 % up
    to 8 shards; plan_dense (the 10^8-record path of bench.py --scaling strong) agrees with plan()."""
    from sweepga_amd import shard
    rng = np.random.default_rng(2025)
    w = np.exp(0.5 * rng.standard_normal(9900))
    sizes = np.floor(w / w.sum() * 1e8).astype(np.int64)
    for world in (2, 4, 8):
        s = shard.lpt(sizes, world)
        loads = np.bincount(s, weights=sizes, minlength=world)
        assert loads.min() > 0 and loads.max() / loads.mean() < 1.05
    import sweepga_amd as sw
    rec = gen.random_records(np.random.default_rng(3), 20_000, n_genomes=7, chrs_per_genome=2)
    packed = sw.pack_records(gen.records_to_meta(rec))
    pl = shard.plan(packed, 4)
    shard_of_key, counts = shard.plan_dense(packed.cols["q_id"], packed.cols["t_id"], packed.seq_genome_two, packed.n_genome_two, 4)
    g2 = packed.seq_genome_two.astype(np.int64)
    key = g2[packed.cols["q_id"]] * packed.n_genome_two + g2[packed.cols["t_id"]]
    assert counts.sum() == packed.n
    loads_a = np.sort(np.bincount(pl.shard_of_record, minlength=4))
    loads_b = np.sort(np.bincount(shard_of_key[key], minlength=4))
    assert np.array_equal(loads_a, loads_b)
    for k in np.unique(key):   # a genome pair never straddles shards
        assert len(set(shard_of_key[key[key == k]].tolist())) == 1


@pytest.mark.parametrize("world", [2, 8])
def test_local_renumbering_equals_global_merge(world):
    """What bench.py --scaling strong does: every shard computes (lo, hi, first) per genome pair from ITS records only,
    the three vectors are combined with min / max / min (an all_reduce in production) and every shard shifts its own
    chain numbers -> the reference's global numbering (src/paf_filter.rs:517-521), without gathering the records."""
    import sweepga_amd as sw
    from sweepga_amd import shard
    rng = np.random.default_rng(77)
    rec = gen.random_records(rng, 8000, n_genomes=6, chrs_per_genome=2, span=300_000)
    packed = sw.pack_records(gen.records_to_meta(rec))
    ocfg = orc.Config(scaffold_gap=20_000, min_scaffold_length=3_000, scaffold_filter_mode=orc.ONE_TO_ONE, scaffold_max_deviation=15_000)
    want_st, want_ch = orc.apply_filters(ocfg, rec)
    G = packed.n_genome_two
    shard_of_key, _ = shard.plan_dense(packed.cols["q_id"], packed.cols["t_id"], packed.seq_genome_two, G, world)
    g2 = packed.seq_genome_two.astype(np.int64)
    key = g2[packed.cols["q_id"]] * G + g2[packed.cols["t_id"]]
    fn = _oracle_filter_fn(ocfg)
    retained = shard.retained_mask(packed, ocfg.min_block_length, ocfg.min_identity, ocfg.keep_self)
    parts, ranges = [], []
    for r in range(world):
        idx = np.nonzero(shard_of_key[key] == r)[0]
        if len(idx):
            st, ch = fn(shard.subset(packed, idx))
        else:
            st, ch = np.zeros(0, np.uint8), np.zeros(0, np.uint32)
        parts.append((idx, st, ch.astype(np.int64)))
        ranges.append(shard.pair_chain_ranges(ch.astype(np.int64), key[idx], idx, G * G, packed.n, retained[idx]))
    lo = np.min([x[0] for x in ranges], axis=0)
    hi = np.max([x[1] for x in ranges], axis=0)
    first = np.min([x[2] for x in ranges], axis=0)
    shift = shard.chain_shifts(lo, hi, first)
    got_st, got_ch = np.zeros(packed.n, np.uint8), np.zeros(packed.n, np.uint32)
    for idx, st, ch in parts:
        has = ch != 0
        ch[has] += shift[key[idx][has]]
        got_st[idx], got_ch[idx] = st, ch
    assert np.array_equal(got_st, want_st) and np.array_equal(got_ch, want_ch)


def test_subset_keeps_wide_columns():
    """Records with coordinates beyond 2^32 keep their u64 columns (and the `wide` mark that selects swg_filter64) through
    the sharding helpers; plan() only looks at the ids."""
    import sweepga_amd as sw
    from sweepga_amd import shard
    rng = np.random.default_rng(5)
    rec, _ = gen.shifted(gen.random_records(rng, 500, n_genomes=3, chrs_per_genome=2), rng)
    packed = sw.pack_records(gen.records_to_meta(rec))
    assert packed.wide
    pl = shard.plan(packed, 2)
    for r in range(2):
        idx = np.nonzero(pl.shard_of_record == r)[0]
        sub = shard.subset(packed, idx)
        assert sub.wide and sub.cols["q_end"].dtype == np.uint64 and np.array_equal(sub.cols["q_end"], rec.qe[idx])
