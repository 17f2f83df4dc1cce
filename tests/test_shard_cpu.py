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


def _worker(rank, world, port, seed, out_dir):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import sweepga_amd as sw
        from sweepga_amd import shard
        rng = np.random.default_rng(seed)
        rec = gen.random_records(rng, 6000, n_genomes=4, chrs_per_genome=2, span=300_000)
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


@pytest.mark.parametrize("seed", [11, 12])
def test_two_rank_gloo_sharding_equals_unsharded(tmp_path, seed):
    import torch.multiprocessing as mp
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), seed, str(tmp_path)), nprocs=world, join=True)
    rng = np.random.default_rng(seed)
    rec = gen.random_records(rng, 6000, n_genomes=4, chrs_per_genome=2, span=300_000)
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
