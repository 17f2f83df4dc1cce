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


CONFIGS = [
    dict(mapping_filter_mode="OneToOne", scaffold_gap=20_000, min_scaffold_length=3_000, scaffold_filter_mode="OneToOne",
         scaffold_max_deviation=15_000),
    dict(),                                                      # CLI defaults (many:many, 50 kb jump)
    dict(mapping_filter_mode="OneToOne", scaffold_gap=0),        # sweep only: no chain numbers to repair
    dict(mapping_filter_mode="OneToMany", mapping_max_per_query=2, scaffold_gap=10_000, min_scaffold_length=1_000,
         min_block_length=300, min_identity=0.8, scaffold_max_deviation=5_000),
]


@pytest.mark.parametrize("n_ctx", [2, 3, 5])
def test_native_multi_context_equals_single(n_ctx):
    """swg_filter_multi (C++ partition / host threads / merge) over several contexts -- all on the one test GPU --
    must reproduce swg_filter exactly, chain numbers included."""
    import sweepga_amd as sw
    rng = np.random.default_rng(50 + n_ctx)
    rec = gen.random_records(rng, 40_000, n_genomes=5, chrs_per_genome=2, span=500_000)
    packed = sw.pack_records(gen.records_to_meta(rec))
    ctxs = [sw.Context(0) for _ in range(n_ctx)]
    try:
        for kw in CONFIGS:
            kw = {k: (getattr(sw.FilterMode, v) if isinstance(v, str) else v) for k, v in kw.items()}
            f = sw.PafFilter(sw.FilterConfig(**kw), ctx=ctxs[0])
            for keep_self in (False, True):
                f.with_keep_self(keep_self)
                want_st, want_ch = f.filter_columns(packed)
                want = (f.last_stats.n_retained, f.last_stats.n_swept, f.last_stats.n_out)
                st, ch = f.filter_columns_multi(packed, ctxs)
                assert np.array_equal(st, want_st), kw
                assert np.array_equal(ch, want_ch), kw
                assert (f.last_stats.n_retained, f.last_stats.n_swept, f.last_stats.n_out) == want
                assert f.last_stats.n_in == packed.n
    finally:
        for c in ctxs:
            c.close()


def test_native_multi_falls_back_when_prefix_rules_disagree():
    """Names with three '#': prefix-to-last-'#' and first-two-parts differ -> one device, still exact."""
    import sweepga_amd as sw
    rng = np.random.default_rng(9)
    rec = gen.random_records(rng, 5_000, n_genomes=3, chrs_per_genome=2, span=200_000)
    rec.qname = [q.replace("#chr", "#x#chr") for q in rec.qname]
    rec.tname = [t.replace("#chr", "#x#chr") if i % 2 else t for i, t in enumerate(rec.tname)]
    packed = sw.pack_records(gen.records_to_meta(rec))
    ctxs = [sw.Context(0), sw.Context(0)]
    try:
        f = sw.PafFilter(sw.FilterConfig(scaffold_gap=20_000, min_scaffold_length=2_000), ctx=ctxs[0])
        a = f.filter_columns(packed)
        b = f.filter_columns_multi(packed, ctxs)
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    finally:
        for c in ctxs:
            c.close()


def test_cli_devices_flag_byte_identical(tmp_path):
    import os
    import subprocess
    from sweepga_amd import build
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    rng = np.random.default_rng(12)
    rec = gen.random_records(rng, 20_000, n_genomes=4, chrs_per_genome=2, span=400_000)
    paf = tmp_path / "in.paf"
    paf.write_text(gen.records_to_paf(rng, rec))
    flags = ["--num-mappings", "1:1", "--scaffold-jump", "20k", "--scaffold-mass", "2k", "--scaffold-dist", "10k"]
    subprocess.check_call([os.path.join(root, "oracle", "sweepga-ref"), str(paf), "--output-file", str(tmp_path / "ref.paf"), *flags])
    r = subprocess.run([build.CLI, str(paf), "--output-file", str(tmp_path / "gpu.paf"), "--devices", "0,0,0", "--quiet", *flags],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert (tmp_path / "gpu.paf").read_bytes() == (tmp_path / "ref.paf").read_bytes()


def test_the_pinned_ring_wraps_around(tmp_path):
    """swg_filter_multi on an input that is not grouped by query genome: every shard's records are gathered from the caller's
    columns into a pinned ring of three slots and uploaded slot by slot (swg_filter_gathered).  With SWG_RING_CHUNK=2048 a shard of
    ~30,000 records takes 15 chunks -- every slot re-used several times -- and SWG_POISON fills what the flag set does not send;
    SWG_MULTI_SCATTER=1 (the shards copied into host columns first, rounds 2-5) must give the same answer."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = r"""
import sys
sys.path.insert(0, %r)
import numpy as np
import sweepga_amd as sw
from tests import gen
rng = np.random.default_rng(77)
rec = gen.random_records(rng, 90_000, n_genomes=6, chrs_per_genome=2, span=800_000)
packed = sw.pack_records(gen.records_to_meta(rec))
ctxs = [sw.Context(0) for _ in range(3)]
for kw in (dict(), dict(mapping_filter_mode=sw.FilterMode.OneToOne, scaffold_gap=0),
           dict(mapping_filter_mode=sw.FilterMode.OneToOne, scaffold_filter_mode=sw.FilterMode.OneToOne, scaffold_gap=20_000, min_scaffold_length=3_000,
                scaffold_max_deviation=15_000, min_identity=0.75)):
    f = sw.PafFilter(sw.FilterConfig(**kw), ctx=ctxs[0])
    want_st, want_ch = f.filter_columns(packed)
    got_st, got_ch = f.filter_columns_multi(packed, ctxs)
    assert np.array_equal(got_st, want_st) and np.array_equal(got_ch, want_ch), kw
print("ok")
""" % root
    for env in (dict(SWG_RING_CHUNK="2048", SWG_POISON="1"), dict(SWG_MULTI_SCATTER="1")):
        out = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, **env), capture_output=True, text=True, cwd=root)
        assert out.returncode == 0 and "ok" in out.stdout, (env, out.stderr[-2000:])
