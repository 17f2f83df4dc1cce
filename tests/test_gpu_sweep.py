"""GPU parity: the HIP plane sweep (through the C ABI) against the CPU oracle, bit-exact.

Runs on the MI355X box only (-m gpu).  Mismatches are dumped under gpurun_out/ for offline study.
"""
import json
import os

import numpy as np
import pytest

from tests import gen, orc

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "gpurun_out")


def _dump(name, payload):
    os.makedirs(OUT, exist_ok=True)
    with open(os.path.join(OUT, name), "w") as f:
        json.dump(payload, f)


@pytest.fixture(scope="module")
def sw():
    import sweepga_amd
    sweepga_amd.default_context(0)  # fails loudly without the HIP library or a GPU
    return sweepga_amd


def test_device_log_equals_host_libm(sw):
    """Device ln == glibc log bit for bit: all lengths < 2^22, then strided samples up to 2^40."""
    import ctypes as C
    ctx = sw.default_context(0)
    for first, stride, n in ((1, 1, 1 << 22), (1 << 22, 977, 1 << 22), (1 << 32, 1_000_003, 1 << 20)):
        dev = np.zeros(n, dtype=np.float64)
        ctx.check(ctx.lib.swg_log_range(ctx.handle, first, stride, n, dev.ctypes.data_as(C.c_void_p)))
        x = (first + stride * np.arange(n, dtype=np.uint64)).astype(np.float64)
        host = np.zeros(n, dtype=np.float64)
        orc.lib().orc_log_array(C.c_uint64(n), x.ctypes.data_as(C.c_void_p), host.ctypes.data_as(C.c_void_p))
        bad = np.nonzero(dev.view(np.uint64) != host.view(np.uint64))[0]
        assert bad.size == 0, f"{bad.size} log mismatches, first at x={x[bad[0]]}"


KAT = [
    # (maps, k, thr, scoring)  -- vectors of the reference's unit tests (see tests/test_oracle_kat.py)
    ([(100, 200, 300, 400, 0.95)], 1, 0.95, 3),
    ([(100, 200, 300, 400, 0.95), (300, 400, 500, 600, 0.90)], 1, 0.95, 3),
    ([(100, 200, 300, 400, 0.95), (150, 250, 350, 450, 0.90)], 1, 0.95, 3),
    ([(100, 200, 300, 400, 0.95), (100, 200, 500, 600, 0.90), (100, 200, 700, 800, 0.85)], 2, 0.95, 3),
    ([(100, 200, 300, 400, 0.95), (100, 200, 500, 600, 0.90), (100, 200, 700, 800, 0.85)], 1, 1.0, 3),
    ([(100, 200, 300, 400, 0.95), (100, 200, 500, 600, 0.90), (100, 200, 700, 800, 0.85)], 2, 0.5, 3),
    ([(100, 200, 300, 400, 1.0), (100, 200, 500, 600, 1.0), (100, 200, 700, 800, 1.0)], 1, 0.95, 3),
    ([(100, 200, 300, 400, 1.0), (100, 200, 500, 600, 1.0), (100, 200, 700, 800, 1.0)], orc.K_INF, 0.95, 3),
    ([(100, 300, 400, 600, 1.0), (150, 180, 500, 530, 1.0)], 1, 0.95, 3),
    ([(100, 300, 400, 600, 1.0), (100, 300, 700, 900, 1.0), (100, 300, 1000, 1200, 1.0), (100, 300, 1300, 1500, 1.0)], 2, 0.5, 3),
    ([(0, 100, 0, 100, 1.0), (50, 150, 200, 300, 1.0), (120, 220, 400, 500, 1.0), (200, 300, 600, 700, 1.0), (280, 380, 800, 900, 1.0)], 1, 0.95, 3),
    ([(100, 200, 300, 400, 1.0), (100, 190, 500, 590, 1.0), (100, 180, 700, 780, 1.0), (100, 170, 900, 970, 1.0), (100, 160, 1100, 1160, 1.0)], 3, 1.0, 3),
    ([(100, 100, 300, 300, 1.0), (100, 200, 400, 500, 1.0), (100, 300, 600, 800, 1.0)], 1, 0.95, 3),
    ([(1000, 2000, 5000, 6000, 1.0), (1500, 2500, 7000, 8000, 1.0), (3000, 4000, 9000, 10000, 1.0), (3200, 3800, 11000, 11600, 1.0),
      (5000, 5500, 15000, 15500, 1.0), (5000, 5500, 16000, 16500, 1.0), (5000, 5500, 17000, 17500, 1.0), (5000, 5500, 18000, 18500, 1.0),
      (8000, 12000, 20000, 24000, 1.0)], 2, 0.95, 3),
    ([(100, 500, 1000, 1400, 0.70), (100, 200, 2000, 2100, 0.99), (100, 300, 3000, 3200, 0.85)], 1, 0.95, 0),
    ([(100, 200, 1000, 1100, 0.99), (100, 600, 2000, 2500, 0.50), (100, 350, 3000, 3250, 0.75)], 1, 0.95, 1),
    ([(100, 200, 1000, 1100, 0.95), (100, 400, 2000, 2300, 0.60), (100, 300, 3000, 3200, 0.80)], 1, 0.95, 2),
    ([(100, 300, 1000, 1200, 0.90), (100, 280, 2000, 2180, 1.00), (100, 460, 3000, 3360, 0.50)], 1, 0.95, 2),
    ([(100, 200, 1000, 1100, 0.70), (100, 250, 2000, 2150, 0.80), (100, 300, 3000, 3200, 0.90), (100, 180, 4000, 4080, 0.99), (100, 220, 5000, 5120, 0.60)], 2, 0.95, 2),
    ([(100, 101, 1000, 1001, 1.00), (100, 100100, 2000, 102000, 0.01), (100, 1100, 3000, 4000, 0.50)], 1, 0.95, 4),
]


def _maps(sw, rows):
    return [sw.PlaneSweepMapping(i, *r) for i, r in enumerate(rows)]


@pytest.mark.parametrize("case", range(len(KAT)))
def test_reference_vectors(sw, case):
    rows, k, thr, scoring = KAT[case]
    m = _maps(sw, rows)
    for axis, fn_g, fn_o in ((0, sw.plane_sweep_query, orc.plane_sweep_query), (1, sw.plane_sweep_target, orc.plane_sweep_target)):
        got = fn_g(m, k, thr, sw.ScoringFunction(scoring))
        want = fn_o(rows, k, thr, scoring)
        assert got == want, (axis, got, want)
    assert sw.plane_sweep_both(m, k, k, thr, sw.ScoringFunction(scoring)) == orc.plane_sweep_both(rows, k, k, thr, scoring)


def test_empty_and_u64_coordinates(sw):
    assert sw.plane_sweep_query([], 1, 0.95) == []
    U = 2**64 - 1
    # plane_sweep_exact.rs:804-826: a mapping at u64::MAX.  The seam shrinks uncovered stretches, so this runs on the
    # u32 device layout and keeps both, like the reference.
    assert sw.plane_sweep_query(_maps(sw, [(0, 100, 0, 100, 0.95), (U - 100, U, 1000, 1100, 0.9)]), 1, 0.95) == [0, 1]
    # a covered stretch wider than 32 bits cannot be represented: refused, not wrapped
    with pytest.raises(sw.SwgError) as e:
        sw.plane_sweep_query(_maps(sw, [(0, 2**33, 0, 100, 0.95), (5, 2**33 + 7, 1000, 1100, 0.9)]), 1, 0.95)
    assert e.value.code == -5


@pytest.mark.parametrize("seed", range(6))
def test_u64_coordinates_match_oracle(sw, seed):
    """Segments scattered over the whole u64 range (clusters of overlapping intervals separated by huge gaps), both
    axes, against the oracle on the original coordinates."""
    import ctypes as C
    rng = np.random.default_rng(7000 + seed)
    ctx = sw.default_context(0)
    n = int(rng.choice([5, 60, 700, 3000]))
    qs, qe, ts, te, ident = gen.random_segment(rng, n, span=int(rng.choice([2_000, 200_000])), max_len=500)
    off_q = rng.integers(0, 2**62, n, dtype=np.uint64) // np.uint64(2**40) * np.uint64(2**40) * np.uint64(rng.integers(0, 4))
    off_t = np.uint64(2**64 - 1 - int(te.max())) if seed % 2 else np.uint64(0)
    qs, qe = qs + off_q, qe + off_q
    ts, te = ts + off_t, te + off_t
    for k, thr, scoring, axis in ((1, 0.95, 4, 0), (2, 0.5, 3, 1), (1, 0.0, 1, 2), (orc.K_INF, 1.0, 0, 2), (3, 0.95, 2, 0)):
        want = np.zeros(n, dtype=np.uint8)
        want[orc.plane_sweep(axis, qs, qe, ts, te, ident, k_q=k, k_t=k, thr=thr, scoring=scoring)] = 1
        got = np.zeros(n, dtype=np.uint8)
        ctx.check(ctx.lib.swg_plane_sweep(ctx.handle, axis, n, *(a.ctypes.data_as(C.c_void_p) for a in (qs, qe, ts, te, ident)),
                                          k, k, thr, scoring, got.ctypes.data_as(C.c_void_p)))
        assert np.array_equal(got, want), (k, thr, scoring, axis, int((got != want).sum()))


@pytest.mark.parametrize("seed", range(12))
def test_random_segments_bit_exact(sw, seed):
    """Differential fuzz on single segments: k in {1,2,3,7,inf}, thr in {0,.5,.95,1}, all scorings."""
    import ctypes as C
    rng = np.random.default_rng(1000 + seed)
    ctx = sw.default_context(0)
    n = int(rng.choice([2, 3, 17, 64, 257, 600, 1500, 4000]))
    span = int(rng.choice([500, 5_000, 100_000]))
    levels = [0.9, 0.95] if seed % 3 == 0 else None  # force score ties
    qs, qe, ts, te, ident = gen.random_segment(rng, n, span=span, max_len=max(2, span // 4), ident_levels=levels)
    fails = []
    for k in (1, 2, 3, 7, orc.K_INF):
        for thr in (0.0, 0.5, 0.95, 1.0):
            for scoring in range(5):
                if (k, thr, scoring) != (1, 0.95, 3) and rng.random() < 0.6:
                    continue
                for axis in (0, 1, 2):
                    want = np.zeros(n, dtype=np.uint8)
                    want[orc.plane_sweep(axis, qs, qe, ts, te, ident, k_q=k, k_t=k, thr=thr, scoring=scoring)] = 1
                    got = np.zeros(n, dtype=np.uint8)
                    ctx.check(ctx.lib.swg_plane_sweep(ctx.handle, axis, n, *(a.ctypes.data_as(C.c_void_p) for a in (qs, qe, ts, te, ident)),
                                                      k, k, thr, scoring, got.ctypes.data_as(C.c_void_p)))
                    if not np.array_equal(got, want):
                        bad = np.nonzero(got != want)[0]
                        fails.append(dict(k=int(min(k, 10**9)), thr=thr, scoring=scoring, axis=axis, n=n, nbad=int(bad.size),
                                          first=[int(b) for b in bad[:10]], got=[int(got[b]) for b in bad[:10]]))
    if fails:
        _dump(f"sweep_fuzz_seed{seed}.json", dict(fails=fails, qs=qs.tolist(), qe=qe.tolist(), ts=ts.tolist(),
                                                   te=te.tolist(), ident=ident.tolist()))
    assert not fails, fails[:3]


@pytest.mark.parametrize("seed,n", [(1, 20_000), (2, 60_000)])
def test_large_single_segment(sw, seed, n):
    """One deep segment (many carry-ins per tile) against the oracle, k = 1 and k = 2."""
    import ctypes as C
    rng = np.random.default_rng(seed)
    ctx = sw.default_context(0)
    qs, qe, ts, te, ident = gen.random_segment(rng, n, span=2_000_000, max_len=40_000, zero_frac=0.001, dup_frac=0.01)
    for k, thr in ((1, 0.95), (2, 0.5)):
        for axis in (0, 1):
            want = np.zeros(n, dtype=np.uint8)
            want[orc.plane_sweep(axis, qs, qe, ts, te, ident, k_q=k, k_t=k, thr=thr)] = 1
            got = np.zeros(n, dtype=np.uint8)
            ctx.check(ctx.lib.swg_plane_sweep(ctx.handle, axis, n, *(a.ctypes.data_as(C.c_void_p) for a in (qs, qe, ts, te, ident)),
                                              k, k, thr, 3, got.ctypes.data_as(C.c_void_p)))
            bad = np.nonzero(got != want)[0]
            if bad.size:
                _dump(f"sweep_large_seed{seed}_k{k}_axis{axis}.json", dict(nbad=int(bad.size), first=bad[:50].tolist()))
            assert bad.size == 0, (k, axis, bad.size)


@pytest.mark.parametrize("seed", range(3))
def test_deep_segment_general_k(sw, seed):
    """Deep segments (hundreds of carry-ins per tile) for 2 <= k: the pruned tile kernel (k <= 16: stars, candidates, member
    changes) and the plain one (k = 17, and everything under SWG_KN_PLAIN), with ties, zero lengths and every threshold kind."""
    import ctypes as C
    rng = np.random.default_rng(100 + seed)
    ctx = sw.default_context(0)
    n = [12_000, 25_000, 40_000][seed]
    qs, qe, ts, te, ident = gen.random_segment(rng, n, span=[300_000, 1_000_000, 2_500_000][seed], max_len=[20_000, 30_000, 60_000][seed],
                                               zero_frac=0.002, dup_frac=0.05, ident_levels=[None, [0.8, 0.9, 0.95, 0.99, 1.0], None][seed])
    fails = []
    for k, thr, scoring in ((2, 0.95, 3), (3, 0.5, 3), (3, 1.0, 0), (4, 0.0, 1), (5, 0.7, 2), (8, 0.95, 4), (9, 0.6, 3), (2, 0.1, 3), (12, 0.8, 3), (16, 0.5, 1), (17, 0.9, 3)):
        for axis in (0, 1):
            want = np.zeros(n, dtype=np.uint8)
            want[orc.plane_sweep(axis, qs, qe, ts, te, ident, k_q=k, k_t=k, thr=thr, scoring=scoring)] = 1
            got = np.zeros(n, dtype=np.uint8)
            ctx.check(ctx.lib.swg_plane_sweep(ctx.handle, axis, n, *(a.ctypes.data_as(C.c_void_p) for a in (qs, qe, ts, te, ident)),
                                              k, k, thr, scoring, got.ctypes.data_as(C.c_void_p)))
            bad = np.nonzero(got != want)[0]
            if bad.size:
                fails.append(dict(k=k, thr=thr, scoring=scoring, axis=axis, nbad=int(bad.size), first=bad[:10].tolist(),
                                  got=got[bad[:10]].tolist()))
    if fails:
        _dump(f"sweep_deep_k_seed{seed}.json", dict(fails=fails))
    assert not fails, fails[:4]


@pytest.mark.parametrize("seed", range(6))
def test_mapping_filter_no_scaffold(sw, seed):
    """apply_filters with scaffold_gap = 0 (plane sweep only) on multi-genome records."""
    rng = np.random.default_rng(50 + seed)
    n = int(rng.choice([1, 2, 50, 3000, 20_000]))
    rec = gen.random_records(rng, n, n_genomes=int(rng.integers(1, 5)), chrs_per_genome=int(rng.integers(1, 4)),
                             pansn=bool(seed % 2 == 0))
    meta = gen.records_to_meta(rec)
    modes = [(sw.FilterMode.OneToOne, None, None), (sw.FilterMode.OneToMany, 1, None), (sw.FilterMode.ManyToMany, 2, 3),
             (sw.FilterMode.ManyToMany, None, None)]
    for mode, mq, mt in modes:
        for thr in (0.5, 0.95):
            for keep_self in (False, True):
                cfg = sw.FilterConfig(mapping_filter_mode=mode, mapping_max_per_query=mq, mapping_max_per_target=mt,
                                      overlap_threshold=thr, scaffold_gap=0, min_block_length=100 if seed % 2 else 0,
                                      min_identity=0.75 if seed % 3 == 0 else 0.0)
                f = sw.PafFilter(cfg).with_keep_self(keep_self)
                status, chain = f.filter_columns(sw.pack_records(meta))
                ocfg = orc.Config(mapping_filter_mode=int(mode), mapping_max_per_query=mq or 0, mapping_max_per_target=mt or 0,
                                  overlap_threshold=thr, scaffold_gap=0, min_block_length=cfg.min_block_length,
                                  min_identity=cfg.min_identity, keep_self=keep_self)
                ost, och = orc.apply_filters(ocfg, rec)
                bad = np.nonzero(status != ost)[0]
                if bad.size:
                    _dump(f"mapfilter_seed{seed}.json", dict(mode=int(mode), mq=mq, mt=mt, thr=thr, keep_self=keep_self,
                                                             nbad=int(bad.size), first=bad[:20].tolist()))
                assert bad.size == 0, (int(mode), mq, mt, thr, keep_self, bad.size)
                assert np.array_equal(chain, och)


def test_reversed_interval_does_not_depend_on_unrelated_records():
    """Reversed intervals (start > end) are malformed PAF and not modelled -- but what the filter answers
    for such a record must not depend on records that have nothing to do with it.  Round 3's shortcut for the unlimited sweep
    counted only exactly-zero-length records as degenerate, the per-axis path every start >= end: whether a reversed record was
    kept then depended on whether some OTHER record of the input happened to have zero length."""
    import sweepga_amd as sw
    from tests import gen
    rng = np.random.default_rng(91)
    rec = gen.random_records(rng, 3_000, n_genomes=3, chrs_per_genome=2, span=300_000, zero_frac=0.0)
    k = 1234
    rec.qs[k], rec.qe[k] = rec.qe[k] + 10, rec.qs[k]   # reversed on the query axis
    cfg = sw.FilterConfig(scaffold_gap=0)              # many:many, no scaffolding: the unlimited sweep decides
    f = sw.PafFilter(cfg)
    base, _ = f.filter_columns(sw.pack_records(gen.records_to_meta(rec)))
    base = base.copy()
    # the same records plus one zero-length record in a genome pair of its own
    import copy
    rec2 = copy.copy(rec)
    rec2.qname = rec.qname + ["zz#1#chr1"]
    rec2.tname = rec.tname + ["yy#1#chr1"]
    for name, val in (("qs", 500), ("qe", 500), ("ts", 700), ("te", 900), ("block_length", 200), ("matches", 180)):
        setattr(rec2, name, np.append(getattr(rec, name), np.uint64(val)))
    rec2.identity = np.append(rec.identity, 0.9)
    rec2.strand = np.append(rec.strand, np.uint8(ord("+")))
    rec2.rank = np.arange(len(rec2.qname), dtype=np.uint64)
    more, _ = f.filter_columns(sw.pack_records(gen.records_to_meta(rec2)))
    assert np.array_equal(more[:len(base)], base)


def test_reversed_interval_against_the_oracle():
    """The same malformed record held to the ORACLE.  The reference keeps a reversed interval in an unlimited sweep: its End event
    precedes its Begin event, `remove` finds nothing, and the interval entered later is never taken out of the tree again
    (src/plane_sweep_exact.rs:300-349) -- the oracle restates that.  The device evaluates the sweep in closed form over
    [start, end) and counts start >= end as "never active": it DROPS the record.  That is the one documented divergence
    (DESIGN.md section 4); this test pins both halves of it -- the reversed record's two answers, and that every other record of
    the input gets the oracle's answer, with and without the scaffold stage behind the sweep."""
    import sweepga_amd as sw
    from tests import gen, orc
    rng = np.random.default_rng(91)
    rec = gen.random_records(rng, 3_000, n_genomes=3, chrs_per_genome=2, span=300_000, zero_frac=0.0)
    k = 1234
    rec.qs[k], rec.qe[k] = rec.qe[k] + 10, rec.qs[k]   # reversed on the query axis
    others = np.arange(len(rec)) != k
    packed = sw.pack_records(gen.records_to_meta(rec))
    # many:many, no scaffolding: the unlimited sweep alone decides
    st, ch = sw.PafFilter(sw.FilterConfig(scaffold_gap=0)).filter_columns(packed)
    ost, och = orc.apply_filters(orc.Config(scaffold_gap=0), rec)
    assert ost[k] == 3 and st[k] == 0            # reference: kept (unassigned); device: dropped
    assert np.array_equal(st[others], ost[others]) and np.array_equal(ch[others], och[others])
    # 1:1 on both axes: the reversed record stays in the reference's tree from its start on with the score of a wrapped
    # length (2^64 - 10: it outranks everything) -- the records of ITS sweep segments may differ, every other segment must not
    st, _ = sw.PafFilter(sw.FilterConfig(mapping_filter_mode=sw.FilterMode.OneToOne, scaffold_gap=0)).filter_columns(packed)
    ost, _ = orc.apply_filters(orc.Config(mapping_filter_mode=orc.ONE_TO_ONE, scaffold_gap=0), rec)
    genome = lambda name: name.rsplit("#", 1)[0]
    seg = np.array([(q == rec.qname[k] and genome(t) == genome(rec.tname[k])) or (t == rec.tname[k] and genome(q) == genome(rec.qname[k]))
                    for q, t in zip(rec.qname, rec.tname)])
    assert np.array_equal(st[~seg], ost[~seg])
