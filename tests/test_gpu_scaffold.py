"""GPU parity of the scaffold path (chaining, scaffold sweep, numbering, anchors, rescue) against the
CPU oracle: status AND chain numbers must match exactly.  -m gpu only."""
import json
import os
import tempfile

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
    sweepga_amd.default_context(0)
    return sweepga_amd


def _cfg_pair(sw, **kw):
    """(FilterConfig for the device, orc.Config for the oracle) from one set of values."""
    c = sw.FilterConfig(**kw)
    o = orc.Config(min_block_length=c.min_block_length, mapping_filter_mode=int(c.mapping_filter_mode),
                   mapping_max_per_query=c.mapping_max_per_query or 0, mapping_max_per_target=c.mapping_max_per_target or 0,
                   scaffold_filter_mode=int(c.scaffold_filter_mode), scaffold_max_per_query=c.scaffold_max_per_query or 0,
                   scaffold_max_per_target=c.scaffold_max_per_target or 0, overlap_threshold=c.overlap_threshold,
                   scaffold_gap=c.scaffold_gap, min_scaffold_length=c.min_scaffold_length,
                   scaffold_overlap_threshold=c.scaffold_overlap_threshold, scaffold_max_deviation=c.scaffold_max_deviation,
                   scoring_function=int(c.scoring_function), min_identity=c.min_identity,
                   min_scaffold_identity=c.min_scaffold_identity)
    return c, o


def test_union_find_sets(sw):
    # tests/test_binary_search_optimization.rs:222-226 expectation and the two edge cases :279-384
    uf = sw.UnionFind(5)
    for x, y in [(0, 1), (1, 2), (3, 4)]:
        uf.union(x, y)
    assert uf.get_sets() == [[0, 1, 2], [3, 4]]
    assert sw.UnionFind(3).get_sets() == [[0], [1], [2]]
    uf = sw.UnionFind(4)
    for x, y in [(0, 1), (1, 2), (2, 3)]:
        uf.union(x, y)
    assert uf.get_sets() == [[0, 1, 2, 3]]
    # long random forest of paths vs the oracle's UnionFind
    rng = np.random.default_rng(3)
    n = 5000
    edges = [(i, i + 1) for i in range(n - 1) if rng.random() < 0.8]
    uf = sw.UnionFind(n)
    for e in edges:
        uf.union(*e)
    assert uf.get_sets() == orc.union_find_sets(n, edges)


def test_plane_sweep_scaffolds_reference_vectors(sw):
    # src/plane_sweep_scaffold.rs:292-371
    c1 = [("chr1", "chr1", 0, 1000, 0, 1000, 0.95), ("chr1", "chr1", 2000, 3000, 2000, 3000, 0.95)]
    assert sw.plane_sweep_scaffolds(c1, sw.FilterMode.OneToOne, 1, 1, 0.5) == orc.plane_sweep_scaffolds(c1, 0, 1, 1, 0.5)
    c2 = [("chr1", "chr1", 0, 1000, 0, 1000, 0.90), ("chr1", "chr1", 900, 1900, 900, 1900, 0.98)]
    assert sw.plane_sweep_scaffolds(c2, sw.FilterMode.OneToOne, 1, 1, 0.95) == orc.plane_sweep_scaffolds(c2, 0, 1, 1, 0.95)


@pytest.mark.parametrize("seed", range(8))
def test_plane_sweep_scaffolds_random(sw, seed):
    rng = np.random.default_rng(700 + seed)
    n = int(rng.choice([1, 2, 30, 400, 3000]))
    names = [f"g{g}#{h}#c{c}" for g in range(3) for h in (1, 2) for c in range(2)] + ["plain1", "plain2", "a#b#c#d"]
    chains = []
    for _ in range(n):
        q, t = rng.choice(names), rng.choice(names)
        qs, ts = int(rng.integers(0, 50_000)), int(rng.integers(0, 50_000))
        ql, tl = int(rng.integers(0, 8000)), int(rng.integers(0, 8000))
        chains.append((str(q), str(t), qs, qs + ql, ts, ts + tl, float(np.round(rng.uniform(0.6, 1.0), 2))))
    for mode, mq, mt in ((0, 1, 1), (2, None, None), (2, 2, 1), (1, 1, None)):
        for thr in (0.5, 0.95):
            got = sw.plane_sweep_scaffolds(chains, sw.FilterMode(mode), mq, mt, thr)
            want = orc.plane_sweep_scaffolds(chains, mode, mq, mt, thr)
            assert got == want, (mode, mq, mt, thr, len(got), len(want))


@pytest.mark.parametrize("seed", range(8))
def test_merge_chains(sw, seed):
    rng = np.random.default_rng(900 + seed)
    n = int(rng.choice([1, 2, 40, 1500, 12_000]))
    rec = gen.random_records(rng, n, n_genomes=int(rng.integers(1, 4)), chrs_per_genome=int(rng.integers(1, 3)),
                             span=int(rng.choice([50_000, 400_000])), max_len=4000, self_frac=0.1)
    meta = gen.records_to_meta(rec)
    for gap in (500, 5_000, 60_000):
        got_of, got = sw.merge_mappings_into_chains(meta, gap)
        want_of, want_cols, want_wid = orc.merge_chains(rec, gap)
        ok = np.array_equal(got_of, want_of)
        if not ok:
            bad = np.nonzero(got_of != want_of)[0]
            _dump(f"merge_chains_seed{seed}_gap{gap}.json", dict(nbad=int(bad.size), first=bad[:20].tolist(),
                                                                 got=got_of[bad[:20]].tolist(), want=want_of[bad[:20]].tolist()))
        assert ok, (gap, n)
        assert np.array_equal(got["query_start"], want_cols[0]) and np.array_equal(got["query_end"], want_cols[1])
        assert np.array_equal(got["target_start"], want_cols[2]) and np.array_equal(got["target_end"], want_cols[3])
        assert np.array_equal(got["weighted_identity"].view(np.uint64), want_wid.view(np.uint64))  # bit-exact f64


@pytest.mark.parametrize("seed,n,span,gap", [(1, 20_000, 400_000, 60_000), (2, 30_000, 3_000_000, 50_000), (3, 9_000, 100_000, 5_000)])
def test_merge_chains_long_dense_units(sw, seed, n, span, gap):
    """One chromosome pair, windows of hundreds to thousands of elements and no cuts: the long-unit path
    (LDS ring, full-window fallback when the 4 listed candidates are all blocked)."""
    rng = np.random.default_rng(seed)
    rec = gen.random_records(rng, n, n_genomes=2, chrs_per_genome=1, span=span, max_len=6000, self_frac=0.0,
                             syntenic_frac=0.9)
    # keep one ordered genome pair only
    keep = [i for i in range(len(rec)) if rec.qname[i].startswith("g0") and rec.tname[i].startswith("g1")]
    sub = orc.Records([rec.qname[i] for i in keep], [rec.tname[i] for i in keep],
                      *(np.ascontiguousarray(getattr(rec, f)[keep]) for f in ("qs", "qe", "ts", "te", "block_length", "identity", "matches", "strand")),
                      np.arange(len(keep), dtype=np.uint64))
    meta = gen.records_to_meta(sub)
    got_of, got = sw.merge_mappings_into_chains(meta, gap)
    want_of, want_cols, want_wid = orc.merge_chains(sub, gap)
    assert np.array_equal(got_of, want_of), int((got_of != want_of).sum())
    assert np.array_equal(got["query_end"], want_cols[1]) and np.array_equal(got["target_start"], want_cols[2])
    assert np.array_equal(got["weighted_identity"].view(np.uint64), want_wid.view(np.uint64))


SCAFFOLD_CFGS = [
    dict(),  # CLI defaults: many:many, jump 50k, mass 10k
    dict(scaffold_gap=20_000, min_scaffold_length=3_000, scaffold_filter_mode=0, scaffold_max_deviation=15_000),
    dict(mapping_filter_mode=0, scaffold_gap=10_000, min_scaffold_length=1_000, scaffold_filter_mode=0,
         scaffold_max_deviation=5_000, scaffold_overlap_threshold=0.3),
    dict(mapping_filter_mode=1, mapping_max_per_query=2, scaffold_gap=30_000, min_scaffold_length=0,
         scaffold_filter_mode=2, scaffold_max_per_query=2, scaffold_max_per_target=1, scaffold_max_deviation=50_000),
    dict(scaffold_gap=5_000, min_scaffold_length=2_000, min_scaffold_identity=0.85, scaffold_filter_mode=0, scoring_function=2),
    dict(mapping_filter_mode=0, overlap_threshold=0.5, scaffold_gap=100_000, min_scaffold_length=20_000,
         scaffold_filter_mode=0, scaffold_max_deviation=100_000, min_block_length=200, min_identity=0.8),
]


@pytest.mark.parametrize("cfg_i", range(len(SCAFFOLD_CFGS)))
@pytest.mark.parametrize("seed", range(4))
def test_full_pipeline(sw, seed, cfg_i):
    rng = np.random.default_rng(40 * cfg_i + seed)
    n = int(rng.choice([1, 3, 200, 5_000, 30_000]))
    rec = gen.random_records(rng, n, n_genomes=int(rng.integers(1, 5)), chrs_per_genome=int(rng.integers(1, 4)),
                             span=int(rng.choice([100_000, 1_000_000])), pansn=bool((seed + cfg_i) % 3), minus_frac=0.3)
    meta = gen.records_to_meta(rec)
    kw = dict(SCAFFOLD_CFGS[cfg_i])
    for k in ("mapping_filter_mode", "scaffold_filter_mode"):
        if k in kw:
            kw[k] = sw.FilterMode(kw[k])
    if "scoring_function" in kw:
        kw["scoring_function"] = sw.ScoringFunction(kw["scoring_function"])
    cfg, ocfg = _cfg_pair(sw, **kw)
    packed = sw.pack_records(meta)
    for keep_self, scaffolds_only in ((False, False), (True, False), (False, True)):
        f = sw.PafFilter(cfg).with_keep_self(keep_self).with_scaffolds_only(scaffolds_only)
        status, chain = f.filter_columns(packed)
        ocfg.keep_self, ocfg.scaffolds_only = keep_self, scaffolds_only
        ost, och = orc.apply_filters(ocfg, rec)
        bad_s = np.nonzero(status != ost)[0]
        bad_c = np.nonzero(chain != och)[0]
        if bad_s.size or bad_c.size:
            _dump(f"pipeline_cfg{cfg_i}_seed{seed}_{int(keep_self)}{int(scaffolds_only)}.json",
                  dict(n=n, nbad_status=int(bad_s.size), nbad_chain=int(bad_c.size), first_s=bad_s[:20].tolist(),
                       got_s=status[bad_s[:20]].tolist(), want_s=ost[bad_s[:20]].tolist(), first_c=bad_c[:20].tolist(),
                       got_c=chain[bad_c[:20]].tolist(), want_c=och[bad_c[:20]].tolist()))
        assert bad_s.size == 0, (cfg_i, seed, keep_self, scaffolds_only, "status", int(bad_s.size), n)
        assert bad_c.size == 0, (cfg_i, seed, keep_self, scaffolds_only, "chain", int(bad_c.size), n)
        st = f.last_stats
        assert st.n_out == int((ost != 0).sum())


def _l(q, qs, qe, t, ts, te, m, b, strand="+"):
    return f"{q}\t100000\t{qs}\t{qe}\t{strand}\t{t}\t100000\t{ts}\t{te}\t{m}\t{b}\t60\tNM:i:{b - m}\tcg:Z:{m}={b - m}X\n"


FIXTURES = {
    # tests/test_scaffold_plane_sweep_filtering.rs:7-56
    "same_pair": (_l("chr1", 10000, 15000, "target_chr1", 10000, 15000, 4750, 5000) + _l("chr1", 15000, 20000, "target_chr1", 15000, 20000, 4750, 5000)
                  + _l("chr1", 12000, 17000, "target_chr1", 30000, 35000, 4900, 5000) + _l("chr1", 17000, 22000, "target_chr1", 35000, 40000, 4900, 5000),
                  dict(min_scaffold_length=1000, scaffold_gap=10000, scaffold_filter_mode=0)),
    # tests/test_scaffold_plane_sweep_filtering.rs:120-169
    "contained": (_l("chr1", 15000, 18000, "target_chr1", 15000, 18000, 2940, 3000) + _l("chr1", 10000, 17500, "target_chr1", 10000, 17500, 7125, 7500)
                  + _l("chr1", 17500, 25000, "target_chr1", 17500, 25000, 7125, 7500),
                  dict(min_scaffold_length=1000, scaffold_gap=10000, scaffold_filter_mode=0)),
    # tests/test_chaining_stability.rs:243-350
    "overlap_penalty": ("querySeq\t10000\t0\t1000\t+\ttargetSeq\t10000\t0\t1000\t950\t1000\t60\n"
                        "querySeq\t10000\t900\t1900\t+\ttargetSeq\t10000\t900\t1900\t950\t1000\t60\n"
                        "querySeq\t10000\t1100\t2100\t+\ttargetSeq\t10000\t1100\t2100\t950\t1000\t60\n",
                        dict(overlap_threshold=0.0, scaffold_gap=10_000, min_scaffold_length=0, scaffold_overlap_threshold=0.0,
                             scaffold_max_deviation=20_000)),
    # an inversion on the diagonal of a forward scaffold + a rescued neighbour + an unrelated pair
    "inversion_rescue": (_l("A#1#c1", 10000, 16000, "B#1#c1", 20000, 26000, 5900, 6000) + _l("A#1#c1", 16500, 23000, "B#1#c1", 26500, 33000, 6400, 6500)
                         + _l("A#1#c1", 23500, 24500, "B#1#c1", 33500, 34500, 990, 1000, "-") + _l("A#1#c1", 40000, 40800, "B#1#c1", 52000, 52800, 790, 800)
                         + _l("A#1#c1", 90000, 90500, "B#1#c2", 100, 600, 490, 500),
                         dict(scaffold_gap=5000, min_scaffold_length=5000, scaffold_filter_mode=0, scaffold_max_deviation=30_000)),
}


@pytest.mark.parametrize("name", sorted(FIXTURES))
def test_filter_paf_byte_identical(sw, name):
    """PafFilter.filter_paf end to end: output file byte-identical to the oracle's."""
    text, kw = FIXTURES[name]
    kw = dict(kw)
    if "scaffold_filter_mode" in kw:
        kw["scaffold_filter_mode"] = sw.FilterMode(kw["scaffold_filter_mode"])
    cfg, ocfg = _cfg_pair(sw, **kw)
    with tempfile.TemporaryDirectory() as d:
        inp, o1, o2 = os.path.join(d, "i.paf"), os.path.join(d, "gpu.paf"), os.path.join(d, "orc.paf")
        with open(inp, "w") as f:
            f.write(text)
        sw.PafFilter(cfg).filter_paf(inp, o1)
        orc.filter_paf(ocfg, inp, o2)
        a, b = open(o1, "rb").read(), open(o2, "rb").read()
        assert a == b, (a.decode(), b.decode())
        assert a  # every fixture keeps something


def test_chaining_equal_distance_candidates(sw):
    """Found by tests/fuzz/fuzz_gpu.py: B's candidates arrive as C (d), D (d, duplicate of C), E (smaller d but already
    taken by A).  The candidate list must keep C before D when E is inserted in front of them; the chain must then
    be B-C-D, not C-D alone (paf_filter.rs:839: strict `<`, the first minimum wins)."""
    rows = [(700, 750, 0, 50), (600, 750, 0, 150), (850, 950, 100, 150), (600, 700, 150, 250), (700, 750, 0, 50)]
    u = lambda k: np.array([r[k] for r in rows], dtype=np.uint64)
    n = len(rows)
    rec = orc.Records(["q"] * n, ["t"] * n, u(0), u(1), u(2), u(3), u(1) - u(0), np.full(n, 0.9), (u(1) - u(0)) * 9 // 10,
                      np.full(n, ord("+"), dtype=np.uint8), np.arange(n, dtype=np.uint64))
    got_of, _ = sw.merge_mappings_into_chains(gen.records_to_meta(rec), 50_000)
    want_of, _, _ = orc.merge_chains(rec, 50_000)
    assert want_of.tolist() == [1, 0, 0, 1, 1]
    assert got_of.tolist() == want_of.tolist()


@pytest.mark.parametrize("first_seed,extras", [(450, False), (3150, False), (5500, False), (0, True), (100_000, True)])
def test_fuzz_slice(sw, first_seed, extras):
    """Slices of tests/fuzz/fuzz_gpu.py (random record sets x random configurations, exact status and chain numbers, every
    8th case also through swg_filter_multi).  The extras=False ranges contain seeds 477, 3190 and 5535, which failed
    before the candidate-list tie fix."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "fuzz"))
    from fuzz_gpu import run_case
    for seed in range(first_seed, first_seed + 60):
        ok, n, kw, keep_self, scaffolds_only, bs, bc = run_case(seed, extras)
        assert ok, (seed, n, bs, bc, kw, keep_self, scaffolds_only)


@pytest.mark.parametrize("seed,n,span,gap,max_len", [(1, 3_000, 300_000, 10_000, 8_000), (2, 40_000, 2_000_000, 10_000, 8_000),
                                                      (3, 60_000, 60_000, 300, 300)])
def test_statistics_count_every_chain(sw, seed, n, span, gap, max_len):
    """swg_stats of a call without a mapping-level sweep (the CLI defaults): n_swept = the retained records, n_chains = ALL
    chains of merge_mappings_into_chains before the span / identity filter (src/paf_filter.rs:437-447), counted where the
    heads are found -- by the chunk labelling for short units and by the generic path for long ones (the third case is one
    dense pair: units of more than 8192 mappings) --, n_chains_kept = the chains that pass.  Expected numbers from the
    independent Python model."""
    from tests import model_apply_filters as model
    from tests.test_model_cpu import _records
    rng = np.random.default_rng(900 + seed)
    rec = gen.random_records(rng, n, n_genomes=2 if seed == 3 else 3, chrs_per_genome=1 if seed == 3 else 2, span=span, max_len=max_len,
                             zero_frac=0.0)  # (no zero-length records: an unlimited mapping sweep would drop those)
    min_len = 5_000 if seed != 3 else 400
    dcfg = sw.FilterConfig(scaffold_gap=gap, min_scaffold_length=min_len, scaffold_max_deviation=0)
    f = sw.PafFilter(dcfg)
    f.filter_columns(sw.pack_records(gen.records_to_meta(rec)))
    md = [m for m in _records(rec) if m["q"] != m["t"] and m["block"] >= dcfg.min_block_length and m["identity"] >= dcfg.min_identity]
    chains = model._merge_mappings_into_chains(md, gap)
    kept = [c for c in chains if c["total"] >= min_len and c["wid"] >= dcfg.min_scaffold_identity]
    st = f.last_stats
    assert (st.n_in, st.n_retained, st.n_swept) == (n, len(md), len(md))
    assert st.n_chains == len(chains) and len(chains) > 0
    assert st.n_chains_kept == len(kept) and 0 < len(kept) < len(chains)
