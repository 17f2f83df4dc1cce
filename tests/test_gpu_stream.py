"""The streamed host path (csrc/swg_stream.hip): ranges of whole query genomes uploaded while their predecessors are
filtered.  Its results -- status AND global chain numbers -- must equal the one-piece call's (SWG_STREAM=0) and the oracle's,
on one context and over several; and an absent identity column (identity = matches / max(block_len, 1) evaluated on the
device) must equal the explicit one."""
import numpy as np
import pytest

from tests import gen, orc
from tests.test_stream_plan_cpu import _sorted_by_query_genome

pytestmark = pytest.mark.gpu

CONFIGS = [
    dict(mapping_filter_mode="OneToOne", scaffold_gap=20_000, min_scaffold_length=3_000, scaffold_filter_mode="OneToOne",
         scaffold_max_deviation=15_000),
    dict(scaffold_gap=20_000, min_scaffold_length=3_000, scaffold_filter_mode="OneToOne", scaffold_max_deviation=15_000),  # c5-like
    dict(),                                                      # CLI defaults (many:many, 50 kb jump)
    dict(mapping_filter_mode="OneToOne", scaffold_gap=0),        # sweep only: no chain numbers
    dict(mapping_filter_mode="OneToMany", mapping_max_per_query=2, scaffold_gap=10_000, min_scaffold_length=1_000,
         min_block_length=300, min_identity=0.8, scaffold_max_deviation=5_000),
    # who reads the value columns (the host paths send only what is read): a chain identity floor alone, a block floor alone
    dict(scaffold_gap=20_000, min_scaffold_length=3_000, min_scaffold_identity=0.93),
    dict(scaffold_gap=20_000, min_scaffold_length=3_000, min_block_length=1_500),
    dict(scaffold_gap=0, min_identity=0.9),
]


def _cfg(sw, kw):
    return sw.FilterConfig(**{k: (getattr(sw.FilterMode, v) if isinstance(v, str) else v) for k, v in kw.items()})


def _ocfg(kw):
    m = {"OneToOne": orc.ONE_TO_ONE, "OneToMany": orc.ONE_TO_MANY}
    return orc.Config(**{k: (m[v] if isinstance(v, str) else v) for k, v in kw.items()})


@pytest.fixture(scope="module")
def grouped():
    import sweepga_amd as sw
    rng = np.random.default_rng(77)
    rec, g = _sorted_by_query_genome(gen.random_records(rng, 60_000, n_genomes=9, chrs_per_genome=2, span=600_000))
    return rec, sw.pack_records(gen.records_to_meta(rec))


@pytest.mark.parametrize("ci", range(len(CONFIGS)))
def test_streamed_equals_one_piece_and_oracle(grouped, ci, monkeypatch):
    import sweepga_amd as sw
    rec, packed = grouped
    kw = CONFIGS[ci]
    f = sw.PafFilter(_cfg(sw, kw))
    monkeypatch.setenv("SWG_STREAM", "0")
    want_st, want_ch = [a.copy() for a in f.filter_columns(packed)]
    want = (f.last_stats.n_retained, f.last_stats.n_swept, f.last_stats.n_chains, f.last_stats.n_chains_kept, f.last_stats.n_out)
    monkeypatch.delenv("SWG_STREAM")
    for chunk in (4_000, 15_000, 25_000):
        monkeypatch.setenv("SWG_STREAM_CHUNK", str(chunk))
        assert len(sw.stream_plan(packed, chunk)) >= 3   # really streamed: at least two ranges
        st, ch = f.filter_columns(packed)
        assert np.array_equal(st, want_st), (kw, chunk)
        assert np.array_equal(ch, want_ch), (kw, chunk)
        s = f.last_stats
        assert (s.n_retained, s.n_swept, s.n_chains, s.n_chains_kept, s.n_out) == want and s.n_in == packed.n
        assert s.h2d_ms > 0 and s.device_ms > 0
    ost, och = orc.apply_filters(_ocfg(kw), rec)
    assert np.array_equal(want_st, ost) and np.array_equal(want_ch, och)


def test_derived_identity_equals_explicit(grouped, monkeypatch):
    """identity = NULL: matches / max(block_len, 1) on the device, bit for bit what the host computed -- same results, one
    piece and streamed, sweep flags (scores read it) and scaffold flags alike."""
    import copy
    import sweepga_amd as sw
    rec, packed = grouped
    assert np.array_equal(packed.cols["identity"], packed.cols["matches"] / np.maximum(packed.cols["block_len"], 1))
    bare = copy.copy(packed)
    bare.cols = dict(packed.cols)
    bare.cols["identity"] = None
    for kw in (CONFIGS[0], CONFIGS[2], CONFIGS[3], CONFIGS[4]):
        f = sw.PafFilter(_cfg(sw, kw))
        for env in ({"SWG_STREAM": "0"}, {"SWG_STREAM_CHUNK": "9000"}):
            for k, v in env.items():
                monkeypatch.setenv(k, v)
            a = [x.copy() for x in f.filter_columns(packed)]
            b = f.filter_columns(bare)
            assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]), (kw, env)
            for k in env:
                monkeypatch.delenv(k)


@pytest.mark.parametrize("n_ctx", [2, 3])
def test_streamed_multi_context(grouped, n_ctx, monkeypatch):
    """swg_filter_multi on grouped records: ranges dealt to the contexts (all on the one test GPU), each streaming its own;
    chain numbers made global on the host."""
    import sweepga_amd as sw
    rec, packed = grouped
    ctxs = [sw.Context(0) for _ in range(n_ctx)]
    try:
        for kw in CONFIGS:
            f = sw.PafFilter(_cfg(sw, kw), ctx=ctxs[0])
            monkeypatch.setenv("SWG_STREAM", "0")
            want_st, want_ch = [a.copy() for a in f.filter_columns(packed)]
            want = (f.last_stats.n_retained, f.last_stats.n_swept, f.last_stats.n_chains_kept, f.last_stats.n_out)
            monkeypatch.delenv("SWG_STREAM")
            monkeypatch.setenv("SWG_STREAM_CHUNK", "5000")
            st, ch = f.filter_columns_multi(packed, ctxs)
            assert np.array_equal(st, want_st), kw
            assert np.array_equal(ch, want_ch), kw
            s = f.last_stats
            assert (s.n_retained, s.n_swept, s.n_chains_kept, s.n_out) == want
            monkeypatch.delenv("SWG_STREAM_CHUNK")
    finally:
        for c in ctxs:
            c.close()


def test_ungrouped_records_take_the_one_piece_path(monkeypatch):
    import sweepga_amd as sw
    rng = np.random.default_rng(78)
    rec = gen.random_records(rng, 20_000, n_genomes=5, chrs_per_genome=2, span=400_000)
    packed = sw.pack_records(gen.records_to_meta(rec))
    monkeypatch.setenv("SWG_STREAM_CHUNK", "2000")
    assert sw.stream_plan(packed, 2000) == []
    kw = CONFIGS[0]
    st, ch = sw.PafFilter(_cfg(sw, kw)).filter_columns(packed)
    ost, och = orc.apply_filters(_ocfg(kw), rec)
    assert np.array_equal(st, ost) and np.array_equal(ch, och)


def test_cli_on_a_grouped_paf_without_dv_tags(tmp_path, monkeypatch):
    """sweepga-gpu on a PAF no line of which carries dv:f: -> the ingest reports the identity column as derived, the command
    line leaves it on the host, and the call is streamed (forced small ranges): byte-identical to the oracle's CLI; the same
    PAF with dv:f: tags (identity column uploaded) as the control."""
    import os
    import subprocess
    import sweepga_amd as sw
    from sweepga_amd import build, paf
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    rng = np.random.default_rng(79)
    rec, _ = _sorted_by_query_genome(gen.random_records(rng, 30_000, n_genomes=6, chrs_per_genome=2, span=400_000))
    flags = ["--scaffold-filter", "1:1", "--scaffold-jump", "20k", "--scaffold-mass", "2k", "--scaffold-dist", "10k"]
    monkeypatch.setenv("SWG_STREAM_CHUNK", "6000")
    for dv in (False, True):
        p = tmp_path / f"in_{int(dv)}.paf"
        p.write_text(gen.records_to_paf(rng, rec, dv_tags=dv))
        with paf.PafFile(str(p)) as h:
            assert bool(h.identity_is_derived) is (not dv)
        ref, out = tmp_path / f"ref_{int(dv)}.paf", tmp_path / f"gpu_{int(dv)}.paf"
        subprocess.check_call([os.path.join(root, "oracle", "sweepga-ref"), str(p), "--output-file", str(ref), *flags])
        for extra in ([], ["--devices", "0,0"]):
            r = subprocess.run([build.CLI, str(p), "--output-file", str(out), "--quiet", *flags, *extra], capture_output=True, text=True)
            assert r.returncode == 0, r.stderr
            assert out.read_bytes() == ref.read_bytes(), (dv, extra)
