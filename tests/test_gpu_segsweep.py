"""The segment-resident k = 1 sweep (csrc/swg_segsort.hip, seg_sweep_body; src/plane_sweep_exact.rs:197-352; opt-in: SWG_SEG_SWEEP=1,
see DESIGN.md section 3.2c for why it is not the default): over a pair-grouped input of more than 65,536 records every (sequence,
genome of the other side) segment of up to 32,768 places is sorted AND swept in one LDS residency -- no sorted columns in memory,
no carry-in routing, no tile kernel.  Against the CPU oracle, with the launch
table showing the path: every size class (one wavefront, 256 threads, 1,024 threads in several batches with carried intervals),
segments read in place and through a list, dead and zero-length records, score ties, every kind of threshold; the longest
segments left to the tile kernels in a compact list; deep data handed back whole; SWG_SEG_SWEEP=0 giving the same answers.
-m gpu only."""
import os
import subprocess
import sys

import numpy as np
import pytest

from tests import gen, orc
from tests.test_gpu_pairs import pair_major

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def sw():
    import sweepga_amd
    sweepga_amd.default_context(0)
    old = os.environ.get("SWG_SEG_SWEEP")
    os.environ["SWG_SEG_SWEEP"] = "1"      # (the library reads the knob at every call)
    yield sweepga_amd
    if old is None:
        del os.environ["SWG_SEG_SWEEP"]
    else:
        os.environ["SWG_SEG_SWEEP"] = old


def sweep_vs_oracle(sw, rec, cfg_kw, keep_self=False):
    kw = {k: (getattr(sw.FilterMode, v) if isinstance(v, str) else v) for k, v in cfg_kw.items()}
    okw = {k: (int(getattr(sw.FilterMode, v)) if isinstance(v, str) else v) for k, v in cfg_kw.items()}
    ctx = sw.default_context(0)
    ctx.profile_reset()
    ctx.profile(True)
    st, ch = sw.PafFilter(sw.FilterConfig(**kw)).with_keep_self(keep_self).filter_columns(sw.pack_records(gen.records_to_meta(rec)))
    ctx.profile(False)
    table = ctx.profile_table()
    ost, och = orc.apply_filters(orc.Config(keep_self=keep_self, **okw), rec)
    bad = np.flatnonzero((st != ost) | (ch != och))
    assert bad.size == 0, (cfg_kw, sorted(k for k in table if k.startswith(("seg_", "sweep_"))), int(bad.size), bad[:10].tolist(),
                           st[bad[:10]].tolist(), ost[bad[:10]].tolist())
    return table


def names_of(table, prefix):
    return sorted(k for k in table if k.startswith(prefix))


SWEEP = {"mapping_filter_mode": "OneToOne", "scaffold_gap": 0}


@pytest.mark.parametrize("thr", [0.0, 0.5, 0.95, 1.0])
def test_every_size_class_against_the_oracle(sw, thr):
    """3 genomes x 3 chromosomes, 120,000 records: segments of a few thousand records (the 256-thread class, several batches with
    carried intervals), plus one genome with 40 small contigs (the one-wavefront class) -- score ties (identities of two
    decimals), equal starts, zero-length records, step-1 floors that kill a fifth of the records."""
    rng = np.random.default_rng(700 + int(thr * 100))
    rec = gen.random_records(rng, 120_000, n_genomes=3, chrs_per_genome=3, span=12_000_000, zero_frac=0.02)
    small = gen.random_records(rng, 30_000, n_genomes=2, chrs_per_genome=40, span=600_000, zero_frac=0.02)
    small.qname = [x.replace("g", "s") for x in small.qname]
    small.tname = [x.replace("g", "s") for x in small.tname]
    both = orc.Records(rec.qname + small.qname, rec.tname + small.tname,
                       *[np.concatenate([getattr(rec, c), getattr(small, c)]) for c in ("qs", "qe", "ts", "te", "block_length", "identity", "matches", "strand")],
                       np.arange(len(rec) + len(small), dtype=np.uint64))
    both.qs = both.qs // 64 * 64           # equal starts
    both.qe = np.maximum(both.qe, both.qs)
    both = pair_major(both, rng)
    for cfg in (dict(SWEEP, overlap_threshold=thr),
                dict(SWEEP, overlap_threshold=thr, min_identity=0.8, min_block_length=300),
                dict(SWEEP, overlap_threshold=thr, scoring_function=2)):     # scores with many ties
        table = sweep_vs_oracle(sw, both, cfg)
        assert names_of(table, "seg_sweep"), sorted(table)
        assert not names_of(table, "sweep_tile") and not names_of(table, "route_"), sorted(table)


def test_the_large_class_runs_in_batches_with_carried_intervals(sw):
    """One genome pair, two chromosomes a side: segments of ~25,000 records at depth ~1.2 -- the 1,024-thread class, five batches,
    intervals carried from batch to batch, long windows left to whole wavefronts; the query axis in place (one run per
    segment), the target axis through the list of several runs."""
    rng = np.random.default_rng(801)
    rec = gen.random_records(rng, 100_000, n_genomes=2, chrs_per_genome=2, span=30_000_000, zero_frac=0.01, self_frac=0.0)
    rec = pair_major(rec, rng)
    for thr in (0.3, 0.95, 1.0):
        table = sweep_vs_oracle(sw, rec, dict(SWEEP, overlap_threshold=thr))
        assert "seg_sweep_big" in table and not names_of(table, "sweep_tile"), sorted(table)
    # ... and the scaffold stage behind it takes the same flags
    table = sweep_vs_oracle(sw, rec, {"mapping_filter_mode": "OneToOne", "scaffold_gap": 5_000, "min_scaffold_length": 2_000, "scaffold_max_deviation": 4_000})
    assert "seg_sweep_big" in table, sorted(table)


def test_the_longest_segments_go_to_the_tile_kernels_in_a_compact_list(sw):
    """A pair of 50,000 records (one segment of more than 32,768 places on either axis) next to ordinary pairs: the resident sweep
    answers the ordinary segments, the long one's begins are sorted into a list of their own and swept by the tile kernels."""
    rng = np.random.default_rng(802)
    big = gen.random_records(rng, 50_000, n_genomes=1, chrs_per_genome=1, span=40_000_000, zero_frac=0.01, self_frac=0.0)
    big.qname = ["x#1#c"] * len(big)
    big.tname = ["y#1#c"] * len(big)
    rec = pair_major(gen.random_records(rng, 70_000, n_genomes=3, chrs_per_genome=2, span=20_000_000, zero_frac=0.01), rng)
    both = orc.Records(rec.qname[:30_000] + big.qname + rec.qname[30_000:], rec.tname[:30_000] + big.tname + rec.tname[30_000:],
                       *[np.concatenate([getattr(rec, c)[:30_000], getattr(big, c), getattr(rec, c)[30_000:]])
                         for c in ("qs", "qe", "ts", "te", "block_length", "identity", "matches", "strand")],
                       np.arange(len(big) + len(rec), dtype=np.uint64))
    both = pair_major(both, rng)
    for thr in (0.5, 1.0):
        table = sweep_vs_oracle(sw, both, dict(SWEEP, overlap_threshold=thr))
        assert names_of(table, "seg_sweep") and "sweep_tile_k1" in table and "combine_begins" in table, sorted(table)


def test_deep_data_is_handed_back_to_the_tile_kernels(sw):
    """30,000 intervals per segment over 40 kbp: hundreds of intervals active everywhere -- more carried intervals than a batch has
    room for.  The resident sweep raises its flag, the axis runs on the tile kernels, the answer is the oracle's; the context
    remembers and does not try again on the next call of that size."""
    rng = np.random.default_rng(803)
    rec = gen.random_records(rng, 70_000, n_genomes=2, chrs_per_genome=1, span=40_000, max_len=4_000, zero_frac=0.0, self_frac=0.0)
    rec = pair_major(rec, rng)
    table = sweep_vs_oracle(sw, rec, dict(SWEEP, overlap_threshold=0.9))
    assert names_of(table, "seg_sweep") and "sweep_tile_k1" in table, sorted(table)
    table = sweep_vs_oracle(sw, rec, dict(SWEEP, overlap_threshold=0.9))
    assert not names_of(table, "seg_sweep") and "sweep_tile_k1" in table, sorted(table)
    # (a call of another size starts afresh)
    small = pair_major(gen.random_records(rng, 70_000 * 3, n_genomes=3, chrs_per_genome=3, span=40_000_000), rng)
    table = sweep_vs_oracle(sw, small, dict(SWEEP))
    assert names_of(table, "seg_sweep") and not names_of(table, "sweep_tile"), sorted(table)


def test_a_segment_beyond_the_longest_class_sends_the_axis_to_the_general_sort(sw):
    """the longest size class passes over its whole segment once per batch of 8,192 records -- quadratic in the
    segment's size on ONE work-group.  One query chromosome against a genome of 60 contigs, pair-major, 1:1: the query axis has a
    single segment of 200,000 records.  Beyond 131,072 places the plan raises its flag and the axis is sorted by the radix
    passes (a few ms); the answer is the oracle's, with and without the resident sweep."""
    rng = np.random.default_rng(804)
    rec = gen.random_records(rng, 200_000, n_genomes=1, chrs_per_genome=60, span=50_000_000, zero_frac=0.0, self_frac=0.0)
    rec.qname = ["q#1#c"] * len(rec)
    rec.tname = [x.replace("g0", "t") for x in rec.tname]
    rec = pair_major(rec, rng)
    import time
    for knob in ("1", "0"):
        os.environ["SWG_SEG_SWEEP"] = knob
        t0 = time.time()
        table = sweep_vs_oracle(sw, rec, dict(SWEEP))
        assert any(k.startswith("begin_gather") for k in table), sorted(table)   # the query axis: the general sort
        assert time.time() - t0 < 20.0
    os.environ["SWG_SEG_SWEEP"] = "1"


def test_knob_off_gives_the_same_answer(sw):
    code = r"""
import numpy as np, sys
sys.path.insert(0, %r)
import sweepga_amd as sw
from tests import gen, orc
from tests.test_gpu_pairs import pair_major
rng = np.random.default_rng(801)
rec = pair_major(gen.random_records(rng, 100000, n_genomes=2, chrs_per_genome=2, span=30000000, zero_frac=0.01, self_frac=0.0), rng)
ctx = sw.default_context(0)
ctx.profile(True)
st, ch = sw.PafFilter(sw.FilterConfig(mapping_filter_mode=sw.FilterMode.OneToOne, scaffold_gap=0, overlap_threshold=0.3)).filter_columns(sw.pack_records(gen.records_to_meta(rec)))
t = ctx.profile_table()
assert not any(k.startswith("seg_sweep") for k in t) and any(k.startswith("seg_sort") for k in t) and "sweep_tile_k1" in t, sorted(t)
ost, och = orc.apply_filters(orc.Config(mapping_filter_mode=int(sw.FilterMode.OneToOne), scaffold_gap=0, overlap_threshold=0.3), rec)
assert np.array_equal(st, ost) and np.array_equal(ch, och)
print("ok")
""" % ROOT
    out = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, SWG_SEG_SWEEP="0"), capture_output=True, text=True, cwd=ROOT)
    assert out.returncode == 0 and "ok" in out.stdout, out.stderr[-2000:]


def test_the_streamed_sweep_over_the_sorted_begins(sw):
    """SWG_SEG_STREAM=1 (opt-in, csrc/swg_segsort.hip seg_stream_body): behind seg_sort, every segment's sorted begins stream
    through LDS chunk by chunk with the intervals that reach further carried along -- sweep_batch, the code the fused kernel
    runs -- instead of the carry-in routing and the tile kernels.  Segments of every size (one chunk, several chunks, the
    longest class), dead records, all thresholds, the scaffold stage behind it; deep data goes on to the tile kernels over the
    same arrays."""
    os.environ["SWG_SEG_SWEEP"] = "0"
    os.environ["SWG_SEG_STREAM"] = "2"      # (2: tried even where an earlier test left the context remembering deep data of this size)
    try:
        rng = np.random.default_rng(811)
        rec = gen.random_records(rng, 150_000, n_genomes=2, chrs_per_genome=2, span=40_000_000, zero_frac=0.02, self_frac=0.0)   # ~37,000 per segment: many chunks
        small = gen.random_records(rng, 40_000, n_genomes=3, chrs_per_genome=8, span=2_000_000, zero_frac=0.02)
        small.qname = [x.replace("g", "s") for x in small.qname]
        small.tname = [x.replace("g", "s") for x in small.tname]
        both = orc.Records(rec.qname + small.qname, rec.tname + small.tname,
                           *[np.concatenate([getattr(rec, c), getattr(small, c)]) for c in ("qs", "qe", "ts", "te", "block_length", "identity", "matches", "strand")],
                           np.arange(len(rec) + len(small), dtype=np.uint64))
        both.qs = both.qs // 64 * 64
        both.qe = np.maximum(both.qe, both.qs)
        both = pair_major(both, rng)
        for cfg in (dict(SWEEP, overlap_threshold=0.0), dict(SWEEP, overlap_threshold=0.95, min_identity=0.8, min_block_length=300), dict(SWEEP, overlap_threshold=1.0),
                    {"mapping_filter_mode": "OneToOne", "scaffold_gap": 5_000, "min_scaffold_length": 2_000, "scaffold_max_deviation": 4_000}):
            table = sweep_vs_oracle(sw, both, cfg)
            assert names_of(table, "seg_stream") and names_of(table, "seg_sort") and not names_of(table, "sweep_tile") and not names_of(table, "route_"), sorted(table)
        deep = pair_major(gen.random_records(rng, 70_000, n_genomes=2, chrs_per_genome=1, span=40_000, max_len=4_000, zero_frac=0.0, self_frac=0.0), rng)
        table = sweep_vs_oracle(sw, deep, dict(SWEEP, overlap_threshold=0.9))
        assert names_of(table, "seg_stream") and "sweep_tile_k1" in table, sorted(table)
    finally:
        os.environ["SWG_SEG_SWEEP"] = "1"
        del os.environ["SWG_SEG_STREAM"]


def test_lone_intervals_settled_inside_the_segment_sort(sw):
    """SWG_SEG_LONE=1 (opt-in): while a segment's sorted starts and ends are in LDS, seg_sort marks the intervals that overlap no
    other interval of their segment -- kept whatever the limit is -- and only the others are compacted for the routing and the
    tile kernels.  k = 1 and k = 2 / 3, every threshold kind, dead and zero-length records, segments of one record, segments of
    several batches and the longest class; an input where nothing overlaps (no tile kernel at all)."""
    os.environ["SWG_SEG_SWEEP"] = "0"
    os.environ["SWG_SEG_LONE"] = "1"
    try:
        rng = np.random.default_rng(812)
        rec = gen.random_records(rng, 150_000, n_genomes=2, chrs_per_genome=2, span=40_000_000, zero_frac=0.02, self_frac=0.0)
        small = gen.random_records(rng, 40_000, n_genomes=3, chrs_per_genome=8, span=2_000_000, zero_frac=0.02)
        small.qname = [x.replace("g", "s") for x in small.qname]
        small.tname = [x.replace("g", "s") for x in small.tname]
        one = gen.random_records(rng, 300, n_genomes=1, chrs_per_genome=1, span=1_000_000, zero_frac=0.3, self_frac=0.0)   # segments of ONE record
        one.qname = [f"o{i}#1#c" for i in range(len(one))]
        one.tname = [f"p{i}#1#c" for i in range(len(one))]
        parts = (rec, small, one)
        both = orc.Records(sum((r.qname for r in parts), []), sum((r.tname for r in parts), []),
                           *[np.concatenate([getattr(r, c) for r in parts]) for c in ("qs", "qe", "ts", "te", "block_length", "identity", "matches", "strand")],
                           np.arange(sum(len(r) for r in parts), dtype=np.uint64))
        both.qs = both.qs // 64 * 64
        both.qe = np.maximum(both.qe, both.qs)
        both = pair_major(both, rng)
        for cfg in (dict(SWEEP, overlap_threshold=0.0), dict(SWEEP, overlap_threshold=0.95, min_identity=0.8, min_block_length=300), dict(SWEEP, overlap_threshold=1.0),
                    {"mapping_filter_mode": "OneToMany", "mapping_max_per_query": 2, "mapping_max_per_target": 3, "scaffold_gap": 0, "overlap_threshold": 0.5},
                    {"mapping_filter_mode": "OneToOne", "scaffold_gap": 5_000, "min_scaffold_length": 2_000, "scaffold_max_deviation": 4_000}):
            table = sweep_vs_oracle(sw, both, cfg)
            assert "begin_compact" in table and names_of(table, "seg_sort"), sorted(table)
        # nothing overlaps anything: every interval is settled by the sort
        sparse = pair_major(gen.random_records(rng, 70_000, n_genomes=2, chrs_per_genome=1, span=2_000_000_000, max_len=300, zero_frac=0.01, self_frac=0.0, syntenic_frac=0.0), rng)
        table = sweep_vs_oracle(sw, sparse, dict(SWEEP))
        assert names_of(table, "seg_sort"), sorted(table)
    finally:
        os.environ["SWG_SEG_SWEEP"] = "1"
        del os.environ["SWG_SEG_LONE"]
