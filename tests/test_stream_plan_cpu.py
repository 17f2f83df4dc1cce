"""swg_stream_plan (csrc/host/stream_plan.h): the record ranges of the streamed host path -- whole query genomes, cut where
the query genome changes -- and the cases in which there is no plan (records not grouped by query genome, the reference's
two genome-prefix rules disagreeing).  Host code: runs without a GPU."""
import numpy as np

from tests import gen


def _sorted_by_query_genome(rec):
    import copy
    g = np.array([int(q.split("#")[0][1:]) for q in rec.qname])
    order = np.argsort(g, kind="stable")
    out = copy.copy(rec)
    for k in ("qs", "qe", "ts", "te", "block_length", "identity", "matches", "strand"):
        setattr(out, k, np.ascontiguousarray(getattr(rec, k)[order]))
    out.qname = [rec.qname[i] for i in order]
    out.tname = [rec.tname[i] for i in order]
    out.rank = np.arange(len(order), dtype=np.uint64)
    return out, g[order]


def test_plan_cuts_at_query_genome_boundaries():
    import sweepga_amd as sw
    rng = np.random.default_rng(3)
    rec, g = _sorted_by_query_genome(gen.random_records(rng, 20_000, n_genomes=7, chrs_per_genome=3))
    packed = sw.pack_records(gen.records_to_meta(rec))
    starts = [0] + [i for i in range(1, len(g)) if g[i] != g[i - 1]] + [len(g)]
    b = sw.stream_plan(packed, 1)           # target 1: every run is a range
    assert b == starts
    b = sw.stream_plan(packed, 5_000)       # ranges of whole runs, at least 5,000 records (a small tail joins its neighbour)
    assert b[0] == 0 and b[-1] == len(g) and set(b) <= set(starts) and len(b) >= 3
    sizes = np.diff(b)
    assert (sizes[:-1] >= 5_000).all() and sizes[-1] >= 2_500
    b = sw.stream_plan(packed, 10 ** 9)     # one range
    assert b == [0, len(g)]


def test_no_plan_when_not_grouped():
    import sweepga_amd as sw
    rng = np.random.default_rng(4)
    rec = gen.random_records(rng, 5_000, n_genomes=4, chrs_per_genome=2)   # query genomes interleaved at random
    packed = sw.pack_records(gen.records_to_meta(rec))
    assert sw.stream_plan(packed, 100) == []
    # grouped except for one stray record of the first genome at the very end: a second run of that genome
    rec2, g = _sorted_by_query_genome(rec)
    rec2.qname[-1] = rec2.qname[0]
    assert sw.stream_plan(sw.pack_records(gen.records_to_meta(rec2)), 100) == []


def test_no_plan_when_prefix_rules_disagree():
    import sweepga_amd as sw
    rng = np.random.default_rng(5)
    rec, _ = _sorted_by_query_genome(gen.random_records(rng, 3_000, n_genomes=3, chrs_per_genome=2))
    rec.qname = [q.replace("#chr", "#x#chr") for q in rec.qname]   # three '#': last-'#' prefix != first-two-parts prefix
    packed = sw.pack_records(gen.records_to_meta(rec))
    assert sw.stream_plan(packed, 100) == []


def test_many_threads_same_plan():
    """Run boundaries that fall on or next to the host threads' slice boundaries."""
    import sweepga_amd as sw
    rng = np.random.default_rng(6)
    rec, g = _sorted_by_query_genome(gen.random_records(rng, 300_000, n_genomes=40, chrs_per_genome=1, max_len=500))
    packed = sw.pack_records(gen.records_to_meta(rec))
    starts = [0] + [i for i in range(1, len(g)) if g[i] != g[i - 1]] + [len(g)]
    assert sw.stream_plan(packed, 1) == starts
