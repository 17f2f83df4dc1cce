"""u64 coordinates (RecordMeta, src/paf_filter.rs:58-62) through the 32-bit device layout: swg_filter64 (host columns,
rebased by host threads), swg_filter_device64 (device columns, rebased by two kernels), the PAF / .1aln front ends and
the command line, all against the oracle run on the unrebased u64 records."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from tests import gen, orc
from tests.test_gpu_scaffold import SCAFFOLD_CFGS, _cfg_pair

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

shifted = gen.shifted


@pytest.fixture(scope="module")
def sw():
    import sweepga_amd
    sweepga_amd.default_context()
    return sweepga_amd


def _cfgs(sw, cfg_i):
    kw = dict(SCAFFOLD_CFGS[cfg_i])
    for k in ("mapping_filter_mode", "scaffold_filter_mode"):
        if k in kw:
            kw[k] = sw.FilterMode(kw[k])
    if "scoring_function" in kw:
        kw["scoring_function"] = sw.ScoringFunction(kw["scoring_function"])
    return _cfg_pair(sw, **kw)


@pytest.mark.parametrize("cfg_i", range(len(SCAFFOLD_CFGS)))
@pytest.mark.parametrize("seed", range(2))
def test_filter64_matches_oracle(sw, seed, cfg_i):
    rng = np.random.default_rng(900 + 10 * cfg_i + seed)
    n = int(rng.choice([1, 300, 6_000, 25_000]))
    rec0 = gen.random_records(rng, n, n_genomes=int(rng.integers(1, 4)), chrs_per_genome=int(rng.integers(1, 4)),
                              span=int(rng.choice([100_000, 1_000_000])), minus_frac=0.3)
    rec, off = shifted(rec0, rng)
    cfg, ocfg = _cfgs(sw, cfg_i)
    packed = sw.pack_records(gen.records_to_meta(rec))
    assert packed.wide == (max(off.values()) > 0 and int(max(rec.qe.max(), rec.te.max())) > 0xFFFFFFFF)
    status, chain = sw.PafFilter(cfg).filter_columns(packed)
    ost, och = orc.apply_filters(ocfg, rec)
    assert np.array_equal(status, ost), int((status != ost).sum())
    assert np.array_equal(chain, och), int((chain != och).sum())


def test_a_sequence_touched_over_more_than_2_32_bases(sw, tmp_path):
    """SURVEY H6, RecordMeta's u64 fields (src/paf_filter.rs:58-62): a sequence whose mappings against DIFFERENT genomes lie more
    than 2^32 bases apart does not fit one constant per sequence; the front ends then rebase per sweep segment -- (sequence, genome
    of the other side), which every comparison of the filter nests in (tests/test_wide_cpu.py shows the invariance on the
    oracle).  Status and chain numbers against the oracle on the u64 records: host columns (swg_filter64), device columns
    (swg_filter_device64) and a PAF through the command line, for the CLI defaults, a scaffold flag set with a rescue, the 1:1
    sweep alone and the full pipeline behind it."""
    from sweepga_amd import build
    from sweepga_amd._lib import SwgRecords, SwgStats
    rng = np.random.default_rng(4711)
    rec0 = gen.random_records(rng, 30_000, n_genomes=4, chrs_per_genome=2, span=1_500_000, minus_frac=0.3)
    rec = gen.shifted_by_axis(rec0, rng)
    # (the premise: some sequence's query coordinates spread over 2^32 bases or more)
    spread = {}
    for q, a, b in zip(rec.qname, rec.qs, rec.qe):
        lo, hi = spread.get(q, (int(a), int(b)))
        spread[q] = (min(lo, int(a)), max(hi, int(b)))
    assert max(hi - lo for lo, hi in spread.values()) >= 2**32
    packed = sw.pack_records(gen.records_to_meta(rec))
    assert packed.wide
    cfgs = [dict(),
            dict(scaffold_gap=3_000, min_scaffold_length=4_000, scaffold_max_deviation=5_000),
            dict(mapping_filter_mode="OneToOne", scaffold_gap=0),
            dict(mapping_filter_mode="OneToOne", scaffold_filter_mode="OneToOne", scaffold_gap=5_000, min_scaffold_length=1_000,
                 scaffold_max_deviation=30_000)]
    hip = Hip()
    try:
        r = SwgRecords()
        n = len(rec)
        r.n = n
        for k in ("q_id", "t_id", "q_start", "q_end", "t_start", "t_end", "identity", "matches", "block_len", "strand"):
            setattr(r, k, hip.up(packed.cols[k]))
        r.n_seq = packed.n_seq
        r.seq_genome_last = hip.up(packed.seq_genome_last)
        r.n_genome_last = packed.n_genome_last
        r.seq_genome_two = hip.up(packed.seq_genome_two)
        r.n_genome_two = packed.n_genome_two
        d_status, d_chain = hip.alloc(n), hip.alloc(4 * n)
        ctx = sw.default_context()
        for kw in cfgs:
            cfg = sw.FilterConfig(**{k: (getattr(sw.FilterMode, v) if isinstance(v, str) else v) for k, v in kw.items()})
            ocfg = orc.Config(**{k: (int(getattr(sw.FilterMode, v)) if isinstance(v, str) else v) for k, v in kw.items()})
            ost, och = orc.apply_filters(ocfg, rec)
            status, chain = sw.PafFilter(cfg).filter_columns(packed)                       # host columns: swg_filter64
            assert np.array_equal(status, ost) and np.array_equal(chain, och), (kw, int((status != ost).sum()))
            stats = SwgStats()
            cc = cfg.to_c(False, False)
            ctx.check(ctx.lib.swg_filter_device64(ctx.handle, C.byref(r), C.byref(cc), C.c_void_p(d_status), C.c_void_p(d_chain),
                                                  C.byref(stats)))                        # device columns: two more kernels
            ctx.synchronize()
            status, chain = hip.down(d_status, np.uint8, n), hip.down(d_chain, np.uint32, n)
            assert np.array_equal(status, ost) and np.array_equal(chain, och), (kw, "device", int((status != ost).sum()))
    finally:
        hip.free()
    # the same records as a PAF through the command line, byte for byte against the oracle's CLI (which keeps u64)
    inp = tmp_path / "wide.paf"
    inp.write_text(gen.records_to_paf(rng, rec), newline="")
    for flags in ([], ["--num-mappings", "1:1", "--scaffold-jump", "0"],
                  ["--num-mappings", "1:1", "--scaffold-filter", "1:1", "--scaffold-dist", "20000"]):
        a, b = tmp_path / "a.paf", tmp_path / "b.paf"
        out = subprocess.run([build.CLI, str(inp), "--output-file", str(a), "--quiet", *flags], capture_output=True, text=True)
        assert out.returncode == 0, out.stderr
        subprocess.check_call([os.path.join(ROOT, "oracle", "sweepga-ref"), str(inp), "--output-file", str(b), *flags])
        assert a.read_bytes() == b.read_bytes() and a.stat().st_size > 0, flags


def test_filter64_range_errors(sw):
    """What stays refused: ONE sweep segment -- a sequence against one genome -- touched over 2^32 bases or more, and a matches /
    block length that does not fit 32 bits."""
    from sweepga_amd import RecordMeta, SwgError
    f = sw.PafFilter(sw.FilterConfig())
    a = RecordMeta(0, "g1#1#a", "g2#1#b", 10, 2000, 0, 2000, 2000, 0.9, 1800, 2000, "+")
    far = RecordMeta(1, "g1#1#a", "g2#1#b", 2**32 + 10, 2**32 + 2000, 0, 2000, 2000, 0.9, 1800, 2000, "+")
    with pytest.raises(SwgError, match="against one genome touch spans 2\\^32"):
        f.filter_columns(sw.pack_records([a, far]))                      # the stretch [10, 2^32 + 2000) of one segment does not fit
    far2 = RecordMeta(1, "g1#1#a", "g3#1#b", 2**32 + 10, 2**32 + 2000, 0, 2000, 2000, 0.9, 1800, 2000, "+")
    st, ch = f.filter_columns(sw.pack_records([a, far2]))                # against another genome: a segment of its own
    assert len(st) == 2
    big = RecordMeta(0, "g1#1#a", "g2#1#b", 10, 2000, 0, 2000, 2**32, 0.9, 1800, 2**32, "+")
    with pytest.raises(SwgError, match="block_length >= 2\\^32"):
        f.filter_columns(sw.pack_records([big]))
    ok = RecordMeta(0, "g1#1#a", "g2#1#b", 2**32 + 10, 2**32 + 2000, 2**50, 2**50 + 2000, 2000, 0.9, 1800, 2000, "+")
    st, ch = f.filter_columns(sw.pack_records([ok, ok]))
    assert len(st) == 2


class Hip:
    """hipMalloc / hipMemcpy / hipFree of the runtime libsweepga_gpu.so itself is linked against."""

    def __init__(self):
        from sweepga_amd import _lib
        self.lib = _lib.load()   # dlsym on the library's handle reaches its dependencies: the very runtime it runs on
        self.lib.hipMalloc.restype = self.lib.hipMemcpy.restype = self.lib.hipFree.restype = C.c_int
        self.lib.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
        self.lib.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        self.lib.hipFree.argtypes = [C.c_void_p]
        self.held = []

    def up(self, a):
        a = np.ascontiguousarray(a)
        p = C.c_void_p()
        assert self.lib.hipMalloc(C.byref(p), max(a.nbytes, 8)) == 0
        assert self.lib.hipMemcpy(p, a.ctypes.data, a.nbytes, 1) == 0
        self.held.append(p)
        return p.value

    def alloc(self, nbytes):
        p = C.c_void_p()
        assert self.lib.hipMalloc(C.byref(p), max(nbytes, 8)) == 0
        self.held.append(p)
        return p.value

    def down(self, ptr, dtype, n):
        out = np.zeros(n, dtype=dtype)
        assert self.lib.hipMemcpy(out.ctypes.data, C.c_void_p(ptr), out.nbytes, 2) == 0
        return out

    def free(self):
        for p in self.held:
            self.lib.hipFree(p)
        self.held = []


@pytest.mark.parametrize("cfg_i", [0, 2, 5])
def test_filter_device64_matches_oracle(sw, cfg_i):
    """Device-resident u64 columns: seq_lo + rebase kernels, then the same pipeline."""
    from sweepga_amd._lib import SwgRecords, SwgStats
    rng = np.random.default_rng(4242 + cfg_i)
    n = 40_000
    rec0 = gen.random_records(rng, n, n_genomes=3, chrs_per_genome=2, span=2_000_000, minus_frac=0.3)
    # grouped by query (a wavefront then names one query: the one-atomic-per-wavefront path) in the first half, shuffled after
    order = np.concatenate([np.argsort(np.array(rec0.qname[:n // 2]), kind="stable"), np.arange(n // 2, n)])
    rec0 = orc.Records([rec0.qname[i] for i in order], [rec0.tname[i] for i in order],
                       *(np.ascontiguousarray(getattr(rec0, f)[order]) for f in ("qs", "qe", "ts", "te", "block_length", "identity",
                                                                                "matches", "strand")), np.arange(n, dtype=np.uint64))
    rec, _ = shifted(rec0, rng)
    cfg, ocfg = _cfgs(sw, cfg_i)
    packed = sw.pack_records(gen.records_to_meta(rec))
    assert packed.wide
    hip = Hip()
    try:
        r = SwgRecords()
        r.n = n
        for k in ("q_id", "t_id", "q_start", "q_end", "t_start", "t_end", "identity", "matches", "block_len", "strand"):
            setattr(r, k, hip.up(packed.cols[k]))
        r.n_seq = packed.n_seq
        r.seq_genome_last = hip.up(packed.seq_genome_last)
        r.n_genome_last = packed.n_genome_last
        r.seq_genome_two = hip.up(packed.seq_genome_two)
        r.n_genome_two = packed.n_genome_two
        d_status, d_chain = hip.alloc(n), hip.alloc(4 * n)
        ctx = sw.default_context()
        stats = SwgStats()
        cc = cfg.to_c(False, False)
        ctx.check(ctx.lib.swg_filter_device64(ctx.handle, C.byref(r), C.byref(cc), C.c_void_p(d_status), C.c_void_p(d_chain),
                                              C.byref(stats)))
        ctx.synchronize()
        status, chain = hip.down(d_status, np.uint8, n), hip.down(d_chain, np.uint32, n)
        ost, och = orc.apply_filters(ocfg, rec)
        assert np.array_equal(status, ost), int((status != ost).sum())
        assert np.array_equal(chain, och), int((chain != och).sum())
        assert stats.n_out == int((ost != 0).sum())
        # a stretch that does not fit: the device reports the record
        qe = packed.cols["q_end"].copy()
        worst = int(np.argmax(qe))
        qe[worst] += np.uint64(2**32)
        r.q_end = hip.up(qe)
        rc = ctx.lib.swg_filter_device64(ctx.handle, C.byref(r), C.byref(cc), C.c_void_p(d_status), C.c_void_p(d_chain), C.byref(stats))
        assert rc == -5 and f"record {worst}:".encode() in ctx.lib.swg_last_error(ctx.handle)
    finally:
        hip.free()


def test_cli_on_wide_paf(sw, tmp_path):
    """sweepga-gpu on a PAF whose coordinates sit beyond 2^32: byte-identical to the oracle CLI (which keeps u64)."""
    from sweepga_amd import build
    rng = np.random.default_rng(77)
    rec0 = gen.random_records(rng, 8_000, n_genomes=3, chrs_per_genome=2, span=1_000_000, minus_frac=0.3)
    rec, _ = shifted(rec0, rng)
    inp = tmp_path / "wide.paf"
    inp.write_text(gen.records_to_paf(rng, rec), newline="")
    ref_cli = os.path.join(ROOT, "oracle", "sweepga-ref")
    gpu_cli = build.CLI
    for flags in ([], ["--num-mappings", "1:1", "--scaffold-jump", "0"],
                  ["--num-mappings", "1:1", "--scaffold-filter", "1:1", "--scaffold-dist", "20000"],
                  ["--min-aln-identity", "ani50", "--scaffold-jump", "10k", "--scaffold-mass", "2k"]):
        a, b = tmp_path / "a.paf", tmp_path / "b.paf"
        r = subprocess.run([gpu_cli, str(inp), "--output-file", str(a), "--quiet", *flags], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        subprocess.check_call([ref_cli, str(inp), "--output-file", str(b), *flags])
        assert a.read_bytes() == b.read_bytes() and a.stat().st_size > 0, flags


def test_aln_records_wide(sw):
    """.1aln derivation with coordinates beyond 2^32 -> rebased columns -> the filter, against the oracle on the u64 values."""
    from sweepga_amd import AlnRecords
    rng = np.random.default_rng(31)
    rec0 = gen.random_records(rng, 5_000, n_genomes=2, chrs_per_genome=2, span=500_000, minus_frac=0.3, zero_frac=0.0)
    rec, off = shifted(rec0, rng)
    strand = "".join(chr(int(c)) for c in rec.strand)
    with AlnRecords(rec.qname, rec.tname, rec.qs, rec.qe, rec.ts, rec.te, rec.matches, strand) as a:
        assert a.seq_offsets is not None
        packed = a.packed()
    cfg, ocfg = _cfgs(sw, 1)
    status, chain = sw.PafFilter(cfg).filter_columns(packed)
    # what extract_1aln_metadata derives (src/unified_filter.rs:107-123): block = spans' sum, identity = matches / query span
    qspan = rec.qe - rec.qs
    orec = orc.Records(rec.qname, rec.tname, rec.qs, rec.qe, rec.ts, rec.te, qspan + (rec.te - rec.ts),
                       np.where(qspan > 0, rec.matches / np.maximum(qspan, 1), 0.0), rec.matches, rec.strand, rec.rank)
    ost, och = orc.apply_filters(ocfg, orec)
    assert np.array_equal(status, ost) and np.array_equal(chain, och)
