"""End to end through the C++ host: `sweepga-gpu <paf> --output-file ...` must write byte-for-byte what the
oracle's `sweepga-ref` (CPU restatement of the reference binary's filter path) writes, for the same flags."""
import os
import subprocess

import numpy as np
import pytest

from tests import gen

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

FLAG_SETS = [
    [],                                                                   # CLI defaults
    ["--num-mappings", "1:1", "--scaffold-jump", "0"],                    # BASELINE config "sweep"
    ["--num-mappings", "1:1", "--scaffold-filter", "1:1", "--scaffold-dist", "20000"],   # "full"
    ["--num-mappings", "1", "--overlap", "0.5", "--scaffold-jump", "10k", "--scaffold-mass", "2k", "--scaffold-dist", "5k"],
    ["--num-mappings", "2:3", "--scoring", "length-ani", "--scaffold-jump", "20000", "--scaffold-mass", "1000",
     "--scaffold-filter", "2:1", "--scaffold-overlap", "0.3", "--min-aln-length", "200", "--min-aln-identity", "80"],
    ["--self", "--scaffolds-only", "--scaffold-jump", "30k", "--scaffold-mass", "5k", "--scoring", "matches"],
    ["--scoring", "ani", "--num-mappings", "1:many", "--scaffold-jump", "0", "--min-aln-identity", "0.9"],
    # tree sparsification of the input before the filter (src/main.rs:3640-3688): ranks then count the surviving lines
    ["--sparsify", "tree:1:1:0.3", "--num-mappings", "1:1", "--scaffold-jump", "10k", "--scaffold-mass", "2k", "--scaffold-dist", "5k"],
    ["--sparsify", "knn:2", "--scaffold-jump", "0"],
]


@pytest.fixture(scope="module")
def bins():
    from sweepga_amd import build
    return build.CLI, os.path.join(ROOT, "oracle", "sweepga-ref")


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_cli_output_byte_identical(bins, tmp_path, seed):
    cli, ref = bins
    rng = np.random.default_rng(4242 + seed)
    n = int(rng.choice([300, 5000, 40_000]))
    rec = gen.random_records(rng, n, n_genomes=int(rng.integers(2, 5)), chrs_per_genome=int(rng.integers(1, 4)),
                             span=int(rng.choice([200_000, 2_000_000])), pansn=bool(seed % 2 == 0))
    paf = tmp_path / "in.paf"
    paf.write_text(gen.records_to_paf(rng, rec))
    for k, flags in enumerate(FLAG_SETS):
        o1, o2 = tmp_path / f"gpu{k}.paf", tmp_path / f"ref{k}.paf"
        # SWG_DEBUG: the CLI poisons its (uncleared) result columns and checks that the filter wrote every entry
        r = subprocess.run([cli, str(paf), "--output-file", str(o1), "--quiet", *flags], capture_output=True, text=True,
                           env=dict(os.environ, SWG_DEBUG="1") if k % 2 == 0 else None)
        assert r.returncode == 0, r.stderr[-2000:]
        subprocess.check_call([ref, str(paf), "--output-file", str(o2), *flags])
        a, b = o1.read_bytes(), o2.read_bytes()
        assert a == b, (flags, len(a), len(b))
    assert os.path.getsize(tmp_path / "gpu0.paf") > 0


@pytest.mark.parametrize("suffix,threads", [(".paf", 1), (".paf", 7), (".paf.gz", 3)])
def test_filter_paf_native_matches_oracle_and_python_mirror(tmp_path, suffix, threads):
    """PafFilter.filter_paf = swg_filter_paf (native ingest -> GPU filter -> native egress), against the oracle's
    filter_paf (paf_filter.rs:278-289) and the Python extract/apply/write mirror."""
    import gzip
    from sweepga_amd import FilterConfig, FilterMode, PafFilter
    from tests import orc
    rng = np.random.default_rng(77)
    rec = gen.random_records(rng, 30_000, n_genomes=4, chrs_per_genome=3)
    text = gen.records_to_paf(rng, rec)
    plain = tmp_path / "in.paf"
    plain.write_text(text)
    src = tmp_path / ("in2" + suffix)
    src.write_bytes(gzip.compress(text.encode()) if suffix.endswith(".gz") else text.encode())
    cfg = FilterConfig(scaffold_gap=20_000, min_scaffold_length=5_000, scaffold_max_deviation=10_000,
                       mapping_filter_mode=FilterMode.OneToOne, mapping_max_per_query=1, mapping_max_per_target=1)
    f = PafFilter(cfg)
    timing = f.filter_paf(src, tmp_path / "native.paf", threads=threads)
    assert set(timing) == {"load", "parse", "filter", "write"} and f.last_stats.n_in >= f.last_stats.n_retained > 0
    f.filter_paf_python(plain, tmp_path / "py.paf")
    ocfg = orc.Config(scaffold_gap=20_000, min_scaffold_length=5_000, scaffold_max_deviation=10_000,
                      mapping_filter_mode=orc.ONE_TO_ONE, mapping_max_per_query=1, mapping_max_per_target=1)
    orc.filter_paf(ocfg, str(plain), str(tmp_path / "orc.paf"))
    want = (tmp_path / "orc.paf").read_bytes()
    assert len(want) > 0
    assert (tmp_path / "native.paf").read_bytes() == want
    assert (tmp_path / "py.paf").read_bytes() == want


@pytest.mark.parametrize("knob", ["SWG_SORT_FALLBACK", "SWG_SORT_WIDE", "SWG_SORT_PAIRS", "SWG_SORT_BITS8", "SWG_CHAIN_DEEP", "SWG_CHAIN_OLD", "SWG_KN_PLAIN", "SWG_TILE_512",
                                  "SWG_TILE_256", "SWG_SLOTS", "SWG_CHAIN_DEEP+SWG_CAND_GENERIC", "SWG_SORTA_PAIRS", "SWG_SORT_DROP10", "SWG_WALK_PLAIN"])
def test_cli_with_other_sort_paths(bins, tmp_path, knob):
    """SWG_SORT_FALLBACK=1 forces the three-kernel radix sort, SWG_SORT_WIDE=1 the 64-bit look-back words of the
    onesweep pass (otherwise only used for n >= 2^30), SWG_SORT_PAIRS=1 keeps the sweep's begins in 12-byte (key, index) pairs
    for every pass (otherwise 8-byte packed words after the first), SWG_SORT_BITS8=1 keeps the packed passes to 8-bit digits
    (otherwise 9-bit ones where they save a pass), SWG_CHAIN_OLD=1 the per-step selection kernels of round 2
    instead of the batch walk, SWG_CHAIN_DEEP=1 the wavefront-per-element candidate kernel of deep
    chaining groups (otherwise only used when groups average more than 8192 mappings), SWG_KN_PLAIN=1 sends every tile of a
    2 <= k < inf sweep through the plain tile kernel (otherwise only the tiles the pruned kernel leaves), SWG_TILE_512=1 gives
    every k = 1 sweep 512-begin tiles (otherwise only deep data: 64 carry-ins per 256-begin tile on average), SWG_TILE_256=1
    256-begin tiles (otherwise 128-begin ones on sparse data like this), SWG_SLOTS=1 writes the 32-byte record slots also when
    nothing sweeps and has the scaffold stage's first gather read them (otherwise only with >= 2^17 records per possible
    sequence pair), SWG_CAND_GENERIC=1 keeps the deep candidate kernel on its generic batch loop (otherwise only for gap limits
    of 2^31 and more), SWG_SORTA_PAIRS=1 sorts A behind a mapping sweep as 12-byte (key, index) pairs (otherwise 8-byte (group, index)
    words, the keys rebuilt from the record slots), SWG_SORT_DROP10=1 caps sort A's truncation at 10 bits (otherwise up to 16 on
    sparse keys), SWG_WALK_PLAIN=1 has the fused walk build its candidate lists lane by lane (otherwise only for gap limits beyond
    2^22; below, the wavefront's windows go through one loop on packed keys)."""
    cli, ref = bins
    rng = np.random.default_rng(99)
    rec = gen.random_records(rng, 60_000, n_genomes=3, chrs_per_genome=2, span=1_000_000)
    paf = tmp_path / "in.paf"
    paf.write_text(gen.records_to_paf(rng, rec))
    for k, flags in enumerate(FLAG_SETS[:5] + [["--num-mappings", "3:2", "--scaffold-jump", "0"]]):
        o1, o2 = tmp_path / f"gpu{k}.paf", tmp_path / f"ref{k}.paf"
        r = subprocess.run([cli, str(paf), "--output-file", str(o1), "--quiet", *flags], capture_output=True, text=True,
                           env={**os.environ, **{k1: "1" for k1 in knob.split("+")}})
        assert r.returncode == 0, r.stderr
        subprocess.check_call([ref, str(paf), "--output-file", str(o2), *flags])
        assert o1.read_bytes() == o2.read_bytes(), flags


def test_warmup_reserves_and_changes_no_result(tmp_path):
    """swg_warmup (arena + staging block for a hinted size, code objects through one small built-in call): the first real
    call then needs no grow-and-rerun, and its results are those of a cold context."""
    import sweepga_amd as sw
    rng = np.random.default_rng(5)
    rec = gen.random_records(rng, 20_000, n_genomes=3, chrs_per_genome=2)
    packed = sw.pack_records(gen.records_to_meta(rec))
    cfg = sw.FilterConfig(mapping_filter_mode=sw.FilterMode.OneToOne, scaffold_max_deviation=5_000)
    cold, warm = sw.Context(0), sw.Context(0)
    try:
        want = sw.PafFilter(cfg, ctx=cold).filter_columns(packed)
        assert warm.memory_info() == (0, 0)
        warm.warmup(200_000, 64, True)                    # (a hint: small inputs have fixed costs beyond bytes per record)
        cap, _ = warm.memory_info()
        assert cap >= 200_000 * 200
        got = sw.PafFilter(cfg, ctx=warm).filter_columns(packed)
        cap2, peak = warm.memory_info()
        assert cap2 == cap and 0 < peak <= cap            # the reserved arena was enough
        assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1])
        warm.warmup(0, 0, False)                          # no hint: code objects only
    finally:
        cold.close()
        warm.close()


def test_memory_info_reserve_and_filter_paf_errors(tmp_path):
    """swg_memory_info / swg_reserve, and error propagation of swg_filter_paf (missing input, unwritable output)."""
    import sweepga_amd as sw
    ctx = sw.Context(0)
    try:
        assert ctx.memory_info() == (0, 0)
        ctx.reserve(64 << 20)
        cap, _ = ctx.memory_info()
        assert cap >= 64 << 20
        rng = np.random.default_rng(3)
        paf = tmp_path / "in.paf"
        paf.write_text(gen.records_to_paf(rng, gen.random_records(rng, 2000)))
        f = sw.PafFilter(sw.FilterConfig(), ctx=ctx)
        f.filter_paf(paf, tmp_path / "out.paf")
        cap2, peak = ctx.memory_info()
        assert cap2 >= cap and 0 < peak <= cap2          # the reserved arena was enough: no grow-and-rerun
        with pytest.raises(sw.SwgError, match="cannot open"):
            f.filter_paf(tmp_path / "missing.paf", tmp_path / "o.paf")
        with pytest.raises(sw.SwgError, match="cannot create"):
            f.filter_paf(paf, tmp_path / "no_such_dir" / "o.paf")
        f.filter_paf(paf, tmp_path / "out2.paf")           # the context is still usable after the errors
        assert (tmp_path / "out2.paf").read_bytes() == (tmp_path / "out.paf").read_bytes()
    finally:
        ctx.close()
