"""A pytest slice of tests/fuzz/fuzz_large.py: shapes the bench workload lacks (one giant deep pair, many chromosomes per
genome, thousands of tiny pairs, coordinate ties on a grid, minus strand only, names without '#') x four flag sets, whole
filter on the GPU vs the oracle, exact status and chain numbers."""
import importlib.util
import os
import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RECORDS = 250_000


def _fuzz_large():
    spec = importlib.util.spec_from_file_location("fuzz_large", os.path.join(ROOT, "tests", "fuzz", "fuzz_large.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


@pytest.fixture(scope="module")
def results():
    import sweepga_amd as sw
    from tests import gen, orc
    fl = _fuzz_large()
    out, threads = {}, []
    for si, shape in enumerate(fl.SHAPES):
        rng = np.random.default_rng(1000 + si)
        n = int(RECORDS * shape["scale"])
        rec = gen.random_records(rng, n, n_genomes=shape["n_genomes"], chrs_per_genome=shape["chrs_per_genome"], span=shape["span"],
                                 max_len=shape["max_len"], syntenic_frac=shape["syntenic_frac"], pansn=shape.get("pansn", True),
                                 minus_frac=shape.get("minus_frac", 0.2))
        if "grid" in shape:
            for a in (rec.qs, rec.qe, rec.ts, rec.te):
                a[:] = a // shape["grid"] * shape["grid"]
        packed = sw.pack_records(gen.records_to_meta(rec))
        for cname, kw in fl.CONFIGS:
            kwg = {k: (getattr(sw.FilterMode, v) if isinstance(v, str) else v) for k, v in kw.items()}
            st, ch = sw.PafFilter(sw.FilterConfig(**kwg)).filter_columns(packed)
            okw = {k: (int(getattr(sw.FilterMode, v)) if isinstance(v, str) else v) for k, v in kw.items()}
            slot = dict(st=st.copy(), ch=ch.copy())
            out[(shape["name"], cname)] = slot

            def work(slot=slot, okw=okw, rec=rec):
                slot["ost"], slot["och"] = orc.apply_filters(orc.Config(**okw), rec)
            th = threading.Thread(target=work)
            th.start()
            threads.append(th)
    for th in threads:
        th.join()
    return out


def _cases():
    fl = _fuzz_large()
    return [(s["name"], c) for s in fl.SHAPES for c, _ in fl.CONFIGS]


@pytest.mark.parametrize("shape,config", _cases())
def test_large_shape_matches_oracle(results, shape, config):
    r = results[(shape, config)]
    assert np.array_equal(r["st"], r["ost"]), int((r["st"] != r["ost"]).sum())
    assert np.array_equal(r["ch"], r["och"]), int((r["ch"] != r["och"]).sum())


def _many_genome_records(rng, n_seq, n_pairs, per_pair):
    from tests import orc
    """Names without '#': every sequence is its own genome under both prefix rules (paf_filter.rs:1022-1030,
    plane_sweep_scaffold.rs:13-22).  n_pairs (query, target) contig pairs with `per_pair` roughly syntenic records each."""
    names = [f"contig{i:06d}" for i in range(n_seq)]
    qn, tn, qs, ql, ts, strand = [], [], [], [], [], []
    for _ in range(n_pairs):
        q, t = rng.choice(n_seq, 2, replace=False)
        k = int(rng.integers(1, per_pair * 2))
        base_q, base_t = int(rng.integers(0, 500_000)), int(rng.integers(0, 500_000))
        step = rng.integers(500, 30_000, k).cumsum()
        minus = rng.random() < 0.3
        for j in range(k):
            qn.append(names[q])
            tn.append(names[t])
            qs.append(base_q + int(step[j]))
            ql.append(int(rng.integers(200, 6_000)))
            ts.append(max(0, base_t + (int(step[-1] - step[j]) if minus else int(step[j])) + int(rng.normal(0, 800))))
            strand.append(ord("-") if minus != (rng.random() < 0.05) else ord("+"))
    n = len(qn)
    perm = rng.permutation(n)
    u = lambda a: np.ascontiguousarray(np.asarray(a, dtype=np.uint64)[perm])
    qs, ql, ts = u(qs), u(ql), u(ts)
    ident = np.round(0.75 + 0.25 * rng.random(n), 4)
    matches = np.floor(ident * ql).astype(np.uint64)
    return orc.Records([qn[i] for i in perm], [tn[i] for i in perm], qs, qs + ql, ts, ts + ql, ql,
                       np.ascontiguousarray(matches / np.maximum(ql, 1)), matches, np.asarray(strand, dtype=np.uint8)[perm],
                       np.arange(n, dtype=np.uint64))


@pytest.mark.parametrize("cfg_name", ["default", "full", "loose"])
def test_more_than_2_14_genomes(cfg_name):
    """~20,000 contigs without '#' = 20,000 genomes: the genome-pair tables switch from dense G x G to open addressing over
    the pairs that occur (swg_scaffold.hip PairTable).  Status and chain numbers against the oracle."""
    import sweepga_amd as sw
    from tests import gen, orc
    from tests.test_gpu_scaffold import _cfg_pair
    rng = np.random.default_rng(2014)
    rec = _many_genome_records(rng, 60_000, 12_000, 4)
    kw = {"default": dict(),
          "full": dict(mapping_filter_mode=sw.FilterMode.OneToOne, scaffold_filter_mode=sw.FilterMode.OneToOne, scaffold_gap=50_000,
                       min_scaffold_length=10_000, scaffold_max_deviation=20_000),
          "loose": dict(scaffold_gap=80_000, min_scaffold_length=2_000, scaffold_filter_mode=sw.FilterMode.ManyToMany,
                        scaffold_max_deviation=30_000)}[cfg_name]
    cfg, ocfg = _cfg_pair(sw, **kw)
    packed = sw.pack_records(gen.records_to_meta(rec))
    assert packed.n_genome_last > (1 << 14) and packed.n_genome_two > (1 << 14)
    f = sw.PafFilter(cfg)
    status, chain = f.filter_columns(packed)
    ost, och = orc.apply_filters(ocfg, rec)
    assert np.array_equal(status, ost), int((status != ost).sum())
    assert np.array_equal(chain, och), int((chain != och).sum())
    assert int((och != 0).sum()) > 1000          # the case does chain
