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
