"""Both orders of the same records -- grouped by sequence pair and shuffled -- must give the oracle's answer, status and chain
numbers.  Since round 6 a shuffled input of this size is grouped on the device and runs pair-resident (pair_group_records,
swg_pair.hip); with SWG_PAIR_GROUP=0 it takes the global-sort stage, where the input-order probe (input_order_probe_kernel,
swg_filter.hip) decides on the device whether prepare writes the 32-byte record slots for the scaffold stage's first gather."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _permute(rec, perm):
    import copy
    out = copy.copy(rec)
    n = len(perm)
    for k, v in vars(rec).items():
        if isinstance(v, np.ndarray) and v.shape[:1] == (n,):
            setattr(out, k, np.ascontiguousarray(v[perm]))
        elif isinstance(v, list) and len(v) == n:
            setattr(out, k, [v[i] for i in perm])
    out.rank = np.arange(n, dtype=rec.rank.dtype)   # the rank of a record is its place in the input
    return out


@pytest.mark.parametrize("order", ["grouped", "shuffled"])
def test_default_flags_on_grouped_and_shuffled_input(order):
    import sweepga_amd as sw
    from tests import gen, orc
    rng = np.random.default_rng(4242)
    n = 160_000   # (the probe runs from 65,536 records)
    rec = gen.random_records(rng, n, n_genomes=6, chrs_per_genome=2, span=3_000_000, syntenic_frac=0.8)
    names = sorted(set(zip(rec.qname, rec.tname)))
    pair_of = {p: k for k, p in enumerate(names)}
    key = np.array([pair_of[p] for p in zip(rec.qname, rec.tname)], dtype=np.int64)
    perm = np.argsort(key, kind="stable") if order == "grouped" else rng.permutation(n)
    rec = _permute(rec, perm)
    packed = sw.pack_records(gen.records_to_meta(rec))
    for cfg_kw in ({}, {"scaffold_filter_mode": "OneToOne", "scaffold_max_deviation": 20_000}):
        kw = {k: (getattr(sw.FilterMode, v) if isinstance(v, str) else v) for k, v in cfg_kw.items()}
        st, ch = sw.PafFilter(sw.FilterConfig(**kw)).filter_columns(packed)
        okw = {k: (int(getattr(sw.FilterMode, v)) if isinstance(v, str) else v) for k, v in cfg_kw.items()}
        ost, och = orc.apply_filters(orc.Config(**okw), rec)
        assert np.array_equal(st, ost), (order, cfg_kw, int((st != ost).sum()))
        assert np.array_equal(ch, och), (order, cfg_kw, int((ch != och).sum()))


def test_shuffled_input_on_the_global_sort_stage():
    """The same shuffled records with the device-side grouping switched off: the global-sort stage and its input-order probe."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = "import sys; sys.path.insert(0, %r); from tests.test_gpu_input_order import test_default_flags_on_grouped_and_shuffled_input as t; t('shuffled'); print('ok')" % root
    out = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, SWG_PAIR_GROUP="0"), capture_output=True, text=True, cwd=root)
    assert out.returncode == 0 and "ok" in out.stdout, out.stderr[-2000:]
