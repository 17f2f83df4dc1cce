"""BASELINE.json configs[2] at FULL size, every flag set: 10^7 mappings in one chromosome pair through swg_filter, status AND
chain numbers of every record against the oracle's answer, which was computed once on a CPU box
(tools/make_sbig1_golden.py -> tests/golden/sbig1_full_size.json: sha256 of the two result columns + counts; the records come
from numpy's PCG64, tests/sbig1_numpy.py, the same bytes on every machine).  Seconds on the GPU box."""
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden", "sbig1_full_size.json")


@pytest.fixture(scope="module")
def setup():
    import sweepga_amd as sw
    from sweepga_amd.filter import PackedRecords
    from tests import sbig1_numpy
    gold = json.load(open(GOLDEN))
    n = next(iter(gold["expected"].values()))["n"]
    assert gold["numpy"].split(".")[0] == np.__version__.split(".")[0], "the generator's stream is tied to numpy's major version"
    cols = sbig1_numpy.gen(n)
    table = np.arange(2, dtype=np.uint32)
    packed = PackedRecords(n=n, cols=cols, n_seq=2, seq_genome_last=table, n_genome_last=2, seq_genome_two=table.copy(),
                           n_genome_two=2)
    return sw, packed, gold, sbig1_numpy


@pytest.mark.parametrize("flags", ["sweep", "default", "full"])
def test_full_size_sbig1_equals_the_oracle(setup, flags):
    sw, packed, gold, sbig1_numpy = setup
    kw = dict(gold["flags"][flags])
    for k in ("mapping_filter_mode", "scaffold_filter_mode"):
        if k in kw:
            kw[k] = sw.FilterMode(kw[k])
    status, chain = sw.PafFilter(sw.FilterConfig(**kw)).filter_columns(packed)
    got = sbig1_numpy.fingerprint(status, chain)
    want = gold["expected"][flags]
    assert got["n"] == want["n"] == 10_000_000
    assert (got["kept"], got["scaffold"], got["rescued"], got["max_chain"]) == \
           (want["kept"], want["scaffold"], want["rescued"], want["max_chain"])
    assert got["status_sha256"] == want["status_sha256"]
    assert got["chain_sha256"] == want["chain_sha256"]
