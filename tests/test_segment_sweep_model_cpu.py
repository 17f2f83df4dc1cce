"""The restatement of the axis sweep that DESIGN.md section 13 plans to run segment-resident (tools/model_segment_sweep.py)
against the oracle (src/plane_sweep_exact.rs:197-433) on random segments: ties on scores and positions, zero-length and nested
intervals, k = 1, 2, 3 and unlimited, thresholds below and at 1."""
import os
import sys

import numpy as np
import pytest

from tests import orc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
from model_segment_sweep import segment_sweep, segment_sweep_k1_resident  # noqa: E402


def random_segment(rng, n, span, grid, malformed=False):
    qs = rng.integers(0, span, n) // grid * grid
    ql = rng.integers(0, span // 4, n) // grid * grid
    ql[rng.random(n) < 0.05] = 0                                   # zero-length
    ts = rng.integers(0, span, n) // grid * grid
    tl = np.where(rng.random(n) < 0.7, ql, rng.integers(0, span // 4, n) // grid * grid)
    ident = np.round(rng.uniform(0.7, 1.0, n), 2 if rng.random() < 0.5 else 6)   # (two decimals: score ties)
    ident[rng.random(n) < 0.03] = 0.0                              # -inf scores
    qe, te = qs + ql, ts + tl
    if malformed:                                                  # end before start: inserted at its start, never erased
        m = rng.random(n) < 0.04
        qe = np.where(m & (qs > 0), qs // 2, qe)
        te = np.where(m & (ts > 0), ts // 2, te)
    return qs.astype(np.uint64), qe.astype(np.uint64), ts.astype(np.uint64), te.astype(np.uint64), ident


@pytest.mark.parametrize("seed", range(80))
def test_model_equals_the_oracle(seed):
    rng = np.random.default_rng(5000 + seed)
    n = int(rng.choice([2, 3, 8, 40, 200]))
    qs, qe, ts, te, ident = random_segment(rng, n, int(rng.choice([50, 1_000, 100_000])), int(rng.choice([1, 1, 10])), malformed=seed % 4 == 3)
    scoring = int(rng.choice([orc.LOG_LENGTH_IDENTITY, orc.IDENTITY, orc.LENGTH, orc.LENGTH_IDENTITY]))
    score = [orc.score(int(qs[i]), int(qe[i]), float(ident[i]), scoring) for i in range(n)]
    for k in (1, 2, 3, None):
        for thr in (0.3, 0.95, 1.0):
            for axis, (a, b) in ((0, (qs, qe)), (1, (ts, te))):
                want = orc.plane_sweep(axis, qs, qe, ts, te, ident, k_q=k or 2 ** 63, k_t=k or 2 ** 63, thr=thr, scoring=scoring)
                got = segment_sweep(a, b, score, k, thr)
                assert got == sorted(want), (seed, n, k, thr, axis, got[:10], sorted(want)[:10])


@pytest.mark.parametrize("seed", range(60))
def test_resident_k1_model_equals_the_oracle(seed):
    """segment_sweep_k1_resident is the executable model of seg_sweep_body (csrc/swg_segsort.hip): batches, carried intervals, one
    evaluation per interval over the batch's range of positions.  Tiny batches so that every interval is carried somewhere."""
    rng = np.random.default_rng(9000 + seed)
    n = int(rng.choice([2, 3, 8, 40, 200, 500]))
    qs, qe, ts, te, ident = random_segment(rng, n, int(rng.choice([50, 1_000, 100_000])), int(rng.choice([1, 1, 10])))
    scoring = int(rng.choice([orc.LOG_LENGTH_IDENTITY, orc.IDENTITY, orc.LENGTH, orc.LENGTH_IDENTITY]))
    score = [orc.score(int(qs[i]), int(qe[i]), float(ident[i]), scoring) for i in range(n)]
    checked = 0
    for thr in (0.0, 0.3, 0.95, 1.0):
        for axis, (a, b) in ((0, (qs, qe)), (1, (ts, te))):
            want = sorted(orc.plane_sweep(axis, qs, qe, ts, te, ident, k_q=1, k_t=1, thr=thr, scoring=scoring))
            for cap, cmax in ((8, 10 ** 9), (3, 10 ** 9), (64, 10 ** 9), (10 ** 9, 0)):
                got = segment_sweep_k1_resident(a, b, score, thr, cap, cmax)
                if got is None:          # (a run of equal starts longer than a batch: the kernel hands the axis back)
                    continue
                checked += 1
                assert got == want, (seed, n, thr, axis, cap, got[:10], want[:10])
    assert checked
