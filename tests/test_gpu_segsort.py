"""The segmented sort of the sweep's begins (csrc/swg_segsort.hip) is only chosen for grouped inputs of 2^20 records and
more; SWG_SEGSORT=1 forces it at every size.  The sweep / scaffold / large-shape parity suites are run again under that
switch in a fresh process (the library reads the switch once): every path of the sort -- runs, buckets of many small
segments, single-segment buckets, giants, dead records -- then sits under oracle comparisons.  SWG_SEGSORT=0 (never) is
checked the same way on the sweep suite."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("value,files", [("1", ["tests/test_gpu_sweep.py", "tests/test_gpu_scaffold.py", "tests/test_gpu_large_shapes.py",
                                                "tests/test_gpu_scaffold_kat.py"]),
                                         ("0", ["tests/test_gpu_sweep.py"])])
def test_parity_suites_under_segsort_switch(value, files):
    r = subprocess.run([sys.executable, "-m", "pytest", *files, "-m", "gpu", "-x", "-q"], cwd=ROOT, capture_output=True, text=True,
                       env={**os.environ, "SWG_SEGSORT": value}, timeout=1500)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-1000:]
