import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
# The host paths do not send the value columns (identity / matches / block_len) a flag set does not read; under this knob the
# library fills what it did not send with 0xff bytes, so that every GPU test doubles as a check that nobody reads them anyway.
os.environ.setdefault("SWG_POISON", "1")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: long-running CPU test")


@pytest.fixture(scope="session", autouse=True)
def _build_oracle():
    """The oracle is the checker: build it once per session if its .so is missing/stale."""
    odir = os.path.join(ROOT, "oracle")
    so = os.path.join(odir, "liboracle.so")
    srcs = [os.path.join(odir, f) for f in os.listdir(odir) if f.endswith((".cpp", ".h"))]
    if not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        subprocess.check_call(["make", "-C", odir, "-s"])
    # the oracle's command line and the two host binaries are git-ignored build products: make sure they exist even if
    # a snapshot of the tree came without them (the HIP library itself is never rebuilt here)
    if not os.path.exists(os.path.join(odir, "sweepga-ref")) or not os.path.exists(os.path.join(odir, "alnstats-ref")):
        subprocess.check_call(["make", "-C", odir, "-s"])
    lib = os.path.join(ROOT, "sweepga_amd", "libsweepga_gpu.so")
    if os.path.exists(lib):
        from sweepga_amd import build
        if not os.path.exists(build.CLI):
            build.build_cli()
        if not os.path.exists(build.SYNTH):
            build.build_synth()
        if not os.path.exists(build.STATS):
            build.build_alnstats()
    yield
