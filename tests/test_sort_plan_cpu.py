"""Digit plans of the radix sorts for every key width (tests/native/plan_check.cpp): host code of the library, no GPU."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def plan_check(tmp_path_factory):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    from sweepga_amd import build
    build.build()
    out = str(tmp_path_factory.mktemp("plan") / "plan_check")
    lib_dir = os.path.join(ROOT, "sweepga_amd")
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-O1", "-std=c++17", os.path.join(ROOT, "tests", "native", "plan_check.cpp"), "-o", out,
                           "-L", lib_dir, "-lsweepga_gpu", "-Wl,-rpath," + lib_dir])
    return out


@pytest.mark.parametrize("env", [{}, {"SWG_SORT_BITS8": "1"}])
def test_plans_tile_the_key_bits(plan_check, env):
    r = subprocess.run([plan_check], capture_output=True, text=True, env={**os.environ, **env}, timeout=60)
    assert r.returncode == 0, r.stdout + r.stderr
    last = r.stdout.strip().splitlines()[-1]
    assert last.startswith("0 bad")
    if env:
        assert last.endswith(" 0 widths with 9-bit digits")
    else:
        assert not last.endswith(" 0 widths with 9-bit digits")
