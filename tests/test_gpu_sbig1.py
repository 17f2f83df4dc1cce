"""BASELINE.json configs[2] ("S-big1"): ONE (query, target) chromosome pair, 248,956,422 bp, 10^7 mappings
(seed 1234, depth ~165): a single query-axis segment and a single target-axis segment.

Full size through swg_filter_device, record for record against the oracle (src/plane_sweep_exact.rs:268-433 via
src/paf_filter.rs:972-1123), for `--num-mappings 1:1 --scaffold-jump 0`.  For the scaffold flag sets the literal oracle
needs hours at 10^7 -- not for the chaining scan (O(n x window), minutes) but for the inversion capture, which loops over
kept '+' chains x '-' mappings of the pair (src/paf_filter.rs:535-597: 1.4 * 10^6 x 10^6 here).  The full-size check of
those flag sets therefore lives outside the pytest budget (tools/sbig1_full_parity.py, ~15 min, the oracle's indexed form
of that one step; results in profiles/r03_sbig1_full_size_parity_{default,full}.json); here they are checked (status AND
chain numbers) with the literal oracle on
  * 10^6 mappings on the full-length chromosomes (depth ~16), and
  * 2 x 10^5 mappings with the chromosome length scaled by n / 10^7, i.e. at S-big1's own depth (long chaining units,
    the block-speculative path of swg_scaffold.hip),
for the default flags and for `--num-mappings 1:1 --scaffold-filter 1:1 --scaffold-dist 20000`.
All oracle runs go on their own host threads at once (one group = one oracle thread)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CASES = {  # name: (pipeline, n, chromosome length)
    "sweep_full_size": ("sweep", 10_000_000, 248_956_422),
    "default_1e6_full_length": ("default", 1_000_000, 248_956_422),
    "default_2e5_same_depth": ("default", 200_000, 4_979_128),
    "full_1e6_full_length": ("full", 1_000_000, 248_956_422),
    "full_2e5_same_depth": ("full", 200_000, 4_979_128),
    "sweep_1e6_same_depth": ("sweep", 1_000_000, 24_895_642),
    # the 2 <= k < inf tile kernel on deep data: --num-mappings 3:2 --scaffold-jump 0
    "k3_2_3e5_same_depth": ("k32", 300_000, 7_468_692),
}


@pytest.fixture(scope="module")
def results():
    r = subprocess.run([sys.executable, "-m", "tests.sbig1_check", json.dumps(CASES)], capture_output=True, text=True, cwd=ROOT,
                       timeout=1500)
    assert r.returncode == 0, r.stderr[-3000:]
    return json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])


@pytest.mark.parametrize("name", list(CASES))
def test_sbig1_matches_oracle(results, name):
    r = results[name]
    assert r["n"] == CASES[name][1]
    assert r["n_out_device"] == r["n_out_oracle"]
    assert r["status_mismatches"] == 0
    assert r["chain_mismatches"] == 0
