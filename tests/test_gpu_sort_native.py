"""The radix sorts of csrc/swg_sort.hip on their own (tests/native/sort_bench.cpp): random keys of a given width, values the
identity; every result is checked on the host element by element -- ordered by (key, value), i.e. sorted AND stable, keys
recomputed from the values, the values a permutation.  Tile edges (8192-element tiles), one to eight passes, the packed
8-byte passes (8- and 9-bit digits) and the 12-byte ones, 64-bit look-back words, the three-kernel fallback."""
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def sort_bench():
    from sweepga_amd import build
    return build.build_sort_bench()


CASES = [  # n, key bits, packed, extra environment
    (5_000, 20, 1, {}), (8_192, 24, 1, {}), (8_193, 33, 1, {}), (16_385, 17, 1, {}), (1_000_003, 48, 1, {}), (3_000_000, 42, 1, {}),
    (8_193, 33, 2, {}), (3_000_000, 42, 2, {}),
    # 9-bit digits where they save a pass: 44 bits = 8 + 9+9+9+9, 41 = 8 + 9+8+8+8, 26 = 8 + 9+9, 25 = 8 + 9+8
    (1_000_003, 44, 1, {}), (3_000_000, 41, 1, {}), (300_000, 26, 1, {}), (8_193, 25, 2, {}), (8_192, 26, 1, {}),
    (3_000_000, 42, 1, {"SWG_SORT_BITS8": "1"}),
    (5_000, 9, 0, {}), (8_192, 17, 0, {}), (8_193, 64, 0, {}), (1_000_003, 56, 0, {}), (3_000_000, 43, 0, {}),
    (3_000_000, 40, 0, {"SWG_SORT_WIDE": "1"}), (300_000, 40, 0, {"SWG_SORT_FALLBACK": "1"}), (300_000, 40, 1, {"SWG_SORT_PAIRS": "1"}),
]


@pytest.mark.parametrize("n,bits,packed,env", CASES)
def test_sort_is_sorted_stable_and_a_permutation(sort_bench, n, bits, packed, env):
    e = dict(os.environ)
    e.update(env)
    r = subprocess.run([sort_bench, str(n), str(bits), "2", str(packed)], capture_output=True, text=True, env=e, timeout=300)
    if packed and env.get("SWG_SORT_PAIRS"):  # the knob makes the packed entry point decline: the caller's cue for the 12-byte passes
        assert r.returncode == 2 and "sort returned" in r.stderr
        return
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout.count("sorted, stable, a permutation") == 2 and "WRONG" not in r.stdout
