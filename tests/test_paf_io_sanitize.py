"""ASan + UBSan over the host-side code: PAF ingest/egress, tree sparsification, alnstats, .1aln derivation (sanitizers run
on the CPU build only)."""
import gzip
import os
import subprocess

import numpy as np
import pytest

from tests import gen
from tests.test_paf_io_cpu import EDGE_TEXT, bgzf_bytes

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def harness(tmp_path_factory):
    out = tmp_path_factory.mktemp("san") / "paf_io_sanitize"
    cmd = ["g++", "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-fno-omit-frame-pointer",
           os.path.join(ROOT, "tests", "native", "paf_io_sanitize.cpp"),
           os.path.join(ROOT, "sweepga_amd", "csrc", "host", "paf_io.cpp"),
           os.path.join(ROOT, "sweepga_amd", "csrc", "host", "tree_filter.cpp"),
           os.path.join(ROOT, "sweepga_amd", "csrc", "host", "alnstats.cpp"), "-o", str(out), "-lz", "-lpthread"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        pytest.skip("sanitizer build unavailable: " + r.stderr[-300:])
    return str(out)


def test_paf_io_under_asan_ubsan(harness, tmp_path):
    rng = np.random.default_rng(77)
    big = gen.records_to_paf(rng, gen.random_records(rng, 8000)) * 2
    files = {
        "edge.paf": EDGE_TEXT.encode(),
        "empty.paf": b"",
        "newlines.paf": b"\n\n\r\n\n",
        "no_final_newline.paf": b"a\t1\t2\t3\t+\tb\t4\t5\t6\t7\t8\t9",
        "tabs.paf": b"\t\t\t\t\t\t\t\t\t\t\t\t\t\n" + b"\t" * 10 + b"\n" + b"x" * 70000 + b"\n",
        "tags.paf": b"a\t1\t2\t3\t+\tb\t4\t5\t6\t7\t8\t9\tdv:f:\tcg:Z:\tdv:f:1e400\tcg:Z:99999999999999999999=\tcg:Z:5=\t\n",
        "hash.paf": b"#\t#\t0\t1\t+\t##\t#\t0\t1\t1\t1\t0\n#a#b#c#d\t1\t0\t1\t-\t#\t1\t0\t1\t1\t1\t0\n",
        "big.paf": big.encode(),
        "big.paf.gz": gzip.compress(big.encode()),
        "big2.paf.bgz": bgzf_bytes(big.encode(), block=30000),
        "trunc.paf.gz": gzip.compress(big.encode())[:5000],
        "garbage.paf.gz": b"\x1f\x8b" + bytes(rng.integers(0, 256, 3000, dtype=np.uint8)),
        # coordinates beyond 2^32: the 64-bit pass and the per-sequence rebasing of the ingest
        "wide.paf": gen.records_to_paf(rng, gen.shifted(gen.random_records(rng, 3000), rng)[0]).encode(),
        "too_wide.paf": b"q\t9\t0\t5\t+\tt\t9\t2\t6\t3\t4\t0\nq\t9\t1\t4294967296\t+\tt\t9\t2\t6\t3\t4\t0\n",
    }
    paths = []
    for name, blob in files.items():
        (tmp_path / name).write_bytes(blob)
        paths.append(str(tmp_path / name))
    env = {**os.environ, "ASAN_OPTIONS": "detect_leaks=1:abort_on_error=0", "UBSAN_OPTIONS": "print_stacktrace=1"}
    r = subprocess.run([harness, *paths, str(tmp_path / "missing.paf")], capture_output=True, text=True, env=env)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    assert "ok " in r.stdout and "ERROR: AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr
    # truncated gzip, garbage gzip, a mapped stretch beyond 2^32, missing file; three thread counts each
    assert r.stdout.count("open failed") == 4 * 3
