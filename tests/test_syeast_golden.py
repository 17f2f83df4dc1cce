"""BASELINE.json configs[0] / configs[1] on the committed S-yeast fixture (tests/golden/syeast.paf.gz, made by
tools/make_syeast_fixture.py from the reference's data/scerevisiae8.fa.gz.fai names and lengths).

  configs[0]  "--scaffold-jump 0, CPU reference filter (plumbing, no GPU)" -> the oracle reproduces the
              committed output hashes (guards the oracle against regressions);
  configs[1]  "default pipeline, 1 MI355X, bit-exact vs CPU" -> the GPU command line writes byte-identical files.
"""
import gzip
import hashlib
import json
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")


@pytest.fixture(scope="module")
def fixture(tmp_path_factory):
    exp = json.load(open(os.path.join(GOLD, "syeast_expected.json")))
    text = gzip.open(os.path.join(GOLD, "syeast.paf.gz"), "rb").read()
    assert hashlib.sha256(text).hexdigest() == exp["input_sha256"]
    p = tmp_path_factory.mktemp("syeast") / "syeast.paf"
    p.write_bytes(text)
    return str(p), exp


def test_oracle_reproduces_golden(fixture, tmp_path):
    paf, exp = fixture
    ref = os.path.join(ROOT, "oracle", "sweepga-ref")
    for name, e in exp["flag_sets"].items():
        out = tmp_path / (name + ".paf")
        subprocess.check_call([ref, paf, "--output-file", str(out), *e["flags"]])
        data = out.read_bytes()
        assert data.count(b"\n") == e["kept"], name
        assert hashlib.sha256(data).hexdigest() == e["sha256"], name


@pytest.mark.gpu
def test_gpu_cli_matches_golden(fixture, tmp_path):
    from sweepga_amd import build
    paf, exp = fixture
    for name, e in exp["flag_sets"].items():
        out = tmp_path / (name + ".paf")
        r = subprocess.run([build.CLI, paf, "--output-file", str(out), "--quiet", *e["flags"]], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        data = out.read_bytes()
        assert data.count(b"\n") == e["kept"], name
        assert hashlib.sha256(data).hexdigest() == e["sha256"], name
