"""The C++ host (sweepga_amd/bin/sweepga-gpu): builds, parses flags like the reference, and fails loudly
without a GPU (no CPU filter inside)."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def cli():
    from sweepga_amd import build
    build.build()
    return build.CLI


def test_help_and_flag_errors(cli, tmp_path):
    r = subprocess.run([cli, "--help"], capture_output=True, text=True)
    assert r.returncode == 0 and "usage: sweepga-gpu" in r.stdout
    r = subprocess.run([cli], capture_output=True, text=True)
    assert r.returncode == 2
    r = subprocess.run([cli, "x.paf", "--bogus"], capture_output=True, text=True)
    assert r.returncode == 2 and "unknown flag" in r.stderr
    p = tmp_path / "a.paf"
    p.write_text("a\t1\t0\t10\t+\tb\t1\t0\t10\t9\t10\t60\n")
    r = subprocess.run([cli, str(p), "--num-mappings", "0"], capture_output=True, text=True)
    assert r.returncode == 1  # the reference exits(1) on a bare 0 (main.rs:281-288)
    r = subprocess.run([cli, str(p), "--min-aln-identity", "ani50"], capture_output=True, text=True)
    assert r.returncode == 3 and "no usable GPU" in r.stderr  # the ANI pre-pass runs on the device
    r = subprocess.run([cli, str(p), "--min-aln-identity", "ninety"], capture_output=True, text=True)
    assert r.returncode == 2 and "Invalid identity value" in r.stderr


def test_no_filter_copies_input(cli, tmp_path):  # main.rs:3461-3470
    p = tmp_path / "a.paf"
    o = tmp_path / "o.paf"
    p.write_bytes(b"a\t1\t0\t10\t+\tb\t1\t0\t10\t9\t10\t60\r\nshort\n")
    r = subprocess.run([cli, str(p), "--no-filter", "--output-file", str(o)], capture_output=True, text=True)
    assert r.returncode == 0
    assert o.read_bytes() == b"a\t1\t0\t10\t+\tb\t1\t0\t10\t9\t10\t60\nshort\n"


def test_without_gpu_exits_loudly(cli, tmp_path):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    p = tmp_path / "a.paf"
    p.write_text("a\t1\t0\t10\t+\tb\t1\t0\t10\t9\t10\t60\n")
    r = subprocess.run([cli, str(p)], capture_output=True, text=True)
    assert r.returncode == 3 and "no CPU fallback" in r.stderr and r.stdout == ""
