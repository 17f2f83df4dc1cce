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


def test_no_filter_copies_input_to_stdout(cli, tmp_path):  # main.rs:3461-3473: stdout, --output-file is not consulted
    import os
    ref = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "sweepga-ref")
    p = tmp_path / "a.paf"
    o = tmp_path / "o.paf"
    p.write_bytes(b"a\t1\t0\t10\t+\tb\t1\t0\t10\t9\t10\t60\r\nshort\n")
    for exe in (cli, ref):
        r = subprocess.run([exe, str(p), "--no-filter", "--output-file", str(o)], capture_output=True)
        assert r.returncode == 0
        assert r.stdout == b"a\t1\t0\t10\t+\tb\t1\t0\t10\t9\t10\t60\nshort\n"
        assert not o.exists()


def test_numeric_flags_are_checked(cli, tmp_path):
    """clap rejects `--overlap abc`; so does this command line (also --scaffold-overlap, --device, --threads)."""
    p = tmp_path / "a.paf"
    p.write_text("a\t1\t0\t10\t+\tb\t1\t0\t10\t9\t10\t60\n")
    for flag, v in (("--overlap", "abc"), ("--overlap", ""), ("--scaffold-overlap", "0.5x"), ("--device", "one"), ("--threads", "-3")):
        r = subprocess.run([cli, str(p), "--no-filter", flag, v], capture_output=True, text=True)
        assert r.returncode == 2 and "invalid value for " + flag in r.stderr, (flag, v, r.returncode, r.stderr)


def test_without_gpu_exits_loudly(cli, tmp_path):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    p = tmp_path / "a.paf"
    p.write_text("a\t1\t0\t10\t+\tb\t1\t0\t10\t9\t10\t60\n")
    r = subprocess.run([cli, str(p)], capture_output=True, text=True)
    assert r.returncode == 3 and "no CPU fallback" in r.stderr and r.stdout == ""


def test_sparsify_flag_is_validated_only(cli, tmp_path):
    """--sparsify on the PAF path (src/knn_graph.rs:59-160, src/main.rs:3494-3509): `none` / `all` / fractions / `random:`
    have no effect on the filter, pre-alignment strategies are refused after the --no-filter shortcut, garbage is a usage
    error, and `tree:` / `knn:` tree-filter the PAF before the filter (src/main.rs:3640-3688)."""
    import os
    ref = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "sweepga-ref")
    p = tmp_path / "a.paf"
    p.write_text("a\t1\t0\t10\t+\tb\t1\t0\t10\t9\t10\t60\n")
    for v, want in (("none", 0), ("all", 0), ("0.5", 0), ("random:0.3", 0), ("tree:2:1:0.1", 0), ("knn:3", 0), ("auto", 0),
                    ("1.5", 2), ("tree:0", 2), ("random:2", 2), ("bogus", 2)):
        for exe in (cli, ref):
            r = subprocess.run([exe, str(p), "--no-filter", "--sparsify", v], capture_output=True, text=True)
            assert r.returncode == want, (exe, v, r.returncode, r.stderr)
    for v in ("auto", "giant:0.9", "connectivity:0.5", "wfmash:auto", "wfmash:0.2"):
        for exe in (cli, ref):
            r = subprocess.run([exe, str(p), "--sparsify", v], capture_output=True, text=True)
            assert r.returncode == 1 and "not valid for post-alignment" in r.stderr, (exe, v, r.returncode, r.stderr)
    # tree / knn: the input is tree-filtered first (src/main.rs:3640-3688).  The oracle's command line runs on the CPU: two
    # genome pairs, tree:1 keeps the better neighbour of every genome; the GPU command line needs its device for the filter.
    t = tmp_path / "t.paf"
    mk = lambda q, tt, m: f"{q}\t9000\t0\t5000\t+\t{tt}\t9000\t0\t5000\t{m}\t5000\t60\n"
    t.write_text(mk("A#1#c", "B#1#c", 4900) + mk("A#1#c", "C#1#c", 4000) + mk("B#1#c", "C#1#c", 4500))
    r = subprocess.run([ref, str(t), "--sparsify", "tree:1", "--scaffold-jump", "0"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    # A's best neighbour is B (0.98), B's is A, C's is B (0.90 > 0.80): A-C is dropped
    assert [ln.split("\t")[0] + ">" + ln.split("\t")[5] for ln in r.stdout.splitlines()] == ["A#1#c>B#1#c", "B#1#c>C#1#c"]
    import torch
    if not torch.cuda.is_available():
        r = subprocess.run([cli, str(t), "--sparsify", "tree:1", "--scaffold-jump", "0"], capture_output=True, text=True)
        assert r.returncode == 3 and "no usable GPU" in r.stderr