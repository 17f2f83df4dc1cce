"""bench.py contract (one JSON line with `roofline` and `cpu_baseline`) and, through its all-threads parity leg, a
multi-million-record parity check of all three flag sets on a scaled-down S-pan workload.  Also smoke()."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_bench(*args):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], capture_output=True, text=True, cwd=ROOT,
                       timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0])


@pytest.mark.parametrize("pipeline", ["sweep", "full", "default"])
def test_bench_line_and_parity(pipeline):
    n = 4_000_000
    d = run_bench("--mappings", str(n), "--genomes", "21", "--steps", "2", "--warmup", "1", "--others", "0", "--pipeline", pipeline,
                  "--cpu-sample", "300000", "--parity-mappings", str(n))
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1 and d["higher_is_better"] is True
    assert d["scaling"] == "weak" and d["vs_baseline"] is None and d["data"] == "synthetic"
    assert "workload" in d["config"] and "model" not in d["config"]
    assert abs(d["value"] - n / (d["ms_per_step"] * 1e-3)) / d["value"] < 1e-6
    rf = d["roofline"]
    assert rf["bound"] == "hbm" and rf["peak"] == 8000.0 and rf["unit"] == "GB/s"
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-12 and rf["achieved"] > 0
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] == 1 and cb["value"] > 0 and cb["sample"]
    assert d["parity_vs_oracle_on_sample"] is True
    pa = d["parity_all_threads"]
    assert pa["mappings_checked"] == n and pa["status_equal"] is True
    assert pa["chain_partition_equal"] is (None if pipeline == "sweep" else True)
    assert d["counts"]["in"] == n and 0 < d["counts"]["out"] < n


def test_smoke_entry():
    r = subprocess.run([sys.executable, "-c", "import __graft_entry__ as g; g.smoke(); print('smoke ok')"], capture_output=True,
                       text=True, cwd=ROOT, timeout=600)
    assert r.returncode == 0 and "smoke ok" in r.stdout, r.stderr[-2000:]
