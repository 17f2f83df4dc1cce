"""bench.py contract (ONE stdout line under 4 KB with `roofline` and `cpu_baseline`; the full report -- the three flag sets
under `pipelines`, the S-big1 leg, the PCIe leg -- in the --detail file) and, through its all-threads parity leg, a multi-million-record parity check of all three flag sets on
a scaled-down S-pan workload.  Also --gpus validation, the strong-scaling mode on one GPU, and smoke()."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_bench(*args, env=None, expect_rc=0):
    """-> the full report (the --detail file) with the parsed stdout line under "_line" and its length under "_line_bytes"."""
    import tempfile
    e = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        e.pop(k, None)
    e.update(env or {})
    with tempfile.TemporaryDirectory() as tmp:
        detail = os.path.join(tmp, "detail.json")
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args, "--detail", detail], capture_output=True,
                           text=True, cwd=ROOT, timeout=900, env=e)
        assert r.returncode == expect_rc, r.stderr[-2000:]
        if expect_rc != 0:
            return r
        lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
        assert len(lines) == 1 and lines[0].startswith("{"), r.stdout[-2000:]   # stdout is the ONE line and nothing else
        d = json.load(open(detail))
    d["_line"], d["_line_bytes"] = json.loads(lines[0]), len(lines[0])
    return d


@pytest.fixture(scope="module")
def line():
    n = 4_000_000
    return n, run_bench("--mappings", str(n), "--genomes", "21", "--steps", "2", "--warmup", "1", "--cpu-sample", "300000",
                        "--parity-mappings", str(n), "--sbig1", "300000", "--sbig1-parity-sweep", "300000",
                        "--sbig1-parity-scaffold", "60000", "--e2e", "0")


def test_stdout_line_is_small_and_complete(line):
    """The driver keeps an 8 KB tail of stdout (round 2's 20 KB line could not be parsed): the line stays under 4 KB and
    still carries the contract keys, `roofline`, `cpu_baseline` and one scalar per other leg."""
    n, d = line
    ln = d["_line"]
    assert d["_line_bytes"] < 4096, d["_line_bytes"]
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "ms_per_step_unprofiled", "higher_is_better",
              "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline", "cpu_baseline_all_cores",
              "sweep_ms_per_step", "full_ms_per_step", "c5_ms_per_step", "sbig1_sweep_ms", "sbig1_default_ms", "sbig1_full_ms", "pcie_default_ms",
              "pcie_sweep_ms", "by_query_default_ms", "shuffled_default_ms", "multichrom_default_ms", "parity", "parity_ok", "detail"):
        assert k in ln, k
    assert ln["value"] == pytest.approx(d["value"], rel=1e-6) and ln["ms_per_step"] == pytest.approx(d["ms_per_step"], abs=1e-3)
    assert ln["parity_ok"] is True and len(ln["parity"]) == 10 and ln["parity"]["span_c5"]["ok"] is True
    # the reordered inputs: every record's answer as in the pair-major run, besides the oracle on a few genome pairs
    assert ln["parity"]["by_query_default"]["same_as_grouped"] is True and ln["parity"]["shuffled_default"]["same_as_grouped"] is True
    assert ln["parity"]["multichrom_default"]["ok"] is True
    assert ln["parity"]["span_c5"]["checked"] == n
    for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "kernel_avg_ms", "pipeline_frac"):
        assert k in ln["roofline"], k
    assert ln["cpu_baseline"]["kind"] == "port" and ln["cpu_baseline"]["cores"] == 1
    assert ln["ms_per_step_unprofiled"] > 0 and ln["ms_per_step_all_events"] > 0
    # the timed region carries HIP events around the dominant kernel only: its live duration is in the roofline object, next
    # to the one measured with events around every launch (pass A, which also produced the kernel table)
    rf = d["roofline"]
    assert rf["kernel_avg_ms"] > 0 and rf["kernel_avg_ms_all_events"] > 0
    assert abs(rf["kernel_avg_ms"] - rf["kernel_avg_ms_all_events"]) / rf["kernel_avg_ms_all_events"] < 0.25
    assert rf["kernel"] in d["kernels_ms_per_step"]


def test_bench_line_contract(line):
    n, d = line
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline", "pipelines", "sbig1", "pcie_inclusive"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1 and d["higher_is_better"] is True
    assert d["scaling"] == "weak" and d["vs_baseline"] is None and d["data"] == "synthetic"
    assert "workload" in d["config"] and "model" not in d["config"]
    assert d["config"]["flags"] == "(defaults)" and "pipeline=default" in d["config"]["workload"]   # headline = the default flags
    assert abs(d["value"] - n / (d["ms_per_step"] * 1e-3)) / d["value"] < 1e-6
    assert d["value"] == d["pipelines"]["default"]["value"]
    rf = d["roofline"]
    assert rf["bound"] == "hbm" and rf["peak"] == 8000.0 and rf["unit"] == "GB/s"
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-12 and rf["achieved"] > 0
    assert rf["pipeline_achieved"] > 0 and rf["algorithmic_bytes_per_mapping"] == 47
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] == 1 and cb["value"] > 0 and cb["sample"]
    for p in ("default", "sweep"):
        assert d["pcie_inclusive"][p]["value"] > 0 and d["pcie_inclusive"][p]["h2d_ms"] > 0


@pytest.mark.parametrize("pipeline", ["sweep", "full", "default", "c5"])
def test_bench_pipelines_and_parity(line, pipeline):
    n, d = line
    p = d["pipelines"][pipeline]
    assert p["ms_per_step"] > 0 and abs(p["value"] - n / (p["ms_per_step"] * 1e-3)) / p["value"] < 1e-6
    assert p["roofline"]["algorithmic_bytes_per_mapping"] == (33 if pipeline == "sweep" else 47)
    assert p["cpu_baseline"]["cores"] == 1 and p["cpu_baseline_all_cores"]["cores"] >= 1
    assert p["parity_vs_oracle_on_sample"] is True
    pa = p["parity_all_threads"]
    assert pa["mappings_checked"] == n and pa["status_equal"] is True
    assert pa["chain_partition_equal"] is (None if pipeline == "sweep" else True)
    assert p["counts"]["in"] == n and 0 < p["counts"]["out"] < n
    if pipeline == "c5":  # BASELINE.json configs[4] as written: many:many mappings, so every record is chained (members = input)
        assert p["flags"] == "--scaffold-filter 1:1 --scaffold-dist 20000" and p["counts"]["swept"] == n
        assert p["counts"]["out"] > d["pipelines"]["full"]["counts"]["out"]


@pytest.mark.parametrize("pipeline", ["sweep", "full", "default"])
def test_bench_sbig1_leg(line, pipeline):
    _, d = line
    p = d["sbig1"]["pipelines"][pipeline]
    assert p["ms_per_step"] > 0 and p["counts"]["in"] == 300000
    assert p["parity"]["status_equal"] is True
    assert p["parity"]["chain_equal"] is (None if pipeline == "sweep" else True)


def test_gpus_flag_is_validated():
    # more GPUs than the box has: refuses instead of measuring one GPU and calling it N
    import torch
    r = run_bench("--gpus", str(torch.cuda.device_count() + 1), "--mappings", "100000", expect_rc=2)
    assert "GPU" in r.stderr
    # under a launcher with a different world size: refuses as well
    r = run_bench("--gpus", "2", "--mappings", "100000", env={"RANK": "0", "WORLD_SIZE": "1", "LOCAL_RANK": "0"}, expect_rc=2)
    assert "WORLD_SIZE" in r.stderr


def test_strong_scaling_mode_one_gpu():
    n = 2_000_000
    d = run_bench("--scaling", "strong", "--mappings", str(n), "--genomes", "21", "--steps", "2", "--warmup", "1")
    assert d["scaling"] == "strong" and d["n_gpus"] == 1 and d["_line_bytes"] < 4096 and d["_line"]["strong"]["mappings_total"] == n
    ss = d["strong_scaling"]
    assert ss["mappings_total"] == n and ss["shard_mappings_rank0"] == n and ss["load_max_over_mean"] == 1.0
    for p in ("default", "sweep", "full", "c5"):
        assert ss["pipelines"][p]["ms_per_step"] > 0 and len(ss["pipelines"][p]["per_rank_ms"]) == 1
    assert ss["pipelines"]["default"]["renumber_s"] is not None and ss["pipelines"]["sweep"]["renumber_s"] is None


def test_two_rank_launch_rehearsal():
    """The path an N-GPU driver run takes -- bench.py --gpus N spawns torch.distributed.run before touching a GPU, every
    rank joins a process group, barriers, MAX-over-ranks timing, (strong) LPT shards + the all_reduce(MIN) chain
    renumbering -- executed end to end with 2 ranks on this box's one GPU (`--rehearse`: rank r on device r mod 1, gloo
    instead of RCCL).  Strong mode: the fingerprint of (record, status, GLOBAL chain number) over both shards must equal
    the unsharded run's, for every flag set.  Not a scaling measurement."""
    n = 2_000_000
    common = ("--mappings", str(n), "--genomes", "21", "--steps", "2", "--warmup", "1")
    one = run_bench("--scaling", "strong", *common)
    two = run_bench("--scaling", "strong", "--gpus", "2", "--rehearse", *common)
    assert two["n_gpus"] == 2 and two["scaling"] == "strong" and "gloo" in two["_line"]["rehearsal"]
    ss = two["strong_scaling"]
    assert ss["mappings_total"] == n and 0 < ss["shard_mappings_rank0"] < n and 1.0 <= ss["load_max_over_mean"] < 1.05
    assert sum(ss["loads"]) == n
    for p in ("default", "sweep", "full", "c5"):
        e = ss["pipelines"][p]
        assert len(e["per_rank_ms"]) == 2 and all(x > 0 for x in e["per_rank_ms"])
        assert e["ms_per_step"] >= max(e["per_rank_ms"]) * 0.999   # the reported time is the MAX over ranks
        assert e["result_checksum"] == one["strong_scaling"]["pipelines"][p]["result_checksum"], p
    # weak mode: every rank filters its own shard of n mappings; value = all ranks' mappings / max-over-ranks time
    w = run_bench("--gpus", "2", "--rehearse", "--only", *common)
    assert w["n_gpus"] == 2 and w["scaling"] == "weak" and w["_line"]["n_gpus"] == 2
    assert abs(w["value"] - 2 * n / (w["ms_per_step"] * 1e-3)) / w["value"] < 1e-6
    # ... and the host-buffer leg runs on every rank at once, reported as the slowest rank's time over all ranks' records
    pc = w["pcie_inclusive"]["default"]
    assert pc["ranks"] == 2 and pc["ms"] > 0 and abs(pc["value"] - 2 * n / (pc["ms"] * 1e-3)) / pc["value"] < 1e-6
    assert w["_line"]["pcie_default_ms"] == pytest.approx(pc["ms"], abs=0.01)


def test_smoke_entry():
    r = subprocess.run([sys.executable, "-c", "import __graft_entry__ as g; g.smoke(); print('smoke ok')"], capture_output=True,
                       text=True, cwd=ROOT, timeout=600)
    assert r.returncode == 0 and "smoke ok" in r.stderr + r.stdout, r.stderr[-2000:]
