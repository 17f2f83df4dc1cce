"""bench.py's stdout line stays under 4 KB (the driver keeps an 8 KB tail; round 2's 20 KB line was cut and unparsable).
CPU check: summary_line() over a committed full report of a 10^8-mapping run."""
import importlib.util
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_summary_line_of_a_full_report_is_small():
    b = _bench()
    full = json.load(open(os.path.join(ROOT, "profiles", "r02_v3_bench_100m.json")))
    assert len(json.dumps(full)) > 8192   # the report that broke the driver's parse
    full["ms_per_step_unprofiled"] = full["ms_per_step"]
    line = b.summary_line(full, os.path.join(ROOT, "gpurun_out", "bench_detail.json"))
    text = json.dumps(line)
    assert len(text) < b.MAX_LINE_BYTES == 4096
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline", "cpu_baseline_all_cores", "parity_ok"):
        assert k in line, k
    assert line["value"] == round(full["value"], 4) and line["roofline"]["kernel"] == full["roofline"]["kernel"]
    assert line["roofline"]["frac"] == round(full["roofline"]["frac"], 5)
    assert set(line["config"]) <= {"workload", "flags", "mappings_per_gpu", "groups_per_gpu"} and "model" not in line["config"]
    assert line["sbig1_default_ms"] == round(full["sbig1"]["pipelines"]["default"]["ms_per_step"], 4)
    assert line["parity_ok"] is True and line["detail"] == "gpurun_out/bench_detail.json"


def test_line_degrades_instead_of_failing():
    """A report that outgrows the limit still yields ONE parsable line under it (optional keys go first, `truncated` says
    which) -- never an exception after a multi-minute run."""
    b = _bench()
    full = json.load(open(os.path.join(ROOT, "profiles", "r02_v3_bench_100m.json")))
    full["ms_per_step_unprofiled"] = full["ms_per_step"]
    line = b.summary_line(full, None)
    line["cpu_baseline"]["sample"] = "x" * 3000
    line["cpu_baseline_all_cores"]["sample"] = "y" * 3000
    line["config"]["workload"] = "w" * 3000
    text = b.fit_line(line)
    out = json.loads(text)
    assert len(text) < b.MAX_LINE_BYTES and out["truncated"]
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in out, k
    small = b.summary_line(full, None)
    assert "truncated" not in json.loads(b.fit_line(small))
