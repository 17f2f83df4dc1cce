#!/usr/bin/env python3
"""Per-kernel HBM traffic from two rocprofv3 PMC runs (FETCH_SIZE and WRITE_SIZE collected in SEPARATE passes,
as /opt/skills/guides/MI355X_MICROARCH.md prescribes) -> the JSON bench.py reads for `roofline.traffic`.

  rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/pmc_fetch -- python3 bench.py --steps 1 --warmup 0 --cpu-sample 0 --others 0
  rocprofv3 --kernel-trace --pmc WRITE_SIZE -d gpurun_out/pmc_write -- python3 bench.py --steps 1 --warmup 0 --cpu-sample 0 --others 0
  python tools/pmc_traffic.py gpurun_out/pmc_fetch gpurun_out/pmc_write 100000000 > profiles/rNN_vMM_hbm_traffic_sweep_100m.json

Both counters are in KB per dispatch; hbm_bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 (the gfx950 FETCH_SIZE x2
correction, calibrated on os_pass which reads exactly 12 B per pair)."""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict


def short_name(name):
    n = name.replace("(anonymous namespace)::", "").replace("void ", "")
    m = re.match(r"([A-Za-z0-9_:]+)(<[^(]*>)?", n)
    base, targs = (m.group(1), m.group(2) or "") if m else (n, "")
    base = base.split("::")[-1]
    if base.endswith("_kernel"):
        base = base[:-7]
    if base == "sweep_tile":
        base += "_k1" if "true" in targs else "_kn"
    # the names below are the labels of the library's own per-launch table (SWG_LAUNCH): one label = one kernel = one rocprof
    # name, so that bench.py divides a label's PMC bytes by the duration of the SAME launches
    t = [x.strip() for x in targs.strip("<>").split(",")] if targs else []
    if base == "chain_walk" and len(t) >= 3 and t[2] == "true":
        base = "chain_walk_spec"      # blocks of long units, speculative rounds (template <BIGW, FUSED, SPEC>)
    if base == "os_pass_packed" and len(t) >= 2 and t[1] == "true":
        base = "os_pass_packed_first"  # the pass that reads (key, value) pairs and writes packed words
    if base == "pair_sort" and t:      # pair_sort_kernel<NT, ES, ER, NBK, NBIN, PERM>: by size class, PERM = through the hash grouping
        base = "pair_sort_" + ("p" if t[-1] == "true" else "") + {"64": "s", "256": "m", "1024": "l"}.get(t[0], t[0])
    if base in ("pair_finish", "pair_chains") and t:    # <NT, ...>: 64 / 256 = the small size classes
        base += {"64": "_s", "256": "_m"}.get(t[0], "")
    if base in ("pair_key1", "pair_key2", "pair_rank1", "pair_rank_count", "pair_base", "pair_base_count", "pair_sizes", "pair_number_small"):
        base = "pair_number"          # the chain_N numbering over the pair table: one label in the library
    if base in ("seg_sort", "seg_sweep") and t:   # seg_sort_kernel<NT, ...> / seg_sweep_kernel<NT, ...> (the large class and the longest segments: *_big)
        base += {"64": "_s", "256": "_m"}.get(t[0], "")
    if base == "seg_stream_small":
        base = "seg_stream_s"
    if base in ("run_alive", "run_key"):
        base = "seg_" + base
    if base == "assign_numbers_runs":
        base = "assign_numbers"
    if base == "pair_starts":          # (the run list's first step, launched under the label of its second)
        base = "pair_runs"
    if base == "pair_out":             # (large inputs: the numbering's last step also brings the results to input order)
        base = "pair_renumber"
    if base == "pair_long_plan":
        base = "spec_plan"
    if base == "pair_long_verdict":
        base = "spec_final"
    if base in ("fill_u32", "fill_u64"):
        base = "fill"
    if base == "iota_u32":
        base = "iota"
    return base


def collect(d, counter):
    """{label: [launches, sum of counter values]} and the launches per pipeline EXECUTION.  Every execution of the filter
    starts with one `prepare` launch, so the dispatches (in Dispatch_Id order) between two of them are one execution.  A call
    whose scratch arena overflowed runs its pipeline a second time (run_with_arena: grow, run again), i.e. a profile of c calls
    can hold c + 1 executions, the first of them cut short; the launches per call are therefore read off the LAST execution
    (steady state), and the per-call totals are launches-per-call x the average bytes per launch."""
    acc = defaultdict(lambda: [0, 0.0])
    rows = []
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if row.get("Counter_Name") != counter:
                continue
            raw = row["Kernel_Name"]
            if any(t in raw for t in ("at::", "rocprim::", "thrust::", "hipcub::", "c10::")):
                continue  # torch's kernels (the synthetic record generator), not the library's
            k = short_name(row["Kernel_Name"])
            a = acc[k]
            a[0] += 1
            a[1] += float(row["Counter_Value"])
            rows.append((int(row["Dispatch_Id"]), k))
    rows.sort()
    executions, last = 0, defaultdict(int)
    # A call starts with pair_boundary (large inputs: the pair plan), pair_hash (small ones) or prepare (no plan at all); a call
    # the pair-resident path hands over to the global-sort stage has pair_boundary / pair_hash AND, later, prepare -- that
    # prepare is not a new call.  (A trace may mix call shapes.)
    # With SWG_CALL_MARKER=1 (tools/profile_round.sh sets it) every call opens with the empty launch `call_begin`: unambiguous.
    # Without it, the heuristic above -- which takes a call without a plan for a hand-over when it follows one with a plan.
    marked = any(k == "call_begin" for _, k in rows)
    by_plan = False
    for _, k in rows:
        if marked:
            start = k == "call_begin"
        else:
            start = k in ("pair_boundary", "pair_hash") or (k == "prepare" and not (by_plan and "prepare" not in last))
        if start:
            executions += 1
            last = defaultdict(int)
            by_plan = k != "prepare"
        last[k] += 1
    acc.pop("call_begin", None)
    last.pop("call_begin", None)
    return acc, executions, dict(last)


def main():
    fetch_dir, write_dir, n = sys.argv[1], sys.argv[2], int(sys.argv[3])
    # filter calls inside one profiled bench.py run (--steps 1 --warmup 0: all-events pass + timed region + no-events pass + statistics = 4)
    calls = int(sys.argv[4]) if len(sys.argv) > 4 else 4
    (fe, ex_f, last_f), (wr, ex_w, last_w) = collect(fetch_dir, "FETCH_SIZE"), collect(write_dir, "WRITE_SIZE")
    executions = max(ex_f, ex_w)
    kernels = {}
    total = 0.0
    for k in sorted(set(fe) | set(wr)):
        f = fe[k][1] / fe[k][0] if fe[k][0] else 0.0
        w = wr[k][1] / wr[k][0] if wr[k][0] else 0.0
        launches = max(fe[k][0], wr[k][0])
        # launches of one call: those of the last (complete, steady-state) execution; a profile without `prepare` (the seams)
        # falls back to launches / calls
        lpc = float(max(last_f.get(k, 0), last_w.get(k, 0))) if executions else launches / calls
        per_launch = (2.0 * f + w) * 1024.0
        kernels[k] = {"launches_profiled": launches, "launches_per_call": lpc, "fetch_size_kb_per_launch": f,
                      "write_size_kb_per_launch": w, "hbm_bytes_per_launch": per_launch,
                      "hbm_bytes_per_call": per_launch * lpc}
        total += per_launch * lpc
    # the library these counters were taken FROM: tools/profile_round.sh leaves its sha256 in both counter directories when it
    # collects them (lib_sha256.txt); without that, or when the two passes ran different libraries, nothing is stamped and
    # bench.py reports no `traffic` (it only does when the library it runs has the stamped digest)
    def stamp(d):
        try:
            return open(os.path.join(d, "lib_sha256.txt")).read().split()[0]
        except (OSError, IndexError):
            return None
    sha = stamp(fetch_dir) if stamp(fetch_dir) and stamp(fetch_dir) == stamp(write_dir) else None
    print(json.dumps({"_how": __doc__.strip(), "lib_sha256": sha, "n_mappings": n, "calls_profiled": calls, "pipeline_executions_profiled": executions,
                      "hbm_bytes_per_call_all_kernels": total, "kernels": kernels}, indent=1))


if __name__ == "__main__":
    main()
