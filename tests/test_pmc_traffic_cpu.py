"""tools/pmc_traffic.py: the launches of one call come from the LAST pipeline execution of the profiled process (a first call
whose arena overflowed runs its pipeline twice; a first call may take a fall-back it never takes again), the bytes per launch
are averages over all launches, 2 x FETCH_SIZE + WRITE_SIZE."""
import csv
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEAD = ["Correlation_Id", "Dispatch_Id", "Agent_Id", "Queue_Id", "Process_Id", "Thread_Id", "Grid_Size", "Kernel_Id", "Kernel_Name",
        "Workgroup_Size", "LDS_Block_Size", "Scratch_Size", "VGPR_Count", "Accum_VGPR_Count", "SGPR_Count", "Counter_Name",
        "Counter_Value", "Start_Timestamp", "End_Timestamp"]


def _write(d, counter, dispatches):
    os.makedirs(os.path.join(d, "runc"), exist_ok=True)
    with open(os.path.join(d, "runc", "1_counter_collection.csv"), "w", newline="") as f:
        w = csv.writer(f, quoting=csv.QUOTE_NONNUMERIC)
        w.writerow(HEAD)
        for k, (name, value) in enumerate(dispatches, 1):
            w.writerow([k, k, "Agent 2", 2, 1, 1, 256, 7, name, 256, 0, 0, 32, 0, 32, counter, value, 0, 1])


def test_launches_per_call_come_from_the_last_execution(tmp_path):
    ns = "void (anonymous namespace)::"
    torch_k = "void at::native::(anonymous namespace)::some_generator_kernel<double>(int)"
    # execution 1: cut short after the first sort pass, with a fall-back launch; executions 2 and 3: the steady state
    first = [(torch_k, 999.0), (ns + "prepare_kernel(unsigned long)", 10.0), (ns + "os_pass_packed9_kernel<unsigned int>(int)", 4.0),
             (ns + "gather_all_words_kernel(unsigned long)", 100.0), (ns + "gather_all_words_kernel(unsigned long)", 100.0)]
    steady = [(ns + "prepare_kernel(unsigned long)", 10.0)] + [(ns + "os_pass_packed9_kernel<unsigned int>(int)", 4.0)] * 3 + \
             [(ns + "gather_all_words_kernel(unsigned long)", 100.0), ("void swg_scaf::(anonymous namespace)::chain_walk_kernel<256, true, true>(unsigned int)", 7.0)]
    fe, wr = str(tmp_path / "fetch"), str(tmp_path / "write")
    _write(fe, "FETCH_SIZE", first + steady + steady)
    _write(wr, "WRITE_SIZE", [(n, v / 2) for n, v in first + steady + steady])
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "pmc_traffic.py"), fe, wr, "1000", "2"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    j = json.loads(r.stdout)
    assert j["pipeline_executions_profiled"] == 3 and j["calls_profiled"] == 2
    k = j["kernels"]
    assert "some_generator" not in k                                   # torch's kernels are not the library's
    assert k["prepare"]["launches_per_call"] == 1.0 and k["prepare"]["launches_profiled"] == 3
    assert k["os_pass_packed9"]["launches_per_call"] == 3.0            # (7 launches in all: 1 + 3 + 3)
    assert k["gather_all_words"]["launches_per_call"] == 1.0           # (4 launches in all: the first call's two, then one each)
    assert k["chain_walk_spec"]["launches_per_call"] == 1.0            # template <BIGW, FUSED, SPEC = true> has its own label
    assert k["gather_all_words"]["hbm_bytes_per_launch"] == (2 * 100.0 + 50.0) * 1024
    want = sum(v["hbm_bytes_per_launch"] * v["launches_per_call"] for v in k.values())
    assert abs(j["hbm_bytes_per_call_all_kernels"] - want) < 1e-6


def test_mixed_call_shapes_and_the_collection_time_stamp(tmp_path):
    """A trace may mix call shapes: a call starts with pair_boundary (large inputs), pair_hash (small ones) or
    prepare (no pair plan), and the prepare of a call that the pair path handed over to the global-sort stage is not a new call:
    with SWG_CALL_MARKER=1 (tools/profile_round.sh) every call opens with the empty launch `call_begin`, and the trace is cut there.  The
    JSON is stamped with the digest profile_round.sh left next to the counters when it collected them, and only when both passes
    carry the same one."""
    ns = "void swg_scaf::(anonymous namespace)::"
    a = "void (anonymous namespace)::"
    big = [(ns + "pair_boundary_kernel(unsigned int)", 8.0), (ns + "pair_sort_big_kernel(int)", 50.0)]
    handed = [(ns + "pair_boundary_kernel(unsigned int)", 8.0), (ns + "pair_sort_big_kernel(int)", 50.0), (a + "prepare_kernel(unsigned long)", 10.0),
              (a + "gather_all_words_kernel(unsigned long)", 100.0)]
    small = [(ns + "pair_hash_kernel(unsigned int)", 1.0), (ns + "pair_sort_kernel<64, 16, 16, 256, 64, true>(int)", 2.0)]
    plain = [(a + "prepare_kernel(unsigned long)", 10.0), (a + "gather_all_words_kernel(unsigned long)", 100.0)]
    mark = [(a + "call_begin_kernel()", 0.0)]
    seq = mark + big + mark + handed + mark + small + mark + plain + mark + big
    fe, wr = str(tmp_path / "fetch"), str(tmp_path / "write")
    _write(fe, "FETCH_SIZE", seq)
    _write(wr, "WRITE_SIZE", seq)
    run = lambda: json.loads(subprocess.run([sys.executable, os.path.join(ROOT, "tools", "pmc_traffic.py"), fe, wr, "1000", "5"],   # noqa: E731
                                            capture_output=True, text=True, check=True).stdout)
    j = run()
    assert j["pipeline_executions_profiled"] == 5                       # (the handed-over call's prepare opens no sixth)
    assert j["kernels"]["pair_sort_big"]["launches_per_call"] == 1.0 and j["kernels"]["prepare"]["launches_per_call"] == 0.0
    assert "call_begin" not in j["kernels"]
    assert j["lib_sha256"] is None                                      # nothing was left at collection time: not stamped
    for d, digest in ((fe, "ab" * 32), (wr, "ab" * 32)):
        open(os.path.join(d, "lib_sha256.txt"), "w").write(digest + "  libsweepga_gpu.so\n")
    assert run()["lib_sha256"] == "ab" * 32
    open(os.path.join(wr, "lib_sha256.txt"), "w").write("cd" * 32 + "  libsweepga_gpu.so\n")
    assert run()["lib_sha256"] is None                                  # the two passes ran different libraries
