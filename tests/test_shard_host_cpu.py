"""Host side of swg_filter_multi (sweepga_amd/csrc/host/shard_host.h: plan, scatter, merge on host threads) without a GPU:
tests/native/shard_host_bench.cpp runs the three phases around a stand-in for the per-device filter call and checks pair
ids (first-appearance order), statuses and the globally renumbered chain ids against a serial restatement of the protocol
(src/paf_filter.rs:517-521: kept chains are numbered genome pair by genome pair in the order the pairs first appear)."""
import json
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def exe(tmp_path_factory):
    out = str(tmp_path_factory.mktemp("shard") / "shard_host_bench")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-pthread", "-Wall", "-Wextra", "-o", out,
                           os.path.join(ROOT, "tests", "native", "shard_host_bench.cpp")])
    return out


@pytest.mark.parametrize("records,genomes,shards,threads", [
    (1, 2, 2, 1), (50_000, 5, 3, 1), (200_000, 7, 8, 4), (1_000_000, 12, 8, 3), (3_000_000, 30, 4, 8), (700_000, 3, 16, 5)])
def test_plan_scatter_merge_against_serial_protocol(exe, records, genomes, shards, threads):
    r = subprocess.run([exe, str(records), str(genomes), str(shards), str(threads), "1"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads(r.stdout.strip().splitlines()[-1])
    assert d["ok"] is True and d["checked"] is True and d["records"] == records and d["shards"] == shards
    if d["pairs"] >= 20 * shards:
        assert d["load_max_over_mean"] < 1.1   # LPT by mapping count
