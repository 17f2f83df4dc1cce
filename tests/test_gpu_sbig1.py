"""BASELINE.json configs[2] ("S-big1"): ONE (query, target) chromosome pair, 248,956,422 bp, 10^7 mappings
(seed 1234, depth ~165): a single query-axis segment and a single target-axis segment.

Full size through swg_filter_device, record for record against the oracle (src/plane_sweep_exact.rs:268-433 via
src/paf_filter.rs:972-1123), for `--num-mappings 1:1 --scaffold-jump 0`.  The reference's chaining scan is
O(n x window) (src/paf_filter.rs:784-851) and at this depth a window holds ~2,000 mappings, so the oracle needs hours for
the scaffold flag sets at 10^7; those are checked (status AND chain numbers) on
  * 10^6 mappings on the full-length chromosomes (depth ~16), and
  * 2 x 10^5 mappings with the chromosome length scaled by n / 10^7, i.e. at S-big1's own depth (long chaining units,
    the block-speculative path of swg_scaffold.hip),
for the default flags and for `--num-mappings 1:1 --scaffold-filter 1:1 --scaffold-dist 20000`.
All oracle runs go on their own host threads at once (one group = one oracle thread)."""
import ctypes as C
import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

CASES = {  # name: (pipeline, n, chromosome length)
    "sweep_full_size": ("sweep", 10_000_000, 248_956_422),
    "default_1e6_full_length": ("default", 1_000_000, 248_956_422),
    "default_2e5_same_depth": ("default", 200_000, 4_979_128),
    "full_1e6_full_length": ("full", 1_000_000, 248_956_422),
    "full_2e5_same_depth": ("full", 200_000, 4_979_128),
    "sweep_1e6_same_depth": ("sweep", 1_000_000, 24_895_642),
}


@pytest.fixture(scope="module")
def results():
    import torch
    import bench
    import sweepga_amd as sw
    from sweepga_amd import _lib
    from tests import orc
    device = torch.device("cuda", 0)
    ctx = sw.Context(0)
    out, threads = {}, []
    for name, (pipeline, n, chr_len) in CASES.items():
        cols, _ = bench.gen_shard(torch, n, 2, 1234, device, chr_len=chr_len, single_pair=True)
        run = bench.Runner(torch, sw, _lib, ctx, device, None, cols, n, 2)
        cfg = bench.make_config(sw, pipeline)
        run.step(cfg.to_c(), with_stats=True)
        ctx.synchronize()
        st, ch = run.status[:n].cpu().numpy(), run.chain[:n].cpu().numpy()
        host = bench._host_cols(cols, 0, n)
        ost, och = np.zeros(n, np.uint8), np.zeros(n, np.uint32)
        th = threading.Thread(target=orc.apply_filters_ids, args=(bench._oracle_config(cfg), host, bench.SBIG1_NAMES, 0, n, ost, och))
        th.start()
        threads.append(th)
        out[name] = dict(st=st, ch=ch, ost=ost, och=och, n_out=int(run.stats.n_out), scaffold=bool(cfg.scaffold_gap))
        del run, cols
    for th in threads:
        th.join()
    return out


@pytest.mark.parametrize("name", list(CASES))
def test_sbig1_matches_oracle(results, name):
    r = results[name]
    assert r["n_out"] == int((r["ost"] != 0).sum())
    assert np.array_equal(r["st"], r["ost"]), int((r["st"] != r["ost"]).sum())
    if r["scaffold"]:
        assert np.array_equal(r["ch"], r["och"]), int((r["ch"] != r["och"]).sum())
    else:
        assert not r["ch"].any()
