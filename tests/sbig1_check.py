"""Helper of tests/test_gpu_sbig1.py: runs in a FRESH process (torch must initialise HIP before libsweepga_gpu.so is loaded:
torch ships its own HIP runtime and fails to find the device once another copy has been initialised in the process).
Generates the S-big1 cases on the GPU, filters them through swg_filter_device and checks every record against the oracle
(one host thread per case, all at once).  Prints one JSON object."""
import json
import sys
import threading

import numpy as np


def main(cases):
    import torch
    assert torch.cuda.is_available()
    import bench
    import sweepga_amd as sw
    from sweepga_amd import _lib
    import os
    from tests import orc
    if os.environ.get("SBIG1_FAST_INVERSION"):  # tools/sbig1_full_parity.py: step 4b through the oracle's bucket index
        orc.set_fast_inversion(True)
    device = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    ctx = sw.Context(0)
    out, threads, keep = {}, [], {}
    for name, (pipeline, n, chr_len) in cases.items():
        cols, _ = bench.gen_shard(torch, n, 2, 1234, device, chr_len=chr_len, single_pair=True)
        torch.cuda.synchronize()   # the library runs on its own stream: the generator's kernels must have finished
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
        keep[name] = (st, ch, ost, och, int(run.stats.n_out), bool(cfg.scaffold_gap), host)
        del run, cols
    for th in threads:
        th.join()
    for name, (st, ch, ost, och, n_out, scaffold, _) in keep.items():
        out[name] = {"n": len(st), "n_out_device": n_out, "n_out_oracle": int((ost != 0).sum()),
                     "status_mismatches": int((st != ost).sum()),
                     "chain_mismatches": int((ch != och).sum()) if scaffold else int((ch != 0).sum())}
    print(json.dumps(out))


if __name__ == "__main__":
    main(json.loads(sys.argv[1]))
