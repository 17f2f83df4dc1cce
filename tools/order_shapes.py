"""The default flags on S-pan's records in orders and shapes away from the bench workload (one line each: ms per call, the path
taken, the longest kernels): pair-major (the bench), shuffled, by query in query order with the targets interleaved (what wfmash
writes), and 100 genomes x 20 chromosomes (198,000 sequence pairs).  Through gpurun:  python3 tools/order_shapes.py [mappings]
[flag sets] [chromosome counts: only those shapes]"""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
import sweepga_amd as sw  # noqa: E402
from sweepga_amd import _lib  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
pipes = sys.argv[2].split(",") if len(sys.argv) > 2 else ["default"]
device = torch.device("cuda:0")
ctx = sw.Context(0)


def run(tag, cols, G):
    rec = bench.make_records(_lib, cols, n, G)
    status = torch.zeros(n, dtype=torch.uint8, device=device)
    chain = torch.zeros(n, dtype=torch.int32, device=device)
    for p in pipes:
        ccfg = bench.make_config(sw, p).to_c()
        best = None
        for it in range(4):
            if it == 3:
                ctx.profile_reset()
                ctx.profile(True)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            ctx.check(ctx.lib.swg_filter_device(ctx.handle, C.byref(rec), C.byref(ccfg), status.data_ptr(), chain.data_ptr(), None))
            ctx.synchronize()
            dt = time.perf_counter() - t0
            if it and it < 3:
                best = dt if best is None or dt < best else best
        ctx.profile(False)
        t = ctx.profile_table()
        top = sorted(t.items(), key=lambda kv: -kv[1][1])[:8]
        path = "pair" if "pair_renumber" in t and not any(k in t for k in ("chain_cuts", "cuts_from_scan", "sortA_keys", "sortA_keys_hist", "sortA_words")) else "global"
        print(tag, p, round(best * 1e3, 2), "ms", path, "kept", int((status != 0).sum()), [(k, v[0], round(v[1], 2)) for k, v in top], flush=True)


if len(sys.argv) > 3:   # only the many-pairs shapes: 100 genomes x C chromosomes for every C named (how the admission rule is measured:
    for c in (int(x) for x in sys.argv[3].split(",")):   # run once as it is and once with SWG_PAIR_MIN_AVG=1 / =100000)
        cols, _ = bench.gen_shard(torch, n, 100, 2025, device, chroms=c)
        run(f"100x{c}", cols, 100)
        del cols
        torch.cuda.empty_cache()
    sys.exit(0)
cols, _ = bench.gen_shard(torch, n, 100, 2025, device)
run("pair-major", cols, 100)
key = cols["q_id"].to(torch.int64) * (1 << 32) + cols["q_start"].to(torch.int64)
order = torch.argsort(key, stable=True)
del key
by_q = {k: (cols[k][order].contiguous() if k in bench.REC_COLS else cols[k]) for k in cols}
del order
run("by-query", by_q, 100)
del by_q
perm = torch.randperm(n, device=device)
shuf = {k: (cols[k][perm].contiguous() if k in bench.REC_COLS else cols[k]) for k in cols}
del perm, cols
run("shuffled", shuf, 100)
del shuf
torch.cuda.empty_cache()
cols, _ = bench.gen_shard(torch, n, 100, 2025, device, chroms=20)
run("100x20", cols, 100)
