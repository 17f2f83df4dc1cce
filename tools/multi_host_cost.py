"""swg_filter_multi on an input that is NOT grouped by query genome (S-pan shuffled): wall time of the whole call with the shards
gathered through the contexts' pinned rings (the default) and with SWG_MULTI_SCATTER=1 (the shards copied into pageable host
columns first, rounds 2-5).  Several contexts on the ONE test GPU stand in for the GPUs of a node: the device work is the same
either way, the difference is the host's.  Through gpurun:
    python3 tools/multi_host_cost.py [mappings] [contexts];  SWG_MULTI_SCATTER=1 python3 tools/multi_host_cost.py ..."""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402
import sweepga_amd as sw  # noqa: E402
from sweepga_amd import _lib  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
k = int(sys.argv[2]) if len(sys.argv) > 2 else 2
dev = torch.device("cuda:0")
cols, _ = bench.gen_shard(torch, n, 100, 2025, dev)
perm = torch.randperm(n, device=dev)
host = {c: (cols[c][perm].contiguous() if c in bench.REC_COLS else cols[c]).cpu().numpy() for c in cols}
del cols, perm
torch.cuda.empty_cache()
rec = _lib.SwgRecords()
rec.n = n
for c in bench.REC_COLS + ("seq_genome_last", "seq_genome_two"):
    setattr(rec, c, host[c].ctypes.data)
rec.n_seq = rec.n_genome_last = rec.n_genome_two = 100
ctxs = [sw.Context(0) for _ in range(k)]
arr = (C.c_void_p * k)(*[c.handle for c in ctxs])
st = np.zeros(n, dtype=np.uint8)
ch = np.zeros(n, dtype=np.uint32)
stats = _lib.SwgStats()
for name in ("default", "sweep"):
    ccfg = bench.make_config(sw, name).to_c()
    best = None
    for it in range(3):
        t0 = time.perf_counter()
        ctxs[0].check(ctxs[0].lib.swg_filter_multi(arr, k, C.byref(rec), C.byref(ccfg), st.ctypes.data, ch.ctypes.data, C.byref(stats)))
        dt = time.perf_counter() - t0
        best = dt if best is None or (it and dt < best) else best
    print("scatter-first" if os.environ.get("SWG_MULTI_SCATTER") else "pinned-ring", name, "contexts", k, "records", n, "wall_s", round(best, 3), "device_ms", round(stats.device_ms, 1),
          "h2d_ms", round(stats.h2d_ms, 1), "kept", int((st != 0).sum()), flush=True)
