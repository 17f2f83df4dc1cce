import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, sweepga_amd as sw
from sweepga_amd import PafFile
from sweepga_amd.filter import PackedRecords
ctx = sw.default_context()
with PafFile(os.path.join("tests", "golden", "syeast.paf.gz")) as pf:
    cols = {k: np.ascontiguousarray(pf.column(k)) for k in ("q_id","t_id","q_start","q_end","t_start","t_end","identity","matches","block_len","strand")}
    packed = PackedRecords(pf.n, cols, int(pf.records.n_seq), np.ascontiguousarray(pf.seq_genome_last), int(pf.records.n_genome_last), np.ascontiguousarray(pf.seq_genome_two), int(pf.records.n_genome_two), None)
    f = sw.PafFilter(sw.FilterConfig())
    for _ in range(3): f.filter_columns(packed)
    ctx.profile_reset(); ctx.profile(True)
    for _ in range(10): f.filter_columns(packed)
    ctx.profile(False)
    for k,(l,ms) in sorted(ctx.profile_table().items(), key=lambda kv:-kv[1][1]): print(k, l, round(ms/10*1000,1), "us per call")
    print("device ms", f.last_stats.device_ms, "h2d", f.last_stats.h2d_ms, "d2h", f.last_stats.d2h_ms)
