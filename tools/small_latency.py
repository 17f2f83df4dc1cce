#!/usr/bin/env python3
"""Latency of one swg_filter call on a small input (tests/golden/syeast.paf.gz, 13,647 mappings -- the size class of
BASELINE.json configs[0]): wall and device milliseconds, median of 20 calls, for the three flag sets of bench.py."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import sweepga_amd as sw
from sweepga_amd import PafFile
ctx = sw.default_context()   # run with SWG_DEBUG=1 to have the library print its read-backs per call
with PafFile(os.path.join(ROOT, "tests", "golden", "syeast.paf.gz")) as pf:
    from sweepga_amd.filter import PackedRecords
    cols = {k: np.ascontiguousarray(pf.column(k)) for k in ("q_id","t_id","q_start","q_end","t_start","t_end","identity","matches","block_len","strand")}
    packed = PackedRecords(pf.n, cols, int(pf.records.n_seq), np.ascontiguousarray(pf.seq_genome_last), int(pf.records.n_genome_last), np.ascontiguousarray(pf.seq_genome_two), int(pf.records.n_genome_two), None)
    for name, cfg in (("default", sw.FilterConfig()), ("sweep", sw.FilterConfig(mapping_filter_mode=sw.FilterMode.OneToOne, scaffold_gap=0)),
                      ("full", sw.FilterConfig(mapping_filter_mode=sw.FilterMode.OneToOne, scaffold_filter_mode=sw.FilterMode.OneToOne, scaffold_gap=50000, min_scaffold_length=10000, scaffold_max_deviation=20000))):
        f = sw.PafFilter(cfg)
        for _ in range(3): f.filter_columns(packed)
        t=[]; d=[]
        for _ in range(20):
            t0=time.perf_counter(); f.filter_columns(packed); t.append((time.perf_counter()-t0)*1e3); d.append(f.last_stats.device_ms)
        ctx.profile_reset(); ctx.profile(True); f.filter_columns(packed); ctx.profile(False)
        launches = sum(v[0] for v in ctx.profile_table().values())
        print(name, "n", pf.n, "wall ms median", round(float(np.median(t)),3), "device ms median", round(float(np.median(d)),3),
              "kernel launches", launches, flush=True)
        os.environ["SWG_DEBUG"] = "1"   # (read once per process by the library: set before the first call to see the read-back count)
