"""Pair-resident stage against the global-sort stage on shapes away from the bench workload: few long pairs, deep pairs
(hundreds of members inside one gap limit).  Run it twice through gpurun, once with SWG_GROUP_FUSED=0:
    python3 tools/shape_probe.py; SWG_GROUP_FUSED=0 python3 tools/shape_probe.py
One line per shape and flag set: knob, genomes, span, records, flags, ms of the second call, the path taken, the five longest kernels."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import sweepga_amd as sw
from tests import gen
from tests.test_gpu_pairs import pair_major
for (ng, span, n) in ((2, 3_000_000, 600_000), (4, 3_000_000, 600_000), (6, 3_000_000, 600_000), (6, 10_000_000, 600_000), (10, 3_000_000, 1_000_000), (10, 30_000_000, 1_000_000)):
    rng = np.random.default_rng(24)
    rec = gen.random_records(rng, n, n_genomes=ng, chrs_per_genome=1, span=span, max_len=20_000, syntenic_frac=0.9, minus_frac=0.2)
    r = pair_major(rec, rng)
    packed = sw.pack_records(gen.records_to_meta(r))
    ctx = sw.default_context(0)
    for cfgname, cfg in (("default", sw.FilterConfig()), ("c5", sw.FilterConfig(scaffold_filter_mode=sw.FilterMode.OneToOne, scaffold_max_deviation=20000))):
        f = sw.PafFilter(cfg)
        for it in range(2):
            ctx.profile_reset(); ctx.profile(True)
            t0 = time.perf_counter(); f.filter_columns(packed); dt = time.perf_counter() - t0
            ctx.profile(False)
            t = ctx.profile_table()
            top = sorted(t.items(), key=lambda kv: -kv[1][1])[:5]
        print(os.environ.get("SWG_GROUP_FUSED", "1"), ng, span, n, cfgname, round(dt * 1e3, 2), "ms", "pair" if "pair_renumber" in t and "chain_cuts" not in t and "cuts_from_scan" not in t else "global", [(k, v[0], round(v[1], 2)) for k, v in top], flush=True)
