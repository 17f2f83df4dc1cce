"""Millions of tiny pairs (all-vs-all of many small contigs): 2*10^7 records in pairs of ~8 records, pair-major.  Times the
default, sweep and full flags on the pair path and with SWG_GROUP_FUSED=0 SWG_SEG_SORT=0 (run it twice)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import sweepga_amd as sw
from sweepga_amd import _lib as lib_mod
import bench
dev = torch.device("cuda:0")
n, ncontig = 20_000_000, 1600
g = torch.Generator(device=dev); g.manual_seed(5)
pair = torch.sort(torch.randint(0, ncontig * ncontig, (n,), generator=g, device=dev))[0]
q = (pair // ncontig).to(torch.int32); t = (pair % ncontig).to(torch.int32)
ln = torch.randint(200, 5000, (n,), generator=g, device=dev, dtype=torch.int32)
qs = torch.randint(0, 2_000_000, (n,), generator=g, device=dev, dtype=torch.int32)
ts = (qs + torch.randint(-20000, 20000, (n,), generator=g, device=dev, dtype=torch.int32)).clamp_(min=0)
matches = (ln.to(torch.float64) * 0.9).to(torch.int32)
table = (torch.arange(ncontig, device=dev, dtype=torch.int32) // 40).contiguous()   # 40 genomes of 40 contigs
cols = dict(q_id=q.contiguous(), t_id=t.contiguous(), q_start=qs, q_end=qs + ln, t_start=ts, t_end=ts + ln,
            identity=(matches.to(torch.float64) / ln.to(torch.float64)).contiguous(), matches=matches, block_len=ln.contiguous(),
            strand=(torch.rand(n, generator=g, device=dev) < 0.1).to(torch.uint8), seq_genome_last=table, seq_genome_two=table.clone())
torch.cuda.synchronize()
ctx = sw.default_context(0)
r = lib_mod.SwgRecords(); r.n = n
for k in bench.REC_COLS: setattr(r, k, cols[k].data_ptr())
r.n_seq = ncontig; r.seq_genome_last = table.data_ptr(); r.n_genome_last = 40; r.seq_genome_two = cols["seq_genome_two"].data_ptr(); r.n_genome_two = 40
import ctypes as C
status = torch.zeros(n, dtype=torch.uint8, device=dev); chain = torch.zeros(n, dtype=torch.int32, device=dev)
for p in ("default", "sweep", "full"):
    ccfg = bench.make_config(sw, p).to_c()
    for it in range(3):
        ctx.profile_reset(); ctx.profile(True)
        t0 = time.perf_counter()
        ctx.check(ctx.lib.swg_filter_device(ctx.handle, C.byref(r), C.byref(ccfg), status.data_ptr(), chain.data_ptr(), None))
        ctx.synchronize(); dt = time.perf_counter() - t0
        ctx.profile(False)
    tb = ctx.profile_table()
    print(os.environ.get("SWG_GROUP_FUSED", "1"), p, round(dt * 1e3, 2), "ms", int((status != 0).sum()), [(k, v[0], round(v[1], 2)) for k, v in sorted(tb.items(), key=lambda kv: -kv[1][1])[:7]], flush=True)
