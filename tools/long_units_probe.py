"""How long the long units of the bench workload are (tools/README.md): per (query, target, strand) group of S-pan the units of the
chaining step -- a member opens a unit when its q_start lies beyond every earlier q_end by more than the gap limit
(src/paf_filter.rs:786-796) -- and of those with LABEL_CAP_ELEMS = 9,216 members and more their count and lengths.  GPU box."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
dev = torch.device("cuda:0")
cols, sizes = bench.gen_shard(torch, n, 100, 1234, dev)
pair = cols["q_id"].to(torch.int64) * 100 + cols["t_id"].to(torch.int64)
key = (pair * 2 + cols["strand"].to(torch.int64)) * (1 << 32) + cols["q_start"].to(torch.int64)
order = torch.argsort(key)
g = (key[order] >> 32)
qs = cols["q_start"][order].to(torch.int64)
qe = cols["q_end"][order].to(torch.int64)
del key, order
# running maximum of q_end inside a group: (group << 32 | q_end) cummax
comp = g * (1 << 32) + qe
run = torch.cummax(comp, 0).values
prev = torch.cat([torch.zeros(1, dtype=torch.int64, device=dev), run[:-1]])
new_group = torch.cat([torch.ones(1, dtype=torch.bool, device=dev), g[1:] != g[:-1]])
cut = new_group | (qs > (prev & 0xffffffff) + 50_000) | ((prev >> 32) != g)
starts = torch.nonzero(cut).flatten()
lens = torch.diff(torch.cat([starts, torch.tensor([n], device=dev)]))
print("units", int(lens.numel()), "mean", float(lens.float().mean()), "max", int(lens.max()))
big = lens[lens >= 9216]
print("long units", int(big.numel()), "members in them", int(big.sum()), "sorted lengths (top 20)", sorted(big.tolist(), reverse=True)[:20])
for lo, hi in ((1024, 2048), (2048, 4096), (4096, 9216)):
    m = lens[(lens >= lo) & (lens < hi)]
    print(f"units of {lo}..{hi}: {int(m.numel())} holding {int(m.sum())}")
print("pair sizes: max", int(sizes.max()), "over 32768:", int((sizes > 32768).sum()), "over 65535:", int((sizes > 65535).sum()))
