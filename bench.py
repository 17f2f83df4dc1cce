#!/usr/bin/env python3
"""bench.py -- PAF mappings/s through the MI355X plane-sweep (+scaffold) filter.

One "step" = one pass of the whole filter (PafFilter::apply_filters, src/paf_filter.rs:379-747)
over one synthetic PAF shard that is already resident in HBM as the SoA of include/sweepga_gpu.h.
Workload (BASELINE.json configs[3], "S-pan"): 100 single-chromosome genomes, 9,900 ordered
non-self genome pairs ("10 k groups"), 10^8 mappings per GPU, lognormal group sizes; flags
`--num-mappings 1:1 --scaffold-jump 0` by default (the sort+sweep path), `--pipeline full` for
1:1 + scaffold chaining + scaffold 1:1 filter + rescue.  Genome pairs are independent, so with
N GPUs every rank filters its own shard (weak scaling, no collective on the data path).

Prints ONE JSON line on rank 0 (see the field list at the bottom).
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

ALGO_BYTES_SWEEP = 33   # 4 x u32 coords + f64 identity + 2 x u32 segment ids in, 1 B flag out
ALGO_BYTES_FULL = 47    # + u32 matches, u32 block_len, u8 strand in; u32 chain id, u8 status out
HBM_PEAK_GBPS = 8000.0  # MI355X_MICROARCH.md: 8 TB/s spec


# BASELINE.json's metric, verbatim
BASELINE_METRIC = "PAF mappings/sec through plane-sweep+scaffold filter, 1/2/4/8 MI355X"


def gen_shard(torch, n, n_genomes, seed, device, chr_len=150_000_000):
    """S-pan shard on the device: group-major order, as an aligner emits pairs."""
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    P = n_genomes * (n_genomes - 1)
    w = torch.exp(0.5 * torch.randn(P, generator=g, device=device, dtype=torch.float64))
    sizes = torch.floor(w / w.sum() * n).to(torch.int64)
    sizes[0] += n - int(sizes.sum())
    pair = torch.repeat_interleave(torch.arange(P, device=device, dtype=torch.int32), sizes)
    q = torch.div(pair, n_genomes - 1, rounding_mode="floor")
    t = pair - q * (n_genomes - 1)
    t = t + (t >= q).to(torch.int32)
    del pair
    ln = torch.exp(7.6009 + 1.2 * torch.randn(n, generator=g, device=device)).clamp_(100, 500_000).to(torch.int32)
    room = (chr_len - ln).to(torch.float32)
    qs = (torch.rand(n, generator=g, device=device) * room).to(torch.int32)
    syn = torch.rand(n, generator=g, device=device) < 0.7
    ts_syn = (qs.to(torch.float32) + 50_000.0 * torch.randn(n, generator=g, device=device))
    ts_syn = torch.minimum(ts_syn.clamp_(min=0), room).to(torch.int32)
    ts_rep = (torch.rand(n, generator=g, device=device) * room).to(torch.int32)
    ts = torch.where(syn, ts_syn, ts_rep)
    del syn, ts_syn, ts_rep, room
    a = torch._standard_gamma(torch.full((n,), 5.0, device=device))
    b = torch._standard_gamma(torch.full((n,), 1.5, device=device))
    ident = 0.70 + 0.30 * (a / (a + b))
    del a, b
    block = ln
    matches = torch.floor(ident.to(torch.float64) * block.to(torch.float64)).to(torch.int32)
    identity = matches.to(torch.float64) / block.to(torch.float64)
    strand = (torch.rand(n, generator=g, device=device) < 0.1).to(torch.uint8)
    table = torch.arange(n_genomes, device=device, dtype=torch.int32)
    cols = dict(q_id=q.contiguous(), t_id=t.contiguous(), q_start=qs, q_end=qs + ln, t_start=ts, t_end=ts + ln,
                identity=identity.contiguous(), matches=matches, block_len=block.contiguous(), strand=strand,
                seq_genome_last=table, seq_genome_two=table.clone())
    return cols, sizes


def make_records(lib_mod, cols, n, n_genomes):
    r = lib_mod.SwgRecords()
    r.n = n
    for k in ("q_id", "t_id", "q_start", "q_end", "t_start", "t_end", "identity", "matches", "block_len", "strand"):
        setattr(r, k, cols[k].data_ptr())
    r.n_seq = n_genomes
    r.seq_genome_last = cols["seq_genome_last"].data_ptr()
    r.n_genome_last = n_genomes
    r.seq_genome_two = cols["seq_genome_two"].data_ptr()
    r.n_genome_two = n_genomes
    return r


def make_config(sw, pipeline):
    if pipeline == "sweep":   # --num-mappings 1:1 --scaffold-jump 0
        return sw.FilterConfig(mapping_filter_mode=sw.FilterMode.OneToOne, scaffold_gap=0)
    if pipeline == "full":    # --num-mappings 1:1 --scaffold-filter 1:1 --scaffold-dist 20000 (jump 50k, mass 10k)
        return sw.FilterConfig(mapping_filter_mode=sw.FilterMode.OneToOne, scaffold_filter_mode=sw.FilterMode.OneToOne,
                               scaffold_gap=50_000, min_scaffold_length=10_000, scaffold_max_deviation=20_000)
    if pipeline == "default":  # all CLI defaults (many:many, jump 50k, mass 10k)
        return sw.FilterConfig()
    raise SystemExit(f"unknown --pipeline {pipeline}")


def _oracle_records(cols, lo, hi, n_names):
    import numpy as np
    from tests import orc
    h = {k: cols[k][lo:hi].cpu().numpy() for k in ("q_id", "t_id", "q_start", "q_end", "t_start", "t_end", "identity",
                                                   "matches", "block_len", "strand")}
    names = [f"g{i:03d}#1#chr1" for i in range(n_names)]
    u = lambda a: np.ascontiguousarray(a.astype(np.uint64))
    return orc.Records([names[i] for i in h["q_id"]], [names[i] for i in h["t_id"]], u(h["q_start"]), u(h["q_end"]),
                       u(h["t_start"]), u(h["t_end"]), u(h["block_len"]), np.ascontiguousarray(h["identity"]),
                       u(h["matches"]), np.where(h["strand"] == 0, ord("+"), ord("-")).astype(np.uint8),
                       u(np.arange(hi - lo)))


def _oracle_config(cfg):
    from tests import orc
    return orc.Config(mapping_filter_mode=int(cfg.mapping_filter_mode), mapping_max_per_query=cfg.mapping_max_per_query or 0,
                      mapping_max_per_target=cfg.mapping_max_per_target or 0,
                      scaffold_filter_mode=int(cfg.scaffold_filter_mode), scaffold_max_per_query=cfg.scaffold_max_per_query or 0,
                      scaffold_max_per_target=cfg.scaffold_max_per_target or 0, overlap_threshold=cfg.overlap_threshold,
                      scaffold_gap=cfg.scaffold_gap, min_scaffold_length=cfg.min_scaffold_length,
                      scaffold_overlap_threshold=cfg.scaffold_overlap_threshold,
                      scaffold_max_deviation=cfg.scaffold_max_deviation, scoring_function=int(cfg.scoring_function),
                      min_identity=cfg.min_identity, min_scaffold_identity=cfg.min_scaffold_identity)


def cpu_baseline_all_cores(cols, sizes, cfg, n_names, first_group, per_thread=150_000, max_threads=32):
    """'What a group-parallel CPU filter would give' (SURVEY.md 8d-ii): the same oracle on T host threads at once,
    every thread on its own whole genome-pair groups (the reference itself filters on one thread)."""
    import threading
    import numpy as np
    from tests import orc
    csum = np.concatenate([[0], sizes.cumsum(0).cpu().numpy()])
    T = max(1, min(os.cpu_count() or 1, max_threads))
    ocfg = _oracle_config(cfg)
    jobs, g = [], first_group
    for _ in range(T):
        g2 = int(np.searchsorted(csum, csum[g] + per_thread)) if g < len(csum) - 1 else g
        g2 = min(max(g2, g + 1), len(csum) - 1)
        if g2 <= g:
            break
        jobs.append(_oracle_records(cols, int(csum[g]), int(csum[g2]), n_names))
        g = g2
    if not jobs:
        return None
    bar = threading.Barrier(len(jobs) + 1)
    th = [threading.Thread(target=orc.apply_filters, args=(ocfg, r), kwargs={"barrier": bar}) for r in jobs]
    for t in th:
        t.start()
    bar.wait()
    t0 = time.perf_counter()
    for t in th:
        t.join()
    wall = time.perf_counter() - t0
    m = sum(len(r) for r in jobs)
    return dict(value=m / wall, unit="mappings/s", cores=len(jobs), kind="port",
                sample=f"{len(jobs)} threads x ~{per_thread} mappings (whole groups), all started together, wall {wall:.2f} s")


def full_parity(cols, sizes, cfg, n_names, status_dev, chain_dev, target, max_threads=64, groups_per_job=25):
    """Parity beyond the single-thread sample: the oracle over whole genome-pair groups on all host threads (groups
    are independent units of the filter), compared record by record with the device results of the timed workload."""
    import numpy as np
    from tests import orc
    T = max(1, min(os.cpu_count() or 1, max_threads))
    csum = np.concatenate([[0], sizes.cumsum(0).cpu().numpy()]).astype(np.int64)
    g_hi = int(np.searchsorted(csum, min(int(target), int(csum[-1])), side="left"))
    g_hi = max(1, min(g_hi, len(csum) - 1))
    bounds = np.unique(np.concatenate([csum[0:g_hi:groups_per_job], [csum[g_hi]]]))
    m = int(bounds[-1])
    h = {k: np.ascontiguousarray(cols[k][:m].cpu().numpy()) for k in ("q_id", "t_id", "q_start", "q_end", "t_start", "t_end",
                                                                       "identity", "matches", "block_len", "strand")}
    names = [f"g{i:03d}#1#chr1" for i in range(n_names)]
    ost, och, wall = orc.apply_filters_by_groups(_oracle_config(cfg), h, names, bounds, T)
    gst = status_dev[:m].cpu().numpy()
    status_equal = bool(np.array_equal(gst, ost))
    chain_equal = None
    if cfg.scaffold_gap:
        job = np.searchsorted(bounds, np.arange(m), side="right").astype(np.int64)
        lab = np.where(och != 0, (job << 32) | och.astype(np.int64), 0)
        chain_equal = bool(orc.same_chain_partition(chain_dev[:m].cpu().numpy(), lab))
    return {"mappings_checked": m, "groups_checked": g_hi, "status_equal": status_equal, "chain_partition_equal": chain_equal,
            "oracle_threads": T, "oracle_wall_s": wall, "oracle_value": m / wall, "unit": "mappings/s"}


def cpu_baseline(cols, sizes, cfg, sample_target, status_dev, chain_dev, n_names):
    """The CPU oracle (port of the reference, 1 thread like the reference's filter) timed on a bounded
    sample: the first whole genome-pair groups of this rank's shard.  Also the parity check."""
    import numpy as np
    from tests import orc
    csum = sizes.cumsum(0).cpu().numpy()
    g = int(np.searchsorted(csum, sample_target)) + 1
    m = int(csum[min(g, len(csum)) - 1])
    rec = _oracle_records(cols, 0, m, n_names)
    ocfg = _oracle_config(cfg)
    ost, och, secs = orc.apply_filters(ocfg, rec, want_seconds=True)
    gst = status_dev[:m].cpu().numpy()
    parity = bool(np.array_equal(gst, ost))
    if parity and cfg.scaffold_gap:
        # chain numbers are global to a call; compare the partition they induce on the sample
        gch = chain_dev[:m].cpu().numpy()
        a = {}
        for x, y in zip(gch.tolist(), och.tolist()):
            if (x == 0) != (y == 0) or a.setdefault(x, y) != y:
                parity = False
                break
    return dict(value=m / secs, unit="mappings/s", cores=1, kind="port",
                sample=f"first {g} genome-pair groups of rank 0's shard = {m} mappings, oracle apply_filters {secs:.2f} s"), parity


def end_to_end(n_lines, ref_lines, threads):
    """PAF file -> PAF file through the C++ host (native ingest, swg_filter, native egress), default flags, next to
    the oracle's CLI on a prefix of the same file; outputs compared byte for byte on that prefix."""
    import hashlib
    import re
    import shutil
    import subprocess
    import tempfile
    from sweepga_amd import build as _build
    ref_bin = os.path.join(ROOT, "oracle", "sweepga-ref")
    if not os.path.exists(_build.CLI):
        _build.build_cli()
    if not os.path.exists(_build.SYNTH):
        _build.build_synth()
    work = tempfile.mkdtemp(prefix="swg_e2e_", dir=os.environ.get("TMPDIR", "/tmp"))
    try:
        paf = os.path.join(work, "in.paf")
        with open(paf, "wb") as f:
            subprocess.check_call([_build.SYNTH, str(n_lines)], stdout=f)
        size = os.path.getsize(paf)
        targ = ["--threads", str(threads)] if threads else []
        best = None
        for _ in range(2):  # second run: page cache warm, as for the CPU side
            t0 = time.perf_counter()
            r = subprocess.run([_build.CLI, paf, "--output-file", os.path.join(work, "gpu.paf"), *targ],
                               capture_output=True, text=True)
            wall = time.perf_counter() - t0
            if r.returncode != 0:
                return {"error": r.stderr.strip()[-300:]}
            best = wall if best is None else min(best, wall)
        m = re.search(r"read ([\d.]+) ms \(load ([\d.]+), parse ([\d.]+)\), filter ([\d.]+) ms \(device ([\d.]+), h2d ([\d.]+), "
                      r"d2h ([\d.]+)\), write ([\d.]+) ms", r.stderr)
        phases = dict(zip(("read_ms", "load_ms", "parse_ms", "filter_ms", "device_ms", "h2d_ms", "d2h_ms", "write_ms"),
                          map(float, m.groups()))) if m else None
        out = {"lines": n_lines, "input_bytes": size, "flags": "(defaults)", "host_threads": threads or os.cpu_count(),
               "wall_s": best, "value": n_lines / best, "unit": "mappings/s (process start to exit, page cache warm)",
               "phases": phases}
        if ref_lines > 0 and os.path.exists(ref_bin):
            sub = os.path.join(work, "sub.paf")
            with open(paf, "rb") as f, open(sub, "wb") as g:
                k = 0
                for line in f:
                    if k >= ref_lines:
                        break
                    g.write(line)
                    k += 1
            t0 = time.perf_counter()
            subprocess.check_call([ref_bin, sub, "--output-file", os.path.join(work, "ref.paf")])
            ref_wall = time.perf_counter() - t0
            subprocess.check_call([_build.CLI, sub, "--output-file", os.path.join(work, "gpu_sub.paf"), "--quiet", *targ])
            sha = lambda p: hashlib.sha256(open(p, "rb").read()).hexdigest()
            out["cpu_reference_cli"] = {"lines": k, "wall_s": ref_wall, "value": k / ref_wall, "unit": "mappings/s",
                                        "kind": "port", "cores": 1}
            out["byte_identical_on_prefix"] = sha(os.path.join(work, "ref.paf")) == sha(os.path.join(work, "gpu_sub.paf"))
        return out
    finally:
        shutil.rmtree(work, ignore_errors=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--mappings", type=int, default=100_000_000, help="mappings per GPU")
    ap.add_argument("--genomes", type=int, default=100)
    ap.add_argument("--pipeline", default="sweep", choices=["sweep", "full", "default"])
    ap.add_argument("--cpu-sample", type=int, default=5_000_000, help="mappings in the CPU-baseline sample (0 = skip)")
    ap.add_argument("--seed", type=int, default=2025)
    ap.add_argument("--others", type=int, default=2, help="timed steps for the other two flag sets (0 = skip them)")
    ap.add_argument("--pcie", action="store_true", help="also time swg_filter (host buffers in/out, PCIe included)")
    ap.add_argument("--shuffle", action="store_true",
                    help="random record order instead of group-major (no locality for the gathers; implies no CPU legs)")
    ap.add_argument("--parity-mappings", type=int, default=-1,
                    help="mappings of the timed workload checked against the oracle on all host threads "
                         "(-1 = auto: 2M (sweep) / 0.5M (scaffold pipelines) per host thread, up to the whole shard; 0 = skip)")
    ap.add_argument("--e2e", type=int, default=0, help="lines of synthetic PAF for the file->file leg (0 = skip)")
    ap.add_argument("--e2e-ref", type=int, default=1_000_000, help="prefix of that file the oracle CLI is timed on")
    ap.add_argument("--threads", type=int, default=0, help="host threads for the e2e leg (0 = all cores)")
    args = ap.parse_args()

    import torch
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: torch.cuda.is_available() is False (no CPU fallback)")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    dist = None
    if world > 1 or "RANK" in os.environ:  # launched by torch.distributed.run: one rank per GPU over RCCL
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29513")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)

    import sweepga_amd as sw
    from sweepga_amd import _lib
    ctx = sw.Context(local_rank)
    n = args.mappings
    cols, sizes = gen_shard(torch, n, args.genomes, args.seed + 7919 * rank, device)
    if args.shuffle:  # the CPU legs index whole groups by position, so they are skipped for a shuffled shard
        perm = torch.randperm(n, device=device)
        for k in ("q_id", "t_id", "q_start", "q_end", "t_start", "t_end", "identity", "matches", "block_len", "strand"):
            cols[k] = cols[k][perm].contiguous()
        del perm
        args.cpu_sample = 0
        args.parity_mappings = 0
    torch.cuda.synchronize()
    rec = make_records(_lib, cols, n, args.genomes)
    cfg = make_config(sw, args.pipeline)
    ccfg = cfg.to_c()
    status = torch.zeros(n, dtype=torch.uint8, device=device)
    chain = torch.zeros(n, dtype=torch.int32, device=device)
    stats = _lib.SwgStats()

    def step(with_stats=False):
        ctx.check(ctx.lib.swg_filter_device(ctx.handle, C.byref(rec), C.byref(ccfg), status.data_ptr(), chain.data_ptr(),
                                            C.byref(stats) if with_stats else None))

    def barrier():
        torch.cuda.synchronize()
        ctx.synchronize()
        if dist is not None:
            dist.barrier()

    for _ in range(args.warmup):
        step()
    barrier()
    ctx.profile_reset()
    ctx.profile(True)   # HIP events around every kernel launch on the library's own stream
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    ctx.synchronize()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    ctx.profile(False)
    prof = ctx.profile_table()
    step(with_stats=True)  # untimed: counts for the report
    ctx.synchronize()
    main_counts = {"in": stats.n_in, "retained": stats.n_retained, "swept": stats.n_swept, "chains": stats.n_chains,
                   "chains_kept": stats.n_chains_kept, "out": stats.n_out, "device_ms_last_step": stats.device_ms}
    status_main, chain_main = status.clone(), chain.clone()

    # the other flag sets of the same workload (shorter, same timing discipline), for the record
    others = {}
    if args.others > 0:
        for name in ("sweep", "full", "default"):
            if name == args.pipeline:
                continue
            ocfg = make_config(sw, name).to_c()

            def ostep(with_stats=False):
                ctx.check(ctx.lib.swg_filter_device(ctx.handle, C.byref(rec), C.byref(ocfg), status.data_ptr(),
                                                    chain.data_ptr(), C.byref(stats) if with_stats else None))
            ostep()
            barrier()
            t1 = time.perf_counter()
            for _ in range(args.others):
                ostep()
            ctx.synchronize()
            dt = time.perf_counter() - t1
            if dist is not None:
                tt = torch.tensor([dt], dtype=torch.float64, device=device)
                dist.all_reduce(tt, op=dist.ReduceOp.MAX)
                dt = float(tt.item())
            ostep(with_stats=True)
            ctx.synchronize()
            others[name] = {"value": n * world / (dt / args.others), "unit": "mappings/s", "ms_per_step": dt / args.others * 1e3,
                            "steps": args.others, "out": stats.n_out, "chains": stats.n_chains, "chains_kept": stats.n_chains_kept}

    pcie = None
    if args.pcie and rank == 0:
        import numpy as np
        host = {k: v.cpu().numpy() for k, v in cols.items()}
        hrec = _lib.SwgRecords()
        hrec.n = n
        for k in ("q_id", "t_id", "q_start", "q_end", "t_start", "t_end", "identity", "matches", "block_len", "strand",
                  "seq_genome_last", "seq_genome_two"):
            setattr(hrec, k, host[k].ctypes.data)
        hrec.n_seq = hrec.n_genome_last = hrec.n_genome_two = args.genomes
        hst = np.zeros(n, dtype=np.uint8)
        hch = np.zeros(n, dtype=np.uint32)
        hs = _lib.SwgStats()
        for _ in range(2):
            t1 = time.perf_counter()
            ctx.check(ctx.lib.swg_filter(ctx.handle, C.byref(hrec), C.byref(ccfg), hst.ctypes.data, hch.ctypes.data, C.byref(hs)))
            dt = time.perf_counter() - t1
        pcie = {"value": n / dt, "unit": "mappings/s", "ms": dt * 1e3, "h2d_ms": hs.h2d_ms, "d2h_ms": hs.d2h_ms,
                "device_ms": hs.device_ms, "note": "swg_filter: pageable host buffers in and out, second call"}

    e2e = end_to_end(args.e2e, args.e2e_ref, args.threads) if (args.e2e > 0 and rank == 0 and world == 1) else None

    if rank == 0:
        algo = ALGO_BYTES_SWEEP if args.pipeline == "sweep" else ALGO_BYTES_FULL
        ms_per_step = elapsed / args.steps * 1e3
        total_kernel_ms = sum(ms for _, ms in prof.values())
        dom = max(prof.items(), key=lambda kv: kv[1][1]) if prof else ("none", (1, float("nan")))
        dom_name, (dom_launches, dom_ms) = dom
        dom_avg_ms = dom_ms / max(dom_launches, 1)
        achieved = algo * n / (dom_avg_ms * 1e-3) / 1e9
        # HBM bytes per launch of that kernel from the committed rocprofv3 PMC run (same command, same n)
        traffic = None
        try:
            tj = json.load(open(os.path.join(ROOT, "profiles", "r01_v30_hbm_traffic_sweep_100m.json")))
            if tj.get("n_mappings") == n and args.pipeline == "sweep" and dom_name in tj["kernels"]:
                traffic = tj["kernels"][dom_name]["hbm_bytes_per_launch"]
        except (OSError, ValueError, KeyError):
            traffic = None
        pipe_achieved = algo * n / (ms_per_step * 1e-3) / 1e9
        cpu, parity, cpu_mt = (None, None, None)
        if world > 1:  # the CPU legs belong to the N=1 run (rank 0 would keep the other ranks waiting at the last barrier)
            args.cpu_sample = 0
            if args.parity_mappings < 0:
                args.parity_mappings = 0
        if args.cpu_sample > 0:
            cpu, parity = cpu_baseline(cols, sizes, cfg, args.cpu_sample, status_main, chain_main, args.genomes)
            cpu_mt = cpu_baseline_all_cores(cols, sizes, cfg, args.genomes, 0)
        parity_full = None
        pm = args.parity_mappings
        if pm < 0:
            per_thread = 2_000_000 if args.pipeline == "sweep" else 500_000  # ~10 s of oracle time either way
            pm = min(n, per_thread * min(os.cpu_count() or 1, 64)) if args.cpu_sample > 0 else 0
        if pm > 0:
            parity_full = full_parity(cols, sizes, cfg, args.genomes, status_main, chain_main, pm)
        out = {
            "metric": BASELINE_METRIC,
            "value": n * world / (elapsed / args.steps),
            "unit": "mappings/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u32 coordinates, f64 scores",
            "data": "synthetic",
            "config": {"workload": f"BASELINE.json configs[3] (synthetic 100 M mappings across 10 k (q,t) groups, 100-genome pangenome "
                                   f"shape; S-pan in SURVEY.md 8d): {n} mappings per GPU over {args.genomes * (args.genomes - 1)} "
                                   f"genome-pair groups ({args.genomes} single-chromosome genomes), pipeline={args.pipeline}",
                       "flags": {"sweep": "--num-mappings 1:1 --scaffold-jump 0",
                                 "full": "--num-mappings 1:1 --scaffold-filter 1:1 --scaffold-dist 20000",
                                 "default": "(defaults)"}[args.pipeline],
                       "mappings_per_gpu": n, "groups_per_gpu": args.genomes * (args.genomes - 1)},
            "roofline": {"bound": "hbm", "kernel": dom_name, "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic,
                         "traffic_unit": "bytes per launch (rocprofv3 FETCH_SIZE x2 + WRITE_SIZE, profiles/r01_v30_hbm_traffic_sweep_100m.json)",
                         "kernel_avg_ms": dom_avg_ms, "kernel_launches_per_step": dom_launches / args.steps,
                         "algorithmic_bytes_per_mapping": algo, "units_per_launch": n,
                         "pipeline_achieved": pipe_achieved, "pipeline_frac": pipe_achieved / HBM_PEAK_GBPS,
                         "kernel_ms_per_step": total_kernel_ms / args.steps},
            "cpu_baseline": cpu,
            "cpu_baseline_all_cores": cpu_mt,
            "parity_vs_oracle_on_sample": parity,
            "parity_all_threads": parity_full,
            "counts": main_counts,
            "other_pipelines": others,
            "pcie_inclusive": pcie,
            "end_to_end": e2e,
            "arena_bytes_per_mapping": {"capacity": ctx.memory_info()[0] / n, "peak_last_call": ctx.memory_info()[1] / n},
            "kernels_ms_per_step": {k: round(v[1] / args.steps, 4) for k, v in sorted(prof.items(), key=lambda kv: -kv[1][1])},
        }
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
