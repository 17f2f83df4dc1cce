#!/usr/bin/env python3
"""bench.py -- PAF mappings/s through the MI355X plane-sweep + scaffold filter.

One "step" = one pass of the whole filter (PafFilter::apply_filters, src/paf_filter.rs:379-747) over one synthetic
record set that is already resident in HBM as the SoA of include/sweepga_gpu.h.

Workload of the headline (BASELINE.json configs[3], "S-pan" in SURVEY.md 8d): 100 single-chromosome genomes, 9,900
ordered non-self genome pairs ("10 k groups"), 10^8 mappings per GPU, lognormal group sizes.  Three flag sets of the
reference's command line are timed on it with the same K / W:
    default  `sweepga <paf> --output-file ...` with every flag at its default (many:many, jump 50 k, mass 10 k)   <- `value`
    sweep    `--num-mappings 1:1 --scaffold-jump 0`                      (the sort + plane-sweep path alone)
    full     `--num-mappings 1:1 --scaffold-filter 1:1 --scaffold-dist 20000`   (the whole scaffold path behind a 1:1 sweep)
    c5       `--scaffold-filter 1:1 --scaffold-dist 20000`   (BASELINE.json configs[4] as SURVEY.md 8d C5 states it: many:many
             mappings, so all 10^8 records are chained and every one of them is a rescue candidate)
and BASELINE.json configs[2] ("S-big1": ONE chromosome pair, 10^7 mappings, depth ~165) is timed beside them.

N GPUs: one process per GPU (launched by torch.distributed.run, or by this script itself when it is started with
--gpus N outside a launcher).  `--scaling weak` (default): every rank filters its own 10^8-mapping shard.
`--scaling strong`: ONE 10^8 record set, genome pairs bin-packed over the ranks by sweepga_amd.shard (LPT), each
rank filters its device-resident shard, kept-chain counts are exchanged (one all_gather of two small vectors) and
every rank renumbers its own chains.  Genome pairs are independent units of the filter, so there is no other
collective on the data path.

Prints ONE JSON line (< 4 KB) on rank 0; the full report (per-kernel tables, counts, every leg) goes to --detail.
"""
import argparse
import ctypes as C
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

ALGO_BYTES_SWEEP = 33   # 4 x u32 coords + f64 identity + 2 x u32 segment ids in, 1 B flag out
ALGO_BYTES_FULL = 47    # + u32 matches, u32 block_len, u8 strand in; u32 chain id, u8 status out
HBM_PEAK_GBPS = 8000.0  # MI355X_MICROARCH.md: 8 TB/s spec
PROFILE_TAG = "r06_v5"     # profiles/<tag>_hbm_traffic_<pipeline>_<100m|sbig1_10m>.json: rocprofv3 PMC bytes per launch

# BASELINE.json's metric, verbatim
BASELINE_METRIC = "PAF mappings/sec through plane-sweep+scaffold filter, 1/2/4/8 MI355X"
PIPELINES = ("default", "sweep", "full", "c5")
FLAGS = {"sweep": "--num-mappings 1:1 --scaffold-jump 0",
         "full": "--num-mappings 1:1 --scaffold-filter 1:1 --scaffold-dist 20000",
         "c5": "--scaffold-filter 1:1 --scaffold-dist 20000",
         "default": "(defaults)",
         "k32": "--num-mappings 3:2 --scaffold-jump 0"}
REC_COLS = ("q_id", "t_id", "q_start", "q_end", "t_start", "t_end", "identity", "matches", "block_len", "strand")
SBIG1_LEN = 248_956_422


def gen_shard(torch, n, n_genomes, seed, device, chr_len=150_000_000, single_pair=False, chroms=1):
    """Synthetic records on the device, group-major order, as an aligner emits pairs.
    S-pan: n_genomes single-chromosome genomes, every ordered non-self pair, lognormal(0.5) group sizes.
    S-big1 (single_pair): every record maps sequence 0 onto sequence 1 (one query segment, one target segment).
    chroms > 1: every genome has that many chromosomes (sequence id = genome * chroms + chromosome, of chr_len / chroms bases),
    homologous chromosomes map onto each other: n_genomes * (n_genomes - 1) * chroms sequence pairs."""
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    if device.type == "cuda":
        torch.cuda.manual_seed(seed)   # _standard_gamma below draws from the device's default generator
    if single_pair:
        sizes = torch.tensor([n], dtype=torch.int64, device=device)
        q = torch.zeros(n, dtype=torch.int32, device=device)
        t = torch.ones(n, dtype=torch.int32, device=device)
    else:
        P = n_genomes * (n_genomes - 1) * chroms
        w = torch.exp(0.5 * torch.randn(P, generator=g, device=device, dtype=torch.float64))
        sizes = torch.floor(w / w.sum() * n).to(torch.int64)
        if chroms > 1:   # (the flooring's remainder one record per pair: on ONE pair of a 7.5 Mbp chromosome it would be a deep pair of its own)
            sizes[: n - int(sizes.sum())] += 1
        else:            # (S-pan as every round measured it)
            sizes[0] += n - int(sizes.sum())
        pair = torch.repeat_interleave(torch.arange(P, device=device, dtype=torch.int32), sizes)
        c = pair % chroms
        pair = torch.div(pair, chroms, rounding_mode="floor")
        q = torch.div(pair, n_genomes - 1, rounding_mode="floor")
        t = pair - q * (n_genomes - 1)
        t = t + (t >= q).to(torch.int32)
        if chroms > 1:
            q, t = q * chroms + c, t * chroms + c
            chr_len = chr_len // chroms
        del pair, c
    ln = torch.exp(7.6009 + 1.2 * torch.randn(n, generator=g, device=device)).clamp_(100, 500_000).to(torch.int32)
    ln = torch.minimum(ln, torch.tensor(max(chr_len // 2, 100), dtype=torch.int32, device=device))
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
    table = torch.div(torch.arange(n_genomes * chroms, device=device, dtype=torch.int32), chroms, rounding_mode="floor")
    cols = dict(q_id=q.contiguous(), t_id=t.contiguous(), q_start=qs, q_end=qs + ln, t_start=ts, t_end=ts + ln,
                identity=identity.contiguous(), matches=matches, block_len=block.contiguous(), strand=strand,
                seq_genome_last=table, seq_genome_two=table.clone())
    if device.type == "cuda":
        torch.cuda.synchronize(device)   # the library works on its own stream: the columns must be complete before it reads them
    return cols, sizes


def span_names(n_genomes):
    return [f"g{i:03d}#1#chr1" for i in range(n_genomes)]


SBIG1_NAMES = ["hgA#1#chr1", "hgB#1#chr1"]


def make_records(lib_mod, cols, n, n_genomes):
    r = lib_mod.SwgRecords()
    r.n = n
    for k in REC_COLS:
        setattr(r, k, cols[k].data_ptr())
    r.n_seq = int(cols["seq_genome_last"].numel())   # (n_genomes sequences unless the genomes have several chromosomes)
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
    if pipeline == "c5":      # --scaffold-filter 1:1 --scaffold-dist 20000, mappings many:many (SURVEY.md 8d C5 = configs[4])
        return sw.FilterConfig(scaffold_filter_mode=sw.FilterMode.OneToOne, scaffold_gap=50_000, min_scaffold_length=10_000,
                               scaffold_max_deviation=20_000)
    if pipeline == "default":  # all CLI defaults (many:many, jump 50k, mass 10k)
        return sw.FilterConfig()
    if pipeline == "k32":      # --num-mappings 3:2 --scaffold-jump 0: the 2 <= k < inf tile kernel (--pipeline k32 --only)
        return sw.FilterConfig(mapping_filter_mode=sw.FilterMode.OneToMany, mapping_max_per_query=3, mapping_max_per_target=2,
                               scaffold_gap=0)
    raise SystemExit(f"unknown pipeline {pipeline}")


# ---- CPU side (the oracle = port of the reference; checker and cpu_baseline only) ---------------------------------
def _host_cols(cols, lo, hi):
    import numpy as np
    return {k: np.ascontiguousarray(cols[k][lo:hi].cpu().numpy()) for k in REC_COLS}


def _oracle_records(cols, lo, hi, names):
    import numpy as np
    from tests import orc
    h = _host_cols(cols, lo, hi)
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


def cpu_baseline_all_cores(cols, sizes, cfg, names, first_group, per_thread=150_000, total_cap=16_000_000):
    """'What a group-parallel CPU filter would give' (SURVEY.md 8d-ii): the same oracle on ALL host threads at once
    (os.cpu_count(), stated in `cores`), every thread on its own run of whole genome-pair groups (the reference itself
    filters on one thread).  Wall time of the whole threaded run, best of 2; the oracle is a String-keyed port like the
    reference (names are cloned and hashed per record), so this includes the allocator contention such a port has."""
    import numpy as np
    from tests import orc
    csum = np.concatenate([[0], sizes.cumsum(0).cpu().numpy()]).astype(np.int64)
    T = max(1, os.cpu_count() or 1)
    per_thread = max(20_000, min(per_thread, total_cap // T))
    bounds, g = [int(csum[first_group])], first_group
    for _ in range(T):
        g2 = int(np.searchsorted(csum, csum[g] + per_thread)) if g < len(csum) - 1 else g
        g2 = min(max(g2, g + 1), len(csum) - 1)
        if g2 <= g:
            break
        bounds.append(int(csum[g2]))
        g = g2
    if len(bounds) < 2:
        return None
    lo, hi = bounds[0], bounds[-1]
    h = _host_cols(cols, lo, hi)
    rel = np.asarray(bounds, dtype=np.int64) - lo
    ocfg = _oracle_config(cfg)
    walls = []
    for _ in range(2):
        _, _, wall = orc.apply_filters_by_groups(ocfg, h, names, rel, len(rel) - 1)
        walls.append(wall)
    m = hi - lo
    return dict(value=m / min(walls), unit="mappings/s", cores=len(rel) - 1, kind="port",
                sample=f"{len(rel) - 1} threads (= all {T} host threads) x ~{per_thread} mappings of whole groups, String-keyed port, "
                       f"wall {min(walls):.2f} s (best of 2: {walls[0]:.2f}, {walls[1]:.2f})")


def full_parity(cols, sizes, cfg, names, status_dev, chain_dev, target, max_threads=64, groups_per_job=25):
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
    h = _host_cols(cols, 0, m)
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


def cpu_baseline(cols, sizes, cfg, sample_target, status_dev, chain_dev, names):
    """The CPU oracle (port of the reference, 1 thread like the reference's filter) timed on a bounded
    sample: the first whole genome-pair groups of this rank's shard.  Also a parity check."""
    import numpy as np
    from tests import orc
    csum = sizes.cumsum(0).cpu().numpy()
    g = int(np.searchsorted(csum, sample_target)) + 1
    m = int(csum[min(g, len(csum)) - 1])
    rec = _oracle_records(cols, 0, m, names)
    ocfg = _oracle_config(cfg)
    ost, och, secs = orc.apply_filters(ocfg, rec, want_seconds=True)
    gst = status_dev[:m].cpu().numpy()
    parity = bool(np.array_equal(gst, ost))
    if parity and cfg.scaffold_gap:
        # chain numbers are global to a call; compare the partition they induce on the sample
        parity = bool(orc.same_chain_partition(chain_dev[:m].cpu().numpy(), och))
    return dict(value=m / secs, unit="mappings/s", cores=1, kind="port",
                sample=f"first {min(g, len(csum))} genome-pair groups of rank 0's shard = {m} mappings, oracle apply_filters {secs:.2f} s"), parity


def end_to_end(n_lines, ref_lines, threads):
    """PAF file -> PAF file through the C++ host (native ingest, swg_filter, native egress), default flags, next to
    the oracle's CLI on a prefix of the same file; outputs compared byte for byte on that prefix."""
    import hashlib
    import re
    import shutil
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
            # lines grouped by query genome, the way an aligner writes its output query by query (rounds 1-3 drew the query
            # genome of every line at random, which no aligner does and which keeps swg_filter from streaming its upload)
            subprocess.check_call([_build.SYNTH, str(n_lines), "100", "2025", "150000000", "query"], stdout=f)
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
                      r"d2h ([\d.]+)\), write ([\d.]+) ms(?: \| device start-up ([\d.]+) ms beside the read \(create ([\d.]+), "
                      r"warm-up ([\d.]+)\), ([\d.]+) ms waited for)?", r.stderr)
        phases = {k: float(v) for k, v in zip(("read_ms", "load_ms", "parse_ms", "filter_ms", "device_ms", "h2d_ms", "d2h_ms", "write_ms",
                                                "device_startup_ms", "create_ms", "warmup_ms", "startup_wait_ms"), m.groups())
                  if v is not None} if m else None
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


# ---- timing ----------------------------------------------------------------------------------------------------
class Runner:
    """One context, one record set in HBM; times K calls of swg_filter_device per flag set."""

    def __init__(self, torch, sw, lib_mod, ctx, device, dist, cols, n, n_seq, coll_device=None):
        self.torch, self.sw, self.lib_mod, self.ctx, self.device, self.dist = torch, sw, lib_mod, ctx, device, dist
        self.coll_device = coll_device or device   # where collective buffers live (the CPU under --rehearse: gloo)
        self.cols, self.n = cols, n
        self.rec = make_records(lib_mod, cols, n, n_seq)
        self.status = torch.zeros(max(n, 1), dtype=torch.uint8, device=device)
        self.chain = torch.zeros(max(n, 1), dtype=torch.int32, device=device)
        self.stats = lib_mod.SwgStats()

    def step(self, ccfg, with_stats=False):
        self.ctx.check(self.ctx.lib.swg_filter_device(self.ctx.handle, C.byref(self.rec), C.byref(ccfg), self.status.data_ptr(),
                                                      self.chain.data_ptr(), C.byref(self.stats) if with_stats else None))

    def barrier(self):
        self.torch.cuda.synchronize()
        self.ctx.synchronize()
        if self.dist is not None:
            self.dist.barrier()

    def time(self, pipeline, steps, warmup, keep_results=False):
        """W untimed steps, barrier + synchronize, exactly K timed steps, synchronize, MAX over ranks."""
        torch, ctx = self.torch, self.ctx
        cfg = make_config(self.sw, pipeline)
        ccfg = cfg.to_c()
        for _ in range(warmup):
            self.step(ccfg)

        def timed(k):
            """barrier + synchronize, exactly k steps, synchronize, MAX over ranks -> (seconds, this rank's seconds)"""
            self.barrier()
            t0 = time.perf_counter()
            for _ in range(k):
                self.step(ccfg)
            ctx.synchronize()
            torch.cuda.synchronize()
            local = time.perf_counter() - t0
            allr = local
            if self.dist is not None:
                tt = torch.tensor([local], dtype=torch.float64, device=self.coll_device)
                self.dist.all_reduce(tt, op=self.dist.ReduceOp.MAX)
                allr = float(tt.item())
            return allr, local

        # pass A (not the headline): HIP events around EVERY launch on the library's stream -> the per-kernel table, and which
        # kernel is the dominant one.  ~200 event records per call cost about 1 ms of a ~100-launch pipeline.
        ctx.profile_reset()
        ctx.profile_select(None)
        ctx.profile(True)
        all_events, _ = timed(steps)
        ctx.profile(False)
        prof = ctx.profile_table()
        prof_units = ctx.profile_units()
        dom = max(prof.items(), key=lambda kv: kv[1][1])[0] if prof else None
        # pass B = THE TIMED REGION: the same K steps with HIP events around the dominant kernel's launches only (on the stream
        # it is launched on): its live average duration for `roofline`, and a step time that is not inflated by the other events
        ctx.profile_reset()
        ctx.profile_select(dom)
        ctx.profile(True)
        elapsed, local_elapsed = timed(steps)
        ctx.profile(False)
        dom_live = ctx.profile_table().get(dom) if dom else None
        ctx.profile_select(None)
        # pass C: no events at all -- what a caller gets
        plain, _ = timed(steps)
        self.step(ccfg, with_stats=True)  # untimed: counts for the report
        ctx.synchronize()
        s = self.stats
        out = {"cfg": cfg, "elapsed": elapsed, "local_elapsed": local_elapsed, "ms_per_step": elapsed / steps * 1e3,
               "ms_per_step_all_events": all_events / steps * 1e3, "ms_per_step_unprofiled": plain / steps * 1e3,
               "prof": prof, "prof_units": prof_units, "dom": dom, "dom_live": dom_live,
               "counts": {"in": s.n_in, "retained": s.n_retained, "swept": s.n_swept, "chains": s.n_chains,
                          "chains_kept": s.n_chains_kept, "out": s.n_out, "device_ms_last_step": s.device_ms}}
        if keep_results:
            out["status"], out["chain"] = self.status[:self.n].clone(), self.chain[:self.n].clone()
        return out


_LIB_SHA = []


def LIB_SHA256():
    """sha256 of the shared library this process runs (tools/pmc_traffic.py stamps the traffic JSON with the same digest)."""
    if not _LIB_SHA:
        import hashlib
        p = os.path.join(ROOT, "sweepga_amd", "libsweepga_gpu.so")
        _LIB_SHA.append(hashlib.sha256(open(p, "rb").read()).hexdigest() if os.path.exists(p) else None)
    return _LIB_SHA[0]


def roofline(pipeline, n, steps, t, wl="100m"):
    """Dominant kernel of one flag set = the launch label with the most time per step.  One label is one kernel function
    (one rocprof name; tools/pmc_traffic.py maps names to the same labels), so every figure below averages over the SAME
    launches: `kernel_avg_ms` = HIP-event time of that label / its launches, `traffic` = that kernel's rocprofv3 PMC bytes
    per launch (profiles/<tag>_hbm_traffic_*.json, same label, same launches per call -- checked, else null).
    `achieved` follows the contract: ALGORITHMIC bytes of one call (33 or 47 B per mapping x the mappings one
    launch works on) / kernel_avg_ms.  `kernel_own_*` = traffic / kernel_avg_ms; `pipeline_*` the end-to-end figure."""
    algo = ALGO_BYTES_SWEEP if pipeline in ("sweep", "k32") else ALGO_BYTES_FULL
    prof = t["prof"]
    total_kernel_ms = sum(ms for _, ms in prof.values())
    dom_name, (dom_launches, dom_ms) = max(prof.items(), key=lambda kv: kv[1][1]) if prof else ("none", (1, float("nan")))
    dom_avg_ms_all_events = dom_ms / max(dom_launches, 1)   # (pass A: events around every launch)
    if t.get("dom_live") and t.get("dom") == dom_name:      # the timed region's own measurement (events around this kernel only)
        dom_launches, dom_ms = t["dom_live"]
    dom_avg_ms = dom_ms / max(dom_launches, 1)
    per_step = dom_launches / steps
    # units one launch works on: n for the per-record kernels; the sort passes run on sub-problems of different sizes (the
    # library counts the pairs of every pass), so their average launch is charged the average number of pairs
    units = t.get("prof_units", {}).get(dom_name, 0) / max(prof[dom_name][0] if prof else 1, 1) or n
    achieved = algo * units / (dom_avg_ms * 1e-3) / 1e9
    traffic, tnote, total_traffic, tfile = None, None, None, f"profiles/{PROFILE_TAG}_hbm_traffic_{pipeline}_{wl}.json"
    try:
        tj = json.load(open(os.path.join(ROOT, tfile)))
        if tj.get("lib_sha256") != LIB_SHA256():
            # counters of another build of the library say nothing about this one's launches: no traffic figure at all
            tnote = f"{tfile} was collected from library {str(tj.get('lib_sha256'))[:12]}, this run loads {str(LIB_SHA256())[:12]}"
            tj = {"kernels": {}}
        elif tj.get("n_mappings") != n:
            tnote = f"{tfile} is for {tj.get('n_mappings')} mappings"
        elif dom_name not in tj["kernels"]:
            tnote = tnote or f"{dom_name} not in {tfile}"
        else:
            k = tj["kernels"][dom_name]
            lpc = k.get("launches_per_call", k["launches_profiled"] / tj.get("calls_profiled", 3))
            if abs(lpc - per_step) > 1e-6:
                tnote = f"{tfile} has {lpc:g} launches of {dom_name} per call, this run {per_step:g}"
            else:
                traffic = k["hbm_bytes_per_launch"]
        total_traffic = tj.get("hbm_bytes_per_call_all_kernels")
    except (OSError, ValueError, KeyError) as e:
        tnote = f"{tfile}: {type(e).__name__}"
    pipe_achieved = algo * n / (t["ms_per_step"] * 1e-3) / 1e9
    own = traffic / (dom_avg_ms * 1e-3) / 1e9 if traffic else None
    return {"bound": "hbm", "kernel": dom_name, "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic, "traffic_note": tnote,
            "traffic_unit": f"HBM bytes per launch of that kernel (rocprofv3 FETCH_SIZE x2 + WRITE_SIZE, {tfile})",
            "kernel_avg_ms": dom_avg_ms, "kernel_avg_ms_all_events": dom_avg_ms_all_events, "kernel_launches_per_step": per_step,
            "algorithmic_bytes_per_mapping": algo, "units_per_launch": units,
            "kernel_own_achieved": own, "kernel_own_frac": own / HBM_PEAK_GBPS if own else None,
            "pipeline_achieved": pipe_achieved, "pipeline_frac": pipe_achieved / HBM_PEAK_GBPS,
            "pipeline_traffic": total_traffic, "pipeline_traffic_over_algorithmic": total_traffic / (algo * n) if total_traffic else None,
            "kernel_ms_per_step": total_kernel_ms / steps}


def kernels_table(t, steps):
    return {k: round(v[1] / steps, 4) for k, v in sorted(t["prof"].items(), key=lambda kv: -kv[1][1])}


# ---- configs[2]: S-big1 --------------------------------------------------------------------------------------------
def sbig1_leg(torch, sw, lib_mod, ctx, device, args):
    """BASELINE.json configs[2]: 10^7 mappings in ONE (query, target) pair of 248,956,422-bp chromosomes (seed 1234).
    Timed at full size for the three flag sets.  Parity inside the bench run is bounded: the oracle needs ~2 min for
    the full 10^7 sweep and hours for the scaffold flags at this depth (its chaining scan is O(n x window)), so the bench
    checks S-big1-shaped instances of the SAME depth (chromosome length scaled with n); the full-size record-for-record
    check of the sweep flags lives in tests/test_gpu_sbig1.py."""
    import threading
    import numpy as np
    from tests import orc
    n = args.sbig1
    cols, sizes = gen_shard(torch, n, 2, 1234, device, chr_len=SBIG1_LEN, single_pair=True)
    run = Runner(torch, sw, lib_mod, ctx, device, None, cols, n, 2)
    steps, warm = max(1, min(args.steps, 5)), 1
    out = {"workload": f"BASELINE.json configs[2] (S-big1): {n} mappings in one pair {SBIG1_NAMES[0]} -> {SBIG1_NAMES[1]}, "
                       f"{SBIG1_LEN} bp, seed 1234", "steps": steps, "warmup": warm, "pipelines": {}}
    for p in ("sweep", "default", "full"):
        t = run.time(p, steps, warm)
        out["pipelines"][p] = {"flags": FLAGS[p], "ms_per_step": t["ms_per_step"], "ms_per_step_all_events": t["ms_per_step_all_events"],
                               "ms_per_step_unprofiled": t["ms_per_step_unprofiled"],
                               "value": n / (t["ms_per_step"] * 1e-3),
                               "unit": "mappings/s", "counts": t["counts"], "roofline": roofline(p, n, steps, t, "sbig1_10m"),
                               "kernels_ms_per_step": kernels_table(t, steps)}
    del run, cols
    if args.cpu_sample > 0:
        jobs = [("sweep", args.sbig1_parity_sweep), ("default", args.sbig1_parity_scaffold), ("full", args.sbig1_parity_scaffold)]
        res, threads = {}, []

        def check(p, m):
            c, _ = gen_shard(torch, m, 2, 1234, device, chr_len=max(int(SBIG1_LEN * (m / 1e7)), 1_000_000), single_pair=True)
            torch.cuda.synchronize()   # the library runs on its own stream: the generator's kernels must have finished
            r = Runner(torch, sw, lib_mod, sw.Context(device.index or 0), device, None, c, m, 2)
            cfg = make_config(sw, p)
            r.step(cfg.to_c())
            r.ctx.synchronize()
            st, ch = r.status[:m].cpu().numpy(), r.chain[:m].cpu().numpy()
            rec = _oracle_records(c, 0, m, SBIG1_NAMES)
            ost, och, secs = orc.apply_filters(_oracle_config(cfg), rec, want_seconds=True)
            res[p] = {"mappings_checked": m, "same_depth_as_full_size": True, "status_equal": bool(np.array_equal(st, ost)),
                      "chain_equal": bool(np.array_equal(ch, och)) if cfg.scaffold_gap else None,
                      "oracle_s": secs, "cpu_baseline": {"value": m / secs, "unit": "mappings/s", "cores": 1, "kind": "port",
                                                         "sample": f"{m} S-big1-shaped mappings (one pair, depth as at 10^7)"}}
        for p, m in jobs:
            if m > 0:
                th = threading.Thread(target=check, args=(p, m))
                th.start()
                threads.append(th)
        for th in threads:
            th.join()
        for p in res:
            out["pipelines"][p]["parity"] = res[p]
    return out


# ---- strong scaling ------------------------------------------------------------------------------------------------
def _result_checksum(np, idx, status, chain):
    """Order-independent 64-bit fingerprint of (global record index, status, chain number) triples: shards add up to the
    fingerprint of the whole record set, so a sharded run can be compared with the unsharded one."""
    with np.errstate(over="ignore"):
        i = idx.astype(np.uint64) + np.uint64(1)
        h = (i * np.uint64(0x9E3779B97F4A7C15)) ^ (chain.astype(np.uint64) * np.uint64(0xC2B2AE3D27D4EB4F) + status.astype(np.uint64))
        h ^= h >> np.uint64(29)
        h *= np.uint64(0xBF58476D1CE4E5B9)
        return int(np.add.reduce(h, dtype=np.uint64))


def strong_scaling(torch, sw, lib_mod, ctx, device, dist, args, rank, world, coll_device=None):
    """ONE S-pan record set (the same on every rank, seed --seed), genome pairs bin-packed over the ranks by
    sweepga_amd.shard (LPT by mapping count), every rank keeps only its shard in HBM and filters it; the kept-chain
    ranges of all pairs are exchanged with one all_gather and every rank renumbers its own chains."""
    import numpy as np
    from sweepga_amd import shard
    n, G = args.mappings, args.genomes
    cols, _ = gen_shard(torch, n, G, args.seed, device)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    q_h, t_h = cols["q_id"].cpu().numpy(), cols["t_id"].cpu().numpy()
    table = np.arange(G, dtype=np.uint32)
    shard_of_key, counts = shard.plan_dense(q_h, t_h, table, G, world)
    plan_s = time.perf_counter() - t0
    loads = np.bincount(shard_of_key[counts > 0], weights=counts[counts > 0], minlength=world)
    t0 = time.perf_counter()
    key_d = cols["q_id"].to(torch.int64) * G + cols["t_id"].to(torch.int64)
    mine = torch.nonzero(torch.as_tensor(shard_of_key, device=device)[key_d] == rank).flatten()
    scols = {k: cols[k][mine].contiguous() for k in REC_COLS}
    scols["seq_genome_last"], scols["seq_genome_two"] = cols["seq_genome_last"], cols["seq_genome_two"]
    pair_mine = key_d[mine]
    m = int(mine.numel())
    del cols, key_d
    torch.cuda.synchronize()
    partition_s = time.perf_counter() - t0
    coll_device = coll_device or device
    run = Runner(torch, sw, lib_mod, ctx, device, dist, scols, m, G, coll_device)
    res = {}
    idx_h = mine.cpu().numpy()
    for p in PIPELINES:
        t = run.time(p, args.steps, args.warmup, keep_results=True)
        lt = torch.tensor([t["local_elapsed"] / args.steps * 1e3], dtype=torch.float64, device=coll_device)
        per_rank = [torch.zeros_like(lt) for _ in range(world)]
        if dist is not None:
            dist.all_gather(per_rank, lt)
        else:
            per_rank = [lt]
        per_rank_ms = [float(x.item()) for x in per_rank]
        # exchange + local renumbering (chain numbers are global in the reference, src/paf_filter.rs:517-521)
        renumber_s = None
        ch = t["chain"].cpu().numpy().astype(np.int64)
        if t["cfg"].scaffold_gap:
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            pr = pair_mine.cpu().numpy()
            lo, hi, first = shard.pair_chain_ranges(ch, pr, idx_h, G * G, n)
            if dist is not None:
                buf = torch.as_tensor(np.stack([lo, -hi, first]), device=coll_device)
                dist.all_reduce(buf, op=dist.ReduceOp.MIN)   # every pair lives on exactly one rank
                lo, hi, first = buf[0].cpu().numpy(), -buf[1].cpu().numpy(), buf[2].cpu().numpy()
            shift = shard.chain_shifts(lo, hi, first)
            has = ch != 0
            ch[has] += shift[pr[has]]
            renumber_s = time.perf_counter() - t1
        # fingerprint of (record, status, global chain number) over all shards: equal to the unsharded run's (tests compare)
        chk = _result_checksum(np, idx_h, t["status"].cpu().numpy(), ch)
        if dist is not None:
            cb = torch.tensor([np.int64(np.uint64(chk).astype(np.int64))], dtype=torch.int64, device=coll_device)
            dist.all_reduce(cb, op=dist.ReduceOp.SUM)   # wraps modulo 2^64 like the per-shard sums
            chk = int(cb.item()) & 0xFFFFFFFFFFFFFFFF
        res[p] = {"flags": FLAGS[p], "ms_per_step": t["ms_per_step"], "value": n / (t["ms_per_step"] * 1e-3), "unit": "mappings/s",
                  "per_rank_ms": per_rank_ms, "renumber_s": renumber_s, "counts_rank0": t["counts"], "result_checksum": f"{chk:016x}"}
    return {"mappings_total": n, "plan_s": plan_s, "partition_s": partition_s, "shard_mappings_rank0": m,
            "load_max_over_mean": float(loads.max() / loads.mean()), "loads": [int(x) for x in loads], "pipelines": res}


MAX_LINE_BYTES = 4096   # the driver keeps an 8 KB tail of stdout: the ONE line it parses must stay far below that


def _r(x, nd=4):
    return round(x, nd) if isinstance(x, float) else x


def shapes_leg(torch, sw, lib_mod, ctx, device, n, seed, steps=3, parity_genomes=5):
    """The default flags on record sets away from the bench workload's order and shape: S-pan's records
    shuffled; by query sequence in query order with the targets interleaved (what wfmash writes); 100 genomes x 20 chromosomes
    (198,000 homologous chromosome pairs per 10^8 records, pair-major).  One time per shape, and parity against the oracle on
    the records of a few whole genome pairs (a genome pair's answers do not depend on the rest of the input; chain numbers are
    compared as a partition)."""
    import numpy as np
    from tests import orc
    cfg = make_config(sw, "default")
    ccfg = cfg.to_c()
    ocfg = _oracle_config(cfg)
    out = {}

    def same_partition(a, b):
        """chain numbers a, b (int64 tensors, 0 = none) name the same partition of the records"""
        if not bool(((a == 0) == (b == 0)).all()):
            return False
        k = a != 0
        a, b = a[k], b[k]
        pairs = torch.unique(a * (1 << 32) + b).numel()
        return pairs == torch.unique(a).numel() == torch.unique(b).numel()

    def run(tag, cols, G, names, genome_of_seq, grouped_check=False, parity_genomes=parity_genomes):
        rec = make_records(lib_mod, cols, n, G)
        status = torch.zeros(n, dtype=torch.uint8, device=device)
        chain = torch.zeros(n, dtype=torch.int32, device=device)
        ctx.profile_reset()
        best = None
        for it in range(steps + 1):
            torch.cuda.synchronize()
            if it == steps:
                ctx.profile(True)
            t0 = time.perf_counter()
            ctx.check(ctx.lib.swg_filter_device(ctx.handle, C.byref(rec), C.byref(ccfg), status.data_ptr(), chain.data_ptr(), None))
            ctx.synchronize()
            dt = time.perf_counter() - t0
            if 0 < it < steps:
                best = dt if best is None else min(best, dt)
        ctx.profile(False)
        table = ctx.profile_table()
        path = "pair-resident" if "pair_renumber" in table and not any(k in table for k in ("chain_cuts", "cuts_from_scan", "sortA_keys", "sortA_keys_hist", "sortA_words")) else "global sorts"
        # parity: every record between the first `parity_genomes` genomes (20 ordered genome pairs), in the order they have in this input
        gq = genome_of_seq[cols["q_id"].long()]
        gt = genome_of_seq[cols["t_id"].long()]
        sel = (gq < parity_genomes) & (gt < parity_genomes) & (gq != gt)
        idx = torch.nonzero(sel, as_tuple=False).flatten()
        h = {k: np.ascontiguousarray(cols[k][idx].cpu().numpy()) for k in REC_COLS}
        u = lambda a: np.ascontiguousarray(a.astype(np.uint64))   # noqa: E731
        orec = orc.Records([names[i] for i in h["q_id"]], [names[i] for i in h["t_id"]], u(h["q_start"]), u(h["q_end"]), u(h["t_start"]),
                           u(h["t_end"]), u(h["block_len"]), np.ascontiguousarray(h["identity"]), u(h["matches"]),
                           np.where(h["strand"] == 0, ord("+"), ord("-")).astype(np.uint8), u(np.arange(len(idx))))
        ost, och = orc.apply_filters(ocfg, orec)
        gst, gch = status[idx].cpu().numpy(), chain[idx].cpu().numpy()
        ok = bool(np.array_equal(gst, ost)) and bool(orc.same_chain_partition(gch, och))
        top = sorted(table.items(), key=lambda kv: -kv[1][1])[:6]
        out[tag] = {"ms_per_step": best * 1e3, "path": path, "parity": {"mappings_checked": int(len(idx)), "ok": ok},
                    "kernels_ms": {k: round(v[1], 3) for k, v in top}}
        if grouped_check:
            # every record, on the device: the same records brought pair-major by a STABLE sort (ties between records are broken
            # by input order, src/plane_sweep_exact.rs: the relative order inside a pair -- with single-chromosome genomes a
            # sweep segment is one pair -- must stay), filtered through the pair-resident path that bench.py checks against
            # the oracle on tens of millions of records; every status equal, the chain numbers the same partition
            key = cols["q_id"].to(torch.int64) * (1 << 32) + cols["t_id"].to(torch.int64)
            order = torch.argsort(key, stable=True)
            del key
            g_cols = {k: (cols[k][order].contiguous() if k in REC_COLS else cols[k]) for k in cols}
            g_rec = make_records(lib_mod, g_cols, n, G)
            st_g = torch.zeros(n, dtype=torch.uint8, device=device)
            ch_g = torch.zeros(n, dtype=torch.int32, device=device)
            torch.cuda.synchronize()   # (the library works on its own stream: the gathered columns must be complete before it reads them)
            ctx.check(ctx.lib.swg_filter_device(ctx.handle, C.byref(g_rec), C.byref(ccfg), st_g.data_ptr(), ch_g.data_ptr(), None))
            ctx.synchronize()
            same = bool((status[order] == st_g).all()) and same_partition(chain[order].to(torch.int64), ch_g.to(torch.int64))
            out[tag]["parity"]["same_as_grouped_at_full_size"] = same
            out[tag]["parity"]["ok"] = ok and same
            del g_cols, g_rec, st_g, ch_g, order
        del status, chain

    cols, _ = gen_shard(torch, n, 100, seed, device)
    table1 = cols["seq_genome_last"].long()
    names1 = span_names(100)
    key = cols["q_id"].to(torch.int64) * (1 << 32) + cols["q_start"].to(torch.int64)
    order = torch.argsort(key, stable=True)
    del key
    by_q = {k: (cols[k][order].contiguous() if k in REC_COLS else cols[k]) for k in cols}
    del order
    run("by_query", by_q, 100, names1, table1, grouped_check=True)
    del by_q
    perm = torch.randperm(n, device=device)
    shuf = {k: (cols[k][perm].contiguous() if k in REC_COLS else cols[k]) for k in cols}
    del cols, perm
    run("shuffled", shuf, 100, names1, table1, grouped_check=True)
    del shuf
    torch.cuda.empty_cache()
    cols, _ = gen_shard(torch, n, 100, seed, device, chroms=20)
    names20 = [f"g{i // 20:03d}#1#chr{i % 20 + 1}" for i in range(100 * 20)]
    run("multichrom", cols, 100, names20, cols["seq_genome_last"].long(), parity_genomes=12)
    del cols
    torch.cuda.empty_cache()
    return out


def summary_line(out, detail_path):
    """The one stdout line (< 4 KB): contract keys, the headline's roofline and CPU baselines, and ONE scalar per other leg.
    Per-kernel tables, counts, notes and samples are in the detail file (`--detail`, default gpurun_out/bench_detail.json)."""
    keep = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "ms_per_step_all_events", "ms_per_step_unprofiled", "higher_is_better",
            "scaling", "vs_baseline", "dtype", "data")
    line = {k: _r(out.get(k)) for k in keep}
    cfg = out.get("config") or {}
    line["config"] = {k: cfg[k] for k in ("workload", "flags", "mappings_per_gpu", "groups_per_gpu") if k in cfg}
    rf = out.get("roofline")
    if rf:
        line["roofline"] = {k: _r(rf.get(k), 5) for k in
                            ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "kernel_avg_ms", "kernel_launches_per_step",
                             "units_per_launch", "algorithmic_bytes_per_mapping", "kernel_own_frac", "pipeline_frac", "kernel_ms_per_step")}
    for k in ("cpu_baseline", "cpu_baseline_all_cores"):
        cb = out.get(k)
        if cb:
            line[k] = {"value": _r(cb["value"], 1), "unit": cb["unit"], "cores": cb["cores"], "kind": cb["kind"], "sample": cb["sample"][:160]}
    pipes = out.get("pipelines") or {}
    par = {}
    for p, e in pipes.items():
        if p != (cfg.get("pipeline") or "default"):
            line[f"{p}_ms_per_step"] = _r(e["ms_per_step"])
            line[f"{p}_pipeline_frac"] = _r(e["roofline"]["pipeline_frac"], 5)
        pa = e.get("parity_all_threads")
        if pa:
            par[f"span_{p}"] = {"checked": pa["mappings_checked"], "ok": bool(pa["status_equal"] and pa["chain_partition_equal"] is not False)}
        elif e.get("parity_vs_oracle_on_sample") is not None:
            par[f"span_{p}"] = {"checked": "sample", "ok": bool(e["parity_vs_oracle_on_sample"])}
    sb = out.get("sbig1")
    if sb:
        for p, e in sb["pipelines"].items():
            line[f"sbig1_{p}_ms"] = _r(e["ms_per_step"])
            pa = e.get("parity")
            if pa:
                par[f"sbig1_{p}"] = {"checked": pa["mappings_checked"], "ok": bool(pa["status_equal"] and pa["chain_equal"] is not False)}
    sh = out.get("shapes")
    if sh:
        for tag, e in sh.items():   # shuffled_default_ms, by_query_default_ms, multichrom_default_ms
            line[f"{tag}_default_ms"] = _r(e["ms_per_step"])
            par[f"{tag}_default"] = {"checked": e["parity"]["mappings_checked"], "ok": bool(e["parity"]["ok"])}
            if "same_as_grouped_at_full_size" in e["parity"]:   # every record against the answers for the stably grouped input
                par[f"{tag}_default"]["same_as_grouped"] = bool(e["parity"]["same_as_grouped_at_full_size"])
    pc = out.get("pcie_inclusive")
    if pc:
        for p, e in pc.items():
            line[f"pcie_{p}_ms"] = _r(e["ms"], 2)
    ee = out.get("end_to_end")
    if ee and "wall_s" in ee:
        line["e2e_lines"], line["e2e_wall_s"] = ee["lines"], _r(ee["wall_s"])
        line["e2e_byte_identical_on_prefix"] = ee.get("byte_identical_on_prefix")
        if ee.get("cpu_reference_cli"):
            line["e2e_cpu_cli_mappings_per_s"] = _r(ee["cpu_reference_cli"]["value"], 1)
    ss = out.get("strong_scaling")
    if ss:
        line["strong"] = {"mappings_total": ss["mappings_total"], "load_max_over_mean": _r(ss["load_max_over_mean"], 5),
                          "plan_s": _r(ss["plan_s"]), "partition_s": _r(ss["partition_s"]),
                          "ms_per_step": {p: _r(e["ms_per_step"]) for p, e in ss["pipelines"].items()},
                          "result_checksum": {p: e["result_checksum"] for p, e in ss["pipelines"].items()}}
    if out.get("rehearsal"):
        line["rehearsal"] = out["rehearsal"]
    if par:
        line["parity"] = par
        line["parity_ok"] = all(v["ok"] for v in par.values())
    line["detail"] = os.path.relpath(detail_path, ROOT) if detail_path else None
    return line


def fit_line(line):
    """The stdout line must stay under MAX_LINE_BYTES whatever was measured: optional keys are dropped (least important
    first) until it fits, and the line says so (`truncated`) -- a run never ends without its JSON line."""
    text = json.dumps(line)
    droppable = ("strong", "cpu_baseline_all_cores", "parity", "e2e_cpu_cli_mappings_per_s", "e2e_byte_identical_on_prefix",
                 "e2e_lines", "e2e_wall_s", "ms_per_step_unprofiled", "ms_per_step_all_events")
    dropped = []
    for k in droppable:
        if len(text) < MAX_LINE_BYTES:
            break
        if k in line:
            del line[k]
            dropped.append(k)
            line["truncated"] = dropped
            text = json.dumps(line)
    if len(text) >= MAX_LINE_BYTES:  # still too long: shorten the free-text fields
        for path in (("config", "workload"), ("cpu_baseline", "sample")):
            d = line
            for k in path[:-1]:
                d = d.get(k) or {}
            if isinstance(d.get(path[-1]), str):
                d[path[-1]] = d[path[-1]][:80]
        line["truncated"] = dropped + ["text"]
        text = json.dumps(line)
    if len(text) >= MAX_LINE_BYTES:  # last resort: the contract keys alone
        keep = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                "dtype", "data", "config", "roofline", "cpu_baseline", "parity_ok", "detail")
        line = {k: line[k] for k in keep if k in line}
        line["truncated"] = "all optional keys"
        text = json.dumps(line)
    return text


def spawn_ranks(args, argv):
    """`bench.py --gpus N` outside a launcher: start N ranks as fresh child processes (torch.distributed.run) BEFORE this
    process touches a GPU, and pass their exit code on."""
    import socket
    import torch
    have = torch.cuda.device_count()   # counting devices does not initialise the GPU
    if args.rehearse and have >= 1:
        pass   # rehearsal: ranks share the GPUs there are (rank r on device r mod `have`), gloo instead of RCCL
    elif have < args.gpus:
        print(f"bench.py: --gpus {args.gpus} but only {have} GPU(s) are visible", file=sys.stderr)
        return 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + argv
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--mappings", type=int, default=0, help="mappings per GPU (weak) / in total (strong); 0 = the workload's size")
    ap.add_argument("--genomes", type=int, default=100)
    ap.add_argument("--workload", default="span", choices=["span", "sbig1"],
                    help="span: configs[3] (the headline).  sbig1: configs[2] as the MAIN record set (profiling runs of the dense "
                         "case; --mappings defaults to 10^7, the S-big1 side leg is skipped)")
    ap.add_argument("--pipeline", default="default", choices=list(PIPELINES) + ["k32"], help="flag set reported as `value`")
    ap.add_argument("--only", action="store_true", help="time only --pipeline (profiling runs)")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"])
    ap.add_argument("--cpu-sample", type=int, default=5_000_000, help="mappings in the 1-thread CPU sample (0 = no CPU legs)")
    ap.add_argument("--seed", type=int, default=2025)
    ap.add_argument("--no-pcie", action="store_true", help="skip the swg_filter leg (host buffers in/out, PCIe included)")
    ap.add_argument("--shuffle", action="store_true",
                    help="random record order instead of group-major (no locality for the gathers; implies no CPU legs)")
    ap.add_argument("--parity-mappings", type=int, default=-1,
                    help="mappings of the timed workload checked against the oracle on all host threads "
                         "(-1 = auto: 2M (sweep) / 0.5M (scaffold flag sets) per host thread, up to the whole shard; 0 = skip)")
    ap.add_argument("--shapes", type=int, default=1, help="1: the default flags on other orders / shapes of the records (shapes_leg); 0 = skip")
    ap.add_argument("--sbig1", type=int, default=10_000_000, help="mappings of the S-big1 leg (configs[2]); 0 = skip")
    ap.add_argument("--sbig1-parity-sweep", type=int, default=1_000_000)
    ap.add_argument("--sbig1-parity-scaffold", type=int, default=200_000)
    ap.add_argument("--e2e", type=int, default=10_000_000, help="lines of synthetic PAF for the file->file leg (0 = skip)")
    ap.add_argument("--e2e-ref", type=int, default=1_000_000, help="prefix of that file the oracle CLI is timed on")
    ap.add_argument("--threads", type=int, default=0, help="host threads for the e2e leg (0 = all cores)")
    ap.add_argument("--rehearse", action="store_true",
                    help="run the N-rank launch path on fewer GPUs than ranks: rank r uses device r mod (visible GPUs) and the "
                         "process group is gloo (CPU buffers) instead of RCCL.  Exercises spawn -> torch.distributed.run -> "
                         "init_process_group -> barrier / MAX-over-ranks / the chain renumbering all_reduce; NOT a scaling number")
    ap.add_argument("--detail", default="", help="file for the full report (per-kernel tables, counts, every leg); "
                                                 "default gpurun_out/bench_detail.json.  stdout carries ONE line < 4 KB")
    args = ap.parse_args()

    launched = "RANK" in os.environ
    if args.gpus > 1 and not launched:
        sys.exit(spawn_ranks(args, sys.argv[1:]))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE is {world}; refusing to report a {world}-GPU number as "
              f"{args.gpus}-GPU", file=sys.stderr)
        sys.exit(2)

    # stdout carries the ONE JSON line and nothing else: native libraries write there too (gloo's "[Gloo] Rank 0 is connected
    # to ..." lines, RCCL with NCCL_DEBUG=INFO), so file descriptor 1 is pointed at stderr for the whole run and the line goes
    # to a private duplicate of the original stdout
    sys.stdout.flush()
    line_out = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)

    import torch
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: torch.cuda.is_available() is False (no CPU fallback)")
    dev_index = local_rank % max(torch.cuda.device_count(), 1) if args.rehearse else local_rank
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    coll_device = torch.device("cpu") if args.rehearse else device
    dist = None
    if launched:  # launched by torch.distributed.run: one rank per GPU over RCCL
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29513")
        if args.rehearse:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)

    import sweepga_amd as sw
    from sweepga_amd import _lib
    ctx = sw.Context(dev_index)
    if args.workload == "sbig1":
        args.genomes, args.sbig1 = 2, 0
        if (args.mappings or 10_000_000) > 2_000_000:
            args.cpu_sample = 0   # one group = one oracle thread: minutes at 10^7 (tests/test_gpu_sbig1.py does that check)
    n, G = args.mappings or (10_000_000 if args.workload == "sbig1" else 100_000_000), args.genomes
    args.mappings = n
    names = SBIG1_NAMES if args.workload == "sbig1" else span_names(G)
    order = [args.pipeline] + ([] if args.only else [p for p in PIPELINES if p != args.pipeline])
    out = None

    if args.scaling == "strong":
        ss = strong_scaling(torch, sw, _lib, ctx, device, dist, args, rank, world, coll_device)
        if rank == 0:
            head = ss["pipelines"][args.pipeline]
            out = {"metric": BASELINE_METRIC, "value": head["value"], "unit": "mappings/s", "n_gpus": world, "steps": args.steps,
                   "warmup": args.warmup, "ms_per_step": head["ms_per_step"], "higher_is_better": True, "scaling": "strong",
                   "vs_baseline": None, "dtype": "u32 coordinates, f64 scores", "data": "synthetic",
                   "config": {"workload": f"BASELINE.json configs[3]/[4]: ONE S-pan record set of {n} mappings over {G * (G - 1)} "
                                          f"genome-pair groups, sharded by genome pair over {world} GPU(s) (LPT by mapping count), "
                                          f"pipeline={args.pipeline}", "flags": FLAGS[args.pipeline]},
                   "strong_scaling": ss}
    else:
        if args.workload == "sbig1":
            cols, sizes = gen_shard(torch, n, 2, 1234 + 7919 * rank, device, chr_len=SBIG1_LEN, single_pair=True)
        else:
            cols, sizes = gen_shard(torch, n, G, args.seed + 7919 * rank, device)
        if args.shuffle:  # the CPU legs index whole groups by position, so they are skipped for a shuffled shard
            perm = torch.randperm(n, device=device)
            for k in REC_COLS:
                cols[k] = cols[k][perm].contiguous()
            del perm
            args.cpu_sample = 0
            args.parity_mappings = 0
        torch.cuda.synchronize()
        run = Runner(torch, sw, _lib, ctx, device, dist, cols, n, G, coll_device)
        cpu_legs = world == 1 and args.cpu_sample > 0   # the CPU legs belong to the N=1 run
        timed = {p: run.time(p, args.steps, args.warmup, keep_results=cpu_legs) for p in order}

        # Host buffers in and out (swg_filter), PCIe included.  N = 1: four legs.  N > 1: EVERY rank runs the headline flags'
        # leg on its own shard at the same time (barrier first), and the line carries the MAX over ranks -- the scaling curve
        # of the path a host binding takes (all GPUs' uploads share the host's memory system and PCIe root complexes), next
        # to the kernel-only one
        pcie = None
        if not args.no_pcie and (world > 1 or rank == 0):
            import numpy as np
            host = {k: v.cpu().numpy() for k, v in cols.items()}
            hrec = _lib.SwgRecords()
            hrec.n = n
            for k in REC_COLS + ("seq_genome_last", "seq_genome_two"):
                setattr(hrec, k, host[k].ctypes.data)
            hrec.n_seq = hrec.n_genome_last = hrec.n_genome_two = G
            hst = np.zeros(n, dtype=np.uint8)
            hch = np.zeros(n, dtype=np.uint32)
            hs = _lib.SwgStats()
            pcie = {}
            # <flags>: as a host binding calls it.  <flags>_derived: identity = NULL (= matches / max(block_len, 1), what the
            # ingest reports for a PAF without dv:f: tags; the synthetic records are of that kind) -- 8 B per record less to
            # upload.  <flags>_one_piece: SWG_STREAM=0, the unstreamed call (upload, filter, download one after the other).
            legs = [(args.pipeline, False, False), (args.pipeline, True, False), (args.pipeline, False, True), ("sweep", False, False)]
            if world > 1:
                legs = legs[:1]
            ident_ptr = hrec.identity
            for pname, derived, one_piece in dict.fromkeys(legs):
                ccfg = make_config(sw, pname).to_c()
                hrec.identity = None if derived else ident_ptr
                if one_piece:
                    os.environ["SWG_STREAM"] = "0"
                best = None
                for _ in range(3):
                    if dist is not None:
                        dist.barrier()   # all ranks start their calls together
                    t1 = time.perf_counter()
                    ctx.check(ctx.lib.swg_filter(ctx.handle, C.byref(hrec), C.byref(ccfg), hst.ctypes.data, hch.ctypes.data, C.byref(hs)))
                    dt = time.perf_counter() - t1
                    if dist is not None:   # a repetition counts as its slowest rank
                        tt = torch.tensor([dt], dtype=torch.float64, device=coll_device)
                        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
                        dt = float(tt.item())
                    if best is None or dt < best[0]:
                        best = (dt, hs.h2d_ms, hs.d2h_ms, hs.device_ms)
                os.environ.pop("SWG_STREAM", None)
                key = pname + ("_derived" if derived else "") + ("_one_piece" if one_piece else "")
                pcie[key] = {"value": n * world / best[0], "unit": "mappings/s", "ranks": world, "ms": best[0] * 1e3, "h2d_ms": best[1], "d2h_ms": best[2],
                             "device_ms": best[3], "flags": FLAGS[pname],
                             "note": "swg_filter: pageable host buffers in and out (what a host binding calls), best of 3; "
                                     "columns the flag set does not read are not transferred; ranges of whole query genomes are "
                                     "uploaded while their predecessors are filtered unless _one_piece"}
            hrec.identity = ident_ptr
            del host, hst, hch

        if rank == 0:
            pipes = {}
            for p in order:
                t = timed[p]
                e = {"flags": FLAGS[p], "ms_per_step": t["ms_per_step"], "ms_per_step_all_events": t["ms_per_step_all_events"],
                     "ms_per_step_unprofiled": t["ms_per_step_unprofiled"],
                     "value": n * world / (t["ms_per_step"] * 1e-3), "unit": "mappings/s", "steps": args.steps, "warmup": args.warmup, "counts": t["counts"],
                     "roofline": roofline(p, n, args.steps, t, "sbig1_10m" if args.workload == "sbig1" else "100m"),
                     "kernels_ms_per_step": kernels_table(t, args.steps)}
                if cpu_legs:
                    cb, par = cpu_baseline(cols, sizes, t["cfg"], args.cpu_sample if p == "sweep" else args.cpu_sample // 4,
                                           t["status"], t["chain"], names)
                    e["cpu_baseline"], e["parity_vs_oracle_on_sample"] = cb, par
                    e["cpu_baseline_all_cores"] = cpu_baseline_all_cores(cols, sizes, t["cfg"], names, 0)
                    pm = args.parity_mappings
                    if pm < 0:
                        per_thread = 2_000_000 if p == "sweep" else 500_000  # ~10 s of oracle time either way
                        pm = min(n, per_thread * min(os.cpu_count() or 1, 64))
                    e["parity_all_threads"] = full_parity(cols, sizes, t["cfg"], names, t["status"], t["chain"], pm) if pm > 0 else None
                    del t["status"], t["chain"]
                pipes[p] = e
            head = pipes[args.pipeline]
            out = {
                "metric": BASELINE_METRIC,
                "value": head["value"],
                "unit": "mappings/s",
                "n_gpus": world,
                "steps": args.steps,
                "warmup": args.warmup,
                "ms_per_step": head["ms_per_step"],
                "ms_per_step_all_events": head["ms_per_step_all_events"],
                "ms_per_step_unprofiled": head["ms_per_step_unprofiled"],
                "higher_is_better": True,
                "scaling": "weak",
                "vs_baseline": None,
                "dtype": "u32 coordinates, f64 scores",
                "data": "synthetic",
                "config": {"workload": (f"BASELINE.json configs[2] (S-big1): {n} mappings per GPU in one chromosome pair of {SBIG1_LEN} bp, "
                                        f"pipeline={args.pipeline}") if args.workload == "sbig1" else
                                       (f"BASELINE.json configs[3] (synthetic 100 M mappings across 10 k (q,t) groups, 100-genome pangenome "
                                        f"shape; S-pan in SURVEY.md 8d): {n} mappings per GPU over {G * (G - 1)} "
                                        f"genome-pair groups ({G} single-chromosome genomes), pipeline={args.pipeline}"),
                           "flags": FLAGS[args.pipeline], "pipeline": args.pipeline, "mappings_per_gpu": n, "groups_per_gpu": G * (G - 1)},
                "roofline": head["roofline"],
                "cpu_baseline": head.get("cpu_baseline"),
                "cpu_baseline_all_cores": head.get("cpu_baseline_all_cores"),
                "parity_vs_oracle_on_sample": head.get("parity_vs_oracle_on_sample"),
                "parity_all_threads": head.get("parity_all_threads"),
                "counts": head["counts"],
                "pipelines": pipes,
                "pcie_inclusive": pcie,
                "arena_bytes_per_mapping": {"capacity": ctx.memory_info()[0] / n, "peak_last_call": ctx.memory_info()[1] / n},
                "kernels_ms_per_step": head["kernels_ms_per_step"],
            }
        del run, cols
        if rank == 0 and world == 1:
            torch.cuda.empty_cache()
            out["shapes"] = shapes_leg(torch, sw, _lib, ctx, device, n, args.seed) if (args.shapes and args.workload == "span" and not args.shuffle and not args.only) else None
            out["sbig1"] = sbig1_leg(torch, sw, _lib, ctx, device, args) if args.sbig1 > 0 else None
            out["end_to_end"] = end_to_end(args.e2e, args.e2e_ref, args.threads) if args.e2e > 0 else None

    if rank == 0 and args.rehearse:
        out["rehearsal"] = (f"{world} rank(s) on {torch.cuda.device_count()} GPU(s), gloo process group: the launch path only, "
                            f"not a scaling measurement")
    if rank == 0:
        detail = args.detail or os.path.join(ROOT, "gpurun_out", "bench_detail.json")
        try:
            os.makedirs(os.path.dirname(os.path.abspath(detail)), exist_ok=True)
            with open(detail, "w") as f:
                json.dump(out, f)
        except OSError as e:
            print(f"bench.py: could not write {detail}: {e}", file=sys.stderr)
            detail = None
        print(fit_line(summary_line(out, detail)), file=line_out, flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
