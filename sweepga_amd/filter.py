"""Host-side mirror of the reference's filter interface over the C ABI (include/sweepga_gpu.h).

Names and argument meaning follow the reference so that tests read like its own:
  FilterConfig / FilterMode / ScoringFunction ..... src/paf_filter.rs:20-49, src/filter_types.rs
  PlaneSweepMapping, plane_sweep_query/target/both . src/plane_sweep_exact.rs:10-18, 268, 355, 436
  plane_sweep_scaffolds ............................ src/plane_sweep_scaffold.rs:47
  PafFilter.apply_filters / filter_paf ............. src/paf_filter.rs:278-289, 379-747
  SequenceIndex .................................... src/sequence_index.rs:7-31
All compute goes through libsweepga_gpu.so; nothing here evaluates the filter on the CPU.
"""
import ctypes as C
import enum
import os
from dataclasses import dataclass, field
from typing import List, NamedTuple, Optional

import numpy as np

from . import _lib
from ._lib import K_INF, SwgConfig, SwgError, SwgRecords, SwgStats, default_context

USIZE_MAX = K_INF


class ScoringFunction(enum.IntEnum):  # src/filter_types.rs:8-14
    Identity = 0
    Length = 1
    LengthIdentity = 2
    LogLengthIdentity = 3
    Matches = 4


class FilterMode(enum.IntEnum):  # src/filter_types.rs:18-22
    OneToOne = 0
    OneToMany = 1
    ManyToMany = 2


class ChainStatus(enum.IntEnum):  # src/mapping.rs:82-86 (+ Dropped = not in the result map)
    Dropped = 0
    Scaffold = 1
    Rescued = 2
    Unassigned = 3


STATUS_TAG = {ChainStatus.Scaffold: "scaffold", ChainStatus.Rescued: "rescued", ChainStatus.Unassigned: "unassigned"}


@dataclass
class FilterConfig:
    """src/paf_filter.rs:20-49.  Fields the filter never reads (chain_gap, plane_sweep_secondaries,
    sparsity, no_merge, prefix_delimiter, skip_prefix) are accepted and ignored, as in the reference."""
    chain_gap: int = 0
    min_block_length: int = 0
    mapping_filter_mode: FilterMode = FilterMode.ManyToMany
    mapping_max_per_query: Optional[int] = None
    mapping_max_per_target: Optional[int] = None
    plane_sweep_secondaries: int = 0
    scaffold_filter_mode: FilterMode = FilterMode.ManyToMany
    scaffold_max_per_query: Optional[int] = None
    scaffold_max_per_target: Optional[int] = None
    overlap_threshold: float = 0.95
    sparsity: float = 1.0
    no_merge: bool = True
    scaffold_gap: int = 50_000
    min_scaffold_length: int = 10_000
    scaffold_overlap_threshold: float = 0.5
    scaffold_max_deviation: int = 0
    prefix_delimiter: str = "#"
    skip_prefix: bool = False
    scoring_function: ScoringFunction = ScoringFunction.LogLengthIdentity
    min_identity: float = 0.0
    min_scaffold_identity: float = 0.0

    def to_c(self, keep_self=False, scaffolds_only=False) -> SwgConfig:
        c = SwgConfig()
        c.min_block_length = self.min_block_length
        c.mapping_filter_mode = int(self.mapping_filter_mode)
        c.mapping_max_per_query = self.mapping_max_per_query or 0
        c.mapping_max_per_target = self.mapping_max_per_target or 0
        c.scaffold_filter_mode = int(self.scaffold_filter_mode)
        c.scaffold_max_per_query = self.scaffold_max_per_query or 0
        c.scaffold_max_per_target = self.scaffold_max_per_target or 0
        c.overlap_threshold = self.overlap_threshold
        c.scaffold_gap = self.scaffold_gap
        c.min_scaffold_length = self.min_scaffold_length
        c.scaffold_overlap_threshold = self.scaffold_overlap_threshold
        c.scaffold_max_deviation = self.scaffold_max_deviation
        c.scoring_function = int(self.scoring_function)
        c.min_identity = self.min_identity
        c.min_scaffold_identity = self.min_scaffold_identity
        c.keep_self = int(bool(keep_self))
        c.scaffolds_only = int(bool(scaffolds_only))
        return c


class PlaneSweepMapping(NamedTuple):  # src/plane_sweep_exact.rs:10-18
    idx: int
    query_start: int
    query_end: int
    target_start: int
    target_end: int
    identity: float
    flags: int = 0


def _u64(a):
    return np.ascontiguousarray(np.asarray(a, dtype=np.uint64))


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def _sweep(axis, mappings, kq, kt, thr, scoring, ctx):
    ctx = ctx or default_context()
    n = len(mappings)
    if n == 0:
        return []
    qs = _u64([m.query_start for m in mappings])
    qe = _u64([m.query_end for m in mappings])
    ts = _u64([m.target_start for m in mappings])
    te = _u64([m.target_end for m in mappings])
    ident = np.ascontiguousarray(np.asarray([m.identity for m in mappings], dtype=np.float64))
    keep = np.zeros(n, dtype=np.uint8)
    ctx.check(ctx.lib.swg_plane_sweep(ctx.handle, axis, n, _ptr(qs), _ptr(qe), _ptr(ts), _ptr(te), _ptr(ident),
                                      int(kq), int(kt), float(thr), int(scoring), _ptr(keep)))
    return [int(i) for i in np.nonzero(keep)[0]]


def plane_sweep_query(mappings, mappings_to_keep, overlap_threshold, scoring=ScoringFunction.LogLengthIdentity,
                      ctx=None) -> List[int]:
    """src/plane_sweep_exact.rs:268-352"""
    return _sweep(0, mappings, mappings_to_keep, 1, overlap_threshold, scoring, ctx)


def plane_sweep_target(mappings, mappings_to_keep, overlap_threshold, scoring=ScoringFunction.LogLengthIdentity,
                       ctx=None) -> List[int]:
    """src/plane_sweep_exact.rs:355-433"""
    return _sweep(1, mappings, 1, mappings_to_keep, overlap_threshold, scoring, ctx)


def plane_sweep_both(mappings, query_mappings_to_keep, target_mappings_to_keep, overlap_threshold,
                     scoring=ScoringFunction.LogLengthIdentity, ctx=None) -> List[int]:
    """src/plane_sweep_exact.rs:436-461"""
    return _sweep(2, mappings, query_mappings_to_keep, target_mappings_to_keep, overlap_threshold, scoring, ctx)


class SequenceIndex:
    """src/sequence_index.rs:7-31: name <-> u32 id, ids in first-appearance order.  Also derives the
    two genome-prefix id tables the device needs."""

    def __init__(self):
        self.name_to_id = {}
        self.names = []

    def get_or_insert(self, name: str) -> int:
        i = self.name_to_id.get(name)
        if i is None:
            i = len(self.names)
            self.name_to_id[name] = i
            self.names.append(name)
        return i

    def __len__(self):
        return len(self.names)

    @staticmethod
    def prefix_last(name: str) -> str:  # src/paf_filter.rs:1022-1030
        p = name.rfind("#")
        return name if p < 0 else name[: p + 1]

    @staticmethod
    def prefix_two(name: str) -> str:  # src/plane_sweep_scaffold.rs:13-22
        parts = name.split("#")
        return f"{parts[0]}#{parts[1]}#" if len(parts) >= 2 else name

    def genome_tables(self):
        def table(fn):
            ids, out = {}, np.zeros(max(len(self.names), 1), dtype=np.uint32)
            for i, nm in enumerate(self.names):
                out[i] = ids.setdefault(fn(nm), len(ids))
            return out, max(len(ids), 1)

        last, n_last = table(self.prefix_last)
        two, n_two = table(self.prefix_two)
        return last, n_last, two, n_two


@dataclass
class RecordMeta:
    """src/paf_filter.rs:54-71"""
    rank: int
    query_name: str
    target_name: str
    query_start: int
    query_end: int
    target_start: int
    target_end: int
    block_length: int
    identity: float
    matches: int
    alignment_length: int
    strand: str
    chain_id: Optional[str] = None
    chain_status: ChainStatus = ChainStatus.Unassigned


def parse_cigar_counts(cigar: str):
    """src/paf.rs:32-64 -> (matches, mismatches, insertions, deletions) or None on a number error."""
    m = x = i = d = 0
    num = ""
    for ch in cigar:
        if ch.isascii() and ch.isdigit():
            num += ch
        else:
            if not num or int(num) >= 2**64:
                return None
            c = int(num)
            num = ""
            if ch == "=":
                m += c
            elif ch == "X":
                x += c
            elif ch == "I":
                i += c
            elif ch == "D":
                d += c
    return m, x, i, d


def _parse_u64(s, default):
    body = s[1:] if s.startswith("+") else s
    if body and body.isascii() and body.isdigit() and int(body) < 2**64:
        return int(body)
    return default


def _parse_f64(s):
    if not s or any(c in s for c in " \t\nxX(_"):
        return None
    try:
        return float(s)
    except ValueError:
        return None


def parse_paf_line(line: str, rank: int) -> Optional[RecordMeta]:
    """One iteration of extract_metadata, src/paf_filter.rs:298-373."""
    f = line.split("\t")
    if len(f) < 11:
        return None
    matches = _parse_u64(f[9], 0)
    block = _parse_u64(f[10], 1)
    identity = matches / max(block, 1)
    exact = matches
    for tag in f[11:]:
        if tag.startswith("dv:f:"):
            dv = _parse_f64(tag[5:])
            if dv is not None:
                identity = 1.0 - dv
        elif tag.startswith("cg:Z:"):
            cc = parse_cigar_counts(tag[5:])
            if cc is not None and cc[0] > 0:
                exact = cc[0]
                identity = cc[0] / max(block, 1)
    return RecordMeta(rank, f[0], f[5], _parse_u64(f[2], 0), _parse_u64(f[3], 0), _parse_u64(f[7], 0),
                      _parse_u64(f[8], 0), block, identity, exact, block, "+" if f[4] == "+" else "-")


def read_lines(path):
    """BufRead::lines(): split on '\\n'; a '\\r' goes only together with the '\\n' after it (a last line without a
    newline keeps a trailing '\\r')."""
    with open(path, "rb") as fh:
        data = fh.read().decode("utf-8", errors="surrogateescape")
    lines = data.split("\n")
    unterminated = bool(lines) and lines[-1] != ""
    if lines and lines[-1] == "":
        lines.pop()
    last = len(lines) - 1
    return [ln[:-1] if ln.endswith("\r") and not (unterminated and i == last) else ln for i, ln in enumerate(lines)]


@dataclass
class PackedRecords:
    """Column (SoA) form handed to the C ABI (swg_records)."""
    n: int
    cols: dict
    n_seq: int
    seq_genome_last: np.ndarray
    n_genome_last: int
    seq_genome_two: np.ndarray
    n_genome_two: int
    index: SequenceIndex = field(default=None)
    wide: bool = False   # q_start .. t_end, matches, block_len are uint64 (swg_records64; same struct layout, wider columns)

    def to_c(self) -> SwgRecords:
        r = SwgRecords()
        r.n = self.n
        for k in ("q_id", "t_id", "q_start", "q_end", "t_start", "t_end", "identity", "matches", "block_len", "strand"):
            v = self.cols[k]   # identity may be None: matches / max(block_len, 1), evaluated on the device (swg_records)
            setattr(r, k, v.ctypes.data if v is not None else None)
        r.n_seq = self.n_seq
        r.seq_genome_last = self.seq_genome_last.ctypes.data
        r.n_genome_last = self.n_genome_last
        r.seq_genome_two = self.seq_genome_two.ctypes.data
        r.n_genome_two = self.n_genome_two
        return r


def stream_plan(packed: PackedRecords, target_records: int):
    """The record ranges a streamed swg_filter call would use (swg_stream_plan; host code, no GPU): a list of bounds, or []
    when the records are not grouped by query genome."""
    from . import _lib
    lib = _lib.load()
    cap = packed.n + 2
    bounds = np.zeros(cap, dtype=np.uint64)
    k = C.c_uint64(0)
    rec = packed.to_c()
    lib.swg_stream_plan.restype = C.c_int
    rc = lib.swg_stream_plan(C.byref(rec), C.c_uint64(target_records), bounds.ctypes.data_as(C.c_void_p), C.c_uint64(cap), C.byref(k))
    if rc != 0:
        raise RuntimeError(f"swg_stream_plan failed: {rc}")
    return [int(x) for x in bounds[:k.value + 1]] if k.value else []


def pack_records(metadata: List[RecordMeta]) -> PackedRecords:
    """Interns names; the u32 layout of swg_records when every value fits, else RecordMeta's own u64 widths
    (swg_records64: the library rebases each sequence's coordinates)."""
    idx = SequenceIndex()
    n = len(metadata)
    q_id = np.fromiter((idx.get_or_insert(m.query_name) for m in metadata), dtype=np.uint32, count=n)
    t_id = np.fromiter((idx.get_or_insert(m.target_name) for m in metadata), dtype=np.uint32, count=n)

    def col32(get, what):
        return np.fromiter((get(m) for m in metadata), dtype=np.uint64, count=n)

    cols = {
        "q_id": q_id, "t_id": t_id,
        "q_start": col32(lambda m: m.query_start, "query_start"),
        "q_end": col32(lambda m: m.query_end, "query_end"),
        "t_start": col32(lambda m: m.target_start, "target_start"),
        "t_end": col32(lambda m: m.target_end, "target_end"),
        "identity": np.fromiter((m.identity for m in metadata), dtype=np.float64, count=n),
        "matches": col32(lambda m: m.matches, "matches"),
        "block_len": col32(lambda m: m.block_length, "block_length"),
        "strand": np.fromiter((0 if m.strand == "+" else 1 for m in metadata), dtype=np.uint8, count=n),
    }
    six = ("q_start", "q_end", "t_start", "t_end", "matches", "block_len")
    wide = bool(n) and max(int(cols[k].max()) for k in six) > 0xFFFFFFFF
    if not wide:
        for k in six:
            cols[k] = cols[k].astype(np.uint32)
    cols = {k: np.ascontiguousarray(v) for k, v in cols.items()}
    last, n_last, two, n_two = idx.genome_tables()
    return PackedRecords(n, cols, max(len(idx), 1), last, n_last, two, n_two, idx, wide)


class PafFilter:
    """src/paf_filter.rs:229-289"""

    def __init__(self, config: FilterConfig, ctx=None):
        self.config = config
        self.keep_self = False
        self.scaffolds_only = False
        self._ctx = ctx
        self.last_stats = None

    def with_keep_self(self, keep_self: bool):
        self.keep_self = keep_self
        return self

    def with_scaffolds_only(self, scaffolds_only: bool):
        self.scaffolds_only = scaffolds_only
        return self

    @property
    def ctx(self):
        if self._ctx is None:
            self._ctx = default_context()
        return self._ctx

    def extract_metadata(self, path) -> List[RecordMeta]:
        """src/paf_filter.rs:292-376 (plain-text PAF)."""
        out = []
        for rank, line in enumerate(read_lines(path)):
            m = parse_paf_line(line, rank)
            if m is not None:
                out.append(m)
        return out

    def filter_columns(self, packed: PackedRecords):
        """apply_filters on packed columns -> (status[n] uint8, chain[n] uint32)."""
        n = packed.n
        status = np.zeros(max(n, 1), dtype=np.uint8)
        chain = np.zeros(max(n, 1), dtype=np.uint32)
        stats = SwgStats()
        rec = packed.to_c()
        cfg = self.config.to_c(self.keep_self, self.scaffolds_only)
        ctx = self.ctx
        entry = ctx.lib.swg_filter64 if packed.wide else ctx.lib.swg_filter
        ctx.check(entry(ctx.handle, C.byref(rec), C.byref(cfg), _ptr(status), _ptr(chain), C.byref(stats)))
        self.last_stats = stats
        return status[:n], chain[:n]

    def filter_columns_multi(self, packed: PackedRecords, contexts):
        """The same over several contexts (one per GPU of the node): swg_filter_multi shards the genome pairs over
        them and makes the chain numbers global again; results are identical to filter_columns."""
        n = packed.n
        status = np.zeros(max(n, 1), dtype=np.uint8)
        chain = np.zeros(max(n, 1), dtype=np.uint32)
        stats = SwgStats()
        rec = packed.to_c()
        cfg = self.config.to_c(self.keep_self, self.scaffolds_only)
        handles = (C.c_void_p * len(contexts))(*[c.handle for c in contexts])
        entry = contexts[0].lib.swg_filter_multi64 if packed.wide else contexts[0].lib.swg_filter_multi
        contexts[0].check(entry(handles, len(contexts), C.byref(rec), C.byref(cfg), _ptr(status), _ptr(chain), C.byref(stats)))
        self.last_stats = stats
        return status[:n], chain[:n]

    def apply_filters(self, metadata: List[RecordMeta]):
        """src/paf_filter.rs:379-747 -> {rank: RecordMeta} with chain_id / chain_status set."""
        packed = pack_records(metadata)
        status, chain = self.filter_columns(packed)
        passing = {}
        for m, st, ch in zip(metadata, status, chain):
            if st == ChainStatus.Dropped:
                continue
            r = RecordMeta(**{**m.__dict__})
            r.chain_status = ChainStatus(int(st))
            r.chain_id = f"chain_{int(ch)}" if ch else None
            passing[m.rank] = r
        return passing

    def write_filtered_output(self, input_path, output_path, passing):
        """src/paf_filter.rs:1689-1726"""
        with open(output_path, "wb") as out:
            for rank, line in enumerate(read_lines(input_path)):
                meta = passing.get(rank)
                if meta is None:
                    continue
                if meta.chain_id is not None:
                    line += f"\tch:Z:{meta.chain_id}"
                line += f"\tst:Z:{STATUS_TAG[meta.chain_status]}"
                out.write(line.encode("utf-8", errors="surrogateescape") + b"\n")

    def filter_paf(self, input_path, output_path, threads=0):
        """src/paf_filter.rs:278-289.  Ingest, filter and egress all run in the native library
        (host threads for the text, the GPU for apply_filters); returns {"load","parse","filter","write"} ms."""
        cfg = self.config.to_c(self.keep_self, self.scaffolds_only)
        stats = SwgStats()
        timing = (C.c_double * 4)()
        ctx = self.ctx
        rc = ctx.lib.swg_filter_paf(ctx.handle, os.fsencode(input_path), os.fsencode(output_path), C.byref(cfg),
                                    int(threads), C.byref(stats), timing)
        if rc != 0:
            raise SwgError(rc, (ctx.lib.swg_paf_last_error() or b"").decode())
        self.last_stats = stats
        return dict(zip(("load", "parse", "filter", "write"), timing))

    def filter_paf_python(self, input_path, output_path):
        """The same three steps through the Python mirrors of extract_metadata / write_filtered_output
        (kept for API parity and as a cross-check of the native ingest)."""
        metadata = self.extract_metadata(input_path)
        passing = self.apply_filters(metadata)
        self.write_filtered_output(input_path, output_path, passing)


def plane_sweep_scaffolds(chains, filter_mode, max_per_query, max_per_target, overlap_threshold,
                          scoring_function=ScoringFunction.LogLengthIdentity, ctx=None) -> List[int]:
    """src/plane_sweep_scaffold.rs:47-94.  chains: objects/tuples exposing the ScaffoldLike accessors
    (query_name, target_name, query_start, query_end, target_start, target_end, identity).
    Returns the kept indices in the reference's output order."""
    ctx = ctx or default_context()
    n = len(chains)
    if n == 0:
        return []

    def get(c, k, pos):
        return getattr(c, k) if hasattr(c, k) else c[pos]

    idx = SequenceIndex()
    q_id = np.fromiter((idx.get_or_insert(get(c, "query_name", 0)) for c in chains), dtype=np.uint32, count=n)
    t_id = np.fromiter((idx.get_or_insert(get(c, "target_name", 1)) for c in chains), dtype=np.uint32, count=n)
    _, _, two, n_two = idx.genome_tables()
    qs = _u64([get(c, "query_start", 2) for c in chains])
    qe = _u64([get(c, "query_end", 3) for c in chains])
    ts = _u64([get(c, "target_start", 4) for c in chains])
    te = _u64([get(c, "target_end", 5) for c in chains])
    ident = np.ascontiguousarray(np.asarray([get(c, "identity", 6) for c in chains], dtype=np.float64))
    order = np.zeros(n, dtype=np.uint64)
    nk = C.c_uint64(0)
    ctx.check(ctx.lib.swg_plane_sweep_scaffolds(ctx.handle, n, _ptr(q_id), _ptr(t_id), len(idx), _ptr(two), n_two,
                                                _ptr(qs), _ptr(qe), _ptr(ts), _ptr(te), _ptr(ident), int(filter_mode),
                                                max_per_query or 0, max_per_target or 0, float(overlap_threshold),
                                                int(scoring_function), _ptr(order), C.byref(nk)))
    return [int(x) for x in order[: nk.value]]


def merge_mappings_into_chains(metadata: List[RecordMeta], max_gap: int, ctx=None):
    """src/paf_filter.rs:750-933 -> (chain_of[n] in all_chains order, dict of per-chain columns)."""
    ctx = ctx or default_context()
    packed = pack_records(metadata)
    n = packed.n
    chain_of = np.zeros(max(n, 1), dtype=np.uint32)
    cols = [np.zeros(max(n, 1), dtype=np.uint32) for _ in range(4)]
    wid = np.zeros(max(n, 1), dtype=np.float64)
    nc = C.c_uint64(0)
    rec = packed.to_c()
    ctx.check(ctx.lib.swg_merge_chains(ctx.handle, C.byref(rec), int(max_gap), _ptr(chain_of), _ptr(cols[0]),
                                       _ptr(cols[1]), _ptr(cols[2]), _ptr(cols[3]), _ptr(wid), C.byref(nc)))
    k = nc.value
    return chain_of[:n], dict(query_start=cols[0][:k], query_end=cols[1][:k], target_start=cols[2][:k],
                              target_end=cols[3][:k], weighted_identity=wid[:k])


class UnionFind:
    """src/union_find.rs over the device: unions are recorded and get_sets() labels connected components on
    the GPU.  Sets come back ordered by their smallest member (== the reference's root order for the union
    sequences the filter produces; see include/sweepga_gpu.h)."""

    def __init__(self, n, ctx=None):
        self.n = n
        self.edges = []
        self._ctx = ctx

    def union(self, x, y):
        self.edges.append((x, y))

    def get_sets(self):
        ctx = self._ctx or default_context()
        xs = np.ascontiguousarray(np.asarray([e[0] for e in self.edges], dtype=np.uint32))
        ys = np.ascontiguousarray(np.asarray([e[1] for e in self.edges], dtype=np.uint32))
        set_of = np.zeros(max(self.n, 1), dtype=np.uint32)
        ns = C.c_uint64(0)
        ctx.check(ctx.lib.swg_union_find_sets(ctx.handle, self.n, len(self.edges), _ptr(xs), _ptr(ys), _ptr(set_of),
                                              C.byref(ns)))
        sets = [[] for _ in range(ns.value)]
        for i in range(self.n):
            sets[int(set_of[i])].append(i)
        return sets
