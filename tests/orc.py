"""ctypes front-end to the CPU oracle (oracle/liboracle.so).  Test infrastructure only."""
import ctypes as C
import os
from dataclasses import dataclass

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_SO = os.environ.get("SWG_ORACLE_SO") or os.path.join(ROOT, "oracle", "liboracle.so")  # the override is for oracle development
K_INF = 2**64 - 1

IDENTITY, LENGTH, LENGTH_IDENTITY, LOG_LENGTH_IDENTITY, MATCHES = range(5)
ONE_TO_ONE, ONE_TO_MANY, MANY_TO_MANY = range(3)
DROPPED, SCAFFOLD, RESCUED, UNASSIGNED = range(4)


class OrcConfig(C.Structure):
    _fields_ = [
        ("min_block_length", C.c_uint64),
        ("mapping_filter_mode", C.c_int32),
        ("mapping_max_per_query", C.c_uint64),
        ("mapping_max_per_target", C.c_uint64),
        ("scaffold_filter_mode", C.c_int32),
        ("scaffold_max_per_query", C.c_uint64),
        ("scaffold_max_per_target", C.c_uint64),
        ("overlap_threshold", C.c_double),
        ("scaffold_gap", C.c_uint64),
        ("min_scaffold_length", C.c_uint64),
        ("scaffold_overlap_threshold", C.c_double),
        ("scaffold_max_deviation", C.c_uint64),
        ("scoring_function", C.c_int32),
        ("min_identity", C.c_double),
        ("min_scaffold_identity", C.c_double),
        ("keep_self", C.c_int32),
        ("scaffolds_only", C.c_int32),
    ]


@dataclass
class Config:
    """Mirror of the used fields of FilterConfig (paf_filter.rs:20-49); defaults = CLI defaults."""
    min_block_length: int = 0
    mapping_filter_mode: int = MANY_TO_MANY
    mapping_max_per_query: int = 0
    mapping_max_per_target: int = 0
    scaffold_filter_mode: int = MANY_TO_MANY
    scaffold_max_per_query: int = 0
    scaffold_max_per_target: int = 0
    overlap_threshold: float = 0.95
    scaffold_gap: int = 50000
    min_scaffold_length: int = 10000
    scaffold_overlap_threshold: float = 0.5
    scaffold_max_deviation: int = 0
    scoring_function: int = LOG_LENGTH_IDENTITY
    min_identity: float = 0.0
    min_scaffold_identity: float = 0.0
    keep_self: bool = False
    scaffolds_only: bool = False

    def c(self):
        o = OrcConfig()
        for name, _ in OrcConfig._fields_:
            setattr(o, name, getattr(self, name))
        return o


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(_SO)
        _lib.orc_score.restype = C.c_double
        _lib.orc_score.argtypes = [C.c_uint64, C.c_uint64, C.c_double, C.c_int]
        _lib.orc_log.restype = C.c_double
        _lib.orc_log.argtypes = [C.c_double]
        _lib.orc_plane_sweep_scaffolds.restype = C.c_int64
        _lib.orc_union_find_sets.restype = C.c_int64
        _lib.orc_plane_sweep_core.restype = C.c_int64
        _lib.orc_apply_filters.restype = C.c_int64
        _lib.orc_merge_chains.restype = C.c_int64
        _lib.orc_extract_metadata.restype = C.c_int64
        _lib.orc_round_nice.restype = C.c_uint64
        _lib.orc_round_nice.argtypes = [C.c_uint64]
    return _lib


def _u64(a):
    return np.ascontiguousarray(np.asarray(a, dtype=np.uint64))


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _names(names):
    arr = (C.c_char_p * len(names))()
    arr[:] = [n.encode() if isinstance(n, str) else n for n in names]
    return arr


def plane_sweep(axis, qs, qe, ts, te, identity, k_q=1, k_t=1, thr=0.95, scoring=LOG_LENGTH_IDENTITY):
    """axis 0 = plane_sweep_query, 1 = plane_sweep_target, 2 = plane_sweep_both -> kept indices."""
    qs, qe, ts, te = map(_u64, (qs, qe, ts, te))
    ident = np.ascontiguousarray(np.asarray(identity, dtype=np.float64))
    n = len(qs)
    keep = np.zeros(max(n, 1), dtype=np.uint8)
    lib().orc_plane_sweep(C.c_int(axis), C.c_uint64(n), _p(qs), _p(qe), _p(ts), _p(te), _p(ident),
                          C.c_uint64(k_q), C.c_uint64(k_t), C.c_double(thr), C.c_int(scoring), _p(keep))
    return [int(i) for i in np.nonzero(keep[:n])[0]]


def plane_sweep_query(maps, k, thr, scoring=LOG_LENGTH_IDENTITY):
    """maps: list of (qs, qe, ts, te, identity)."""
    a = list(zip(*maps)) if maps else [[], [], [], [], []]
    return plane_sweep(0, a[0], a[1], a[2], a[3], a[4], k_q=k, thr=thr, scoring=scoring)


def plane_sweep_target(maps, k, thr, scoring=LOG_LENGTH_IDENTITY):
    a = list(zip(*maps)) if maps else [[], [], [], [], []]
    return plane_sweep(1, a[0], a[1], a[2], a[3], a[4], k_t=k, thr=thr, scoring=scoring)


def plane_sweep_both(maps, kq, kt, thr, scoring=LOG_LENGTH_IDENTITY):
    a = list(zip(*maps)) if maps else [[], [], [], [], []]
    return plane_sweep(2, a[0], a[1], a[2], a[3], a[4], k_q=kq, k_t=kt, thr=thr, scoring=scoring)


def score(qs, qe, identity, scoring):
    return lib().orc_score(qs, qe, identity, scoring)


def plane_sweep_scaffolds(chains, mode, max_q, max_t, thr, scoring=LOG_LENGTH_IDENTITY):
    """chains: list of (qname, tname, qs, qe, ts, te, identity) -> kept indices in reference order."""
    n = len(chains)
    qn = _names([c[0] for c in chains])
    tn = _names([c[1] for c in chains])
    qs, qe, ts, te = (_u64([c[i] for c in chains]) for i in (2, 3, 4, 5))
    ident = np.ascontiguousarray(np.asarray([c[6] for c in chains], dtype=np.float64))
    order = np.zeros(max(n, 1), dtype=np.uint64)
    k = lib().orc_plane_sweep_scaffolds(C.c_uint64(n), qn, tn, _p(qs), _p(qe), _p(ts), _p(te), _p(ident),
                                        C.c_int(mode), C.c_uint64(max_q or 0), C.c_uint64(max_t or 0),
                                        C.c_double(thr), C.c_int(scoring), _p(order))
    return [int(x) for x in order[:k]]


def union_find_sets(n, unions):
    xs = _u64([u[0] for u in unions])
    ys = _u64([u[1] for u in unions])
    set_of = np.zeros(max(n, 1), dtype=np.uint64)
    k = lib().orc_union_find_sets(C.c_uint64(n), C.c_uint64(len(unions)), _p(xs), _p(ys), _p(set_of))
    sets = [[] for _ in range(k)]
    for i in range(n):
        sets[int(set_of[i])].append(i)
    return sets


def plane_sweep_core(intervals, max_to_keep, thr):
    """intervals: list of (begin, end, score)."""
    n = len(intervals)
    b = np.ascontiguousarray(np.asarray([i[0] for i in intervals], dtype=np.uint32))
    e = np.ascontiguousarray(np.asarray([i[1] for i in intervals], dtype=np.uint32))
    s = np.ascontiguousarray(np.asarray([i[2] for i in intervals], dtype=np.float64))
    order = np.zeros(max(n, 1), dtype=np.uint64)
    k = lib().orc_plane_sweep_core(C.c_uint64(n), _p(b), _p(e), _p(s), C.c_uint64(max_to_keep),
                                   C.c_double(thr), _p(order))
    return [int(x) for x in order[:k]]


@dataclass
class Records:
    """Column form of Vec<RecordMeta> (paf_filter.rs:54-71)."""
    qname: list
    tname: list
    qs: np.ndarray
    qe: np.ndarray
    ts: np.ndarray
    te: np.ndarray
    block_length: np.ndarray
    identity: np.ndarray
    matches: np.ndarray
    strand: np.ndarray  # uint8 '+'/'-'
    rank: np.ndarray

    def __len__(self):
        return len(self.qname)


def parse_paf_text(text):
    """extract_metadata (paf_filter.rs:292-376) in Python for *small* fixtures: same rules."""
    qn, tn, cols, ranks = [], [], [], []
    lines = text.split("\n")
    unterminated = bool(lines) and lines[-1] != ""
    if lines and lines[-1] == "":
        lines.pop()
    for rank, line in enumerate(lines):
        if line.endswith("\r") and not (unterminated and rank == len(lines) - 1):   # a '\r' goes only with its '\n'
            line = line[:-1]
        f = line.split("\t")
        if len(f) < 11:
            continue

        def u(s, d=0):
            s2 = s[1:] if s.startswith("+") else s
            return int(s2) if s2.isascii() and s2.isdigit() and int(s2) < 2**64 else d

        matches, block = u(f[9]), u(f[10], 1)
        ident = matches / max(block, 1)
        for tag in f[11:]:
            if tag.startswith("dv:f:"):
                try:
                    ident = 1.0 - float(tag[5:])
                except ValueError:
                    pass
            elif tag.startswith("cg:Z:"):
                m = C.c_uint64()
                x = C.c_uint64()
                i = C.c_uint64()
                d = C.c_uint64()
                if lib().orc_parse_cigar_counts(tag[5:].encode(), C.byref(m), C.byref(x), C.byref(i), C.byref(d)):
                    if m.value > 0:
                        matches = m.value
                        ident = m.value / max(block, 1)
        qn.append(f[0])
        tn.append(f[5])
        cols.append((u(f[2]), u(f[3]), u(f[7]), u(f[8]), block, ident, matches, ord("+") if f[4] == "+" else ord("-")))
        ranks.append(rank)
    a = list(zip(*cols)) if cols else [[]] * 8
    return Records(qn, tn, _u64(a[0]), _u64(a[1]), _u64(a[2]), _u64(a[3]), _u64(a[4]),
                   np.asarray(a[5], dtype=np.float64), _u64(a[6]), np.asarray(a[7], dtype=np.uint8), _u64(ranks))


def apply_filters(cfg: Config, rec: Records, want_seconds=False, barrier=None):
    """PafFilter::apply_filters -> (status[n], chain[n]) aligned with rec rows.
    `barrier` (threading.Barrier): waited on right before the C call, so several threads can start together
    (ctypes releases the GIL during the call)."""
    n = len(rec)
    status = np.zeros(max(n, 1), dtype=np.uint8)
    chain = np.zeros(max(n, 1), dtype=np.uint32)
    secs = C.c_double(0.0)
    cc = cfg.c()
    strand = np.ascontiguousarray(rec.strand)
    ident = np.ascontiguousarray(rec.identity)
    qn, tn = _names(rec.qname), _names(rec.tname)
    if barrier is not None:
        barrier.wait()
    r = lib().orc_apply_filters(C.byref(cc), C.c_uint64(n), _p(rec.rank), qn, tn,
                                _p(rec.qs), _p(rec.qe), _p(rec.ts), _p(rec.te), _p(rec.block_length),
                                _p(ident), _p(rec.matches), _p(strand), _p(status), _p(chain), C.byref(secs))
    assert r >= 0, "oracle apply_filters failed"
    if want_seconds:
        return status[:n], chain[:n], secs.value
    return status[:n], chain[:n]


def apply_filters_ids(cfg: Config, cols, names, lo, hi, status_out, chain_out):
    """apply_filters over rows [lo, hi) of SoA columns with integer sequence ids (numpy arrays: q_id, t_id, q_start,
    q_end, t_start, t_end, matches, block_len as uint32, identity float64, strand uint8 with 0 = '+').
    Writes status_out[lo:hi] / chain_out[lo:hi]; returns the C-side seconds.  Releases the GIL while running."""
    cc = cfg.c()
    arr = (C.c_char_p * len(names))()
    arr[:] = [n.encode() for n in names]
    secs = C.c_double(0.0)
    st = status_out[lo:hi]
    ch = chain_out[lo:hi]
    f = lib().orc_apply_filters_ids
    f.restype = C.c_int64
    r = f(C.byref(cc), C.c_uint64(lo), C.c_uint64(hi), _p(cols["q_id"]), _p(cols["t_id"]), arr, _p(cols["q_start"]),
          _p(cols["q_end"]), _p(cols["t_start"]), _p(cols["t_end"]), _p(cols["block_len"]), _p(cols["identity"]),
          _p(cols["matches"]), _p(cols["strand"]), _p(st), _p(ch), C.byref(secs))
    assert r >= 0, "oracle apply_filters failed"
    return secs.value


def apply_filters_by_groups(cfg: Config, cols, names, bounds, threads):
    """The oracle over consecutive row ranges bounds[j]..bounds[j+1] (each a union of whole genome-pair groups,
    i.e. independent units of the filter), `threads` ranges at a time.  Returns (status, chain, wall seconds);
    chain numbers are local to each range."""
    import threading
    import time
    n = int(bounds[-1])
    status = np.zeros(max(n, 1), dtype=np.uint8)
    chain = np.zeros(max(n, 1), dtype=np.uint32)
    jobs = list(zip(bounds[:-1], bounds[1:]))
    lock = threading.Lock()
    err = []

    def worker():
        while True:
            with lock:
                if not jobs:
                    return
                lo, hi = jobs.pop()
            try:
                apply_filters_ids(cfg, cols, names, int(lo), int(hi), status, chain)
            except Exception as e:  # noqa: BLE001
                err.append(e)
                return

    t0 = time.perf_counter()
    th = [threading.Thread(target=worker) for _ in range(max(1, threads))]
    for t in th:
        t.start()
    for t in th:
        t.join()
    if err:
        raise err[0]
    return status[:n], chain[:n], time.perf_counter() - t0


def same_chain_partition(a, b):
    """Chain ids are arbitrary labels: two labelings agree iff they induce the same partition (0 = no chain)."""
    a = np.asarray(a, dtype=np.int64)
    b = np.asarray(b, dtype=np.int64)
    if not np.array_equal(a == 0, b == 0):
        return False
    m = a != 0
    if not m.any():
        return True
    pairs = np.unique(np.stack([a[m], b[m]], axis=1), axis=0)
    return len(np.unique(pairs[:, 0])) == len(pairs) == len(np.unique(pairs[:, 1]))


def merge_chains(rec: Records, max_gap):
    n = len(rec)
    chain_of = np.zeros(max(n, 1), dtype=np.uint32)
    cq = [np.zeros(max(n, 1), dtype=np.uint64) for _ in range(5)]
    wid = np.zeros(max(n, 1), dtype=np.float64)
    strand = np.ascontiguousarray(rec.strand)
    k = lib().orc_merge_chains(C.c_uint64(n), _names(rec.qname), _names(rec.tname), _p(rec.qs), _p(rec.qe),
                               _p(rec.ts), _p(rec.te), _p(rec.block_length), _p(rec.matches), _p(strand),
                               C.c_uint64(max_gap), _p(chain_of), _p(cq[0]), _p(cq[1]), _p(cq[2]), _p(cq[3]),
                               _p(cq[4]), _p(wid))
    return chain_of[:n], [c[:k] for c in cq], wid[:k]


def filter_paf(cfg: Config, in_path, out_path):
    cc = cfg.c()
    r = lib().orc_filter_paf(C.byref(cc), in_path.encode(), out_path.encode())
    assert r == 0


def parse_filter_mode(s):
    m = C.c_int32()
    pq = C.c_uint64()
    pt = C.c_uint64()
    ok = lib().orc_parse_filter_mode(s.encode(), C.byref(m), C.byref(pq), C.byref(pt))
    return (m.value, pq.value or None, pt.value or None) if ok else None


def parse_metric_number(s):
    o = C.c_uint64()
    return o.value if lib().orc_parse_metric_number(s.encode(), C.byref(o)) else None


def parse_identity_value(s, ani_percentile=None):
    o = C.c_double()
    if ani_percentile is None:
        return o.value if lib().orc_parse_identity_value(s.encode(), C.byref(o)) else None
    return o.value if lib().orc_parse_identity_value_ani(s.encode(), C.c_double(ani_percentile), C.byref(o)) else None


ANI_ALL, ANI_ORTHOGONAL, ANI_NPERCENTILE = range(3)
NSORT_LENGTH, NSORT_IDENTITY, NSORT_SCORE = range(3)


def parse_ani_method(s):
    """main.rs:296-330 -> (kind, percentile, sort) or None"""
    k, p, so = C.c_int(), C.c_double(), C.c_int()
    if not lib().orc_parse_ani_method(s.encode(), C.byref(k), C.byref(p), C.byref(so)):
        return None
    return k.value, p.value, so.value


def calculate_ani_stats(path, kind, percentile=50.0, sort=NSORT_IDENTITY):
    """main.rs:334-688 -> median per-genome-pair ANI"""
    o = C.c_double()
    r = lib().orc_calculate_ani_stats(str(path).encode(), C.c_int(kind), C.c_double(percentile), C.c_int(sort), C.byref(o))
    if r != 0:
        raise RuntimeError("oracle calculate_ani_stats failed")
    return o.value


def clamp_scaffold_params(jump, mass, avg, adaptive):
    j = C.c_uint64()
    m = C.c_uint64()
    lib().orc_clamp_scaffold_params(C.c_uint64(jump), C.c_uint64(mass), C.c_int(avg is not None),
                                    C.c_uint64(avg or 0), C.c_int(bool(adaptive)), C.byref(j), C.byref(m))
    return j.value, m.value


def set_fast_inversion(on):
    """apply_filters' step 4b (paf_filter.rs:535-597) through a bucket index instead of the literal chains x reverse-mappings
    loop (oracle/sweepga_oracle.cpp; same result, tests/test_oracle_fast_cpu.py).  Process-wide; for full-size checks only."""
    lib().orc_set_fast_inversion(C.c_int(1 if on else 0))
