"""Seeded synthetic inputs for parity tests (numpy; small enough for the CPU oracle)."""
import numpy as np

from tests import orc


def random_segment(rng, n, span=10_000, max_len=2_000, zero_frac=0.03, dup_frac=0.1, ident_levels=None):
    """One sweep segment: (qs, qe, ts, te, identity) with ties, duplicates, zero-length and
    zero-identity records mixed in."""
    qs = rng.integers(0, span, n)
    ql = rng.integers(1, max_len, n)
    ts = rng.integers(0, span, n)
    tl = rng.integers(1, max_len, n)
    if ident_levels:
        ident = rng.choice(np.asarray(ident_levels, dtype=np.float64), n)
    else:
        ident = np.round(rng.uniform(0.7, 1.0, n), 3)
    z = rng.random(n) < zero_frac
    ql[z] = 0
    z2 = rng.random(n) < zero_frac
    tl[z2] = 0
    z3 = rng.random(n) < zero_frac / 2
    ident[z3] = 0.0
    # exact duplicates of earlier records (score + start ties -> index tie-break)
    for i in range(1, n):
        if rng.random() < dup_frac:
            j = rng.integers(0, i)
            qs[i], ql[i], ident[i] = qs[j], ql[j], ident[j]
            if rng.random() < 0.5:
                ts[i], tl[i] = ts[j], tl[j]
    return (qs.astype(np.uint64), (qs + ql).astype(np.uint64), ts.astype(np.uint64), (ts + tl).astype(np.uint64),
            ident.astype(np.float64))


def random_records(rng, n, n_genomes=4, chrs_per_genome=3, span=200_000, max_len=8_000, pansn=True,
                   self_frac=0.02, minus_frac=0.2, syntenic_frac=0.7, zero_frac=0.01):
    """PAF-like records over several genomes/chromosomes -> orc.Records (names + columns)."""
    def name(g, c):
        return f"g{g}#1#chr{c}" if pansn else f"g{g}chr{c}"

    gq = rng.integers(0, n_genomes, n)
    gt = rng.integers(0, n_genomes, n)
    cq = rng.integers(0, chrs_per_genome, n)
    ct = np.where(rng.random(n) < 0.8, cq, rng.integers(0, chrs_per_genome, n))
    same = rng.random(n) < self_frac
    gt = np.where(same, gq, gt)
    ct = np.where(same, cq, ct)
    qs = rng.integers(0, span, n)
    ln = np.minimum(np.exp(rng.normal(np.log(max_len / 8), 1.0, n)).astype(np.int64) + 50, max_len)
    ln[rng.random(n) < zero_frac] = 0
    syn = rng.random(n) < syntenic_frac
    ts = np.where(syn, np.clip(qs + rng.normal(0, 2000, n).astype(np.int64), 0, span), rng.integers(0, span, n))
    tl = np.maximum(ln + rng.integers(-20, 20, n), 0)
    tl[ln == 0] = 0
    ident = np.round(0.7 + 0.3 * rng.beta(5, 1.5, n), 4)
    block = np.maximum(ln, tl).astype(np.uint64)
    matches = np.floor(ident * block).astype(np.uint64)
    ident = matches / np.maximum(block, 1)
    strand = np.where(rng.random(n) < minus_frac, ord("-"), ord("+")).astype(np.uint8)
    qn = [name(int(a), int(b)) for a, b in zip(gq, cq)]
    tn = [name(int(a), int(b)) for a, b in zip(gt, ct)]
    u = lambda a: np.ascontiguousarray(np.asarray(a, dtype=np.uint64))
    return orc.Records(qn, tn, u(qs), u(qs + ln), u(ts), u(ts + tl), u(block), np.ascontiguousarray(ident, dtype=np.float64),
                       u(matches), strand, u(np.arange(n)))


def records_to_meta(rec):
    """orc.Records -> list of sweepga_amd.RecordMeta (host mirror input)."""
    from sweepga_amd import RecordMeta
    out = []
    for i in range(len(rec)):
        out.append(RecordMeta(int(rec.rank[i]), rec.qname[i], rec.tname[i], int(rec.qs[i]), int(rec.qe[i]),
                              int(rec.ts[i]), int(rec.te[i]), int(rec.block_length[i]), float(rec.identity[i]),
                              int(rec.matches[i]), int(rec.block_length[i]), chr(int(rec.strand[i]))))
    return out


def records_to_paf(rng, rec, junk_lines=True, dv_tags=True):
    """orc.Records -> PAF text.  Identity is carried the way real PAFs do it: a third of the lines by the
    matches/block columns only, a third with an extended CIGAR (cg:Z:<m>=<x>X, which overrides column 10),
    a third with a dv:f: tag; plus a few malformed / short lines that must be skipped but still counted."""
    out = []
    for i in range(len(rec)):
        m, b = int(rec.matches[i]), int(rec.block_length[i])
        line = [rec.qname[i], "1000000", str(int(rec.qs[i])), str(int(rec.qe[i])), chr(int(rec.strand[i])), rec.tname[i],
                "1000000", str(int(rec.ts[i])), str(int(rec.te[i])), str(m), str(b), "60"]
        k = int(rng.integers(0, 3))
        if k == 1 and m > 0:
            line += ["NM:i:%d" % (b - m), "cg:Z:%d=%dX" % (m, b - m)]
        elif k == 2 and dv_tags:   # (dv_tags=False: no line's identity is overridden -> the ingest reports it as derived)
            line += ["dv:f:%.4f" % (1.0 - m / max(b, 1)), "tp:A:P"]
        out.append("\t".join(line))
        if junk_lines and rng.random() < 0.01:
            out.append(rng.choice(["", "# comment", "too\tfew\tfields", "q\t1\tx\ty\t+\tt\t1\t0\t0\tz\t\t0"]))
    return "\n".join(out) + "\n"


WIDE_OFFSETS = [0, 2**32 - 1, 2**32 + 5, 5_000_000_000, 2**40 + 123, 2**62, 77]


def shifted(rec, rng):
    """The same records with every sequence moved by its own constant (as query and as target alike) -> (Records, {name: offset})."""
    names = sorted(set(rec.qname) | set(rec.tname))
    off = {nm: WIDE_OFFSETS[int(rng.integers(0, len(WIDE_OFFSETS)))] for nm in names}
    oq = np.array([off[x] for x in rec.qname], dtype=np.uint64)
    ot = np.array([off[x] for x in rec.tname], dtype=np.uint64)
    return orc.Records(rec.qname, rec.tname, rec.qs + oq, rec.qe + oq, rec.ts + ot, rec.te + ot, rec.block_length, rec.identity,
                       rec.matches, rec.strand, rec.rank), off


def shifted_by_axis(rec, rng):
    """The same records with the QUERY coordinates moved by a constant per (query sequence, genome of the target) and the TARGET
    coordinates by a constant per (target sequence, genome of the query) -- the mapping-level sweep's segments
    (src/paf_filter.rs:1037-1100; genome = the name up to its last '#').  A sequence mapped against several genomes is then touched
    over far more than 2^32 bases, each of its segments over as few as before."""
    genome = lambda name: name.rsplit("#", 1)[0] + "#" if "#" in name else name   # noqa: E731
    pick = lambda: WIDE_OFFSETS[int(rng.integers(0, len(WIDE_OFFSETS)))]            # noqa: E731
    off_q, off_t = {}, {}
    oq = np.zeros(len(rec), dtype=np.uint64)
    ot = np.zeros(len(rec), dtype=np.uint64)
    for i, (q, t) in enumerate(zip(rec.qname, rec.tname)):
        kq, kt = (q, genome(t)), (t, genome(q))
        if kq not in off_q:
            off_q[kq] = pick()
        if kt not in off_t:
            off_t[kt] = pick()
        oq[i], ot[i] = off_q[kq], off_t[kt]
    return orc.Records(rec.qname, rec.tname, rec.qs + oq, rec.qe + oq, rec.ts + ot, rec.te + ot, rec.block_length, rec.identity,
                       rec.matches, rec.strand, rec.rank)
