"""Native PAF ingest / egress (sweepga_amd/csrc/host/paf_io.cpp) against the oracle's extract_metadata /
write rules (paf_filter.rs:292-376, 1689-1726) and the Python mirror.  Host code only: runs without a GPU."""
import ctypes as C
import gzip
import os
import struct
import zlib

import numpy as np
import pytest

from sweepga_amd import FilterConfig, PafFile, PafFilter, SwgError
from tests import gen, orc

EDGE_TEXT = (
    "q1\t100\t10\t60\t+\tt1\t100\t5\t55\t45\t50\t60\n"                       # plain
    "q1\t100\t10\t60\t-\tt2\t100\t5\t55\t45\t50\t60\tdv:f:0.25\n"            # dv tag
    "q2\t100\t10\t60\t+\tt1\t100\t5\t55\t45\t50\t60\tcg:Z:30=5X10=\tdv:f:0.5\n"  # both: last writer wins
    "q2\t100\t10\t60\t+\tt1\t100\t5\t55\t45\t50\t60\tdv:f:0.5\tcg:Z:30=5X10=\n"
    "q2\t100\t10\t60\t+\tt1\t100\t5\t55\t45\t50\t60\tcg:Z:10M\n"             # no '=' -> column 10 stands
    "q2\t100\t10\t60\t+\tt1\t100\t5\t55\t45\t50\t60\tcg:Z:=5\n"              # bad cigar -> ignored
    "short\tline\n"                                                          # skipped, still counted
    "\n"
    "# comment\n"
    "q3\t100\tx\t+7\t*\tt3\t100\t-1\t8\tm\t\t60\n"                           # parse failures -> 0 / 0 / block 1
    "q3\t100\t1\t2\t+\tt3\t100\t3\t4\t5\t0\t60\tdv:f:nanx\tdv:f:1e-2\r\n"     # CRLF, block 0 -> denominator 1
    "q1\t100\t10\t60\t+\tt1\t100\t5\t55\t45\t50"                             # 11 fields, no trailing newline
)


def native_vs_oracle(text, tmp_path, threads):
    path = tmp_path / "in.paf"
    path.write_bytes(text.encode())
    n_cap = text.count("\n") + 2
    cols = {k: np.zeros(n_cap, dtype=np.uint64) for k in ("rank", "qs", "qe", "ts", "te", "block", "matches")}
    ident = np.zeros(n_cap, dtype=np.float64)
    strand = np.zeros(n_cap, dtype=np.uint8)
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    f = orc.lib().orc_extract_metadata
    f.restype = C.c_int64
    n = f(str(path).encode(), C.c_uint64(n_cap), p(cols["rank"]), p(cols["qs"]), p(cols["qe"]), p(cols["ts"]),
          p(cols["te"]), p(cols["block"]), p(ident), p(cols["matches"]), p(strand))
    assert n >= 0
    with PafFile(path, threads=threads) as pf:
        assert pf.n == n
        assert (pf.ranks == cols["rank"][:n]).all()
        for name, key in (("q_start", "qs"), ("q_end", "qe"), ("t_start", "ts"), ("t_end", "te"),
                          ("block_len", "block"), ("matches", "matches")):
            assert (pf.column(name).astype(np.uint64) == cols[key][:n]).all(), name
        assert (pf.column("identity").view(np.uint64) == ident[:n].view(np.uint64)).all()
        assert (pf.column("strand") == (strand[:n] != ord("+"))).all()
        names = pf.names
        rec = orc.parse_paf_text(text)
        assert [names[i] for i in pf.column("q_id")] == rec.qname
        assert [names[i] for i in pf.column("t_id")] == rec.tname
        seen = []
        for a, b in zip(rec.qname, rec.tname):
            for nm in (a, b):
                if nm not in seen:
                    seen.append(nm)
        assert names == seen  # SequenceIndex order: first appearance, query before target
        from sweepga_amd import SequenceIndex
        assert list(pf.seq_genome_last[:len(names)]) == _dense([SequenceIndex.prefix_last(x) for x in names])
        assert list(pf.seq_genome_two[:len(names)]) == _dense([SequenceIndex.prefix_two(x) for x in names])
        return pf.n, pf.n_lines


def _dense(keys):
    ids = {}
    return [ids.setdefault(k, len(ids)) for k in keys]


def test_edge_lines(tmp_path):
    n, n_lines = native_vs_oracle(EDGE_TEXT, tmp_path, 1)
    assert (n, n_lines) == (9, 12)


@pytest.mark.parametrize("threads", [1, 2, 5, 8])
def test_random_paf_all_thread_counts(tmp_path, threads):
    rng = np.random.default_rng(100 + threads)
    rec = gen.random_records(rng, 6000, n_genomes=5, chrs_per_genome=4)
    text = gen.records_to_paf(rng, rec) * 3  # > threads * 64 KiB so every slice is exercised
    native_vs_oracle(text, tmp_path, threads)


def test_many_hash_names(tmp_path):
    rng = np.random.default_rng(3)
    lines = []
    for i in range(3000):
        q = "s%d#%d#c%d#x%d" % (rng.integers(0, 4), rng.integers(1, 3), rng.integers(0, 5), rng.integers(0, 2))
        t = ["plain%d" % rng.integers(0, 9), "a#b", "a#b#", "#", "a##c"][int(rng.integers(0, 5))]
        lines.append("%s\t9\t1\t5\t+\t%s\t9\t2\t6\t3\t4\t0" % (q, t))
    native_vs_oracle("\n".join(lines) + "\n", tmp_path, 4)


def test_carriage_return_goes_only_with_its_newline(tmp_path):
    """std's Lines::next pops a '\\r' only after popping a '\\n': the last line of a file without a final newline keeps its
    '\\r', so an 11-field last line has block length "50\\r" (parse failure -> 1) and a tag there reads "dv:f:0.5\\r"."""
    native_vs_oracle("q1\t100\t10\t60\t+\tt1\t100\t5\t55\t45\t50\r\nq1\t100\t10\t60\t+\tt1\t100\t5\t55\t45\t50\r", tmp_path, 2)
    with PafFile(text="q1\t100\t10\t60\t+\tt1\t100\t5\t55\t45\t50\r\nq1\t100\t10\t60\t+\tt1\t100\t5\t55\t45\t50\r") as pf:
        assert [int(x) for x in pf.column("block_len")] == [50, 1]
        assert list(pf.column("identity")) == [45 / 50, 45.0]
    native_vs_oracle("q1\t100\t10\t60\t+\tt1\t100\t5\t55\t45\t50\t60\tdv:f:0.5\r", tmp_path, 1)


def test_empty_and_blank_inputs(tmp_path):
    for text, lines in (("", 0), ("\n", 1), ("\n\n\n", 3), ("a\tb", 1)):
        path = tmp_path / "e.paf"
        path.write_text(text)
        with PafFile(path, threads=4) as pf:
            assert (pf.n, pf.n_lines) == (0, lines)
            assert pf.write(tmp_path / "e.out", np.zeros(0, dtype=np.uint8)) == 0
        assert (tmp_path / "e.out").read_bytes() == b""


def bgzf_bytes(data, block=40000):
    out = b""
    chunks = [data[i:i + block] for i in range(0, len(data), block)] + [b""]  # EOF marker block
    for c in chunks:
        co = zlib.compressobj(6, zlib.DEFLATED, -15)
        cd = co.compress(c) + co.flush()
        bsize = 12 + 6 + len(cd) + 8 - 1
        out += (b"\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff" + struct.pack("<H", 6) + b"BC" + struct.pack("<HH", 2, bsize)
                + cd + struct.pack("<II", zlib.crc32(c), len(c)))
    return out


def test_gzip_and_bgzf_inputs(tmp_path):  # src/paf.rs:10-30
    rng = np.random.default_rng(8)
    text = gen.records_to_paf(rng, gen.random_records(rng, 5000)).encode()
    plain = tmp_path / "x.paf"
    plain.write_bytes(text)
    with PafFile(plain, threads=2) as ref:
        want = {c: ref.column(c).copy() for c in ("q_id", "t_id", "q_start", "t_end", "identity", "strand")}
        want_names, want_ranks = ref.names, ref.ranks.copy()
    variants = {"x.paf.gz": gzip.compress(text), "y.paf.bgz": bgzf_bytes(text),
                "z.paf.gz": gzip.compress(text[:100000]) + gzip.compress(text[100000:]),  # multi-member
                "w.paf": bgzf_bytes(text)}  # gzip magic without the extension
    for name, blob in variants.items():
        (tmp_path / name).write_bytes(blob)
        with PafFile(tmp_path / name, threads=4) as pf:
            assert pf.names == want_names and (pf.ranks == want_ranks).all(), name
            for c, v in want.items():
                assert (pf.column(c) == v).all(), (name, c)
    (tmp_path / "bad.paf.gz").write_bytes(bgzf_bytes(text)[:-40] + b"\0" * 12)
    with pytest.raises(SwgError):
        PafFile(tmp_path / "bad.paf.gz")


@pytest.mark.parametrize("threads", [1, 3, 8])
def test_writer_matches_python_mirror(tmp_path, threads):  # src/paf_filter.rs:1689-1726
    rng = np.random.default_rng(21)
    rec = gen.random_records(rng, 9000)
    text = gen.records_to_paf(rng, rec).replace("\n", "\r\n", 50)
    path = tmp_path / "in.paf"
    path.write_bytes(text.encode())
    f = PafFilter(FilterConfig())
    meta = f.extract_metadata(path)
    status = rng.integers(0, 4, len(meta)).astype(np.uint8)
    chain = np.where(rng.random(len(meta)) < 0.6, rng.integers(1, 2**32, len(meta)), 0).astype(np.uint32)
    chain[:5] = [0, 1, 9, 10, 4294967295]
    from sweepga_amd.filter import ChainStatus
    passing = {}
    for m, st, ch in zip(meta, status, chain):
        if st:
            m.chain_status = ChainStatus(int(st))
            m.chain_id = "chain_%d" % ch if ch else None
            passing[m.rank] = m
    f.write_filtered_output(path, tmp_path / "py.out", passing)
    with PafFile(path, threads=threads) as pf:
        assert pf.write(tmp_path / "native.out", status, chain) == len(passing)
        assert pf.write(tmp_path / "native_nochain.out", status) == len(passing)
    assert (tmp_path / "native.out").read_bytes() == (tmp_path / "py.out").read_bytes()
    assert b"ch:Z:" not in (tmp_path / "native_nochain.out").read_bytes()


def test_open_buffer_and_errors(tmp_path):
    with PafFile(text=EDGE_TEXT, threads=2) as pf:
        assert (pf.n, pf.n_lines) == (9, 12)
    with pytest.raises(SwgError, match="cannot open"):
        PafFile(tmp_path / "missing.paf")
    # a sequence touched over 2^32 bases by the mappings against ONE genome, and a match count >= 2^32, are beyond the 32-bit layout
    # (against different genomes: rebased per sweep segment, tests/test_gpu_wide.py)
    with pytest.raises(SwgError, match="sequence q that the mappings against one genome touch spans 2\\^32 bases or more \\(query_end on line 2"):
        PafFile(text="q\t9\t0\t5\t+\tt\t9\t2\t6\t3\t4\t0\nq\t9\t1\t4294967296\t+\tt\t9\t2\t6\t3\t4\t0\n")
    with pytest.raises(SwgError, match="matches >= 2\\^32 on line 1"):
        PafFile(text="q\t9\t0\t5\t+\tt\t9\t2\t6\t4294967296\t4\t0\n")
    with PafFile(text=EDGE_TEXT) as pf:
        with pytest.raises(ValueError):
            pf.write(tmp_path / "o", np.zeros(3, dtype=np.uint8))
        with pytest.raises(SwgError, match="cannot create"):
            pf.write(tmp_path / "no_such_dir" / "o", np.ones(9, dtype=np.uint8))


@pytest.mark.parametrize("threads", [1, 3])
def test_wide_coordinates_are_rebased_per_sequence(tmp_path, threads):
    """RecordMeta is u64 (src/paf_filter.rs:58-62).  A file with values >= 2^32 is parsed into 64-bit columns and every
    sequence's coordinates are taken relative to the smallest coordinate the sequence has anywhere (as query or target);
    everything else equals the oracle's extract_metadata."""
    rng = np.random.default_rng(5)
    base = {"gA#1#c1": 5_000_000_000, "gA#1#c2": 0, "gB#1#c1": 2**33 + 17, "gB#1#c2": 123, "gC#1#x": 2**40}
    names = list(base)
    lines = []
    for _ in range(3000):
        q, t = names[rng.integers(0, 5)], names[rng.integers(0, 5)]
        qs = base[q] + int(rng.integers(0, 3_000_000_000))
        ts = base[t] + int(rng.integers(0, 3_000_000_000))
        ql, tl = int(rng.integers(1, 50_000)), int(rng.integers(1, 50_000))
        m = int(rng.integers(1, ql + 1))
        lines.append(f"{q}\t99\t{qs}\t{qs + ql}\t{'+-'[int(rng.integers(0, 2))]}\t{t}\t99\t{ts}\t{ts + tl}\t{m}\t{max(ql, tl)}\t60")
    text = "\n".join(lines) + "\n"
    ref = orc.parse_paf_text(text)
    lo = {}
    for i in range(len(lines)):
        for nm, v in ((ref.qname[i], int(ref.qs[i])), (ref.tname[i], int(ref.ts[i]))):
            lo[nm] = min(lo.get(nm, v), v)
    with PafFile(text=text, threads=threads) as pf:
        assert pf.n == len(lines)
        nm = pf.names
        off = pf.seq_offsets
        assert off is not None and {nm[i]: int(off[i]) for i in range(len(nm))} == lo
        q = [nm[i] for i in pf.column("q_id")]
        t = [nm[i] for i in pf.column("t_id")]
        assert q == list(ref.qname) and t == list(ref.tname)
        oq = np.array([lo[x] for x in q], dtype=np.uint64)
        ot = np.array([lo[x] for x in t], dtype=np.uint64)
        assert np.array_equal(pf.column("q_start").astype(np.uint64) + oq, ref.qs)
        assert np.array_equal(pf.column("q_end").astype(np.uint64) + oq, ref.qe)
        assert np.array_equal(pf.column("t_start").astype(np.uint64) + ot, ref.ts)
        assert np.array_equal(pf.column("t_end").astype(np.uint64) + ot, ref.te)
        assert pf.is_rebased    # the handle says so, and gives RecordMeta's own values back
        for name, want in (("q_start", ref.qs), ("q_end", ref.qe), ("t_start", ref.ts), ("t_end", ref.te)):
            assert np.array_equal(pf.absolute(name), want)
        assert np.array_equal(pf.column("matches"), ref.matches) and np.array_equal(pf.column("block_len"), ref.block_length)
        assert np.array_equal(pf.column("identity"), ref.identity)
    with PafFile(text=text.replace("5000", "4", 1) if False else "\n".join(lines[:5]) + "\n") as pf:
        assert pf.seq_offsets is not None
    with PafFile(text="q\t9\t1\t5\t+\tt\t9\t2\t6\t3\t4\t0\n") as pf:
        assert pf.seq_offsets is None        # nothing reached 2^32: columns are the file's own values
        assert not pf.is_rebased and int(pf.absolute("q_start")[0]) == 1


def test_fuzz_slice_against_oracle(tmp_path):
    """A slice of tests/fuzz/fuzz_paf_io.py: hostile field values and tag soups, native ingest == oracle extract_metadata."""
    import subprocess
    import sys
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tests", "fuzz", "fuzz_paf_io.py"), "--minutes", "0.1", "--seed", "500000"],
                       capture_output=True, text=True, cwd=root)
    assert r.returncode == 0 and "'failures': 0" in r.stdout, r.stdout[-500:] + r.stderr[-500:]


def test_writing_onto_the_mapped_input(tmp_path):
    """--output-file equal to the input path: the writer must not truncate the file it has mapped (the reference filters
    into a temporary file first, src/main.rs:3630-3636).  The result replaces the input once it is complete."""
    rng = np.random.default_rng(8)
    rec = gen.random_records(rng, 5000)
    text = gen.records_to_paf(rng, rec, junk_lines=False)
    a, b = tmp_path / "a.paf", tmp_path / "b.paf"
    a.write_text(text)
    b.write_text(text)
    with PafFile(b) as pf:
        status = (rng.random(pf.n) < 0.5).astype(np.uint8) * 3
        want_n = pf.write(tmp_path / "want.out", status)
    for target in (a, tmp_path / "link.paf"):   # the same path, and another name of the same inode
        if target.name == "link.paf":
            os.link(a, target)
        with PafFile(a) as pf:
            assert pf.write(target, status) == want_n
        assert target.read_bytes() == (tmp_path / "want.out").read_bytes()
        assert not [p for p in os.listdir(tmp_path) if ".swg_tmp." in p]
        a.write_text(text)
        if target.name == "link.paf":
            os.unlink(target)
