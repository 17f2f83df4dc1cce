""".1aln record derivation (src/unified_filter.rs:83-142): hand-derived known answers for the oracle's restatement, and
the C ABI (swg_aln_open, host code) against the oracle on random decoded alignments.  The .1aln DECODER is the
reference's un-vendored fastga-rs / onecode dependency: parity of decoding is unpinned; this pins what the reference does
with a decoded alignment."""
import ctypes as C

import numpy as np
import pytest

from tests import orc


def oracle_records(qn, tn, qs, qe, ts, te, matches, strand, cap=256):
    n = len(qn)

    def u(a):
        return np.ascontiguousarray(np.asarray(a, dtype=np.uint64))

    qs, qe, ts, te, matches = map(u, (qs, qe, ts, te, matches))
    block = np.zeros(max(n, 1), dtype=np.uint64)
    ident = np.zeros(max(n, 1), dtype=np.float64)
    names = C.create_string_buffer(2 * max(n, 1) * cap)
    qa = (C.c_char_p * max(n, 1))(*[s.encode("utf-8") for s in qn])
    ta = (C.c_char_p * max(n, 1))(*[s.encode("utf-8") for s in tn])
    f = orc.lib().orc_records_from_1aln
    f.restype = C.c_int64
    r = f(C.c_uint64(n), qa, ta, orc._p(qs), orc._p(qe), orc._p(ts), orc._p(te), orc._p(matches),
          C.c_char_p(bytes(ord(c) for c in strand) or b"\0"), orc._p(block), orc._p(ident), names, C.c_uint64(cap))
    assert r == n
    raw = names.raw

    def get(k):
        return raw[k * cap:(k + 1) * cap].split(b"\0", 1)[0].decode("utf-8")

    return [get(i) for i in range(n)], [get(n + i) for i in range(n)], block[:n], ident[:n]


# (header, name after split_whitespace().next().unwrap_or(full)) -- src/unified_filter.rs:83-92
NAME_KAT = [
    ("chr1", "chr1"),
    ("HG002#1#chr1 Homo sapiens isolate", "HG002#1#chr1"),
    ("ctg7\tlen=500", "ctg7"),
    ("  lead", "lead"),                    # leading white space is skipped, not an empty first word
    ("", ""),                              # no word: unwrap_or(&full)
    (" \t ", " \t "),                      # white space only: the whole header is kept
    ("a\u00a0b", "a"),                     # U+00A0 NO-BREAK SPACE is White_Space
    ("a\u200bb", "a\u200bb"),              # U+200B ZERO WIDTH SPACE is not
    ("x\u3000y z", "x"),                   # U+3000 IDEOGRAPHIC SPACE
    ("n\u0085m", "n"),                     # U+0085 NEXT LINE
    ("q\u2028r", "q"),                     # U+2028 LINE SEPARATOR
    ("\u00e9t\u00e9 1", "\u00e9t\u00e9"),  # multi-byte characters that are not white space
]


def test_oracle_names_hand_derived():
    qn = [h for h, _ in NAME_KAT]
    z = np.zeros(len(qn), dtype=np.uint64)
    got_q, got_t, _, _ = oracle_records(qn, qn[::-1], z, z, z, z, z, "+" * len(qn))
    assert got_q == [w for _, w in NAME_KAT]
    assert got_t == [w for _, w in NAME_KAT][::-1]


def test_oracle_columns_hand_derived():
    """block_length = query_span + target_span (:107-112); identity = matches / query_span, 0.0 for an empty span
    (:119-123), NOT matches / block_length."""
    qs, qe = [100, 0, 5, 7], [1100, 10, 5, 4]          # spans 1000, 10, 0, and 4 - 7 wrapping
    ts, te = [2000, 50, 9, 0], [2900, 75, 19, 0]       # spans 900, 25, 10, 0
    matches = [950, 10, 3, 1]
    _, _, block, ident = oracle_records(["a"] * 4, ["b"] * 4, qs, qe, ts, te, matches, "+-+-")
    assert [int(x) for x in block] == [1900, 35, 10, (2**64 - 3)]
    assert ident[0] == 950 / 1000 and ident[1] == 1.0 and ident[2] == 0.0
    assert ident[3] == 1 / float(2**64 - 3)


def test_abi_names_hand_derived():
    from sweepga_amd import AlnRecords
    qn = [h for h, _ in NAME_KAT]
    z = np.zeros(len(qn), dtype=np.uint64)
    with AlnRecords(qn, qn, z, z, z, z, z, "+" * len(qn)) as a:
        names = a.names
        assert [names[i] for i in a.column("q_id")] == [w for _, w in NAME_KAT]


def test_abi_matches_oracle_on_random_alignments():
    from sweepga_amd import AlnRecords
    rng = np.random.default_rng(17)
    n = 4000
    heads = [f"g{g}#1#chr{c}" + rng.choice(["", " desc", "\tx y", " "]) for g in range(4) for c in range(3)]
    heads += ["  padded#1#c", "plain", "x y"]
    qn = [heads[i] for i in rng.integers(0, len(heads), n)]
    tn = [heads[i] for i in rng.integers(0, len(heads), n)]
    qs = rng.integers(0, 1_000_000, n)
    ql = rng.integers(0, 50_000, n)
    ts = rng.integers(0, 1_000_000, n)
    tl = rng.integers(0, 50_000, n)
    matches = (ql * rng.uniform(0.7, 1.0, n)).astype(np.int64)
    strand = "".join(rng.choice(["+", "-"], n))
    want_q, want_t, want_block, want_ident = oracle_records(qn, tn, qs, qs + ql, ts, ts + tl, matches, strand)
    with AlnRecords(qn, tn, qs, qs + ql, ts, ts + tl, matches, strand) as a:
        names = a.names
        assert [names[i] for i in a.column("q_id")] == want_q
        assert [names[i] for i in a.column("t_id")] == want_t
        assert np.array_equal(a.column("block_len").astype(np.uint64), want_block)
        assert np.array_equal(a.column("identity"), want_ident)           # bit-identical f64
        assert np.array_equal(a.column("q_start"), qs) and np.array_equal(a.column("t_end"), ts + tl)
        assert np.array_equal(a.column("matches"), matches)
        assert np.array_equal(a.column("strand"), np.array([0 if c == "+" else 1 for c in strand], dtype=np.uint8))
        # SequenceIndex order: first appearance, query before target of the same record
        seen, order = set(), []
        for q, t in zip(want_q, want_t):
            for nm in (q, t):
                if nm not in seen:
                    seen.add(nm)
                    order.append(nm)
        assert names == order


def test_abi_range_and_errors():
    from sweepga_amd import AlnRecords, SwgError
    with AlnRecords([], [], [], [], [], [], [], "") as a:
        assert a.n == 0
    with pytest.raises(SwgError, match="2\\^32"):
        AlnRecords(["a"], ["b"], [0], [2**32], [0], [5], [1], "+")
    # reversed coordinates wrap as in release Rust (:107-112): span 4 - 7 = 2^64 - 3, block = 2^64 - 3 + 5 = 2 (mod 2^64)
    with AlnRecords(["a"], ["b"], [7], [4], [0], [5], [1], "+") as a:
        assert int(a.column("block_len")[0]) == 2 and a.column("identity")[0] == 1 / float(2**64 - 3)
