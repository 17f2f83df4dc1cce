"""Tree sparsification of a PAF before the filter (--sparsify tree:<near>[:<far>[:<random>]], src/tree_filter.rs:13-285,
src/main.rs:3640-3688): hand-derived answers for the oracle's restatement, the product (swg_paf_tree_filter, host code)
against the oracle on random PAFs, and the hash of the random selection (SipHash-1-3 = DefaultHasher) against an
independent Python implementation checked on the SipHash paper's own vector.  The reference holds no vector for this pass
beyond extract_genome_prefix (src/tree_filter.rs:446-451): parity unpinned; ties in identity fall to the neighbour's prefix
in ascending order (the reference's order is arbitrary there)."""
import ctypes as C

import numpy as np
import pytest

from tests import orc

M64 = (1 << 64) - 1


def _rotl(x, b):
    return ((x << b) | (x >> (64 - b))) & M64


def siphash(c, d, k0, k1, data):
    v0, v1, v2, v3 = k0 ^ 0x736f6d6570736575, k1 ^ 0x646f72616e646f6d, k0 ^ 0x6c7967656e657261, k1 ^ 0x7465646279746573

    def rnd(v0, v1, v2, v3):
        v0 = (v0 + v1) & M64; v1 = _rotl(v1, 13) ^ v0; v0 = _rotl(v0, 32)
        v2 = (v2 + v3) & M64; v3 = _rotl(v3, 16) ^ v2
        v0 = (v0 + v3) & M64; v3 = _rotl(v3, 21) ^ v0
        v2 = (v2 + v1) & M64; v1 = _rotl(v1, 17) ^ v2; v2 = _rotl(v2, 32)
        return v0, v1, v2, v3

    n = len(data)
    for i in range(0, n - n % 8, 8):
        m = int.from_bytes(data[i:i + 8], "little")
        v3 ^= m
        for _ in range(c):
            v0, v1, v2, v3 = rnd(v0, v1, v2, v3)
        v0 ^= m
    b = ((n & 0xff) << 56) | int.from_bytes(data[n - n % 8:], "little")
    v3 ^= b
    for _ in range(c):
        v0, v1, v2, v3 = rnd(v0, v1, v2, v3)
    v0 ^= b
    v2 ^= 0xff
    for _ in range(d):
        v0, v1, v2, v3 = rnd(v0, v1, v2, v3)
    return v0 ^ v1 ^ v2 ^ v3


def oracle_tree(text, kn, kf, rf):
    raw = text.encode()
    out = C.create_string_buffer(len(raw) + 16)
    f = orc.lib().orc_tree_filter_text
    f.restype = C.c_int64
    r = f(raw, C.c_uint64(len(raw)), C.c_uint64(kn), C.c_uint64(kf), C.c_double(rf), out, C.c_uint64(len(raw) + 16))
    assert r >= 0
    return out.raw[:r].decode()


def product_tree(text, kn, kf, rf):
    from sweepga_amd import _lib
    lib = _lib.load()
    raw = text.encode()
    p, n = C.c_void_p(), C.c_uint64()
    assert lib.swg_paf_tree_filter(raw, len(raw), kn, kf, rf, C.byref(p), C.byref(n)) == 0
    try:
        return C.string_at(p, n.value).decode()
    finally:
        lib.swg_free(p)


def line(q, t, matches, block):
    return f"{q}\t1000\t0\t{block}\t+\t{t}\t1000\t0\t{block}\t{matches}\t{block}\t60"


def test_default_hasher_is_siphash13():
    k = bytes(range(16))   # the SipHash paper's test vector (SipHash-2-4)
    assert siphash(2, 4, int.from_bytes(k[:8], "little"), int.from_bytes(k[8:], "little"), bytes(range(15))) == 0xa129ca6149be45e5
    f = orc.lib().orc_default_hash_str_pair
    f.restype = C.c_uint64
    f.argtypes = [C.c_char_p, C.c_char_p]
    for a, b in ((b"A#1#", b"B#1#"), (b"", b""), (b"HG002#1#", b"NA12878#2#"), (b"abcdefgh", b"ijklmnopq")):
        assert f(a, b) == siphash(1, 3, 0, 0, a + b"\xff" + b + b"\xff")   # impl Hash for str: bytes, then 0xff


# Four genomes.  Pair identities = sum(matches) / sum(block) over BOTH directions (src/tree_filter.rs:39-75):
#   A-B: (900 + 80) / (1000 + 100) = 0.8909     A-C: 700 / 1000 = 0.70     A-D: 500 / 1000 = 0.50
#   B-C: 950 / 1000 = 0.95                       B-D: 600 / 1000 = 0.60     C-D: (400 + 450) / 2000 = 0.425
# tree:1 -> nearest neighbour of A = B, of B = C, of C = B, of D = B (0.60 > 0.50 > 0.425): pairs {A-B, B-C, B-D}.
# tree:1:1 adds the farthest: A -> D (0.50), B -> D (0.60 is B's lowest), C -> D, D -> C: {A-D, B-D, C-D} on top.
# Lines of the same genome, '#' comments, empty and short lines never survive (src/tree_filter.rs:222-231, 176-181).
KAT = "\n".join([
    line("A#1#c1", "B#1#c1", 900, 1000),     # 0  A-B
    line("B#1#c1", "A#1#c2", 80, 100),       # 1  A-B (other direction, other chromosome)
    line("A#1#c1", "C#1#c1", 700, 1000),     # 2  A-C
    "# comment\tx\tx\tx\tx\tx\tx\tx\tx\tx\tx\tx",
    line("A#1#c1", "D#1#c1", 500, 1000),     # 4  A-D
    line("B#1#c1", "C#1#c1", 950, 1000),     # 5  B-C
    "",
    line("B#1#c2", "D#1#c1", 600, 1000),     # 7  B-D
    line("C#1#c1", "D#1#c1", 400, 1000),     # 8  C-D
    line("D#1#c1", "C#1#c1", 450, 1000),     # 9  C-D
    line("A#1#c1", "A#1#c2", 999, 1000),     # 10 same genome: never kept
    "too\tfew\tfields",
]) + "\n"
L = KAT.split("\n")


def test_oracle_hand_derived():
    keep = lambda idx: "".join(L[i] + "\n" for i in idx)
    assert oracle_tree(KAT, 1, 0, 0.0) == keep([0, 1, 5, 7])
    assert oracle_tree(KAT, 1, 1, 0.0) == keep([0, 1, 4, 5, 7, 8, 9])
    assert oracle_tree(KAT, 3, 0, 0.0) == keep([0, 1, 2, 4, 5, 7, 8, 9])      # every neighbour of everyone
    assert oracle_tree(KAT, 0, 1, 0.0) == keep([4, 7, 8, 9])                  # farthest only: A-D, B-D, C-D
    assert oracle_tree(KAT, 1, 0, 1.0) == keep([0, 1, 2, 4, 5, 7, 8, 9])      # random fraction 1: threshold u64::MAX, every pair
    # CRLF input: BufRead::lines strips it, the writer emits "\n"
    assert oracle_tree(KAT.replace("\n", "\r\n"), 1, 0, 0.0) == keep([0, 1, 5, 7])


def test_oracle_random_fraction_follows_the_hash():
    pairs = {("A#1#", "B#1#"): [0, 1], ("A#1#", "C#1#"): [2], ("A#1#", "D#1#"): [4], ("B#1#", "C#1#"): [5], ("B#1#", "D#1#"): [7],
             ("C#1#", "D#1#"): [8, 9]}
    for rf in (0.1, 0.37, 0.5, 0.9):
        thr = min(int(rf * float(1 << 64)), M64)
        want = set([0, 1, 5, 7])      # tree:1
        for (a, b), idx in pairs.items():
            if siphash(1, 3, 0, 0, a.encode() + b"\xff" + b.encode() + b"\xff") <= thr:
                want |= set(idx)
        assert oracle_tree(KAT, 1, 0, rf) == "".join(L[i] + "\n" for i in sorted(want))


@pytest.mark.parametrize("seed", range(6))
def test_product_matches_oracle(seed):
    rng = np.random.default_rng(seed)
    genomes = [f"g{i}#{h}#" for i in range(int(rng.integers(2, 9))) for h in (1, 2)][:int(rng.integers(2, 12))]
    names = [g + f"chr{c}" for g in genomes for c in range(3)] + ["plain1", "plain2", "x#y", "z#1"]
    rows = []
    for _ in range(int(rng.integers(1, 3000))):
        q, t = rng.choice(names, 2)
        b = int(rng.choice([100, 1000, 1000, 5000]))          # few distinct values -> identity ties between pairs
        m = int(b * rng.choice([0.5, 0.8, 0.8, 0.9, 0.95]))
        s = line(q, t, m, b)
        r = rng.random()
        if r < 0.02:
            s = "#" + s
        elif r < 0.04:
            s = "\t".join(s.split("\t")[:9])
        elif r < 0.06:
            s = s.replace(f"\t{m}\t{b}\t", "\tNaN\t\t")        # unparsable numbers: 0 and 1
        rows.append(s)
    text = "\n".join(rows) + ("\n" if rng.random() < 0.8 else "")
    for kn, kf, rf in ((1, 0, 0.0), (2, 1, 0.0), (0, 2, 0.0), (3, 3, 0.25), (1, 0, 0.6), (50, 0, 0.0)):
        assert product_tree(text, kn, kf, rf) == oracle_tree(text, kn, kf, rf), (seed, kn, kf, rf)
    assert product_tree("", 1, 0, 0.0) == ""
