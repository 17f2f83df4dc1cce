"""The arithmetic behind the fused walk's packed-key loop (sweepga_amd/csrc/swg_chain.hip, chain_walk_kernel, round 6), on the CPU:
a candidate (d, j - i) with d = q_gap^2 + r_gap^2 is ONE 64-bit key, bit 62 | d << 16 | (j - i), computed as
(q_gap << 8)^2 + (r_gap << 8)^2 + (bit 62 | j - i); read as an IEEE double it is a positive NORMAL number (bit 62 sets the exponent
field whatever d is), positive doubles order like their bit patterns, so a chain of min / max pairs on doubles keeps the four
smallest keys in (d, j) order -- what v_min_f64 / v_max_f64 do on the device -- and +infinity stands for a rejected pair.
Checked here against plain sorting for gaps up to the limit the device uses it for (2^22), including the borders."""
import numpy as np

KEY_BIT = np.uint64(1) << np.uint64(62)
KEY_INF = np.uint64(0x7FF0000000000000)
KC = 4


def pack(q_gap, r_gap, off):
    qa = (q_gap.astype(np.uint64) << np.uint64(8)) & np.uint64(0xFFFFFFFF)   # (32-bit registers on the device)
    ra = (r_gap.astype(np.uint64) << np.uint64(8)) & np.uint64(0xFFFFFFFF)
    return qa * qa + (ra * ra + (KEY_BIT | off.astype(np.uint64)))


def keep_four(keys):
    """the device's insertion: kb[c] = min(kb[c], t), t = max(kb[c], t), on doubles"""
    kb = np.full(KC, KEY_INF, dtype=np.uint64).view(np.float64)
    for k in keys.view(np.float64):
        t = k
        for c in range(KC):
            lo, hi = min(kb[c], t), max(kb[c], t)
            kb[c], t = lo, hi
    return kb.view(np.uint64)


def test_keys_are_normal_doubles_that_order_like_integers():
    rng = np.random.default_rng(1)
    G = 1 << 22
    q = np.concatenate([rng.integers(0, G + 1, 5000), [0, 0, G, G, 1, G - 1]]).astype(np.uint64)
    r = np.concatenate([rng.integers(0, G + 1, 5000), [0, G, 0, G, 1, G - 1]]).astype(np.uint64)
    off = np.concatenate([rng.integers(1, 1 << 16, 5000), [1, 65535, 1, 65535, 7, 9]]).astype(np.uint64)
    k = pack(q, r, off)
    d = q * q + r * r
    assert np.array_equal((k & ~KEY_BIT) >> np.uint64(16), d) and np.array_equal(k & np.uint64(0xFFFF), off)   # unpacking is exact
    assert int(k.max()) < int(KEY_INF)                                   # below +infinity's pattern
    f = k.view(np.float64)
    assert np.all(np.isfinite(f)) and np.all(f >= np.finfo(np.float64).tiny)   # positive normal numbers: no denormal mode matters
    o_int, o_flt = np.argsort(k, kind="stable"), np.argsort(f, kind="stable")
    assert np.array_equal(k[o_int], k[o_flt])                            # the same order either way


def test_the_min_max_chain_keeps_the_four_best_in_d_then_j_order():
    rng = np.random.default_rng(2)
    for trial in range(300):
        n = int(rng.integers(0, 40))
        gap = int(rng.choice([1, 500, 50_000, 1 << 22]))
        q = rng.integers(0, gap + 1, n).astype(np.uint64)
        r = rng.integers(0, gap + 1, n).astype(np.uint64)
        if n > 3 and trial % 3 == 0:      # ties on d: the earlier j must win
            q[1], r[1] = q[0], r[0]
            q[3], r[3] = r[0], q[0]
        ok = rng.random(n) < 0.7
        off = np.arange(1, n + 1, dtype=np.uint64)                       # j - i ascends with the scan
        keys = np.where(ok, pack(q, r, off), KEY_INF)
        got = keep_four(keys)
        want = sorted((int(q[x] * q[x] + r[x] * r[x]), int(off[x])) for x in range(n) if ok[x])[:KC]
        got_pairs = [((int(g) & ~int(KEY_BIT)) >> 16, int(g) & 0xFFFF) for g in got if g != KEY_INF]
        assert got_pairs == want, (trial, got_pairs, want)
        assert int((got == KEY_INF).sum()) == KC - len(want)
