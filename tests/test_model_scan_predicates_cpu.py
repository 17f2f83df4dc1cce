"""The algebra behind round 4's rewritten inner loops, checked on the CPU against the formulas they replace (which restate
src/paf_filter.rs:798-836, :570-580 and :686-718):

* a chaining gap as |a - b| with one limit per side (cand_scan_fast, the fused window loop of chain_walk_kernel) against the
  original select chain, for every kind of gap limit (small, > 2^32, u64::MAX);
* distances that fit 32 bits when the limit is at most 46340, and the early cut read from min(gap, 65535)^2;
* the scaffold stage's two floating-point tests as integer thresholds found by bisection (fp_thresholds_kernel): the same
  IEEE double operations in Python."""
import math
import random

M32 = (1 << 32) - 1
M64 = (1 << 64) - 1


def gap_original(a_j, b_i, max_gap):
    """(ok, gap) of one axis as the kernels computed it before: a_j = the later element's start, b_i = the earlier one's end."""
    gap32 = min(max_gap, M32)
    fifth32 = min(max_gap // 5, M32)
    wrap = max_gap == M64
    ge = a_j >= b_i
    ov = (b_i - a_j) & M32
    inn = ge or ov <= fifth32
    g = (a_j - b_i) if ge else (ov if inn else 0)
    return (inn or wrap) and g <= gap32, g


def gap_new(a_j, b_i, max_gap):
    gap32 = min(max_gap, M32)
    fifth32 = min(max_gap // 5, M32)
    g = abs(a_j - b_i)
    return g <= (gap32 if a_j >= b_i else fifth32), g


def test_gap_predicate_is_the_original_one():
    rng = random.Random(7)
    limits = [0, 1, 4, 5, 6, 1000, 46340, 46341, 50000, (1 << 31) - 1, 1 << 31, M32, M32 + 1, 5 * M32, 5 * M32 + 5, M64 - 1, M64]
    for _ in range(200_000):
        max_gap = rng.choice(limits)
        b = rng.choice([0, 1, 1000, 1 << 31, M32 - 3, M32, rng.randint(0, M32)])
        delta = rng.choice([0, 1, -1, max_gap // 5, max_gap // 5 + 1, -(max_gap // 5), -(max_gap // 5) - 1, max_gap, max_gap + 1,
                            rng.randint(-(1 << 20), 1 << 20), rng.randint(-M32, M32)])
        a = min(max(b + delta, 0), M32)
        ok0, g0 = gap_original(a, b, max_gap)
        ok1, g1 = gap_new(a, b, max_gap)
        assert ok0 == ok1, (a, b, max_gap)
        if ok0:
            assert g0 == g1, (a, b, max_gap)


def test_distances_fit_32_bits_up_to_46340_and_the_cut_reads_them():
    assert 2 * 46340 * 46340 < M32 and 2 * 46341 * 46341 > M32       # the largest distance stays below "empty"
    assert 65535 * 65535 > 2 * 46340 * 46340 and 65535 * 65535 < M32  # ... and below the capped square, which is not "empty"
    rng = random.Random(8)
    for _ in range(200_000):
        sd3 = rng.choice([0, 1, rng.randint(0, 2 * 46340 * 46340), 2 * 46340 * 46340])
        qg = rng.choice([0, 1, 46340, 65534, 65535, 65536, rng.randint(0, M32), math.isqrt(sd3), math.isqrt(sd3) + 1, max(math.isqrt(sd3) - 1, 0)])
        fast = min(qg, 65535) ** 2 >= sd3
        assert fast == (sd3 <= qg * qg), (sd3, qg)
        assert not (min(qg, 65535) ** 2 >= M32)                       # an empty list never cuts


def perp(deviation):
    pd = float(deviation) / 1.4142135623730951
    return M64 if pd >= 18446744073709551616.0 else int(pd)


def dist(s2):
    dd = math.sqrt(float(s2))
    return M64 if dd >= 18446744073709551616.0 else int(dd)


def largest_ok(ok):
    if ok(M64):
        return M64
    lo, hi = 0, M64
    while hi - lo > 1:
        mid = lo + ((hi - lo) >> 1)
        if ok(mid):
            lo = mid
        else:
            hi = mid
    return lo


def test_floating_point_distance_tests_as_integer_thresholds():
    rng = random.Random(9)
    for limit in [0, 1, 2, 7, 1000, 20_000, 50_000, 10**9, M32 - 1, M32, M32 + 1, 1 << 40, (1 << 53) + 1, 1 << 62, M64 - 1, M64]:
        for f in (perp, dist):
            t = largest_ok(lambda x: f(x) <= limit)
            assert f(t) <= limit and (t == M64 or f(t + 1) > limit)
            probes = [0, 1, t, max(t - 1, 0), min(t + 1, M64), min(t + 2, M64), t // 2, min(2 * t, M64), M64]
            probes += [rng.randint(0, M64) for _ in range(200)] + [min(max(t + rng.randint(-10**6, 10**6), 0), M64) for _ in range(200)]
            for x in probes:
                assert (f(x) <= limit) == (x <= t), (f.__name__, limit, x, t)
