"""Gap limits at the borders of the chaining kernels' arithmetic (src/paf_filter.rs:786-839): 46340 / 46341 (q^2 + r^2 below /
beyond 2^32), 2^22 and 2^22 + 1 (the last limit whose candidates the fused walk keeps as packed 64-bit keys, and the first
that takes its per-lane lists; SWG_WALK_PLAIN=1: those lists for every limit), 2^31 -+ 1 (the early cut of the deep candidate scan), 2^32 + 5 and 2^63 (beyond any 32-bit coordinate: the limit
cannot bind), u64::MAX (`max_gap + 1` wraps to 0 in release Rust: an overlap beyond a fifth of the limit is distance 0).  One
deep chromosome pair per strand with records placed exactly at, one short of and one beyond every border, against the oracle,
under every path that evaluates d(i, j): the pair-resident walk (default), and the global-sort stage with its per-lane lists,
with the wavefront-per-element candidate kernel forced (SWG_CHAIN_DEEP=1: its three batch loops are chosen by the limit) and
with that kernel's generic loop (SWG_CAND_GENERIC=1).  -m gpu only; the knobs are read once per process: one subprocess each."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GAPS = [46_340, 46_341, 2**22, 2**22 + 1, 2**31 - 1, 2**31, 2**32 + 5, 2**63, 2**64 - 1, 50_000]


def border_records(seed=5):
    """(orc.Records) two sequence pairs ('+' and '-' dominated), ~3,000 records each over 4.2e9 bases: random background with
    spacings around 10^6 (so that limits of 2^31 and more make windows of thousands of records), and for every border value G a
    cluster whose consecutive records are G - 1, G, G + 1 apart on the query and/or the target axis, and overlapping ones whose
    overlap is G / 5 and G / 5 + 1."""
    from tests import orc
    rng = np.random.default_rng(seed)
    rows = []   # (pair, qs, qe, ts, te, strand)
    for pair, minus_frac in ((0, 0.05), (1, 0.95)):
        n_bg = 2_500
        qs = np.sort(rng.integers(0, 4_100_000_000, n_bg))
        ln = rng.integers(200, 30_000, n_bg)
        for k in range(n_bg):
            st = "-" if rng.random() < minus_frac else "+"
            t0 = int(qs[k]) + int(rng.integers(-200_000, 200_000)) if st == "+" else 4_150_000_000 - int(qs[k]) + int(rng.integers(-200_000, 200_000))
            t0 = min(max(t0, 0), 4_200_000_000)
            rows.append((pair, int(qs[k]), int(qs[k] + ln[k]), t0, t0 + int(ln[k]), st))
        base = 50_000_000
        for G in (46_340, 46_341, 2**22, 2**22 + 1, 2**31 - 1, 2**31, 50_000):
            for st in ("+", "-"):
                for dq, dt in ((G - 1, G - 1), (G, G), (G + 1, G), (G, G + 1), (G, 0), (0, G), (G + 1, 0)):
                    a_qs, a_qe = base, base + 1_000
                    b_qs = a_qe + dq
                    if st == "+":
                        a_ts, a_te = base, base + 1_000
                        b_ts = a_te + dt
                        rows.append((pair, a_qs, a_qe, a_ts, a_te, st))
                        rows.append((pair, b_qs, b_qs + 1_000, b_ts, b_ts + 1_000, st))
                    else:   # '-': the successor lies BEFORE its predecessor on the target (t_start[i] - t_end[j], :824-833)
                        b_ts, b_te = base, base + 1_000
                        a_ts = b_te + dt
                        rows.append((pair, a_qs, a_qe, a_ts, a_ts + 1_000, st))
                        rows.append((pair, b_qs, b_qs + 1_000, b_ts, b_te, st))
                    base += 9_000
                # overlaps of exactly G / 5 and G / 5 + 1 on the query axis
                for ov in (G // 5, G // 5 + 1):
                    L = ov + 5_000
                    rows.append((pair, base, base + L, base, base + L, "+"))
                    rows.append((pair, base + L - ov, base + 2 * L, base + L + 10, base + 2 * L + 10, "+"))
                    base += 11_000
    rows.sort(key=lambda r: r[0])   # pair-major, input order otherwise as generated (not sorted by q_start)
    perm = np.concatenate([rng.permutation([k for k, r in enumerate(rows) if r[0] == p]) for p in (0, 1)])
    rows = [rows[int(k)] for k in perm]
    n = len(rows)
    u = lambda v: np.ascontiguousarray(np.asarray(v, dtype=np.uint64))
    qn = [f"a{r[0]}#1#c" for r in rows]
    tn = [f"b{r[0]}#1#c" for r in rows]
    block = u([r[2] - r[1] for r in rows])
    ident = np.round(rng.uniform(0.8, 1.0, n), 3)
    matches = np.floor(ident * block).astype(np.uint64)
    return orc.Records(qn, tn, u([r[1] for r in rows]), u([r[2] for r in rows]), u([r[3] for r in rows]), u([r[4] for r in rows]), block,
                       np.ascontiguousarray(matches / np.maximum(block, 1), dtype=np.float64), u(matches),
                       np.array([ord(r[5]) for r in rows], dtype=np.uint8), u(np.arange(n)))


CODE = r"""
import sys
sys.path.insert(0, %(root)r)
import numpy as np
import sweepga_amd as sw
from tests import gen, orc
from tests.test_gpu_gap_limits import border_records, GAPS
rec = border_records()
packed = sw.pack_records(gen.records_to_meta(rec))
ctx = sw.default_context(0)
for gap in GAPS:
    for mlen in (0, 20_000):
        ctx.profile_reset(); ctx.profile(True)
        st, ch = sw.PafFilter(sw.FilterConfig(scaffold_gap=gap, min_scaffold_length=mlen)).filter_columns(packed)
        ctx.profile(False)
        ost, och = orc.apply_filters(orc.Config(scaffold_gap=gap, min_scaffold_length=mlen), rec)
        bad = np.flatnonzero((st != ost) | (ch != och))
        assert bad.size == 0, (gap, mlen, int(bad.size), bad[:8].tolist(), sorted(ctx.profile_table()))
        took_pairs = "pair_renumber" in ctx.profile_table() and "chain_cuts" not in ctx.profile_table() and "cuts_from_scan" not in ctx.profile_table()
        assert took_pairs == %(pairs)r, (gap, sorted(ctx.profile_table()))
        if %(deep)r:
            assert "chain_candidates_wave" in ctx.profile_table(), sorted(ctx.profile_table())
print("ok", len(rec))
"""


@pytest.mark.parametrize("name,env,pairs,deep", [
    ("pair_path", {}, True, False),
    ("pair_path_plain_lists", {"SWG_WALK_PLAIN": "1"}, True, False),
    ("global_path", {"SWG_GROUP_FUSED": "0"}, False, False),
    ("global_path_plain_lists", {"SWG_GROUP_FUSED": "0", "SWG_WALK_PLAIN": "1"}, False, False),
    ("deep_candidates", {"SWG_GROUP_FUSED": "0", "SWG_CHAIN_DEEP": "1"}, False, True),
    ("deep_candidates_generic", {"SWG_GROUP_FUSED": "0", "SWG_CHAIN_DEEP": "1", "SWG_CAND_GENERIC": "1"}, False, True),
])
def test_gap_limit_borders(name, env, pairs, deep):
    code = CODE % dict(root=ROOT, pairs=pairs, deep=deep)
    out = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, **env), capture_output=True, text=True, cwd=ROOT)
    assert out.returncode == 0 and "ok" in out.stdout, (name, out.stderr[-3000:])
