"""The C++ oracle against an independent brute-force Python restatement of apply_filters (tests/model_apply_filters.py,
written from the Rust text): >= 10^4 random small multi-genome record sets x random flag sets, status and chain numbers.
This is the second pin of the scaffold tail (anchors, inversion capture, never-rescued members, rescue, chain_N numbering:
src/paf_filter.rs:517-747), whose only other ground truth is hand derivation (tests/kat_scaffold.py)."""
import numpy as np
import pytest

from tests import gen, model_apply_filters as model, orc


def _records(rec):
    return [dict(rank=int(rec.rank[i]), q=rec.qname[i], t=rec.tname[i], qs=int(rec.qs[i]), qe=int(rec.qe[i]), ts=int(rec.ts[i]),
                 te=int(rec.te[i]), block=int(rec.block_length[i]), identity=float(rec.identity[i]), matches=int(rec.matches[i]),
                 strand=chr(int(rec.strand[i]))) for i in range(len(rec))]


def _random_cfg(rng):
    kq = [None, 1, 2, 5][int(rng.integers(0, 4))]
    kt = [None, 1, 3][int(rng.integers(0, 3))]
    return dict(min_block_length=int(rng.choice([0, 0, 200])), mapping_filter_mode=int(rng.integers(0, 3)),
                mapping_max_per_query=kq, mapping_max_per_target=kt, scaffold_filter_mode=int(rng.integers(0, 3)),
                scaffold_max_per_query=None if rng.random() < 0.5 else int(rng.integers(1, 4)),
                scaffold_max_per_target=None if rng.random() < 0.5 else int(rng.integers(1, 4)),
                overlap_threshold=float(rng.choice([0.0, 0.3, 0.95, 1.0])),
                scaffold_gap=int(rng.choice([0, 1, 500, 3_000, 20_000, 10_000_000], p=[.06, .04, .2, .3, .3, .1])),
                min_scaffold_length=int(rng.choice([0, 500, 3_000])),
                scaffold_overlap_threshold=float(rng.choice([0.0, 0.5, 1.0])),
                scaffold_max_deviation=int(rng.choice([0, 1, 800, 5_000, 100_000])),
                scoring_function=int(rng.integers(0, 5)), min_identity=float(rng.choice([0.0, 0.0, 0.8])),
                min_scaffold_identity=float(rng.choice([0.0, 0.0, 0.85])), keep_self=bool(rng.random() < 0.3),
                scaffolds_only=bool(rng.random() < 0.1))


def _oracle_cfg(c):
    return orc.Config(min_block_length=c["min_block_length"], mapping_filter_mode=c["mapping_filter_mode"],
                      mapping_max_per_query=c["mapping_max_per_query"] or 0, mapping_max_per_target=c["mapping_max_per_target"] or 0,
                      scaffold_filter_mode=c["scaffold_filter_mode"], scaffold_max_per_query=c["scaffold_max_per_query"] or 0,
                      scaffold_max_per_target=c["scaffold_max_per_target"] or 0, overlap_threshold=c["overlap_threshold"],
                      scaffold_gap=c["scaffold_gap"], min_scaffold_length=c["min_scaffold_length"],
                      scaffold_overlap_threshold=c["scaffold_overlap_threshold"],
                      scaffold_max_deviation=c["scaffold_max_deviation"], scoring_function=c["scoring_function"],
                      min_identity=c["min_identity"], min_scaffold_identity=c["min_scaffold_identity"], keep_self=c["keep_self"],
                      scaffolds_only=c["scaffolds_only"])


def _compare(rec, cfg):
    """-> (#scaffold, #rescued, #captured '-' records) of the case; asserts equality."""
    sc = cfg["scoring_function"]
    got = model.apply_filters(
        _records(rec), cfg,
        lambda maps, k: orc.plane_sweep_query(maps, min(k, orc.K_INF), cfg["overlap_threshold"], sc),
        lambda maps, k: orc.plane_sweep_target(maps, min(k, orc.K_INF), cfg["overlap_threshold"], sc),
        lambda chains: orc.plane_sweep_scaffolds(chains, cfg["scaffold_filter_mode"], cfg["scaffold_max_per_query"],
                                                 cfg["scaffold_max_per_target"], cfg["scaffold_overlap_threshold"], sc))
    st, ch = orc.apply_filters(_oracle_cfg(cfg), rec)
    n_sc = n_re = 0
    for i in range(len(rec)):
        r = int(rec.rank[i])
        want = got.get(r)
        if want is None:
            assert st[i] == orc.DROPPED and ch[i] == 0, (i, st[i], ch[i])
            continue
        status, num, admissible = want
        assert st[i] == status, (i, st[i], status)
        if status == model.RESCUED:
            assert int(ch[i]) in admissible and int(ch[i]) == num, (i, ch[i], num, admissible)
            n_re += 1
        else:
            assert int(ch[i]) == (num or 0), (i, ch[i], num)
            n_sc += status == model.SCAFFOLD
    return n_sc, n_re


@pytest.mark.parametrize("seed", range(10))
def test_oracle_equals_the_python_model(seed):
    rng = np.random.default_rng(50_000 + seed)
    tot_sc = tot_re = 0
    for case in range(1000):
        n = int(rng.choice([1, 2, 5, 12, 30, 70]))
        span = int(rng.choice([3_000, 30_000, 300_000]))
        rec = gen.random_records(rng, n, n_genomes=int(rng.integers(1, 4)), chrs_per_genome=int(rng.integers(1, 3)), span=span,
                                 max_len=int(rng.choice([300, 3_000])), pansn=bool(rng.random() < 0.8),
                                 self_frac=float(rng.choice([0.0, 0.1])), minus_frac=float(rng.choice([0.0, 0.3, 0.6])),
                                 syntenic_frac=float(rng.choice([0.5, 0.95])), zero_frac=float(rng.choice([0.0, 0.05])))
        if rng.random() < 0.25:  # ties: coordinates on a grid, few identity levels
            g = int(rng.choice([50, 500]))
            for a in (rec.qs, rec.qe, rec.ts, rec.te):
                a[:] = a // g * g
            rec.identity[:] = rng.choice([0.8, 0.9], len(rec))
            rec.matches[:] = np.floor(rec.identity * rec.block_length).astype(np.uint64)
        cfg = _random_cfg(rng)
        try:
            sc, re = _compare(rec, cfg)
        except AssertionError as e:
            raise AssertionError(f"seed {seed} case {case} cfg {cfg}: {e}")
        tot_sc += sc
        tot_re += re
    assert tot_sc > 500 and tot_re > 20   # both outcomes were exercised


def test_model_on_the_hand_derived_scaffold_fixtures():
    """The paper derivations of tests/kat_scaffold.py hold for the Python model too (three implementations, one answer)."""
    from tests import kat_scaffold
    names = {"dropped": model.DROPPED, "scaffold": model.SCAFFOLD, "rescued": model.RESCUED}
    base = dict(min_block_length=0, mapping_filter_mode=model.MANY_TO_MANY, mapping_max_per_query=None,
                mapping_max_per_target=None, scaffold_filter_mode=model.MANY_TO_MANY, scaffold_max_per_query=None,
                scaffold_max_per_target=None, overlap_threshold=0.95, scaffold_gap=50_000, min_scaffold_length=10_000,
                scaffold_overlap_threshold=0.5, scaffold_max_deviation=0, scoring_function=orc.LOG_LENGTH_IDENTITY,
                min_identity=0.0, min_scaffold_identity=0.0, keep_self=False, scaffolds_only=False)
    sq = lambda maps, k: orc.plane_sweep_query(maps, min(k, orc.K_INF), cfg["overlap_threshold"])
    st = lambda maps, k: orc.plane_sweep_target(maps, min(k, orc.K_INF), cfg["overlap_threshold"])
    for case in kat_scaffold.CASES:
        cfg = dict(base)
        for k, v in case["cfg"].items():
            cfg[k] = {"OneToOne": model.ONE_TO_ONE, "OneToMany": model.ONE_TO_MANY, "ManyToMany": model.MANY_TO_MANY}[v] \
                if k.endswith("_mode") else v
        recs = [dict(rank=i, q=l[0], qs=l[1], qe=l[2], strand=l[3], t=l[4], ts=l[5], te=l[6], matches=l[7], block=l[8],
                     identity=l[7] / max(l[8], 1)) for i, l in enumerate(case["lines"])]
        ss = lambda chains: orc.plane_sweep_scaffolds(chains, cfg["scaffold_filter_mode"], cfg["scaffold_max_per_query"],
                                                      cfg["scaffold_max_per_target"], cfg["scaffold_overlap_threshold"])
        got = model.apply_filters(recs, cfg, sq, st, ss)
        for i, (status, num) in enumerate(case["expect"]):
            g = got.get(i, (model.DROPPED, None, None))
            assert (g[0], g[1] or 0) == (names[status], num), (case["name"], i, g, status, num)
