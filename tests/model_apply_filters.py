"""A second, independent restatement of PafFilter::apply_filters (src/paf_filter.rs:379-747) in plain Python, written
from the Rust text and NOT from oracle/: dictionaries and loops, no shared helper with the C++ oracle.

Why: the reference cannot be built here (Rust), and the anchors / inversion capture / rescue / chain_N numbering steps
(:517-747) have no exact reference vector -- the C++ oracle and the device kernels were held to hand derivations by the
same author.  This model is the brute-force third opinion: tests/test_model_cpu.py runs it against the C++ oracle on
thousands of random small multi-genome record sets (status and chain numbers).

What it takes from the oracle: ONLY the three plane-sweep seams (plane_sweep_query / plane_sweep_target /
plane_sweep_scaffolds through tests/orc.py), which the reference's own unit tests pin exactly (tests/test_oracle_kat.py).
Everything else -- step-1 retain, genome-pair grouping and intersection (:972-1123), best-buddy chaining and union-find
(:750-933, src/union_find.rs), span / identity filter (:449-455), anchors (:517-528), inversion capture (:535-597),
never-rescued members (:601-604), rescue (:625-732) -- is restated here.

Rescued records: the reference walks the pair's anchors in HashSet order and stops at the first one within range, so any
anchor within range may donate its chain id.  The model returns the SET of admissible ids for them.
"""
import math

U64 = (1 << 64) - 1
DROPPED, SCAFFOLD, RESCUED, UNASSIGNED = 0, 1, 2, 3
ONE_TO_ONE, ONE_TO_MANY, MANY_TO_MANY = 0, 1, 2


def _prefix_last(name):  # :1022-1030: up to and including the last '#', else the whole name
    k = name.rfind("#")
    return name[:k + 1] if k >= 0 else name


class _UnionFind:  # src/union_find.rs:3-64
    def __init__(self, n):
        self.parent = list(range(n))
        self.rank = [0] * n

    def find(self, x):
        if self.parent[x] != x:
            self.parent[x] = self.find(self.parent[x])
        return self.parent[x]

    def union(self, x, y):
        rx, ry = self.find(x), self.find(y)
        if rx == ry:
            return
        if self.rank[rx] < self.rank[ry]:
            self.parent[rx] = ry
        elif self.rank[rx] > self.rank[ry]:
            self.parent[ry] = rx
        else:
            self.parent[ry] = rx
            self.rank[rx] += 1

    def sets(self):  # BTreeMap by root, members in ascending order
        by_root = {}
        for i in range(len(self.parent)):
            by_root.setdefault(self.find(i), []).append(i)
        return [by_root[r] for r in sorted(by_root)]


def _merge_mappings_into_chains(md, max_gap):
    """:750-933.  md: list of record dicts (the plane-swept metadata, in its order) -> chains in all_chains order."""
    groups = {}  # IndexMap: insertion order
    for idx, m in enumerate(md):
        groups.setdefault((m["q"], m["t"], m["strand"]), []).append(idx)
    chains = []
    for (q, t, strand), members in groups.items():
        srt = sorted(members, key=lambda i: md[i]["qs"])  # stable
        n = len(srt)
        best_pred_score = [U64] * n
        best_pred_idx = [None] * n
        for i in range(n):
            mi = md[srt[i]]
            bound = (mi["qe"] + max_gap) & U64
            best_j, best_score = None, U64
            for j in range(i + 1, n):
                mj = md[srt[j]]
                if mj["qs"] > bound:
                    break
                if mj["qs"] >= mi["qe"]:
                    q_gap = mj["qs"] - mi["qe"]
                else:
                    ov = mi["qe"] - mj["qs"]
                    q_gap = ov if ov <= max_gap // 5 else (max_gap + 1) & U64
                if strand == "+":
                    if mj["ts"] >= mi["te"]:
                        r_gap = mj["ts"] - mi["te"]
                    else:
                        ov = mi["te"] - mj["ts"]
                        r_gap = ov if ov <= max_gap // 5 else (max_gap + 1) & U64
                elif mi["ts"] >= mj["te"]:
                    r_gap = mi["ts"] - mj["te"]
                else:
                    ov = mj["te"] - mi["ts"]
                    r_gap = ov if ov <= max_gap // 5 else (max_gap + 1) & U64
                if q_gap <= max_gap and r_gap <= max_gap:
                    d = (q_gap * q_gap + r_gap * r_gap) & U64
                    if d < best_score and d < best_pred_score[j]:
                        best_score, best_j = d, j
            if best_j is not None:
                best_pred_score[best_j] = best_score
                best_pred_idx[best_j] = i
        uf = _UnionFind(n)
        for j in range(n):
            if best_pred_idx[j] is not None:
                uf.union(best_pred_idx[j], j)
        for members_s in uf.sets():
            ms = [md[srt[s]] for s in members_s]
            q_min, q_max = min(m["qs"] for m in ms), max(m["qe"] for m in ms)
            t_min, t_max = min(m["ts"] for m in ms), max(m["te"] for m in ms)
            sm, sb = sum(m["matches"] for m in ms), sum(m["block"] for m in ms)
            total = q_max - q_min
            gap_len = total - sb if total > sb else 0
            lcg = max(math.log(float(gap_len)), 0.0) if gap_len > 0 else 0.0
            eff = float(sb) + lcg
            wid = float(sm) / eff if eff > 0.0 else 0.0
            chains.append(dict(q=q, t=t, strand=strand, qs=q_min, qe=q_max, ts=t_min, te=t_max, total=total, wid=wid,
                               members=[m["rank"] for m in ms]))
    return chains


def _limits(mode, per_q, per_t):  # :1004-1014; None -> usize::MAX
    if mode == ONE_TO_ONE:
        return 1, 1
    if mode == ONE_TO_MANY:
        return (per_q or 1), (per_t or U64)
    return (per_q or U64), (per_t or U64)


def _plane_sweep_mappings(md, cfg, sweep_query, sweep_target):
    """:972-1123 with the two axis sweeps supplied by the caller (the pinned seams)."""
    if len(md) <= 1:
        return list(md)
    kq, kt = _limits(cfg["mapping_filter_mode"], cfg.get("mapping_max_per_query"), cfg.get("mapping_max_per_target"))
    pairs = {}
    for i, m in enumerate(md):
        pairs.setdefault((_prefix_last(m["q"]), _prefix_last(m["t"])), []).append(i)
    kept_all = []
    for idxs in pairs.values():
        q_kept, t_kept = set(), set()
        by_q, by_t = {}, {}
        for i in idxs:
            by_q.setdefault(md[i]["q"], []).append(i)
            by_t.setdefault(md[i]["t"], []).append(i)
        for sub in by_q.values():
            for k in sweep_query([(md[i]["qs"], md[i]["qe"], md[i]["ts"], md[i]["te"], md[i]["identity"]) for i in sub], kq):
                q_kept.add(sub[k])
        for sub in by_t.values():
            for k in sweep_target([(md[i]["qs"], md[i]["qe"], md[i]["ts"], md[i]["te"], md[i]["identity"]) for i in sub], kt):
                t_kept.add(sub[k])
        kept_all.extend(sorted(q_kept & t_kept))
    return [md[i] for i in kept_all]


def apply_filters(records, cfg, sweep_query, sweep_target, sweep_scaffolds):
    """records: list of dicts {rank, q, t, qs, qe, ts, te, block, identity, matches, strand ('+' / '-')}.
    cfg: dict with the FilterConfig fields used (names as in tests/orc.Config) + keep_self, scaffolds_only.
    -> {rank: (status, chain number or None, admissible chain numbers for a rescued record or None)}"""
    # 1. retain :384-388
    md = [m for m in records if m["block"] >= cfg["min_block_length"] and (cfg["keep_self"] or m["q"] != m["t"]) and
          m["identity"] >= cfg["min_identity"]]
    all_original = list(md)
    md = _plane_sweep_mappings(md, cfg, sweep_query, sweep_target)
    out = {}
    if cfg["scaffold_gap"] == 0:  # :409-434
        for m in md:
            out[m["rank"]] = (UNASSIGNED, None, None)
        return out
    chains = _merge_mappings_into_chains(md, cfg["scaffold_gap"])
    chains = [c for c in chains if c["total"] >= cfg["min_scaffold_length"] and c["wid"] >= cfg["min_scaffold_identity"]]
    pre_sweep = set(r for c in chains for r in c["members"])
    if len(chains) > 1:  # :1126-1146
        order = sweep_scaffolds([(c["q"], c["t"], c["qs"], c["qe"], c["ts"], c["te"], c["wid"]) for c in chains])
        chains = [chains[i] for i in order]
    if cfg["scaffolds_only"]:  # :486-513
        ranks = set(m["rank"] for m in all_original)
        for ci, c in enumerate(chains):
            for r in c["members"]:
                if r in ranks:
                    out[r] = (SCAFFOLD, ci + 1, None)  # HashMap::insert: a later chain overwrites (members are disjoint anyway)
        return out
    # step 4 :517-528
    anchor = {}
    for ci, c in enumerate(chains):
        for r in c["members"]:
            anchor[r] = ci + 1
    # step 4b :535-597
    gap = cfg["scaffold_gap"]
    reverse = {}
    for idx, m in enumerate(all_original):
        if m["strand"] == "-":
            reverse.setdefault((m["q"], m["t"]), []).append(idx)
    for ci, c in enumerate(chains):
        if c["strand"] != "+":
            continue
        diag = c["ts"] - c["qs"]
        for idx in reverse.get((c["q"], c["t"]), []):
            m = all_original[idx]
            if m["rank"] in anchor:
                continue
            ext_s = c["qs"] - gap if c["qs"] > gap else 0
            ext_e = min(c["qe"] + gap, U64)
            if m["qe"] < ext_s or m["qs"] > ext_e:
                continue
            qc, tc = (m["qs"] + m["qe"]) // 2, (m["ts"] + m["te"]) // 2
            deviation = abs(tc - qc - diag)
            pd = float(deviation) / math.sqrt(2.0)
            perp = U64 if pd >= 18446744073709551616.0 else int(pd)
            if perp <= gap:
                anchor[m["rank"]] = ci + 1
    never = pre_sweep - set(anchor)  # :601-604
    # step 5 :614-732
    by_pair = {}
    for idx, m in enumerate(all_original):
        by_pair.setdefault((m["q"], m["t"]), []).append(idx)
    anchors_of = {}
    for idx, m in enumerate(all_original):
        if m["rank"] in anchor:
            anchors_of.setdefault((m["q"], m["t"]), []).append(idx)
    D = cfg["scaffold_max_deviation"]
    for key, idxs in by_pair.items():
        pa = anchors_of.get(key, [])
        if not pa:
            continue
        for idx in idxs:  # the reference sorts by query_start first; the outcome per record does not depend on the order
            m = all_original[idx]
            if m["rank"] in anchor:
                out[m["rank"]] = (SCAFFOLD, anchor[m["rank"]], None)
            elif m["rank"] in never:
                continue
            elif D > 0:
                qc, tc = (m["qs"] + m["qe"]) // 2, (m["ts"] + m["te"]) // 2
                ok = []
                for a in pa:
                    am = all_original[a]
                    q_diff = abs(qc - (am["qs"] + am["qe"]) // 2)
                    if q_diff > D:
                        continue
                    t_diff = abs(tc - (am["ts"] + am["te"]) // 2)
                    dd = math.sqrt(float((q_diff * q_diff + t_diff * t_diff) & U64))
                    dist = U64 if dd >= 18446744073709551616.0 else int(dd)
                    if dist <= D:
                        ok.append(a)
                if ok:
                    # oracle / device instance: the lowest-index anchor in range; the reference: any of them
                    out[m["rank"]] = (RESCUED, anchor[all_original[min(ok)]["rank"]],
                                      set(anchor[all_original[a]["rank"]] for a in ok))
    return out
