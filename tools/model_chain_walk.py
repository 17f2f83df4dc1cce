#!/usr/bin/env python3
"""Executable model of chain_walk_kernel's batch logic (sweepga_amd/csrc/swg_chain.hip), checked against the reference's
sequential greedy (src/paf_filter.rs:784-851) on random instances.

The reference walks i = 0..n-1 in order; i takes the first j of its (d, j)-ordered valid list with d < score[j]; then
score[j] = d, pred[j] = i.  The kernel evaluates W consecutive i at once (W = 64 on the device; small here so that every
path is hit): acc bits against the scores before the batch, first acceptable candidate, re-evaluation in lane order of the
lanes that share a j with a lower lane (work list; a lane that moves to a new j wakes the higher lanes holding it), whole
window for a lane whose KC listed candidates are all refused while its window held more, commit by minimum.

Three rules for who starts on the work list (all must reproduce the sequential greedy):
  "count": every lane of a hash slot that more than one lane names (round 3);
  "low":   every lane of a slot except the LOWEST one (round 4, the chunks of short units) -- its choice stands unless a lower
           lane moves to its j, which wakes it;
  "pair":  as "low", and in a slot of exactly two lanes the higher one only if the lower one blocks it -- same j, distance
           not larger (round 4, the blocks of long units).
With "low" and "pair" a lane that moves wakes only the higher lanes it blocks (same j, distance not smaller than its own).

    python3 tools/model_chain_walk.py [cases]
"""
import random
import sys

KC = 4
INF = 1 << 62


def reference(n, valid):
    """valid[i] = list of (d, j) over the valid j of i's window (any order)."""
    score = [INF] * n
    pred = [-1] * n
    for i in range(n):
        best_d, best_j = INF, -1
        for d, j in sorted(valid[i], key=lambda x: x[1]):  # j ascending, strict < keeps the smaller j on ties
            if d < best_d and d < score[j]:
                best_d, best_j = d, j
        if best_j >= 0:
            score[best_j] = best_d
            pred[best_j] = i
    return pred


def batched(n, valid, W, hash_size, rule="count"):
    score = [INF] * n
    pred = [-1] * n
    stats = {"work": 0, "fallback": 0, "moved": 0}
    for i0 in range(0, n, W):
        lanes = list(range(i0, min(i0 + W, n)))
        L = len(lanes)
        lists, nv = [], []
        for i in lanes:
            v = sorted(valid[i])  # (d asc, j asc)
            lists.append(v[:KC])
            nv.append(min(len(v), KC + 1))
        acc = [[d < score[j] for d, j in lists[l]] for l in range(L)]
        fin = []
        for l in range(L):
            c = next((c for c in range(len(lists[l])) if acc[l][c]), None)
            fin.append(lists[l][c] if c is not None else None)
        # hash filter: lanes sharing a slot with another lane (superset of the lanes sharing a j)
        cnt = {}
        for l in range(L):
            if fin[l]:
                cnt[fin[l][1] % hash_size] = cnt.get(fin[l][1] % hash_size, 0) + 1
        low = {}
        for l in range(L):
            if fin[l]:
                low.setdefault(fin[l][1] % hash_size, l)  # lowest lane of the slot

        def starts(l):
            if fin[l] is None:
                return nv[l] > KC
            h = fin[l][1] % hash_size
            if rule == "count":
                return cnt[h] > 1
            if low[h] == l:
                return False
            if rule == "low":
                return True
            return cnt[h] > 2 or (fin[low[h]][1] == fin[l][1] and fin[low[h]][0] <= fin[l][0])

        work = set(l for l in range(L) if starts(l))
        committed = 0

        def commit(upto):
            nonlocal committed
            for l in range(committed, upto):  # atomic minimum; the lane holding the minimum is the predecessor
                if fin[l] and fin[l][0] < score[fin[l][1]]:
                    score[fin[l][1]] = fin[l][0]
            for l in range(committed, upto):
                if fin[l] and score[fin[l][1]] == fin[l][0]:
                    pred[fin[l][1]] = lanes[l]
            committed = upto

        while work:
            l = min(work)
            work.discard(l)
            stats["work"] += 1
            new = None
            for c in range(len(lists[l])):
                if not acc[l][c]:
                    continue
                d, j = lists[l][c]
                if not any(fin[k] and fin[k][1] == j and fin[k][0] <= d for k in range(l)):
                    new = (d, j)
                    break
            if new is None and nv[l] > KC:
                stats["fallback"] += 1
                commit(l)
                best = None
                for d, j in sorted(valid[lanes[l]], key=lambda x: x[1]):
                    if d < score[j] and (best is None or d < best[0]):
                        best = (d, j)
                new = best
            old_j = fin[l][1] if fin[l] else None
            fin[l] = new
            if new and new[1] != old_j:
                stats["moved"] += 1
                work |= set(k for k in range(l + 1, L)
                            if fin[k] and fin[k][1] == new[1] and (rule == "count" or fin[k][0] >= new[0]))
        commit(L)
    return pred, stats


def random_case(rng):
    n = rng.randint(1, 60)
    win = rng.randint(1, 12)
    dmax = rng.choice([3, 8, 50, 1000])  # small ranges force equal distances and many shared targets
    valid = []
    for i in range(n):
        v = []
        for j in range(i + 1, min(n, i + 1 + win)):
            if rng.random() < rng.choice([0.3, 0.7, 1.0]):
                v.append((rng.randint(0, dmax), j))
        valid.append(v)
    return n, valid


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
    rng = random.Random(12345)
    tot = {rule: {"work": 0, "fallback": 0, "moved": 0} for rule in ("count", "low", "pair")}
    for c in range(cases):
        n, valid = random_case(rng)
        want = reference(n, valid)
        for W in (1, 3, 8, 64):
            hs = rng.choice([1, 4, 256])
            for rule in ("count", "low", "pair"):
                got, st = batched(n, valid, W, hs, rule)
                if got != want:
                    print("MISMATCH case", c, "W", W, "rule", rule, n, valid, want, got)
                    return 1
                for k in st:
                    tot[rule][k] += st[k]
    for rule in ("count", "low", "pair"):
        t = tot[rule]
        print(f"{rule:5s}: {cases} cases x 4 widths equal to the sequential greedy; re-evaluated lanes {t['work']}, whole-window "
              f"passes {t['fallback']}, lanes that moved to a new j {t['moved']}")
    return 0


if __name__ == "__main__":
    sys.exit(main())
