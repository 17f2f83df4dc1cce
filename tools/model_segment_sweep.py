"""A model of the segment-resident plane sweep planned in DESIGN.md section 13 (not a product path: the kernels of
csrc/swg_sweep.hip are): the reference's axis sweep (src/plane_sweep_exact.rs:197-352) restated per event position over the
intervals in (start, index) order with the running maximum of their ends beside them -- the data a work-group would hold in LDS.

    kept(x)  <=>  at some event position P of its span x is among the k best of the active set (score descending, start,
                  index), and at no event position is it active, not among them, and overlapping one of them by more than
                  the threshold (FLAG_OVERLAPPED is sticky; DISCARD is only ever set again together with it).

Active at P (events at one position: begins before ends, all of them before the marking): start <= P and not end <= P -- an
interval whose end precedes its start (malformed) is never erased, a zero-length one never active.
tests/test_segment_sweep_model_cpu.py holds this to the oracle on random segments."""
import numpy as np


def axis_overlap(s0, e0, s1, e1):
    """plane_sweep_exact.rs:113-144 (u64 lengths wrap like the reference's subtraction)."""
    ov = max(0, min(e0, e1) - max(s0, s1))
    l0, l1 = (e0 - s0) % (1 << 64), (e1 - s1) % (1 << 64)
    mn = float(min(l0, l1))
    return float(ov) / mn if mn > 0.0 else 0.0


def segment_sweep(start, end, score, k, thr):
    """start / end: the axis' coordinates, score: higher is better (what score_with_function returns), k: how many to keep per
    position (None = no limit) -> sorted list of kept indices."""
    n = len(start)
    if n <= 1:
        return list(range(n))
    start = [int(x) for x in start]
    end = [int(x) for x in end]
    order = sorted(range(n), key=lambda i: (start[i], i))          # the LDS order: (start, index)
    s = [start[i] for i in order]
    e_eff = [end[i] if end[i] >= start[i] else float("inf") for i in order]   # (a malformed interval never ends)
    pm = list(np.maximum.accumulate(np.array([x if x != float("inf") else 2.0 ** 70 for x in e_eff], dtype=np.float64)))
    rank_key = lambda i: (-score[i], start[i], i)                    # noqa: E731  smaller = better
    marked = [False] * n
    overlapped = [False] * n
    positions = sorted(set(start) | set(end))
    lo = 0
    for P in positions:
        while lo < n and pm[lo] <= P:      # everything up to here has ended by P
            lo += 1
        active = []
        j = lo
        while j < n and s[j] <= P:
            i = order[j]
            if end[i] > P or end[i] < start[i]:
                active.append(i)
            j += 1
        if not active:
            continue
        active.sort(key=rank_key)
        top = active if k is None else active[:k]
        for i in top:
            marked[i] = True
        if thr < 1.0:
            tops = set(top)
            for i in active:
                if i in tops:
                    continue
                for t in top:
                    if axis_overlap(start[i], end[i], start[t], end[t]) > thr:
                        overlapped[i] = True
                        break
    return [i for i in range(n) if marked[i] and not overlapped[i]]
