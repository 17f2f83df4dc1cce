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


def segment_sweep_k1_resident(start, end, score, thr, cap_new=8, cmax=4):
    """The k = 1 sweep as csrc/swg_segsort.hip runs it segment-resident (round 6), step for step: the segment's intervals in
    (start, index) order come through LDS in batches of at most cap_new; a batch answers for the positions from its first start
    up to the next batch's first start, and the intervals that reach that far are carried over in front of the next batch (at
    most cmax: more than that -> None, the kernel hands the axis back to the tile kernels) with the flags they have so far.

    One thread per interval t, over the part [a, b) of its span that lies in the batch's range: among the intervals that
    intersect it (a window of slots: from the first slot whose running maximum of ends exceeds a to the last slot that starts
    before b), the BETTER ones (score, then start, then index) are walked in start order with the position `reach` up to which
    they cover [a, b) without a gap; every gap [g, h) is a stretch where t is the top of the active set -- t is marked, and every
    other interval active somewhere in [g, h) (it is worse than t) is tested against t: overlap fraction above the threshold ->
    the sticky `overlapped`.  kept = marked and not overlapped.  (g is always an event position: t's start, the batch's first
    start, or a better interval's end; the set of active intervals only changes at event positions.)
    Device semantics for malformed intervals: end <= start is never active.  Returns the sorted list of kept indices."""
    n = len(start)
    if n <= 1:
        return list(range(n))
    start = [int(x) for x in start]
    end = [int(x) for x in end]
    order = sorted(range(n), key=lambda i: (start[i], i))
    TOP, OVL = 1, 2
    keep = [False] * n
    # batches: consecutive stretches of the sorted order, cut only between different starts (a coarse bin never splits a key)
    batches, a = [], 0
    while a < n:
        b = min(a + cap_new, n)
        while b < n and start[order[b]] == start[order[b - 1]]:
            b -= 1
            if b == a:                      # (a run of equal starts longer than a batch: the dense-bin case, general sort)
                return None
        batches.append(order[a:b])
        a = b
    INF = 1 << 70
    carried = []                             # (interval, flags) in slot order
    for bi, batch in enumerate(batches):
        last = bi + 1 == len(batches)
        s_first = 0 if bi == 0 else start[batch[0]]
        s_next = INF if last else start[batches[bi + 1][0]]
        slots = [i for i, _ in carried] + batch
        F = [f for _, f in carried] + [0] * len(batch)
        nb = len(slots)
        K = [start[i] for i in slots]
        E = [end[i] for i in slots]
        KEY = [-score[i] for i in slots]
        PM, run = [], 0
        for p in range(nb):
            if E[p] > K[p]:
                run = max(run, E[p])
            PM.append(run)
        better = lambda a_, b_: (KEY[a_], a_) < (KEY[b_], b_)   # noqa: E731  slot order = (start, index) order

        for t in range(nb):                  # one thread per slot
            if not E[t] > K[t]:
                continue                     # never active
            a_t, b_t = max(K[t], s_first), min(E[t], s_next)
            if not b_t > a_t:
                continue
            lo = t
            while lo > 0 and PM[lo - 1] > a_t:
                lo -= 1
            hi = t
            while hi + 1 < nb and K[hi + 1] < b_t:
                hi += 1

            def stretch(g, h):
                F[t] |= TOP
                if thr < 1.0:
                    for x in range(lo, hi + 1):
                        if x != t and E[x] > K[x] and K[x] < h and E[x] > g and axis_overlap(K[x], E[x], K[t], E[t]) > thr:
                            F[x] |= OVL

            reach = a_t
            for j in range(lo, hi + 1):
                if reach >= b_t:
                    break
                if j != t and E[j] > K[j] and E[j] > reach and better(j, t):
                    if K[j] > reach:
                        stretch(reach, min(K[j], b_t))
                    reach = E[j]
            if reach < b_t:
                stretch(reach, b_t)
        carried = []
        for p in range(nb):
            if E[p] > K[p] and not last and E[p] > s_next:
                carried.append((slots[p], F[p]))
            else:
                keep[slots[p]] = bool(F[p] & TOP) and not (F[p] & OVL)
        if len(carried) > cmax:
            return None
    return [i for i in range(n) if keep[i]]
