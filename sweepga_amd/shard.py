"""Sharding one record set over several GPUs (one process per GPU), SURVEY.md §8(e).

Genome pairs are independent units of the filter: every sweep segment (paf_filter.rs:1037-1100), chain group
(:761-770), scaffold chromosome pair (plane_sweep_scaffold.rs:116-130) and rescue pair (paf_filter.rs:625-629)
nests inside one genome pair.  So the records are partitioned by genome pair, each rank filters its part with
its own swg_ctx, and no collective is needed on the data path -- only the final gather of per-record results.

Chain numbers (`ch:Z:chain_N`) are global in the reference: kept chains are numbered genome pair by genome
pair, in the order the genome pairs first appear (paf_filter.rs:517-521 over plane_sweep_scaffolds' output
order).  After the gather each shard-local number is shifted by the number of kept chains of all genome pairs
that appear earlier.  That shift is exact when both prefix rules of the reference agree on every name
(`prefix up to the last '#'` == `first two '#' parts`, true for PanSN `sample#hap#contig` and for names
without '#'); otherwise `plan()` refuses to shard and everything goes to rank 0.
"""
from dataclasses import dataclass
from typing import Callable, List, Optional

import numpy as np

from .filter import PackedRecords, SequenceIndex

COLS = ("q_id", "t_id", "q_start", "q_end", "t_start", "t_end", "identity", "matches", "block_len", "strand")


@dataclass
class ShardPlan:
    world: int
    shard_of_record: np.ndarray  # [n] int32
    pair_of_record: np.ndarray   # [n] dense genome-pair id
    n_pairs: int
    sharded: bool                # False: names do not allow exact renumbering -> everything on rank 0


def names_allow_sharding(index: SequenceIndex) -> bool:
    return all(SequenceIndex.prefix_last(nm) == SequenceIndex.prefix_two(nm) for nm in index.names)


def lpt(counts: np.ndarray, world: int) -> np.ndarray:
    """Longest-processing-time bin packing (deterministic: largest first, ties by id): shard of every unit."""
    shard_of = np.zeros(len(counts), dtype=np.int32)
    load = np.zeros(world, dtype=np.int64)
    order = np.lexsort((np.arange(len(counts)), -np.asarray(counts, dtype=np.int64)))
    for p in order:
        if counts[p] == 0:
            continue
        r = int(np.argmin(load))
        shard_of[p] = r
        load[r] += counts[p]
    return shard_of


def plan(packed: PackedRecords, world: int) -> ShardPlan:
    """Longest-processing-time bin packing of genome pairs by mapping count (deterministic)."""
    n = packed.n
    g2 = packed.seq_genome_two
    key = g2[packed.cols["q_id"]].astype(np.uint64) * np.uint64(packed.n_genome_two) + g2[packed.cols["t_id"]].astype(np.uint64)
    uniq, inv, counts = np.unique(key, return_inverse=True, return_counts=True)
    ok = world > 1 and packed.index is not None and names_allow_sharding(packed.index)
    shard_of_pair = lpt(counts, world) if ok else np.zeros(len(uniq), dtype=np.int32)
    return ShardPlan(world, shard_of_pair[inv].astype(np.int32) if n else np.zeros(0, np.int32), inv.astype(np.int64),
                     len(uniq), bool(ok))


def plan_dense(q_id: np.ndarray, t_id: np.ndarray, seq_genome_two: np.ndarray, n_genome_two: int, world: int):
    """The same plan without sorting the records: genome-pair key = genome(q) * G + genome(t) counted into a dense G x G
    table (for the 10^8-record sets of bench.py --scaling strong; G^2 must be small).  Returns (shard of every key,
    mappings of every key)."""
    G = int(n_genome_two)
    if G * G > (1 << 26):
        raise ValueError("plan_dense: genome-pair table too large, use plan()")
    g2 = np.asarray(seq_genome_two, dtype=np.int64)
    key = g2[q_id] * G + g2[t_id]
    counts = np.bincount(key, minlength=G * G).astype(np.int64)
    return lpt(counts, world), counts


def pair_chain_ranges(chain: np.ndarray, pair: np.ndarray, record_index: np.ndarray, n_pairs: int, n_total: int, retained=None):
    """Per genome pair, over the records of ONE shard: lowest / highest shard-local chain number and the first retained
    record (global index).  Pairs without records keep (int64 max, 0, n_total), so the per-pair results of several shards
    combine with min / max / min (every pair lives on exactly one shard)."""
    big = np.iinfo(np.int64).max
    lo = np.full(n_pairs, big, dtype=np.int64)
    hi = np.zeros(n_pairs, dtype=np.int64)
    first = np.full(n_pairs, n_total, dtype=np.int64)
    has = chain != 0
    np.minimum.at(lo, pair[has], chain[has].astype(np.int64))
    np.maximum.at(hi, pair[has], chain[has].astype(np.int64))
    if retained is None:
        np.minimum.at(first, pair, record_index.astype(np.int64))
    else:
        np.minimum.at(first, pair[retained], record_index[retained].astype(np.int64))
    return lo, hi, first


def chain_shifts(lo: np.ndarray, hi: np.ndarray, first: np.ndarray) -> np.ndarray:
    """What to add to a pair's shard-local chain numbers to make them global: kept chains are numbered genome pair by
    genome pair in the order the pairs first appear (src/paf_filter.rs:517-521 over plane_sweep_scaffolds' output order),
    and inside a shard a pair's kept chains are the contiguous numbers lo..hi."""
    shift = np.zeros(len(lo), dtype=np.int64)
    with_chains = np.nonzero(hi > 0)[0]
    order = with_chains[np.argsort(first[with_chains], kind="stable")]
    counts = hi[order] - lo[order] + 1
    offsets = np.concatenate(([0], np.cumsum(counts)[:-1])) if len(order) else np.zeros(0, dtype=np.int64)
    shift[order] = offsets - (lo[order] - 1)
    return shift


def subset(packed: PackedRecords, idx: np.ndarray) -> PackedRecords:
    cols = {k: np.ascontiguousarray(packed.cols[k][idx]) for k in COLS}
    return PackedRecords(len(idx), cols, packed.n_seq, packed.seq_genome_last, packed.n_genome_last,
                         packed.seq_genome_two, packed.n_genome_two, packed.index, packed.wide)


def retained_mask(packed: PackedRecords, min_block_length: int, min_identity: float, keep_self: bool) -> np.ndarray:
    """Step-1 retain predicate (paf_filter.rs:384-388), needed for genome-pair first appearance."""
    c = packed.cols
    m = (c["block_len"].astype(np.uint64) >= np.uint64(min_block_length)) & (c["identity"] >= min_identity)
    if not keep_self:
        m &= c["q_id"] != c["t_id"]
    return m


def merge(packed: PackedRecords, pl: ShardPlan, parts: List[tuple], retained: np.ndarray):
    """parts[r] = (record indices of shard r, status, chain) -> global (status[n], chain[n])."""
    n = packed.n
    status = np.zeros(n, dtype=np.uint8)
    chain = np.zeros(n, dtype=np.uint32)
    for idx, st, ch in parts:
        status[idx] = st
        chain[idx] = ch
    if not pl.sharded or not chain.any():
        return status, chain  # one shard: the numbers are already global
    pair = pl.pair_of_record
    has = chain != 0
    lo, hi, first = pair_chain_ranges(chain, pair, np.arange(n), pl.n_pairs, n, retained)
    shift = chain_shifts(lo, hi, first)
    chain[has] = (chain[has].astype(np.int64) + shift[pair[has]]).astype(np.uint32)
    return status, chain


def filter_sharded(packed: PackedRecords, filter_fn: Callable[[PackedRecords], tuple], rank: int, world: int,
                   min_block_length: int, min_identity: float, keep_self: bool,
                   all_gather: Optional[Callable[[object], list]] = None):
    """Every rank calls this with the same `packed`; rank r filters shard r with `filter_fn` (in production
    PafFilter.filter_columns on the rank's GPU) and the per-shard results are exchanged with `all_gather`
    (torch.distributed.all_gather_object in production).  Returns global (status, chain) on every rank."""
    pl = plan(packed, world)
    mine = np.nonzero(pl.shard_of_record == rank)[0]
    if len(mine):
        st, ch = filter_fn(subset(packed, mine))
        part = (mine, np.asarray(st, dtype=np.uint8), np.asarray(ch, dtype=np.uint32))
    else:
        part = (mine, np.zeros(0, np.uint8), np.zeros(0, np.uint32))
    parts = all_gather(part) if (all_gather is not None and world > 1) else [part]
    return merge(packed, pl, parts, retained_mask(packed, min_block_length, min_identity, keep_self))
