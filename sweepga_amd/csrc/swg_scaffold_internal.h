// Shared declarations of the scaffold stage's translation units (internal):
//   swg_chain.hip           sort A, survivors, best-buddy predecessor selection     (src/paf_filter.rs:761-851)
//   swg_chain_table.hip     labelling, aggregates, all_chains order, span/identity   (src/union_find.rs, paf_filter.rs:854-933, 449-455)
//   swg_scaffold_sweep.hip  plane_sweep_scaffolds + chain numbering                  (src/plane_sweep_scaffold.rs:47-251)
//   swg_scaffold.hip        anchors, inversion capture, rescue, the stage driver     (src/paf_filter.rs:436-747)
//   swg_union_find.hip      UnionFind::get_sets seam                                 (src/union_find.rs:52-63)
#pragma once
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "swg_log.h"
#include "swg_pipeline.h"

namespace swg_scaf {

constexpr int EW = 256;
constexpr uint32_t NONE = 0xffffffffu;
inline unsigned nblk(uint64_t n) { return (unsigned)((n + EW - 1) / EW); }

// small fills; one copy per translation unit (internal linkage)
static __global__ __launch_bounds__(EW) void fill_u32_kernel(uint64_t n, uint32_t* __restrict__ p, uint32_t v) {
  uint64_t i = (uint64_t)blockIdx.x * EW + threadIdx.x;
  if (i < n) p[i] = v;
}
static __global__ __launch_bounds__(EW) void fill_u64_kernel(uint64_t n, uint64_t* __restrict__ p, uint64_t v) {
  uint64_t i = (uint64_t)blockIdx.x * EW + threadIdx.x;
  if (i < n) p[i] = v;
}
static __global__ __launch_bounds__(EW) void iota_u32_kernel(uint64_t n, uint32_t* __restrict__ p) {
  uint64_t i = (uint64_t)blockIdx.x * EW + threadIdx.x;
  if (i < n) p[i] = (uint32_t)i;
}

// The two floating-point tests of the stage as integer thresholds.  Both are monotone in their integer argument --
//   perpendicular distance  (u64)(deviation as f64 / SQRT_2) <= gap     (inversion capture, paf_filter.rs:570-580)
//   Euclidean distance      (u64)sqrt((q^2 + t^2) as f64)    <= D       (rescue, paf_filter.rs:686-718)
// (conversion, division by a positive constant, square root and truncation never decrease) -- so each holds exactly for the
// arguments up to a largest one, found here by bisection WITH THE SAME OPERATIONS on the same unit; the per-candidate loops
// then compare integers instead of an f64 division or square root per candidate (S-big1 `inversion` 2.1 -> 1.5 ms, S-pan c5
// `rescue` 2.61 -> 2.46 ms).  out[0]: largest deviation that passes, out[1]: largest q^2 + t^2.
static __global__ void fp_thresholds_kernel(uint64_t gap, uint64_t D, uint64_t* __restrict__ out) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  auto perp_ok = [&](uint64_t deviation) {
    const double pd = __ddiv_rn((double)deviation, 1.4142135623730951);
    const uint64_t perp = pd >= 18446744073709551616.0 ? ~0ull : (uint64_t)pd;
    return perp <= gap;
  };
  auto dist_ok = [&](uint64_t s2) {
    const double dd = __dsqrt_rn((double)s2);
    const uint64_t dist = dd >= 18446744073709551616.0 ? ~0ull : (uint64_t)dd;
    return dist <= D;
  };
  // largest x with ok(x); ok(0) holds (0 <= gap, 0 <= D)
  uint64_t res[2];
  for (int which = 0; which < 2; ++which) {
    auto ok = [&](uint64_t x) { return which == 0 ? perp_ok(x) : dist_ok(x); };
    uint64_t lo = 0, hi = ~0ull;  // ok(lo); hi: not known
    if (ok(hi)) {
      lo = hi;
    } else {
      while (hi - lo > 1) {  // ok(lo) && !ok(hi)
        const uint64_t mid = lo + ((hi - lo) >> 1);
        if (ok(mid))
          lo = mid;
        else
          hi = mid;
      }
    }
    res[which] = lo;
  }
  out[0] = res[0];
  out[1] = res[1];
}


// genome pair (gq, gt) -> u32, "first appearance" tables of the two prefix rules.  Up to 2^14 genomes: a dense G x G array
// (one load per lookup).  Beyond (names without '#': every contig its own genome): open addressing over the pairs that
// actually occur -- their number is bounded by the (query, target) groups the caller has already counted.
struct PairTable {
  uint32_t* dense;      // [G * G] or nullptr
  unsigned long long* keys;  // sparse: [mask + 1], ~0 = empty
  uint32_t* vals;       // sparse: [mask + 1]
  uint32_t mask;
  uint32_t n_genome;
};
__device__ __forceinline__ uint32_t* pair_slot(const PairTable& t, uint32_t gq, uint32_t gt) {  // inserts when absent
  if (t.dense) return t.dense + ((size_t)gq * t.n_genome + gt);
  const unsigned long long key = (unsigned long long)gq * t.n_genome + gt;
  uint32_t h = (uint32_t)((key * 0x9e3779b97f4a7c15ull) >> 32) & t.mask;
  for (;;) {
    unsigned long long k = __hip_atomic_load(&t.keys[h], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (k == ~0ull) {
      k = atomicCAS(&t.keys[h], ~0ull, key);
      if (k == ~0ull) k = key;
    }
    if (k == key) return t.vals + h;
    h = (h + 1) & t.mask;
  }
}
__device__ __forceinline__ uint32_t pair_get(const PairTable& t, uint32_t gq, uint32_t gt) {  // the pair is present
  if (t.dense) return t.dense[(size_t)gq * t.n_genome + gt];
  const unsigned long long key = (unsigned long long)gq * t.n_genome + gt;
  uint32_t h = (uint32_t)((key * 0x9e3779b97f4a7c15ull) >> 32) & t.mask;
  while (t.keys[h] != key) h = (h + 1) & t.mask;
  return t.vals[h];
}
constexpr uint64_t DENSE_PAIR_LIMIT = uint64_t(1) << 28;  // G * G entries

// Allocates (arena) and clears a PairTable for G genomes of which at most `bound` pairs occur.  (swg_chain_table.hip)
int pair_table_make(swg_ctx* ctx, uint32_t n_genome, uint64_t bound, PairTable* t);

struct ChainTable {
  uint64_t nc = 0;
  uint32_t *qid = nullptr, *tid = nullptr, *qs = nullptr, *qe = nullptr, *ts = nullptr, *te = nullptr;
  double* wid = nullptr;
};

struct ChainBuild {
  // sort A (all `alive` records)
  uint64_t M = 0;
  uint64_t* keyA = nullptr;
  uint32_t* idxA = nullptr;  // original index at A position
  uint32_t *a_qe = nullptr, *a_ts = nullptr, *a_te = nullptr, *a_dpair = nullptr;
  uint64_t n_pairs = 0;
  // survivors (members of chains)
  uint64_t m = 0;
  uint32_t* s_a = nullptr;      // A position
  uint32_t* s_idx = nullptr;    // original index
  uint32_t* s_chain = nullptr;  // index into T of the member's chain, NONE when that chain fails the span / identity filter
                                //   (only materialised when want_s_chain; otherwise chain_of_member() derives it)
  bool want_s_chain = true;
  const uint32_t *m_hd = nullptr, *m_rank_of = nullptr;  // per member: head position / per chain (position order): T index
  const uint8_t* m_ok_head = nullptr;                    // per member position: 1 = heads a chain that passes the filter
  const uint32_t* m_head_of_chain = nullptr;  // [T.nc] member position of the chain's head, in T order
  // chains that pass the span / identity filter (paf_filter.rs:449-455), in all_chains order: only these reach the scaffold
  // sweep, the numbering, the anchors; the others exist as a count
  uint64_t n_chains_all = 0;
  ChainTable T;
  uint8_t* C_strand = nullptr;
  uint32_t* C_dpair = nullptr;
};

// A range of consecutive members walked by one wavefront (swg_chain.hip): a chunk of whole short units (ue == be) or one
// block of a long unit (ue = end of the unit).
struct SpecBlock {
  uint32_t ue;  // end of the unit
  uint32_t bb;  // block begin
  uint32_t be;  // block end
  uint32_t pad;
};
constexpr int BIG_SPAN_SHIFT = 10;  // span_big's granularity = HEAD_SPAN = AGG_SPAN (swg_chain_table.hip)
constexpr uint32_t WALK_CHUNK = 1024;  // a chunk = the units that begin in one WALK_CHUNK-element cell ...
constexpr uint32_t BIG_UNIT = 8192;    // ... all shorter than this (longer units take the block-speculative path)

// What the chain table needs of a chain that passes the span / identity filter, written ONCE by whoever decides the filter at
// the chain's head into the head's slot of a sparse array: one 32-byte sector per passing chain.
struct __attribute__((aligned(32))) HeadRec {
  uint32_t qs, qe, ts, te;
  double wid;
  uint64_t grp;  // (query * n_seq + target) * 2 + strand (not written by the pair-resident path)
};
#ifdef __HIPCC__
// weighted identity of a chain (paf_filter.rs:896-913) from its aggregates
__device__ __forceinline__ double chain_weighted_identity(uint64_t total_length, uint64_t sm, uint64_t sb) {
  const uint64_t gap_length = total_length > sb ? total_length - sb : 0;  // saturating_sub
  double lcg = 0.0;
  if (gap_length > 0) {
    lcg = swg_log_glibc((double)gap_length);
    if (!(lcg > 0.0)) lcg = 0.0;  // .max(0.0)
  }
  const double eff = __dadd_rn((double)sb, lcg);
  return eff > 0.0 ? __ddiv_rn((double)sm, eff) : 0.0;
}
#endif
constexpr uint32_t LABEL_CAP_ELEMS = WALK_CHUNK + BIG_UNIT;  // chain_label_kernel: a chunk holds fewer elements than this

// Chunk-list launchers for the pair-resident path (swg_chain.hip, swg_chain_table.hip)
int pair_walk_launch(swg_ctx* ctx, uint32_t cap_chunks, const uint32_t* n_chunks_dev, const SpecBlock* desc, const uint32_t* s_qs,
                     const uint32_t* s_qe, const uint32_t* s_ts, const uint32_t* s_te, uint64_t max_gap, unsigned long long* bps,
                     uint32_t* pred);
int pair_walk_long_launch(swg_ctx* ctx, uint32_t cap_long, const uint32_t* n_long_dev, const uint32_t* long_list, const SpecBlock* chunks,
                          uint32_t n_members_cap, const uint32_t* s_qs, const uint32_t* s_qe, const uint32_t* s_ts, const uint32_t* s_te,
                          uint64_t max_gap, unsigned long long* own, uint32_t* pred, uint32_t* flags, uint32_t fallback_bit);
int pair_label_launch(swg_ctx* ctx, uint32_t cap_chunks, const uint32_t* n_chunks_dev, const SpecBlock* chunks, const uint32_t* pred,
                      const uint32_t* s_qs, const uint32_t* s_qe, const uint32_t* s_ts, const uint32_t* s_te, const uint32_t* s_m,
                      const uint32_t* s_b, uint64_t min_len, double min_ident, uint32_t* hd, uint8_t* ok_head, HeadRec* rec,
                      unsigned long long* n_heads, uint32_t cap_long, const uint32_t* n_long_dev, const uint32_t* long_list);
// (cap_long / n_long_dev / long_list: the chunks of LABEL_CAP_ELEMS members and more, labelled one work-group each)


// What the predecessor selection (swg_chain.hip) hands to the chain table (swg_chain_table.hip): the members of sort A in
// A order (`s_*`, m entries), their (query, target, strand) groups, and pred[p] = best-buddy predecessor of p (NONE = head).
struct ChainWork {
  uint64_t n_groups = 0;
  uint32_t *s_qs = nullptr, *s_qe = nullptr, *s_ts = nullptr, *s_te = nullptr, *s_m = nullptr, *s_b = nullptr;
  uint64_t* s_grp = nullptr;        // (q * n_seq + t) * 2 + strand
  uint32_t* head_flag = nullptr;    // 1 = first member of its group
  uint32_t* s_gidx = nullptr;       // dense group index
  uint32_t* group_begin = nullptr;  // [n_groups]
  uint32_t* pred = nullptr;
  uint64_t* d_tot = nullptr;        // 4 device scalars
  // chunks of whole short units (chains never leave a unit, so a chunk's chains are complete inside it); the members of
  // long units are flagged in big_member (nullptr: there are none).  n_chunks == 0: no chunk list (the round-2 selection).
  const SpecBlock* chunks = nullptr;
  uint64_t n_chunks = 0;
  const uint8_t* big_member = nullptr;
  const uint8_t* span_big = nullptr;  // [m / 1024 + 1] whether a 1024-element span holds a member of a long unit (with big_member)
};

// merge_mappings_into_chains (paf_filter.rs:750-933) in two halves:
//   chain_predecessors  : sort A over the alive records, members compacted in A order, best-buddy selection -> pred
//   chain_table_build   : chains = paths of pred; heads, aggregates, all_chains order, span / identity filter -> B.T, B.s_chain
// n_alive: number of set `alive` flags when the caller knows it (prepare counts them), ~0 otherwise.
// slots: the 32-byte record slots prepare wrote (matches / block length in their spare words), or nullptr.
int chain_predecessors(swg_ctx* ctx, const swg_records* r, const uint8_t* alive, const uint8_t* member, uint64_t max_gap,
                       int pos_bits, ChainBuild* out, ChainWork* work, const uint32_t* q_order, uint64_t n_alive,
                       const swg_key_ends* slots);
int chain_table_build(swg_ctx* ctx, const swg_records* r, const uint8_t* alive, uint64_t min_len, double min_ident,
                      bool genome_pair_major, ChainBuild* out, const ChainWork& work);
inline int build_chains(swg_ctx* ctx, const swg_records* r, const uint8_t* alive, const uint8_t* member, uint64_t max_gap,
                        uint64_t min_len, double min_ident, int pos_bits, bool genome_pair_major, ChainBuild* out,
                        const uint32_t* q_order = nullptr, uint64_t n_alive = ~0ull, const swg_key_ends* slots = nullptr) {
  ChainWork W;
  SWG_TRY(chain_predecessors(ctx, r, alive, member, max_gap, pos_bits, out, &W, q_order, n_alive, slots));
  if (out->M == 0 || out->m == 0) return SWG_OK;
  return chain_table_build(ctx, r, alive, min_len, min_ident, genome_pair_major, out, W);
}

// plane_sweep_scaffolds (plane_sweep_scaffold.rs:47-251) + chain numbering.  Chains are given in the
// reference's all_chains order (their index is the plane sweep's tie-break `idx`).
// Outputs: C_kept[c] (u8), C_num[c] (1-based position in the reference's output Vec, 0 if dropped).  (swg_scaffold_sweep.hip)
int scaffold_sweep_and_number(swg_ctx* ctx, const ChainTable& T, uint32_t n_seq, const uint32_t* seq_genome2,
                              uint32_t n_g2, int mode, uint64_t max_q, uint64_t max_t, double thr, int scoring,
                              int pos_bits, uint8_t* C_kept, uint32_t* C_num, uint64_t* n_kept_out);

// plane_sweep_both over chains in segments (swg_scaffold_sweep.hip): what scaffold_sweep_and_number runs before the numbering,
// and what the pair-resident path runs over its own chain table when the scaffold filter has limits.
int scaffold_sweep_segments(swg_ctx* ctx, uint64_t nc, const uint64_t* seg, int seg_bits, const uint32_t* qs, const uint32_t* qe,
                            const uint32_t* ts, const uint32_t* te, const double* wid, uint64_t kq, uint64_t kt, double thr, int scoring,
                            int pos_bits, uint8_t* kept, const void* runs = nullptr, uint32_t n_runs = 0);

// The scaffold stage for inputs grouped by chromosome pair (swg_pair.hip): one work-group per pair, the pair's members sorted
// inside LDS.  pair_plan finds the pairs (runs of the input, or a hash table for small inputs) and reads their number back;
// valid = 0: not applicable (not grouped, a pair too long ...).  The plan's device arrays live in the caller's arena frame, so
// it can be made before a mapping sweep (which then knows that nobody will ask for its sorted order) and used after it.
struct PairPlan {
  int valid = 0;
  bool by_hash = false;
  uint32_t n_runs = 0, ncls[4] = {0, 0, 0, 0}, cap = 0;
  void *counters = nullptr, *runs = nullptr, *class_list = nullptr, *perm = nullptr;
  bool not_grouped = false;        // why a plan over the runs of a large input is not valid: a pair has two runs, or there are more
                                   //   runs than pairs the path takes (both: what a grouping of the records by pair would cure)
  const uint32_t* orig = nullptr;  // the records are a pair-major COPY of the caller's: the caller's index of every record (ascending
                                   //   inside a pair), for everything that orders pairs by first appearance (pair_group_records)
};
// Large inputs that are not grouped by (query, target) pair: a stable sort of the record indices by pair and a pair-major copy
// of the columns, on the device (swg_pair.hip).  *ok = 0: not applicable (too many sequences for the sort key).  copy / perm live
// in the caller's arena frame.
int pair_group_records(swg_ctx* ctx, const swg_records* r, swg_records* copy, uint32_t** perm, int* ok);
int pair_ungroup_results(swg_ctx* ctx, uint64_t n, const uint32_t* perm, const uint8_t* st, const uint32_t* ch, uint8_t* status_out, uint32_t* chain_out);
int pair_plan(swg_ctx* ctx, const swg_records* r, const swg_config* cfg, PairPlan* plan);
// *taken = 0: not applicable, or left on a condition found on the device -- the caller runs the global-sort stage
// (swg_scaffold_stage's own path).  plan: from pair_plan, or nullptr (made here).
int scaffold_stage_pairs(swg_ctx* ctx, const swg_records* r, const swg_config* cfg, const uint8_t* alive, const uint8_t* member,
                         bool sweep_assumed_identity, uint8_t* status_out, uint32_t* chain_out, swg_stats* stats, int* taken,
                         const PairPlan* plan = nullptr);

}  // namespace swg_scaf
