// Scaffold stage, pair-resident form (src/paf_filter.rs:436-747 for inputs whose records are grouped by chromosome pair).
//
// Everything the scaffold stage computes nests inside one (query sequence, target sequence) pair: the (query, target, strand)
// groups of merge_mappings_into_chains (paf_filter.rs:761-770), the per-chromosome-pair plane_sweep_both of
// plane_sweep_scaffolds (plane_sweep_scaffold.rs:116-183), the inversion capture (paf_filter.rs:539-597) and the rescue
// (:625-732).  Only the chain_N numbers are global, and those follow from three values per pair.  An aligner writes its PAF
// pair after pair, so a pair is one run of the input -- and its few thousand records fit the LDS of one work-group:
//
//   pair_boundary / pair_runs   the runs of equal (q_id, t_id), checked to be one per pair (else: the global-sort path)
//   pair_sort    one work-group per pair: step-1 retain (paf_filter.rs:384-388), members bucket-sorted in LDS by
//                (strand, q_start, index) (:777), the other columns transposed through LDS into that order (coalesced reads,
//                coalesced writes: no gather), unit cuts and the chunk list of the walk
//   chain_walk   swg_chain.hip's walk over that chunk list (best-buddy predecessors, :784-851)
//   chain_label  swg_chain_table.hip's per-chunk labelling, aggregates and span / identity filter (:854-933, 449-455)
//   pair_finish  one work-group per pair: plane_sweep_both over the pair's chains, pair-local chain numbers, anchors,
//                inversion capture (:535-597)
//   pair_number  chain_N bases: pairs ordered as plane_sweep_scaffolds emits them (genome pair -> chromosome pair, first
//                appearance, plane_sweep_scaffold.rs:116-130), a prefix sum of their kept chains, added in place
//
// The input is read once (47 B per record), every intermediate is addressed by the pair's own offset in the input (no global
// sort key, no compaction), and the host reads back twice: the run list's size, and at the end the statistics together with
// the "cannot be done here" flag (a unit too long for one chunk, a pair too dense for the LDS batches, a degenerate retained
// record under an unlimited mapping sweep) that sends the call to the global-sort path instead.
#include <cmath>
#include <type_traits>

#include "swg_scaffold_internal.h"

namespace swg_scaf {
namespace {

constexpr uint32_t PAIR_CELL = WALK_CHUNK;        // a chunk = the units that begin in one cell of this many members
constexpr uint32_t PAIR_S_MAX = 1024, PAIR_M_MAX = 4096, PAIR_L_MAX = 32768, PAIR_XL_MAX = uint32_t(1) << 18;
constexpr uint32_t PAIR_XL_CELLS = PAIR_XL_MAX / PAIR_CELL;
constexpr uint32_t PAIR_HASH_MAX = 65536;  // inputs up to this many records find their pairs through a hash table
constexpr uint32_t PF_NOT_GROUPED = 1, PF_RUN_OVERFLOW = 2, PF_TOO_LONG = 4, PF_FALLBACK = 8;

struct PairRun {
  uint32_t a, n;  // first record, number of records
};
struct PairInfo {
  uint32_t m;             // members, sorted into [a, a + m)
  uint32_t m_plus;        // '+' members (they come first)
  uint32_t M;             // alive records: the members, then the alive non-members at [a + m, a + M)
  uint32_t first_alive;   // smallest alive record index (NONE: none)
  uint32_t first_mem[2];  // smallest member index per strand (NONE: none)
  uint32_t q, t;
};
struct PairSum {
  uint32_t n_pass;  // chains that pass the span / identity filter
  uint32_t n_kept;  // ... and the scaffold sweep
  uint32_t minmem;  // first member of the first (query, target, strand) group that holds a passing chain
  uint32_t base;    // chain_N numbers of the pairs before this one (pair_number)
};
struct PairCounters {
  uint32_t n_runs;
  uint32_t flags;
  uint32_t n_class[4];
  uint32_t n_chunks;
  uint32_t n_long;
  unsigned long long n_alive, n_members, n_heads, n_kept, n_out;
};

// A work-group barrier that orders LDS accesses only.  __syncthreads() is a work-group fence in front of s_barrier -- it waits
// for EVERY outstanding vector-memory operation (s_waitcnt vmcnt(0)): for the global stores a phase has just issued to be
// acknowledged, and for loads requested ahead for the next phase.  The pair kernels only ever exchange data through LDS, so
// their barriers wait for the LDS queue alone (s_waitcnt lgkmcnt(0), gfx9 encoding) and global memory traffic stays in flight.
__device__ __forceinline__ void lds_barrier() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_s_waitcnt(0xc07f);
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// ---- block-wide scans (NT threads; ws: NT / 64 words of LDS scratch per call site) ---------------------------------
template <int NT>
__device__ __forceinline__ uint32_t block_excl_sum(uint32_t v, uint32_t* ws, uint32_t* total) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  uint32_t inc = v;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const uint32_t t = __shfl_up(inc, d, 64);
    if (lane >= d) inc += t;
  }
  if (NT == 64) {
    *total = __shfl(inc, 63, 64);
    return inc - v;
  }
  lds_barrier();
  if (lane == 63) ws[w] = inc;
  lds_barrier();
  uint32_t off = 0, tot = 0;
#pragma unroll
  for (int k = 0; k < NT / 64; ++k) {
    const uint32_t x = ws[k];
    off += k < w ? x : 0u;
    tot += x;
  }
  *total = tot;
  return off + inc - v;
}
// exclusive running maximum (0 before the first element); *total = maximum over the block
template <int NT>
__device__ __forceinline__ uint64_t block_excl_max(uint64_t v, uint64_t* ws, uint64_t* total) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  uint64_t inc = v;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const uint64_t t = __shfl_up(inc, d, 64);
    if (lane >= d && t > inc) inc = t;
  }
  uint64_t ex = __shfl_up(inc, 1, 64);
  if (lane == 0) ex = 0;
  if (NT == 64) {
    *total = __shfl(inc, 63, 64);
    return ex;
  }
  lds_barrier();
  if (lane == 63) ws[w] = inc;
  lds_barrier();
  uint64_t off = 0, tot = 0;
#pragma unroll
  for (int k = 0; k < NT / 64; ++k) {
    const uint64_t x = ws[k];
    if (k < w && x > off) off = x;
    if (x > tot) tot = x;
  }
  *total = tot;
  return ex > off ? ex : off;
}
template <int NT>
__device__ __forceinline__ uint32_t block_sum(uint32_t v, uint32_t* ws) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  if (NT == 64) return v;
  lds_barrier();
  if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = v;
  lds_barrier();
  uint32_t tot = 0;
#pragma unroll
  for (int k = 0; k < NT / 64; ++k) tot += ws[k];
  return tot;
}

// ---- the runs of equal (q_id, t_id) ---------------------------------------------------------------------------------
// pair_boundary: one bit per record (a run begins here) and the number of set bits per 64-record word -- no counter that every
// wavefront bumps: an input of millions of tiny pairs would serialise on it (a same-address atomic costs ~11 ns: 14 ms for
// 1.25 M of them, measured).  A prefix sum over the words' counts (swg_exclusive_scan_u32) numbers the runs in input order;
// pair_starts lists their first records; pair_runs (a thread per run) takes a run's end from its successor's start, files the
// run under its size class (one atomic per class and work-group) and enters its pair into a hash set: a pair entered twice has
// two runs -- the input is not grouped by chromosome pair.
__global__ __launch_bounds__(256) void pair_boundary_kernel(uint32_t n, const uint32_t* __restrict__ q_id,
                                                            const uint32_t* __restrict__ t_id, unsigned long long* __restrict__ bitmap,
                                                            uint32_t* __restrict__ wcnt) {
  // a wavefront takes 4 x 64 consecutive records (four bitmap words); every load is requested before the first comparison
  constexpr int R = 4;
  const int lane = threadIdx.x & 63;
  const uint32_t base_i = (blockIdx.x * 4u + (threadIdx.x >> 6)) * (64u * R);
  if (base_i >= n) return;
  uint32_t q[R], t[R], pq[R], pt[R];
#pragma unroll
  for (int j = 0; j < R; ++j) {
    const uint32_t i = base_i + (uint32_t)j * 64u + (uint32_t)lane;
    q[j] = i < n ? q_id[i] : 0u;
    t[j] = i < n ? t_id[i] : 0u;
  }
  const uint32_t q_before = base_i ? q_id[base_i - 1] : 0u, t_before = base_i ? t_id[base_i - 1] : 0u;  // (uniform)
#pragma unroll
  for (int j = 0; j < R; ++j) {
    pq[j] = (uint32_t)__shfl_up((int)q[j], 1, 64);
    pt[j] = (uint32_t)__shfl_up((int)t[j], 1, 64);
    const uint32_t lq = j ? (uint32_t)__shfl((int)q[j - 1], 63, 64) : q_before, lt = j ? (uint32_t)__shfl((int)t[j - 1], 63, 64) : t_before;
    if (lane == 0) {
      pq[j] = lq;
      pt[j] = lt;
    }
  }
#pragma unroll
  for (int j = 0; j < R; ++j) {
    const uint32_t i = base_i + (uint32_t)j * 64u + (uint32_t)lane;
    const bool start = i < n && (i == 0 || pq[j] != q[j] || pt[j] != t[j]);
    const unsigned long long mask = __ballot(start);
    if (lane == 0 && base_i + (uint32_t)j * 64u < n) {
      bitmap[i >> 6] = mask;
      wcnt[i >> 6] = (uint32_t)__popcll(mask);
    }
  }
}
// starts[r] = first record of run r (runs numbered in input order by the prefix sums of the words' counts); starts[n_runs] = n
__global__ __launch_bounds__(256) void pair_starts_kernel(uint32_t n, uint32_t n_words, const unsigned long long* __restrict__ bitmap,
                                                          const uint32_t* __restrict__ wpre, const uint64_t* __restrict__ n_runs_dev,
                                                          uint32_t cap, uint32_t* __restrict__ starts) {
  const uint32_t w = blockIdx.x * 256u + threadIdx.x;
  if (w == 0) {
    const uint64_t nr = *n_runs_dev;
    if (nr <= cap) starts[nr] = n;
  }
  if (w >= n_words) return;
  unsigned long long m = bitmap[w];
  uint32_t r = wpre[w];
  while (m) {
    const int b = __builtin_ctzll(m);
    m &= m - 1;
    if (r < cap) starts[r] = (w << 6) + (uint32_t)b;
    ++r;
  }
}
__global__ __launch_bounds__(256) void pair_runs_kernel(uint32_t n, uint32_t cap, const uint64_t* __restrict__ n_runs_dev,
                                                        const uint32_t* __restrict__ starts, const uint32_t* __restrict__ q_id,
                                                        const uint32_t* __restrict__ t_id, unsigned long long* __restrict__ table,
                                                        uint32_t tmask, PairRun* __restrict__ runs, uint32_t* __restrict__ class_list,
                                                        PairCounters* __restrict__ C) {
  __shared__ uint32_t s_cnt[4][4], s_base[4];
  const uint64_t nr64 = *n_runs_dev;
  if (nr64 > cap) {  // more pairs than this path takes (pair_plan's rule): nothing else is looked at
    if (blockIdx.x == 0 && threadIdx.x == 0) {
      C->n_runs = cap + 1u;
      atomicOr(&C->flags, PF_RUN_OVERFLOW);
    }
    return;
  }
  const uint32_t nr = (uint32_t)nr64;
  if (blockIdx.x == 0 && threadIdx.x == 0) C->n_runs = nr;
  const uint32_t k = blockIdx.x * 256u + threadIdx.x;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  int cls = -1;
  if (k < nr) {
    const uint32_t a = starts[k], len = starts[k + 1] - a;
    PairRun r;
    r.a = a;
    r.n = len;
    runs[k] = r;
    if (len > PAIR_XL_MAX) {
      atomicOr(&C->flags, PF_TOO_LONG);
    } else {
      cls = len <= PAIR_S_MAX ? 0 : (len <= PAIR_M_MAX ? 1 : (len <= PAIR_L_MAX ? 2 : 3));
    }
    const unsigned long long key = ((unsigned long long)q_id[a] << 32) | t_id[a];
    uint32_t h = (uint32_t)((key * 0x9e3779b97f4a7c15ull) >> 32) & tmask;
    for (;;) {  // (at most `cap` keys in 2 * cap slots)
      const unsigned long long old = atomicCAS(&table[h], ~0ull, key);
      if (old == ~0ull) break;
      if (old == key) {
        atomicOr(&C->flags, PF_NOT_GROUPED);
        break;
      }
      h = (h + 1) & tmask;
    }
  }
  // the class lists: one atomic per class and work-group
  unsigned long long mk[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    mk[c] = __ballot(cls == c);
    if (lane == 0) s_cnt[wv][c] = (uint32_t)__popcll(mk[c]);
  }
  __syncthreads();
  if (threadIdx.x < 4) {
    const int c = threadIdx.x;
    const uint32_t tot = s_cnt[0][c] + s_cnt[1][c] + s_cnt[2][c] + s_cnt[3][c];
    s_base[c] = tot ? atomicAdd(&C->n_class[c], tot) : 0u;
  }
  __syncthreads();
  if (cls >= 0) {
    uint32_t o = s_base[cls];
    for (int w2 = 0; w2 < wv; ++w2) o += s_cnt[w2][cls];
    const unsigned long long m = cls == 0 ? mk[0] : (cls == 1 ? mk[1] : (cls == 2 ? mk[2] : mk[3]));
    class_list[(size_t)cls * cap + o + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))] = k;
  }
}

// ---- inputs that are not grouped by pair (and small ones in general): the pairs through a hash table ----------------------
// pair_hash: every record's pair is entered into an open-addressing table (key + 1, 0 = empty) and counted; pair_slots: the
// occupied slots become the pairs -- their sizes, offsets into a list of record indices and size classes; pair_perm: every record drops its index into its pair's part of that list.  The order inside a pair is whatever
// the atomics make it: pair_sort orders by (strand, q_start, record index) with the record's own index, so it does not matter.
__global__ __launch_bounds__(256) void pair_hash_kernel(uint32_t n, const uint32_t* __restrict__ q_id, const uint32_t* __restrict__ t_id,
                                                        unsigned long long* __restrict__ table, uint32_t tmask,
                                                        uint32_t* __restrict__ count, uint32_t* __restrict__ slot_of) {
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  if (i >= n) return;
  const unsigned long long key = (((unsigned long long)q_id[i] << 32) | t_id[i]) + 1ull;
  uint32_t h = (uint32_t)((key * 0x9e3779b97f4a7c15ull) >> 32) & tmask;
  for (;;) {  // (at most n keys in >= 2 n slots)
    unsigned long long old = __hip_atomic_load(&table[h], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (old == 0ull) {
      old = atomicCAS(&table[h], 0ull, key);
      if (old == 0ull) old = key;
    }
    if (old == key) break;
    h = (h + 1) & tmask;
  }
  slot_of[i] = h;
  atomicAdd(&count[h], 1u);
}
__global__ __launch_bounds__(256) void pair_slots_kernel(uint32_t tsize, const uint32_t* __restrict__ count, uint32_t* __restrict__ start,
                                                         PairRun* __restrict__ runs, uint32_t* __restrict__ class_list, uint32_t cap,
                                                         PairCounters* __restrict__ C, uint32_t* __restrict__ cursor) {
  // every occupied slot takes the next pair number and the next stretch of the index list (which pair gets which is left to
  // the atomics: nothing depends on the order of the pairs)
  const uint32_t h = blockIdx.x * 256u + threadIdx.x;
  const uint32_t c = h < tsize ? count[h] : 0u;
  if (!c) return;
  const uint32_t k = atomicAdd(&C->n_runs, 1u), o = atomicAdd(cursor, c);
  start[h] = o;
  if (k >= cap) {  // more pairs than this path takes (pair_plan's rule)
    atomicOr(&C->flags, PF_RUN_OVERFLOW);
    return;
  }
  PairRun r;
  r.a = o;
  r.n = c;
  runs[k] = r;
  if (c > PAIR_L_MAX) {
    atomicOr(&C->flags, PF_TOO_LONG);
  } else {
    const int cls = c <= PAIR_S_MAX ? 0 : (c <= PAIR_M_MAX ? 1 : 2);
    class_list[(size_t)cls * cap + atomicAdd(&C->n_class[cls], 1u)] = k;
  }
}
__global__ __launch_bounds__(256) void pair_perm_kernel(uint32_t n, const uint32_t* __restrict__ slot_of, uint32_t* __restrict__ start,
                                                        uint32_t* __restrict__ perm) {
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  if (i < n) perm[atomicAdd(&start[slot_of[i]], 1u)] = i;
}

// ---- pair_sort ------------------------------------------------------------------------------------------------------
// Monotone map of a (strand, key) onto buckets: the strand's share of the buckets, inside it a linear map of the key range.
// Float arithmetic, but monotone in the key whatever the rounding (conversion, multiplication by a positive constant and
// truncation never decrease), which is all the sort needs: a bucket's elements are ordered afterwards.
struct BucketMap {
  uint32_t kmin[2];
  float scale[2];
  uint32_t off[2], nb[2];
};
__device__ __forceinline__ void bucket_map_make(BucketMap& B, uint32_t nbk, uint32_t m0, uint32_t m1, const uint32_t* kmin,
                                                const uint32_t* kmax) {
  const uint32_t m = m0 + m1;
  uint32_t nb0 = m ? (uint32_t)(((uint64_t)nbk * m0) / m) : 0u;
  if (m0 && nb0 == 0) nb0 = 1;
  if (m1 && nb0 >= nbk) nb0 = nbk - 1;
  B.nb[0] = nb0;
  B.nb[1] = nbk - nb0;
  B.off[0] = 0;
  B.off[1] = nb0;
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    B.kmin[s] = kmin[s];
    const float range = (float)(kmax[s] - kmin[s]) + 1.0f;
    B.scale[s] = (float)B.nb[s] / range;
  }
  // work-group-uniform values that came out of LDS reads (vector registers): into scalar registers
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    B.kmin[s] = (uint32_t)__builtin_amdgcn_readfirstlane((int)B.kmin[s]);
    B.scale[s] = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(B.scale[s])));
    B.off[s] = (uint32_t)__builtin_amdgcn_readfirstlane((int)B.off[s]);
    B.nb[s] = (uint32_t)__builtin_amdgcn_readfirstlane((int)B.nb[s]);
  }
}
__device__ __forceinline__ uint32_t bucket_of(const BucketMap& B, uint32_t st, uint32_t k) {
  // (selects, not B.x[st]: an array indexed by a run-time value is sent through scratch memory)
  const uint32_t kmin = st ? B.kmin[1] : B.kmin[0], nb = st ? B.nb[1] : B.nb[0], off = st ? B.off[1] : B.off[0];
  const float scale = st ? B.scale[1] : B.scale[0];
  const float f = (float)(k - kmin) * scale;
  uint32_t b = (uint32_t)f;
  const uint32_t top = nb - 1u;
  b = b < top ? b : top;
  return off + b;
}

struct PairSortArgs {
  const uint32_t *q_id, *t_id, *q_start, *q_end, *t_start, *t_end, *matches, *block_len;
  const double* identity;
  const uint8_t* strand;
  const uint8_t *alive_in, *member_in;  // nullptr: step-1 retain evaluated here / every alive record is a member
  uint64_t min_block;
  int keep_self;
  double min_identity;
  int check_degenerate;  // an unlimited mapping sweep is taken as the identity: a degenerate alive record voids that
  uint64_t max_gap;
  const PairRun* runs;
  const uint32_t* list;
  const uint32_t* perm;  // inputs not grouped by pair: record indices, pair after pair (runs[] then index this list)
  const uint32_t* orig;  // the records are a pair-major copy: the caller's index of every record (first appearances are reported in those)
  uint8_t* code;
  uint32_t *s_qs, *s_qe, *s_ts, *s_te, *s_m, *s_b, *s_idx, *pred;
  PairInfo* info;
  SpecBlock* chunks;
  uint32_t cap_chunks;
  int chunks_by_place;  // the runs are in input order (a plan over runs): a pair's chunks go to slots that follow from its place
  uint32_t* long_list;  // indices (into chunks) of the chunks of LABEL_CAP_ELEMS members and more
  uint32_t cap_long;
  PairCounters* C;
  PairTable gl_first;
  const uint32_t* seq_genome_last;
};

// What a pair_sort work-group does once its members are in order, shared by the two kernels below.
//
// emit_chunks: the chunk list of the walk from the per-cell first unit starts (cellmin; one thread).  The first unit of every
// cell opens a chunk, and so does the first '-' member; a chunk of LABEL_CAP_ELEMS members or more (a long unit) is also put on
// the list of the chunks that chain_label_long_kernel labels.
// Round 6: a pair's chunks take a stretch of the list that follows from its place in the input -- slot a / PAIR_CELL + 2 * (the
// run's index): runs are in input order, a pair of n records has at most n / PAIR_CELL + 2 chunks, and the next pair's stretch
// begins at least that far on -- instead of a returning atomic on one counter at the end of every pair's work-group (its round
// trip was most of the 5.8 us a pair spent here, alone on its CU).  The list is zeroed per call and the kernels that walk it skip
// empty descriptors; its length is its capacity.  (Inputs grouped through the hash table -- at most 65,536 records -- number
// their pairs by atomics: no order to rely on, the counter stays.)
__device__ void emit_chunks(const PairSortArgs& A, uint32_t rk_run, uint32_t a, uint32_t m, uint32_t m_plus, const uint32_t* cellmin, int n_cell) {
  uint32_t prev = NONE, count = 0, n_long = 0, prev2 = NONE;
  const bool mp_pending = m_plus > 0 && m_plus < m;
  auto for_starts = [&](auto&& f) {
    bool pend = mp_pending;
    for (int c = 0; c < n_cell; ++c) {
      const uint32_t v = cellmin[c];
      if (v == NONE) continue;
      if (pend && m_plus <= v) {
        if (m_plus < v) f(m_plus);
        pend = false;
      }
      f(v);
    }
    if (pend) f(m_plus);
  };
  for_starts([&](uint32_t v) {
    ++count;
    if (prev2 != NONE && v - prev2 >= LABEL_CAP_ELEMS) ++n_long;
    prev2 = v;
  });
  if (prev2 != NONE && m - prev2 >= LABEL_CAP_ELEMS) ++n_long;
  const uint32_t slot = A.chunks_by_place ? a / PAIR_CELL + 2u * rk_run : atomicAdd(&A.C->n_chunks, count);
  uint32_t lslot = n_long ? atomicAdd(&A.C->n_long, n_long) : 0u;
  if (slot + count > A.cap_chunks || (n_long && lslot + n_long > A.cap_long)) {  // (the capacities are upper bounds: not reached)
    atomicOr(&A.C->flags, PF_FALLBACK);
    return;
  }
  uint32_t k = 0;
  auto emit = [&](uint32_t b, uint32_t e) {
    SpecBlock d;
    d.bb = a + b;
    d.be = a + e;
    d.ue = d.be;
    d.pad = b >= m_plus ? 1u : 0u;
    if (e - b >= LABEL_CAP_ELEMS) A.long_list[lslot++] = slot + k;
    A.chunks[slot + k++] = d;
  };
  for_starts([&](uint32_t v) {
    if (prev != NONE) emit(prev, v);
    prev = v;
  });
  if (prev != NONE) emit(prev, m);
}
// unit_starts: the thread's E consecutive positions p0 .. p0 + E of a batch of mb members that begins at pair position
// `base`; qs / qe are their sorted q_start / q_end.  Position p opens a unit when its q_start lies beyond every earlier q_end
// of its strand by more than the gap (no window of paf_filter.rs:786-796 can straddle it).  The first unit start of every cell
// goes into cellmin; *carry_max: running maximum of ((strand << 32) | q_end) over the batches so far.
template <int NT, int E>
__device__ __forceinline__ void unit_starts(const uint32_t (&qs)[E], const uint32_t (&qe)[E], uint32_t mb, uint32_t base, uint32_t m_plus,
                                            uint64_t max_gap, uint64_t* ws64, uint32_t* cellmin, uint64_t* carry_max) {
  const uint32_t p0 = (uint32_t)threadIdx.x * E;
  uint64_t tmax = 0;
#pragma unroll
  for (int e = 0; e < E; ++e)
    if (p0 + e < mb) {
      const uint64_t c = ((uint64_t)(base + p0 + e >= m_plus ? 1u : 0u) << 32) | qe[e];
      tmax = c > tmax ? c : tmax;
    }
  uint64_t tot;
  uint64_t prev = block_excl_max<NT>(tmax, ws64, &tot);
  prev = prev > *carry_max ? prev : *carry_max;
#pragma unroll
  for (int e = 0; e < E; ++e)
    if (p0 + e < mb) {
      const uint32_t pa = base + p0 + e;
      uint64_t lim = (prev & 0xffffffffull) + max_gap;
      if (lim < max_gap) lim = ~0ull;  // saturate
      if (pa == 0 || pa == m_plus || (uint64_t)qs[e] > lim) atomicMin(&cellmin[pa / PAIR_CELL], pa);
      const uint64_t c = ((uint64_t)(pa >= m_plus ? 1u : 0u) << 32) | qe[e];
      prev = c > prev ? c : prev;
    }
  *carry_max = tot > *carry_max ? tot : *carry_max;
}
// the thread's E consecutive words of an LDS array, as 16-byte reads (4-way bank conflicts instead of 16-way)
template <int E>
__device__ __forceinline__ void read_block(const uint32_t* lds, uint32_t (&out)[E]) {
  static_assert(E % 4 == 0, "whole 16-byte reads");
  const uint4* v = reinterpret_cast<const uint4*>(lds) + (size_t)threadIdx.x * (E / 4);
#pragma unroll
  for (int j = 0; j < E / 4; ++j) {
    const uint4 x = v[j];
    out[4 * j] = x.x;
    out[4 * j + 1] = x.y;
    out[4 * j + 2] = x.z;
    out[4 * j + 3] = x.w;
  }
}

#ifdef SWG_PAIR_TIMING
__device__ unsigned long long g_pair_t[16];
#define PT_STAMP(k) do { __syncthreads(); if (threadIdx.x == 0) { const unsigned long long t_ = wall_clock64(); atomicAdd(&g_pair_t[k], t_ - pt_last); pt_last = t_; } } while (0)
#define FT_STAMP(k) do { __syncthreads(); if (threadIdx.x == 0) { const unsigned long long t_ = wall_clock64(); atomicAdd(&g_fin_t[k], t_ - pt_last); pt_last = t_; } } while (0)
__device__ unsigned long long g_fin_t[16];
#else
#define PT_STAMP(k) do { } while (0)
#define FT_STAMP(k) do { } while (0)
#endif
// One work-group per pair of at most NT * ER records, sorted in batches of at most NT * ES members (one batch when the pair's
// members fit, which is the rule; otherwise the key range is cut into coarse bins and consecutive bins are glued into batches).
// A thread OWNS the records tid, tid + NT, ... of the run: it loads their columns (coalesced), drops their keys into the
// batch's buckets, learns from the ranking where its records ended up (the slot its key was dropped at names the rank:
// RR), and puts their columns at that place of an LDS buffer that is then written out coalesced -- no gather anywhere.
// Every global load of a phase is requested before the first value is used (a thread's records one after the other would be
// as many memory round trips in a row, and the work-group is alone on its CU: nothing else hides them).
// (the LDS of a pair_sort work-group is one raw block that the body carves up: the two bodies -- this one and the one for the
// longest runs -- share a launch and therefore a block)
// Compile-time offsets only: a pointer that goes through an integer on its way loses its address space, and every LDS access
// behind it becomes a flat instruction.
constexpr size_t lds_align_up(size_t off, size_t align) { return (off + align - 1) / align * align; }
template <int NT, int ES, int ER, int NBK, int NBIN, bool PERM>
constexpr size_t pair_sort_lds_bytes() {
  // K, I, RR, cnt, bins, b_lo, cellmin, ws64, ws, 13 scalars -- plus slack for the alignment of each piece
  return (size_t)NT * ES * (PERM ? 10 : 8) + (size_t)NBK * 4 + (size_t)NBIN * 4 + 17 * 4 + (size_t)((NT * ER + PAIR_CELL - 1) / PAIR_CELL) * 4 +
         (size_t)(NT / 64 + 1) * 12 + 16 * 4 + 128;
}
// PERM: the pair's records are not a run of the input but the entries [a, a + n) of a list of record indices (inputs that are
// not grouped by pair: pair_group_*); the index inside the pair gives way to the record's own index everywhere.
template <int NT, int ES, int ER, int NBK, int NBIN, bool PERM>
__device__ __forceinline__ void pair_sort_body(const PairSortArgs& A, const uint32_t rk_run, char* lds_raw) {
  using IT = typename std::conditional<PERM, uint32_t, uint16_t>::type;
  constexpr int CAP = NT * ES, NREC = NT * ER, MAXB = 16, H = 8;
  constexpr int NCELL = (NREC + (int)PAIR_CELL - 1) / (int)PAIR_CELL;
  static_assert(ER <= 32 && ER % H == 0 && ES % 4 == 0 && NREC <= 65536, "record masks are 32 bits wide, indices 16");
  static_assert(NBIN >= 2 && NBIN <= 4096, "each strand needs a coarse bin of its own (the members are ordered strand first)");
  constexpr size_t O_K = 0, O_I = O_K + (size_t)CAP * 4, O_RR = O_I + (size_t)CAP * sizeof(IT), O_CNT = lds_align_up(O_RR + (size_t)CAP * 2, 4),
                   O_BINS = O_CNT + (size_t)NBK * 4, O_BLO = O_BINS + (size_t)NBIN * 4, O_CELL = O_BLO + (size_t)(MAXB + 1) * 4,
                   O_WS64 = lds_align_up(O_CELL + (size_t)NCELL * 4, 8), O_WS = O_WS64 + (size_t)(NT / 64 + 1) * 8,
                   O_SH = O_WS + (size_t)(NT / 64 + 1) * 4;
  static_assert(O_SH + 13 * 4 <= pair_sort_lds_bytes<NT, ES, ER, NBK, NBIN, PERM>(), "LDS block of the work-group");
  uint32_t* const K = reinterpret_cast<uint32_t*>(lds_raw + O_K);
  IT* const I = reinterpret_cast<IT*>(lds_raw + O_I);
  uint16_t* const RR = reinterpret_cast<uint16_t*>(lds_raw + O_RR);
  uint32_t* const cnt = reinterpret_cast<uint32_t*>(lds_raw + O_CNT);
  uint32_t* const bins = reinterpret_cast<uint32_t*>(lds_raw + O_BINS);
  uint32_t* const b_lo = reinterpret_cast<uint32_t*>(lds_raw + O_BLO);
  uint32_t* const cellmin = reinterpret_cast<uint32_t*>(lds_raw + O_CELL);
  uint64_t* const ws64 = reinterpret_cast<uint64_t*>(lds_raw + O_WS64);
  uint32_t* const ws = reinterpret_cast<uint32_t*>(lds_raw + O_WS);
  uint32_t* const sh_cnt = reinterpret_cast<uint32_t*>(lds_raw + O_SH);
  uint32_t* const sh_kmin = sh_cnt + 4;
  uint32_t* const sh_kmax = sh_cnt + 6;
  uint32_t* const sh_first = sh_cnt + 8;
  uint32_t& sh_nb = sh_cnt[11];
  uint32_t& sh_bad = sh_cnt[12];
#ifdef SWG_PAIR_TIMING
  unsigned long long pt_last = wall_clock64();
#endif
  const int tid = threadIdx.x;
  const PairRun run = A.runs[rk_run];
  const uint32_t a = run.a, n = run.n;
  if (tid < 4) sh_cnt[tid] = 0;
  if (tid < 2) {
    sh_kmin[tid] = 0xffffffffu;
    sh_kmax[tid] = 0;
  }
  if (tid < 3) sh_first[tid] = NONE;
  if (tid == 0) sh_bad = 0;
  for (int c = tid; c < NCELL; c += NT) cellmin[c] = NONE;
  for (int b = tid; b < NBIN; b += NT) bins[b] = 0;
  lds_barrier();
  // the thread's records: e-th record = tid + e * NT; every loop over them runs in groups of H whose loads go out together
  // (the pair's columns as work-group-uniform pointers indexed by a 32-bit offset inside the pair: one scalar base and one
  // 32-bit register per load, not a 64-bit address pair per column and record)
  // (a fresh copy of the thread index per phase: the compiler otherwise computes tid + e * NT for every e once, keeps the 16
  // or 32 values across the whole kernel and ends up spilling them -- they cost one addition each)
  auto fresh_tid = [&]() -> uint32_t {
    uint32_t t = (uint32_t)tid;
    asm volatile("" : "+v"(t));
    return t;
  };
  uint32_t tid_v = (uint32_t)tid, n_v = n;  // (copies that pass through an empty asm per batch: see the batch loop)
  // what a column is indexed by: the record's place in the run (the column pointers then start at the run), or its own index
  auto rec_index = [&](int e) -> uint32_t {
    const uint32_t li = tid_v + (uint32_t)e * NT;
    const uint32_t lc = li < n_v ? li : 0u;
    return PERM ? A.perm[a + lc] : lc;
  };
  const uint32_t cb = PERM ? 0u : a;  // column base
  const uint32_t* c_qs = A.q_start + cb;
  const uint32_t* c_qe = A.q_end + cb;
  const uint32_t* c_ts = A.t_start + cb;
  const uint32_t* c_te = A.t_end + cb;
  const uint32_t* c_m = A.matches + cb;
  const uint32_t* c_b = A.block_len + cb;
  const uint8_t* c_st = A.strand + cb;
  const double* c_id = A.identity ? A.identity + cb : nullptr;
  const uint8_t* c_alive = A.alive_in ? A.alive_in + cb : nullptr;
  const uint8_t* c_member = A.member_in ? A.member_in + cb : nullptr;
  uint32_t* o_qs = A.s_qs + a;
  uint32_t* o_qe = A.s_qe + a;
  uint32_t* o_ts = A.s_ts + a;
  uint32_t* o_te = A.s_te + a;
  uint32_t* o_m = A.s_m + a;
  uint32_t* o_b = A.s_b + a;
  uint32_t* o_idx = A.s_idx + a;
  uint32_t* o_pred = A.pred + a;
  // ---- step-1 retain, members, the key range per strand
  const uint32_t r0 = PERM ? A.perm[a] : a;
  const uint32_t q0 = A.q_id[r0], t0 = A.t_id[r0];
  const bool self_ok = A.keep_self || q0 != t0;
  uint32_t member_mask = 0, strand_mask = 0, extra_mask = 0;
  {
    uint32_t alive_mask = 0, in_mask = 0;
    uint32_t n_mem[2] = {0, 0}, kmin[2] = {0xffffffffu, 0xffffffffu}, kmax[2] = {0, 0}, fst[3] = {NONE, NONE, NONE};
#pragma unroll
    for (int g = 0; g < ER; g += H) {
      if ((uint32_t)g * NT >= n) continue;  // (block-uniform; `continue`, not `break`: the loop must unroll -- its arrays are registers)
      uint8_t stv[H], av[H], mv[H];
      uint32_t qv[H], blv[H], ixv[H];
      double idv[H];
      const bool need_bl = !A.alive_in && (A.min_block != 0 || !A.identity);
#pragma unroll
      for (int e = 0; e < H; ++e) {
        const uint32_t i = ixv[e] = rec_index(g + e);
        stv[e] = c_st[i];
        qv[e] = c_qs[i];
        av[e] = c_alive ? c_alive[i] : (uint8_t)1;
        mv[e] = c_member ? c_member[i] : (uint8_t)1;
        blv[e] = need_bl ? c_b[i] : 0u;
        idv[e] = c_alive ? 1.0 : (c_id ? c_id[i] : (double)c_m[i]);
      }
#pragma unroll
      for (int e = 0; e < H; ++e) {
        const uint32_t li = (uint32_t)tid + (uint32_t)(g + e) * NT;
        if (li >= n) continue;
        const uint32_t gi = PERM ? ixv[e] : a + li;  // the record's own index
        bool alive;
        if (A.alive_in) {
          alive = av[e] != 0;
        } else {
          // identity == nullptr: matches over max(block length, 1), one IEEE division (src/paf_filter.rs:322)
          const double id = A.identity ? idv[e] : __ddiv_rn(idv[e], (double)(blv[e] > 1u ? blv[e] : 1u));
          alive = self_ok && (A.min_block == 0 || (uint64_t)blv[e] >= A.min_block) && id >= A.min_identity;
        }
        const bool member = alive && mv[e] != 0;
        const uint32_t st = stv[e] ? 1u : 0u;
        in_mask |= 1u << (g + e);
        strand_mask |= st << (g + e);
        if (alive) {
          alive_mask |= 1u << (g + e);
          fst[2] = gi < fst[2] ? gi : fst[2];
          if (member) {
            member_mask |= 1u << (g + e);
#pragma unroll
            for (uint32_t s2 = 0; s2 < 2; ++s2)  // (no array indexed by a run-time strand: see bucket_of)
              if (st == s2) {
                ++n_mem[s2];
                kmin[s2] = qv[e] < kmin[s2] ? qv[e] : kmin[s2];
                kmax[s2] = qv[e] > kmax[s2] ? qv[e] : kmax[s2];
                fst[s2] = gi < fst[s2] ? gi : fst[s2];
              }
          }
        }
      }
      asm volatile("" ::: "memory");
    }
    extra_mask = alive_mask & ~member_mask;
    uint32_t c_x = (uint32_t)__popc(extra_mask);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      n_mem[0] += __shfl_xor(n_mem[0], o, 64);
      n_mem[1] += __shfl_xor(n_mem[1], o, 64);
      c_x += __shfl_xor(c_x, o, 64);
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        const uint32_t x = __shfl_xor(kmin[s2], o, 64), y = __shfl_xor(kmax[s2], o, 64);
        kmin[s2] = x < kmin[s2] ? x : kmin[s2];
        kmax[s2] = y > kmax[s2] ? y : kmax[s2];
      }
#pragma unroll
      for (int s2 = 0; s2 < 3; ++s2) {
        const uint32_t x = __shfl_xor(fst[s2], o, 64);
        fst[s2] = x < fst[s2] ? x : fst[s2];
      }
    }
    if ((tid & 63) == 0) {
      if (n_mem[0]) atomicAdd(&sh_cnt[0], n_mem[0]);
      if (n_mem[1]) atomicAdd(&sh_cnt[1], n_mem[1]);
      if (c_x) atomicAdd(&sh_cnt[2], c_x);
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2)
        if (n_mem[s2]) {
          atomicMin(&sh_kmin[s2], kmin[s2]);
          atomicMax(&sh_kmax[s2], kmax[s2]);
        }
#pragma unroll
      for (int s2 = 0; s2 < 3; ++s2)
        if (fst[s2] != NONE) atomicMin(&sh_first[s2], fst[s2]);
    }
  }
  lds_barrier();
  const uint32_t m_plus = (uint32_t)__builtin_amdgcn_readfirstlane((int)sh_cnt[0]);
  const uint32_t m = m_plus + (uint32_t)__builtin_amdgcn_readfirstlane((int)sh_cnt[1]);
  const uint32_t n_x = (uint32_t)__builtin_amdgcn_readfirstlane((int)sh_cnt[2]), M = m + n_x;
  PT_STAMP(1);
  if (tid == NT - 1) {  // (the last wavefront: thread 0's has the chunk list to write at the end)
    PairInfo pi;
    pi.m = m;
    pi.m_plus = m_plus;
    pi.M = M;
    if (A.orig) {  // (a pair-major copy: first appearances in the caller's own record indices -- ascending inside the pair)
#pragma unroll
      for (int j = 0; j < 3; ++j)
        if (sh_first[j] != NONE) sh_first[j] = A.orig[sh_first[j]];
    }
    pi.first_alive = sh_first[2];
    pi.first_mem[0] = sh_first[0];
    pi.first_mem[1] = sh_first[1];
    pi.q = q0;
    pi.t = t0;
    A.info[rk_run] = pi;
    if (sh_first[2] != NONE)  // the genome pair's first alive record (apply_plane_sweep_to_mappings' group order, :1037-1046)
      atomicMin(pair_slot(A.gl_first, A.seq_genome_last[q0], A.seq_genome_last[t0]), sh_first[2]);
    // (the alive records' and the members' totals: summed over `info` by pair_totals_kernel -- a counter that every pair bumps
    // serialises on its cache line: 4.5 of 6.8 ms of the small pairs' sort at 198,000 pairs)
  }
  if (M == 0) return;
  bool degenerate = false;
  // ---- the alive records that are not members (behind a mapping sweep): kept with the pair for the inversion capture and
  // the rescue, behind the members, in input order; bit 31 of their index carries the strand
  if (n_x) {
    uint32_t done = 0;
#pragma unroll 1
    for (int e = 0; e < ER; ++e) {
      if ((uint32_t)e * NT >= n) break;
      const uint32_t li = (uint32_t)tid + (uint32_t)e * NT;
      const bool x = (extra_mask >> e) & 1u;
      uint32_t tot;
      const uint32_t r = block_excl_sum<NT>(x ? 1u : 0u, ws, &tot);
      if (x) {
        const uint32_t p = m + done + r;
        const uint32_t ri = PERM ? A.perm[a + li] : li;
        const uint32_t qs = c_qs[ri], qe = c_qe[ri], ts = c_ts[ri], te = c_te[ri];
        o_qs[p] = qs;
        o_qe[p] = qe;
        o_ts[p] = ts;
        o_te[p] = te;
        o_idx[p] = (PERM ? ri : a + li) | (((strand_mask >> e) & 1u) << 31);
        degenerate |= qs >= qe || ts >= te;
      }
      done += tot;
    }
  }
  if (m == 0) {
    if (A.check_degenerate && __any(degenerate) && (tid & 63) == 0) atomicOr(&A.C->flags, PF_FALLBACK);
    return;
  }
  // ---- coarse bins over (strand, q_start); more than one batch only when the members do not fit one
  BucketMap BM;
  {
    const uint32_t kmn[2] = {sh_kmin[0], sh_kmin[1]}, kmx[2] = {sh_kmax[0], sh_kmax[1]};
    bucket_map_make(BM, NBIN, m_plus, m - m_plus, kmn, kmx);
  }
  // the fine bucket of a key inside the batch [bin_lo, bin_hi): the coarse map refined by a power of two.  f < NBIN <= 2^12
  // has at most 12 integer bits and multiplying a float by 2^12 is exact, so (uint32)(f * 4096) >> 12 == (uint32)f: the fine id
  // names the coarse bin in its high bits and stays monotone in the key.
  auto fine_of = [&](uint32_t st, uint32_t k, int shift, uint32_t first, uint32_t* coarse) -> uint32_t {
    const float f = (float)(k - (st ? BM.kmin[1] : BM.kmin[0])) * (st ? BM.scale[1] : BM.scale[0]);
    const uint32_t top = (st ? BM.nb[1] : BM.nb[0]) - 1u;
    uint32_t g = (uint32_t)(f * 4096.0f);
    if ((g >> 12) > top) g = (top << 12) | 0xfffu;  // (the clamp of bucket_of, in fine units)
    g += (st ? BM.off[1] : BM.off[0]) << 12;
    *coarse = g >> 12;
    const uint32_t b = (g >> shift) - first;
    return b < (uint32_t)NBK ? b : (uint32_t)NBK - 1u;
  };
  uint32_t n_batches = 1;
  if (m > (uint32_t)CAP) {
#pragma unroll
    for (int g = 0; g < ER; g += H) {
      if ((uint32_t)g * NT >= n) continue;
      uint32_t qv[H];
#pragma unroll
      for (int e = 0; e < H; ++e) qv[e] = c_qs[rec_index(g + e)];
#pragma unroll
      for (int e = 0; e < H; ++e)
        if ((member_mask >> (g + e)) & 1u) atomicAdd(&bins[bucket_of(BM, (strand_mask >> (g + e)) & 1u, qv[e])], 1u);
      asm volatile("" ::: "memory");
    }
    lds_barrier();
    {  // bins -> their exclusive prefix sums
      constexpr int PERB = (NBIN + NT - 1) / NT;
      uint32_t c[PERB], sum = 0, tot;
#pragma unroll
      for (int j = 0; j < PERB; ++j) {
        c[j] = tid * PERB + j < NBIN ? bins[tid * PERB + j] : 0u;
        sum += c[j];
      }
      uint32_t off = block_excl_sum<NT>(sum, ws, &tot);
      lds_barrier();
#pragma unroll
      for (int j = 0; j < PERB; ++j)
        if (tid * PERB + j < NBIN) {
          bins[tid * PERB + j] = off;
          off += c[j];
        }
    }
    lds_barrier();
    if (tid == 0) {  // greedy: a batch takes as many bins as fit (a binary search in the prefix sums per batch)
      uint32_t nb = 0, lo = 0;
      b_lo[0] = 0;
      while (lo < (uint32_t)NBIN) {
        const uint32_t start = bins[lo];
        uint32_t l = lo + 1, r = NBIN;
        while (l < r) {
          const uint32_t mid = l + ((r - l + 1) >> 1);
          const uint32_t pm = mid < (uint32_t)NBIN ? bins[mid] : m;
          if (pm - start <= (uint32_t)CAP) l = mid; else r = mid - 1;
        }
        const uint32_t p1 = l < (uint32_t)NBIN ? bins[l] : m;
        if (p1 - start > (uint32_t)CAP || nb + 1 >= (uint32_t)MAXB) {  // one bin denser than a batch: not for this path
          sh_bad = 1;
          break;
        }
        b_lo[++nb] = l;
        lo = l;
      }
      sh_nb = nb;
    }
    lds_barrier();
    if (sh_bad) {
      if (tid == 0) atomicOr(&A.C->flags, PF_FALLBACK);
      return;
    }
    n_batches = (uint32_t)__builtin_amdgcn_readfirstlane((int)sh_nb);
  }
  PT_STAMP(2);
  uint64_t carry_max = 0;
  uint32_t base = 0;
  for (uint32_t bt = 0; bt < n_batches; ++bt) {
    // (the thread index and the run length pass through an empty asm: every batch re-reads the thread's records, and a
    // compiler that sees the same loads in every iteration lifts all of them out of the loop -- some 200 registers held
    // across it.  Not the pointers themselves: behind an asm they lose their address space and the loads become flat ones.)
    {
      uint32_t n_l = n_v;  // (through a vector register: an "s" constraint on a value the compiler may hold in either file is fragile)
      asm volatile("" : "+v"(tid_v), "+v"(n_l), "+v"(member_mask), "+v"(strand_mask));
      n_v = (uint32_t)__builtin_amdgcn_readfirstlane((int)n_l);
    }
    const uint32_t bin_lo = n_batches > 1 ? (uint32_t)__builtin_amdgcn_readfirstlane((int)b_lo[bt]) : 0u;
    const uint32_t bin_hi = n_batches > 1 ? (uint32_t)__builtin_amdgcn_readfirstlane((int)b_lo[bt + 1]) : (uint32_t)NBIN;
    int shift = 0;
    while ((((bin_hi - bin_lo) << 12) >> shift) > (uint32_t)NBK) ++shift;
    const uint32_t first = (bin_lo << 12) >> shift;
    for (int b = tid; b < NBK; b += NT) cnt[b] = 0;
    lds_barrier();
    // ---- count: the batch's members among the thread's records, and their buckets
    uint32_t batch_mask = 0;
#pragma unroll
    for (int g = 0; g < ER; g += H) {
      if ((uint32_t)g * NT >= n) continue;
      uint32_t qv[H];
#pragma unroll
      for (int e = 0; e < H; ++e) qv[e] = c_qs[rec_index(g + e)];
#pragma unroll
      for (int e = 0; e < H; ++e)
        if ((member_mask >> (g + e)) & 1u) {
          uint32_t cb;
          const uint32_t fb = fine_of((strand_mask >> (g + e)) & 1u, qv[e], shift, first, &cb);
          if (cb >= bin_lo && cb < bin_hi) {
            batch_mask |= 1u << (g + e);
            atomicAdd(&cnt[fb], 1u);
          }
        }
      asm volatile("" ::: "memory");  // (keeps the next group's loads behind this group's work: registers)
    }
    lds_barrier();
    PT_STAMP(3);
    uint32_t mb;
    {
      constexpr int PER = NBK / NT;
      static_assert(NBK % NT == 0 && PER >= 1, "bucket counters per thread");
      uint32_t c[PER], sum = 0;
#pragma unroll
      for (int j = 0; j < PER; ++j) {
        c[j] = cnt[tid * PER + j];
        sum += c[j];
      }
      uint32_t off = block_excl_sum<NT>(sum, ws, &mb);
      mb = (uint32_t)__builtin_amdgcn_readfirstlane((int)mb);
#pragma unroll
      for (int j = 0; j < PER; ++j) {
        cnt[tid * PER + j] = off;
        off += c[j];
      }
    }
    lds_barrier();
    // ---- scatter (unordered inside a bucket; cnt[b] ends as the bucket's end); the thread remembers where each key went
    uint32_t slotw[ER / 2];  // two 16-bit slots per word: first the slot of the key, after the ranking the record's final place
#pragma unroll
    for (int j = 0; j < ER / 2; ++j) slotw[j] = 0;
    const uint32_t t_sc = fresh_tid();
#pragma unroll
    for (int g = 0; g < ER; g += H) {
      if ((uint32_t)g * NT >= n) continue;
      uint32_t qv[H], ixv[H];
#pragma unroll
      for (int e = 0; e < H; ++e) ixv[e] = rec_index(g + e);
#pragma unroll
      for (int e = 0; e < H; ++e) qv[e] = c_qs[ixv[e]];
#pragma unroll
      for (int e = 0; e < H; ++e)
        if ((batch_mask >> (g + e)) & 1u) {
          uint32_t cb;
          const uint32_t fb = fine_of((strand_mask >> (g + e)) & 1u, qv[e], shift, first, &cb);
          const uint32_t pos = atomicAdd(&cnt[fb], 1u);
          K[pos] = qv[e];
          I[pos] = (IT)(PERM ? ixv[e] : t_sc + (uint32_t)(g + e) * NT);
          slotw[(g + e) / 2] |= pos << (16 * ((g + e) & 1));
        }
      asm volatile("" ::: "memory");
    }
    lds_barrier();
    PT_STAMP(4);
    const uint32_t plus_here = m_plus > base ? (m_plus - base < mb ? m_plus - base : mb) : 0u;  // '+' members of the batch
    {
      // order inside the buckets: final position = bucket begin + the bucket's elements that order before by (key, index).
      // The thread's ES slots advance together (one round trip of LDS reads per step, not one per slot and step).
      // (!PERM: the slot's record index and its rank share a word, index << 16 | rank; rank 0xffff = an empty slot)
      uint32_t rk[ES], rp[ES], ri[PERM ? ES : 1];
      static_assert(CAP < 0xffff, "ranks are 16 bits wide");
      const uint32_t t_rk = fresh_tid();
#pragma unroll
      for (int e = 0; e < ES; ++e) {
        const uint32_t pos = t_rk + (uint32_t)e * NT;
        rk[e] = pos < mb ? K[pos] : 0u;
        const uint32_t ix = pos < mb ? (uint32_t)I[pos] : 0u;
        if constexpr (PERM) {
          ri[e] = ix;
          rp[e] = 0xffffu;
        } else {
          rp[e] = (ix << 16) | 0xffffu;
        }
      }
      auto count_half = [&](auto off_c) {
        constexpr int OFF = decltype(off_c)::value, HS = ES / 2;
        uint32_t lo[HS], hi[HS], longest = 0;
#pragma unroll
        for (int e = 0; e < HS; ++e) {
          const uint32_t pos = t_rk + (uint32_t)(OFF + e) * NT;
          lo[e] = hi[e] = 0;
          if (pos < mb) {
            uint32_t cb;
            const uint32_t b = fine_of(pos >= plus_here ? 1u : 0u, rk[OFF + e], shift, first, &cb);
            hi[e] = cnt[b];
            lo[e] = b ? cnt[b - 1] : 0u;
            rp[OFF + e] = (rp[OFF + e] & 0xffff0000u) | lo[e];
            longest = hi[e] - lo[e] > longest ? hi[e] - lo[e] : longest;
          }
        }
        for (uint32_t it = 0; it < longest; ++it) {
#pragma unroll
          for (int e = 0; e < HS; ++e) {
            const uint32_t x = lo[e] + it;
            if (x < hi[e]) {
              const uint32_t kx = K[x];
              uint32_t before = kx < rk[OFF + e] ? 1u : 0u;
              if (kx == rk[OFF + e]) {  // (a tie on the key: only then is the other element's index read)
                uint32_t mine;
                if constexpr (PERM) mine = ri[OFF + e]; else mine = rp[OFF + e] >> 16;
                before = (uint32_t)I[x] < mine ? 1u : 0u;
              }
              rp[OFF + e] += before;
            }
          }
        }
      };
      count_half(std::integral_constant<int, 0>{});
      count_half(std::integral_constant<int, ES / 2>{});
      lds_barrier();
#pragma unroll
      for (int e = 0; e < ES; ++e)
        if ((rp[e] & 0xffffu) != 0xffffu) {
          const uint32_t r = rp[e] & 0xffffu;
          K[r] = rk[e];
          if constexpr (PERM) I[r] = ri[e]; else I[r] = (uint16_t)(rp[e] >> 16);
          RR[t_rk + (uint32_t)e * NT] = (uint16_t)r;
        }
    }
    lds_barrier();
    PT_STAMP(5);
    // ---- the sorted q_start and record index out; where the thread's own records went
    const uint32_t t_out = fresh_tid();
#pragma unroll
    for (int e = 0; e < ES; ++e) {
      const uint32_t p = t_out + (uint32_t)e * NT;
      if (p < mb) {
        o_qs[base + p] = K[p];
        o_idx[base + p] = PERM ? (uint32_t)I[p] : a + (uint32_t)I[p];
        o_pred[base + p] = NONE;
      }
    }
    uint32_t qs_r[ES];  // the thread's own ES consecutive positions, for the unit cuts
    read_block<ES>(K, qs_r);
#pragma unroll
    for (int j = 0; j < ER / 2; ++j) {
      const uint32_t w = slotw[j];
      const uint32_t r0 = (batch_mask >> (2 * j)) & 1u ? RR[w & 0xffffu] : 0u, r1 = (batch_mask >> (2 * j + 1)) & 1u ? RR[w >> 16] : 0u;
      slotw[j] = r0 | (r1 << 16);
      asm volatile("" : "+v"(slotw[j]));  // (kept packed: the compiler would otherwise carry the 32 places as 32 registers)
    }
    lds_barrier();
    PT_STAMP(6);
    // ---- the other columns, transposed through LDS: coalesced reads in input order land at their sorted position, coalesced
    // writes follow
    auto put_group = [&](int g, const uint32_t (&v)[H]) {
#pragma unroll
      for (int e = 0; e < H; ++e)
        if ((batch_mask >> (g + e)) & 1u) K[(slotw[(g + e) / 2] >> (16 * ((g + e) & 1))) & 0xffffu] = v[e];
    };
    auto column = [&](const uint32_t* src, uint32_t* dst) {
      tid_v = fresh_tid();
#pragma unroll
      for (int g = 0; g < ER; g += 2 * H) {
        if ((uint32_t)g * NT >= n) continue;
        uint32_t v[H], w[H];
#pragma unroll
        for (int e = 0; e < H; ++e) v[e] = src[rec_index(g + e)];
        if (g + H < ER) {
#pragma unroll
          for (int e = 0; e < H; ++e) w[e] = src[rec_index(g + H + e)];
        }
        put_group(g, v);
        if (g + H < ER) put_group(g + H, w);
        asm volatile("" ::: "memory");
      }
      lds_barrier();
      const uint32_t t_st = fresh_tid();
#pragma unroll
      for (int e = 0; e < ES; ++e) {
        const uint32_t p = t_st + (uint32_t)e * NT;
        if (p < mb) dst[base + p] = K[p];
      }
    };
    column(c_qe, o_qe);
    {
      uint32_t qe_r[ES];
      read_block<ES>(K, qe_r);
#pragma unroll
      for (int e = 0; e < ES; ++e) degenerate |= (uint32_t)tid * ES + e < mb && qs_r[e] >= qe_r[e];
      unit_starts<NT, ES>(qs_r, qe_r, mb, base, m_plus, A.max_gap, ws64, cellmin, &carry_max);  // (its barriers also close the column)
    }
    lds_barrier();
    PT_STAMP(7);
    column(c_ts, o_ts);
    uint32_t ts_r[ES];
    read_block<ES>(K, ts_r);
    lds_barrier();
    PT_STAMP(8);
    column(c_te, o_te);
    {
      uint32_t te_r[ES];
      read_block<ES>(K, te_r);
#pragma unroll
      for (int e = 0; e < ES; ++e) degenerate |= (uint32_t)tid * ES + e < mb && ts_r[e] >= te_r[e];
    }
    lds_barrier();
    PT_STAMP(9);
    if (A.s_m) {  // (nullptr: nobody asks for the chains' weighted identities -- no identity floor, no scaffold filter with limits)
      column(c_m, o_m);
      lds_barrier();
      PT_STAMP(10);
      column(c_b, o_b);
    }
    base += mb;
    lds_barrier();
    PT_STAMP(11);
  }
  if (A.check_degenerate && __any(degenerate) && (tid & 63) == 0) atomicOr(&A.C->flags, PF_FALLBACK);
  if (tid == 0) emit_chunks(A, rk_run, a, m, m_plus, cellmin, NCELL);
  PT_STAMP(12);
}

// Runs longer than one LDS batch: the key range is cut into coarse bins, consecutive bins are glued into batches of at most CAP
// members, every batch is bucket-sorted like a small pair (its members picked out of the run by two passes over the run's
// q_start column), and the other columns are gathered by record index.
template <int NT, int E, int NBK>
constexpr size_t pair_sort_xl_lds_bytes() {
  return (size_t)NT * E * 12 + (size_t)NBK * 4 + 4096 * 4 + 129 * 4 + (size_t)PAIR_XL_CELLS * 4 + (size_t)(NT / 64 + 1) * 12 + 16 * 4 + 128;
}
template <int NT, int E, int NBK>
__device__ __forceinline__ void pair_sort_xl_body(const PairSortArgs& A, const uint32_t rk_run, char* lds_raw) {
  constexpr int CAP = NT * E;
  constexpr int NBIN = 4096, MAXB = 128, U = 8;
  constexpr size_t O_K = 0, O_QE = O_K + (size_t)CAP * 4, O_I = O_QE + (size_t)CAP * 4, O_CNT = O_I + (size_t)CAP * 4,
                   O_BINS = O_CNT + (size_t)NBK * 4, O_BLO = O_BINS + (size_t)NBIN * 4, O_CELL = O_BLO + (size_t)(MAXB + 1) * 4,
                   O_WS64 = lds_align_up(O_CELL + (size_t)PAIR_XL_CELLS * 4, 8), O_WS = O_WS64 + (size_t)(NT / 64 + 1) * 8,
                   O_SH = O_WS + (size_t)(NT / 64 + 1) * 4;
  static_assert(O_SH + 13 * 4 <= pair_sort_xl_lds_bytes<NT, E, NBK>(), "LDS block of the work-group");
  uint32_t* const K = reinterpret_cast<uint32_t*>(lds_raw + O_K);
  uint32_t* const QE = reinterpret_cast<uint32_t*>(lds_raw + O_QE);
  uint32_t* const I = reinterpret_cast<uint32_t*>(lds_raw + O_I);
  uint32_t* const cnt = reinterpret_cast<uint32_t*>(lds_raw + O_CNT);
  uint32_t* const bins = reinterpret_cast<uint32_t*>(lds_raw + O_BINS);
  uint32_t* const b_lo = reinterpret_cast<uint32_t*>(lds_raw + O_BLO);
  uint32_t* const cellmin = reinterpret_cast<uint32_t*>(lds_raw + O_CELL);
  uint64_t* const ws64 = reinterpret_cast<uint64_t*>(lds_raw + O_WS64);
  uint32_t* const ws = reinterpret_cast<uint32_t*>(lds_raw + O_WS);
  uint32_t* const sh_cnt = reinterpret_cast<uint32_t*>(lds_raw + O_SH);
  uint32_t* const sh_kmin = sh_cnt + 4;
  uint32_t* const sh_kmax = sh_cnt + 6;
  uint32_t* const sh_first = sh_cnt + 8;
  uint32_t& sh_nb = sh_cnt[11];
  uint32_t& sh_bad = sh_cnt[12];
#ifdef SWG_PAIR_TIMING
  unsigned long long pt_last = wall_clock64();
#endif
  const int tid = threadIdx.x;
  const PairRun run = A.runs[rk_run];
  const uint32_t a = run.a, n = run.n;
  if (tid < 4) sh_cnt[tid] = 0;
  if (tid < 2) {
    sh_kmin[tid] = 0xffffffffu;
    sh_kmax[tid] = 0;
  }
  if (tid < 3) sh_first[tid] = NONE;
  if (tid == 0) sh_bad = 0;
  for (int c = tid; c < (int)PAIR_XL_CELLS; c += NT) cellmin[c] = NONE;
  for (int b = tid; b < NBIN; b += NT) bins[b] = 0;
  __syncthreads();
  const uint32_t q0 = A.q_id[a], t0 = A.t_id[a];
  const bool self_ok = A.keep_self || q0 != t0;
  // ---- step-1 retain, members, the key range per strand (U records per thread requested together)
  {
    uint32_t c_m[2] = {0, 0}, c_x = 0, kmin[2] = {0xffffffffu, 0xffffffffu}, kmax[2] = {0, 0}, fst[3] = {NONE, NONE, NONE};
    for (uint32_t l0 = 0; l0 < n; l0 += NT * U) {
      uint8_t stv[U], av[U], mv[U];
      uint32_t qv[U], blv[U];
      double idv[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const uint32_t li = l0 + (uint32_t)u * NT + (uint32_t)tid;
        const uint32_t i = a + (li < n ? li : 0u);
        stv[u] = A.strand[i];
        qv[u] = A.q_start[i];
        av[u] = A.alive_in ? A.alive_in[i] : (uint8_t)1;
        mv[u] = A.member_in ? A.member_in[i] : (uint8_t)1;
        blv[u] = (!A.alive_in && (A.min_block != 0 || !A.identity)) ? A.block_len[i] : 0u;
        idv[u] = A.alive_in ? 1.0 : (A.identity ? A.identity[i] : (double)A.matches[i]);
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const uint32_t li = l0 + (uint32_t)u * NT + (uint32_t)tid;
        if (li >= n) continue;
        const uint32_t i = a + li;
        bool alive;
        if (A.alive_in) {
          alive = av[u] != 0;
        } else {
          const double id = A.identity ? idv[u] : __ddiv_rn(idv[u], (double)(blv[u] > 1u ? blv[u] : 1u));
          alive = self_ok && (A.min_block == 0 || (uint64_t)blv[u] >= A.min_block) && id >= A.min_identity;
        }
        const bool member = alive && mv[u] != 0;
        const uint32_t st = stv[u] ? 1u : 0u;
        A.code[i] = alive ? (uint8_t)((member ? 1u : 2u) | (st << 2)) : (uint8_t)0;
        if (alive) {
          if (fst[2] == NONE) fst[2] = i;
          if (member) {
#pragma unroll
            for (uint32_t s2 = 0; s2 < 2; ++s2)
              if (st == s2) {
                ++c_m[s2];
                kmin[s2] = qv[u] < kmin[s2] ? qv[u] : kmin[s2];
                kmax[s2] = qv[u] > kmax[s2] ? qv[u] : kmax[s2];
                if (fst[s2] == NONE) fst[s2] = i;
              }
          } else {
            ++c_x;
          }
        }
      }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      c_m[0] += __shfl_xor(c_m[0], o, 64);
      c_m[1] += __shfl_xor(c_m[1], o, 64);
      c_x += __shfl_xor(c_x, o, 64);
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const uint32_t x = __shfl_xor(kmin[s], o, 64), y = __shfl_xor(kmax[s], o, 64);
        kmin[s] = x < kmin[s] ? x : kmin[s];
        kmax[s] = y > kmax[s] ? y : kmax[s];
      }
#pragma unroll
      for (int s = 0; s < 3; ++s) {
        const uint32_t x = __shfl_xor(fst[s], o, 64);
        fst[s] = x < fst[s] ? x : fst[s];
      }
    }
    if ((tid & 63) == 0) {
      if (c_m[0]) atomicAdd(&sh_cnt[0], c_m[0]);
      if (c_m[1]) atomicAdd(&sh_cnt[1], c_m[1]);
      if (c_x) atomicAdd(&sh_cnt[2], c_x);
#pragma unroll
      for (int s = 0; s < 2; ++s)
        if (c_m[s]) {
          atomicMin(&sh_kmin[s], kmin[s]);
          atomicMax(&sh_kmax[s], kmax[s]);
        }
#pragma unroll
      for (int s = 0; s < 3; ++s)
        if (fst[s] != NONE) atomicMin(&sh_first[s], fst[s]);
    }
  }
  __syncthreads();
  const uint32_t m_plus = sh_cnt[0], m = sh_cnt[0] + sh_cnt[1], n_x = sh_cnt[2], M = m + n_x;
  PT_STAMP(1);
  if (tid == 0) {
    PairInfo pi;
    pi.m = m;
    pi.m_plus = m_plus;
    pi.M = M;
    if (A.orig) {  // (a pair-major copy: first appearances in the caller's own record indices -- ascending inside the pair)
#pragma unroll
      for (int j = 0; j < 3; ++j)
        if (sh_first[j] != NONE) sh_first[j] = A.orig[sh_first[j]];
    }
    pi.first_alive = sh_first[2];
    pi.first_mem[0] = sh_first[0];
    pi.first_mem[1] = sh_first[1];
    pi.q = q0;
    pi.t = t0;
    A.info[rk_run] = pi;
    if (sh_first[2] != NONE) {
      uint32_t* slot = pair_slot(A.gl_first, A.seq_genome_last[q0], A.seq_genome_last[t0]);
      if (*slot > sh_first[2]) atomicMin(slot, sh_first[2]);
    }
    // (the alive records' and the members' totals: summed over `info` by pair_totals_kernel -- a counter that every pair bumps
    // serialises on its cache line: 4.5 of 6.8 ms of the small pairs' sort at 198,000 pairs)
  }
  if (M == 0) return;
  bool degenerate = false;
  if (n_x) {
    uint32_t done = 0;
    for (uint32_t l0 = 0; l0 < n; l0 += NT) {
      const uint32_t li = l0 + tid;
      const uint8_t code = li < n ? A.code[a + li] : (uint8_t)0;
      const bool x = (code & 3u) == 2u;
      uint32_t tot;
      const uint32_t r = block_excl_sum<NT>(x ? 1u : 0u, ws, &tot);
      if (x) {
        const uint32_t i = a + li, p = a + m + done + r;
        const uint32_t qs = A.q_start[i], qe = A.q_end[i], ts = A.t_start[i], te = A.t_end[i];
        A.s_qs[p] = qs;
        A.s_qe[p] = qe;
        A.s_ts[p] = ts;
        A.s_te[p] = te;
        A.s_idx[p] = i | ((uint32_t)(code >> 2) << 31);
        degenerate |= qs >= qe || ts >= te;
      }
      done += tot;
    }
  }
  if (m == 0) {
    if (A.check_degenerate && __any(degenerate) && (tid & 63) == 0) atomicOr(&A.C->flags, PF_FALLBACK);
    return;
  }
  // a pass over the run's members: f(li, strand, q_start), U records per thread requested together
  auto for_members = [&](auto&& f) {
    for (uint32_t l0 = 0; l0 < n; l0 += NT * U) {
      uint8_t cv[U];
      uint32_t qv[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const uint32_t li = l0 + (uint32_t)u * NT + (uint32_t)tid;
        cv[u] = li < n ? A.code[a + li] : (uint8_t)0;
        qv[u] = A.q_start[a + (li < n ? li : 0u)];
      }
#pragma unroll
      for (int u = 0; u < U; ++u)
        if ((cv[u] & 3u) == 1u) f(l0 + (uint32_t)u * NT + (uint32_t)tid, (uint32_t)(cv[u] >> 2), qv[u]);
    }
  };
  // ---- coarse bins -> batches
  BucketMap BM;
  {
    const uint32_t kmn[2] = {sh_kmin[0], sh_kmin[1]}, kmx[2] = {sh_kmax[0], sh_kmax[1]};
    bucket_map_make(BM, NBIN, m_plus, m - m_plus, kmn, kmx);
  }
  for_members([&](uint32_t, uint32_t st, uint32_t k) { atomicAdd(&bins[bucket_of(BM, st, k)], 1u); });
  __syncthreads();
  PT_STAMP(2);
  {  // bins -> their exclusive prefix sums (NBIN / NT consecutive bins per thread)
    constexpr int PERB = NBIN / NT;
    static_assert(NBIN % NT == 0 && PERB >= 1, "bins per thread");
    uint32_t c[PERB], sum = 0, tot;
#pragma unroll
    for (int j = 0; j < PERB; ++j) {
      c[j] = bins[tid * PERB + j];
      sum += c[j];
    }
    uint32_t off = block_excl_sum<NT>(sum, ws, &tot);
    __syncthreads();
#pragma unroll
    for (int j = 0; j < PERB; ++j) {
      bins[tid * PERB + j] = off;
      off += c[j];
    }
  }
  __syncthreads();
  if (tid == 0) {  // greedy: a batch takes as many bins as fit (a binary search in the prefix sums per batch)
    uint32_t nb = 0, lo = 0;
    b_lo[0] = 0;
    while (lo < (uint32_t)NBIN) {
      const uint32_t start = bins[lo];
      uint32_t l = lo + 1, r = NBIN;  // largest hi in (lo, NBIN] with prefix(hi) - prefix(lo) <= CAP; prefix(NBIN) = m
      while (l < r) {
        const uint32_t mid = l + ((r - l + 1) >> 1);
        const uint32_t pm = mid < (uint32_t)NBIN ? bins[mid] : m;
        if (pm - start <= (uint32_t)CAP) l = mid; else r = mid - 1;
      }
      const uint32_t p1 = l < (uint32_t)NBIN ? bins[l] : m;
      if (p1 - start > (uint32_t)CAP || nb + 1 >= (uint32_t)MAXB) {  // one bin denser than a batch (or too many batches): not for this path
        sh_bad = 1;
        break;
      }
      b_lo[++nb] = l;
      lo = l;
    }
    sh_nb = nb;
  }
  __syncthreads();
  if (sh_bad) {
    if (tid == 0) atomicOr(&A.C->flags, PF_FALLBACK);
    return;
  }
  PT_STAMP(3);
  const uint32_t n_batches = sh_nb;
  uint64_t carry_max = 0;
  uint32_t base = 0;
  for (uint32_t bt = 0; bt < n_batches; ++bt) {
    const uint32_t bin_lo = b_lo[bt], bin_hi = b_lo[bt + 1];
    // fine buckets of the batch: the coarse map refined by a power of two, relative to the batch's first bin.  f < NBIN = 2^12
    // has 12 integer bits and multiplying a float by 2^12 is exact, so (uint32)(f * 4096) >> 12 == (uint32)f: the fine id
    // names the coarse bin in its high bits and stays monotone in the key.
    int shift = 0;
    while ((((bin_hi - bin_lo) << 12) >> shift) > (uint32_t)NBK) ++shift;
    const uint32_t first = (bin_lo << 12) >> shift;
    auto fine_of = [&](uint32_t st, uint32_t k, uint32_t* coarse) -> uint32_t {
      const float f = (float)(k - (st ? BM.kmin[1] : BM.kmin[0])) * (st ? BM.scale[1] : BM.scale[0]);
      const uint32_t top = (st ? BM.nb[1] : BM.nb[0]) - 1u;
      uint32_t g = (uint32_t)(f * 4096.0f);
      if ((g >> 12) > top) g = (top << 12) | 0xfffu;  // (the clamp of bucket_of, in fine units)
      g += (st ? BM.off[1] : BM.off[0]) << 12;
      *coarse = g >> 12;
      const uint32_t b = (g >> shift) - first;
      return b < (uint32_t)NBK ? b : (uint32_t)NBK - 1u;
    };
    for (int b = tid; b < NBK; b += NT) cnt[b] = 0;
    __syncthreads();
    for_members([&](uint32_t, uint32_t st, uint32_t k) {
      uint32_t cb;
      const uint32_t fb = fine_of(st, k, &cb);
      if (cb >= bin_lo && cb < bin_hi) atomicAdd(&cnt[fb], 1u);
    });
    __syncthreads();
    PT_STAMP(4);
    uint32_t mb;
    {
      constexpr int PER = NBK / NT;
      static_assert(NBK % NT == 0 && PER >= 1, "bucket counters per thread");
      uint32_t c[PER], sum = 0;
#pragma unroll
      for (int j = 0; j < PER; ++j) {
        c[j] = cnt[tid * PER + j];
        sum += c[j];
      }
      uint32_t off = block_excl_sum<NT>(sum, ws, &mb);
#pragma unroll
      for (int j = 0; j < PER; ++j) {
        cnt[tid * PER + j] = off;
        off += c[j];
      }
    }
    __syncthreads();
    for_members([&](uint32_t li, uint32_t st, uint32_t k) {
      uint32_t cb;
      const uint32_t fb = fine_of(st, k, &cb);
      if (cb >= bin_lo && cb < bin_hi) {
        const uint32_t pos = atomicAdd(&cnt[fb], 1u);
        K[pos] = k;
        I[pos] = li;
      }
    });
    __syncthreads();
    PT_STAMP(5);
    const uint32_t plus_here = m_plus > base ? (m_plus - base < mb ? m_plus - base : mb) : 0u;  // '+' members of the batch
    {
      uint32_t rk[E], rl[E], rr[E];
#pragma unroll
      for (int e = 0; e < E; ++e) {
        const uint32_t pos = (uint32_t)tid + (uint32_t)e * NT;
        rr[e] = NONE;
        if (pos < mb) {
          const uint32_t k = K[pos], li = I[pos];
          uint32_t cb;
          const uint32_t b = fine_of(pos >= plus_here ? 1u : 0u, k, &cb);
          const uint32_t hi = cnt[b], lo = b ? cnt[b - 1] : 0u;
          uint32_t r = lo;
          for (uint32_t x = lo; x < hi; ++x) {
            const uint32_t kx = K[x], lx = I[x];
            r += (kx < k || (kx == k && lx < li)) ? 1u : 0u;
          }
          rk[e] = k;
          rl[e] = li;
          rr[e] = r;
        }
      }
      __syncthreads();
#pragma unroll
      for (int e = 0; e < E; ++e)
        if (rr[e] != NONE) {
          K[rr[e]] = rk[e];
          I[rr[e]] = rl[e];
        }
    }
    __syncthreads();
    PT_STAMP(6);
    // ---- the batch's columns out, gathered by record index (every load of the thread requested together)
    {
      uint32_t li[E], qe[E], ts[E], te[E], mm[E], bb[E];
#pragma unroll
      for (int e = 0; e < E; ++e) {
        const uint32_t p = (uint32_t)tid + (uint32_t)e * NT;
        li[e] = p < mb ? I[p] : 0u;
      }
#pragma unroll
      for (int e = 0; e < E; ++e) {
        const uint32_t i = a + li[e];
        qe[e] = A.q_end[i];
        ts[e] = A.t_start[i];
        te[e] = A.t_end[i];
        mm[e] = A.s_m ? A.matches[i] : 0u;
        bb[e] = A.s_m ? A.block_len[i] : 0u;
      }
#pragma unroll
      for (int e = 0; e < E; ++e) {
        const uint32_t p = (uint32_t)tid + (uint32_t)e * NT;
        if (p >= mb) continue;
        const uint32_t o = a + base + p, qs = K[p];
        QE[p] = qe[e];
        A.s_qs[o] = qs;
        A.s_idx[o] = a + li[e];
        A.pred[o] = NONE;
        A.s_qe[o] = qe[e];
        A.s_ts[o] = ts[e];
        A.s_te[o] = te[e];
        if (A.s_m) {
          A.s_m[o] = mm[e];
          A.s_b[o] = bb[e];
        }
        degenerate |= qs >= qe[e] || ts[e] >= te[e];
      }
    }
    __syncthreads();
    PT_STAMP(7);
    {
      uint32_t qs_r[E], qe_r[E];
      read_block<E>(K, qs_r);
      read_block<E>(QE, qe_r);
      unit_starts<NT, E>(qs_r, qe_r, mb, base, m_plus, A.max_gap, ws64, cellmin, &carry_max);
    }
    base += mb;
    __syncthreads();
    PT_STAMP(8);
  }
  if (A.check_degenerate && __any(degenerate) && (tid & 63) == 0) atomicOr(&A.C->flags, PF_FALLBACK);
  if (tid == 0) emit_chunks(A, rk_run, a, m, m_plus, cellmin, (int)PAIR_XL_CELLS);
  PT_STAMP(9);
}

template <int NT, int ES, int ER, int NBK, int NBIN, bool PERM>
__global__ __launch_bounds__(NT) void pair_sort_kernel(PairSortArgs A) {
  __shared__ __attribute__((aligned(16))) char raw[pair_sort_lds_bytes<NT, ES, ER, NBK, NBIN, PERM>()];
  pair_sort_body<NT, ES, ER, NBK, NBIN, PERM>(A, A.list[blockIdx.x], raw);
}
// The two largest size classes in one launch: the few very long runs first (they last longest), the others fill the chip
// beside them (launched on their own the long runs keep a handful of CUs busy and the rest of the chip waits).
constexpr size_t PAIR_BIG_LDS = pair_sort_lds_bytes<1024, 16, 32, 4096, 1024, false>() > pair_sort_xl_lds_bytes<1024, 8, 4096>()
                                    ? pair_sort_lds_bytes<1024, 16, 32, 4096, 1024, false>()
                                    : pair_sort_xl_lds_bytes<1024, 8, 4096>();
static_assert(PAIR_BIG_LDS <= 160 * 1024, "LDS of a CU");
__global__ __launch_bounds__(1024) void pair_sort_big_kernel(PairSortArgs A, const uint32_t* __restrict__ list_xl, uint32_t n_xl) {
  __shared__ __attribute__((aligned(16))) char raw[PAIR_BIG_LDS];
  if (blockIdx.x < n_xl)
    pair_sort_xl_body<1024, 8, 4096>(A, list_xl[blockIdx.x], raw);
  else
    pair_sort_body<1024, 16, 32, 4096, 1024, false>(A, A.list[blockIdx.x - n_xl], raw);
}

// A call that is being handed over to the global-sort stage (PF_FALLBACK raised by a pair_sort work-group or by the plan of
// the long units): the walk and the labelling take their chunk counts from the device, so clearing them makes the kernels
// still to come return at once instead of walking and labelling 10^8 members for nothing.
// The statistics' totals over the pairs, once per call: alive records, members, kept chains, records in the output.
__global__ __launch_bounds__(1024) void pair_totals_kernel(uint32_t n_runs, const PairInfo* __restrict__ info, const PairSum* __restrict__ sum,
                                                           const uint32_t* __restrict__ n_out_pair, PairCounters* __restrict__ C) {
  __shared__ unsigned long long part[16][4];
  unsigned long long v[4] = {0, 0, 0, 0};
  const bool finished = !(C->flags & PF_FALLBACK);  // (else the sums were not written)
  for (uint32_t k = blockIdx.x * 1024u + threadIdx.x; k < n_runs; k += gridDim.x * 1024u) {
    const PairInfo pi = info[k];
    v[0] += pi.M;
    v[1] += pi.m;
    if (finished) {
      v[2] += pi.M ? sum[k].n_kept : 0u;
      v[3] += n_out_pair[k];
    }
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v[j] += __shfl_xor(v[j], o, 64);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6][j] = v[j];
  }
  __syncthreads();
  if (threadIdx.x < 4) {
    unsigned long long t = 0;
    for (int w = 0; w < 16; ++w) t += part[w][threadIdx.x];
    unsigned long long* dst = threadIdx.x == 0 ? &C->n_alive : threadIdx.x == 1 ? &C->n_members : threadIdx.x == 2 ? &C->n_kept : &C->n_out;
    if (t) atomicAdd(dst, t);  // (a few dozen work-groups; the counters start from zero with every call)
  }
}
__global__ void pair_gate_kernel(PairCounters* __restrict__ C) {
  if (threadIdx.x == 0 && (C->flags & PF_FALLBACK)) {
    C->n_chunks = 0u;
    C->n_long = 0u;
  }
}

// ---- pair_finish ----------------------------------------------------------------------------------------------------
struct PairFinishArgs {
  const PairRun* runs;
  const uint32_t* list;
  const PairInfo* info;
  PairSum* sum;
  const uint32_t *s_qs, *s_qe, *s_ts, *s_te, *s_idx, *hd;
  const uint8_t* ok_head;
  const HeadRec* rec;
  uint32_t* head_num;                    // (the predecessor array, free after the labelling) rank of a kept chain's head
  uint32_t *f_qs, *f_qe, *f_ts, *f_pm;   // the pair's kept '+' chains in q_start order (scratch at the pair's offset)
  uint8_t* status;
  uint32_t* chain;
  int scaffolds_only;
  uint64_t gap;
  uint64_t max_dev;  // largest deviation from a chain's diagonal that the inversion capture accepts (pair_max_deviation)
  uint64_t rescue_d;   // scaffold_max_deviation (0: no rescue)
  uint64_t max_s2;     // largest q^2 + t^2 whose truncated square root is <= rescue_d (pair_max_dist2)
  const uint8_t* kept_in;      // a scaffold sweep with limits ran over the chain table (pair_chains_kernel): kept flag per chain ...
  const uint32_t* chain_base;  // ... whose entries of pair k start here, in the pair's all_chains order; np[2k], np[2k + 1]: the pair's
  const uint32_t* np;          //     passing chains per strand
  uint32_t* fin;       // by position in the pair, or nullptr: the record's result -- status << 30 | pair-local chain number -- instead
                       //   of scattered writes to the output columns (pair_out_kernel brings them to input order through LDS)
  uint32_t* anum;      // by position in the pair: the anchor's (pair-local) chain number, 0 = a rescue candidate, NEVER = never rescued
  uint32_t* n_out_pair;  // [pairs, zeroed] the pair's records in the output
  PairCounters* C;
};
constexpr uint32_t NEVER = 0xfffffffeu;  // a member of a chain that passed the span / identity filter but not the scaffold sweep

// A scaffold sweep with limits (--scaffold-filter 1:1 ...): the chains that pass the span / identity filter go into one chain
// table for the whole input -- every pair a contiguous stretch in ITS all_chains order (the '+' or the '-' chains first, as the
// (query, target, strand) groups appear in the metadata; the stretches themselves in whatever order the atomics hand them out:
// the sweep's segments are the pairs and its index tie-break only ever compares chains of one pair) -- and plane_sweep_both
// runs over that table on the sweep kernels (swg_sweep_axis).  pair_finish then reads the kept flags back per pair.
struct PairChainArgs {
  const PairRun* runs;
  const uint32_t* list;
  const PairInfo* info;
  const uint8_t* ok_head;
  const HeadRec* rec;
  uint32_t *T_qs, *T_qe, *T_ts, *T_te;
  double* T_wid;
  uint64_t* T_seg;
  uint32_t *chain_base, *np;
  uint32_t *cnt, *has, *stretch;  // per pair: its chains in the table, whether it has any; the exclusive prefix sums of `has`
  PairRun* T_runs;             // the pairs' stretches of the table (those that hold a chain)
  unsigned long long* totals;  // [0] chains in the table, [1] largest coordinate, [2] stretches
  const PairCounters* C;
};
// PHASE 0 counts a pair's passing chains; the host's prefix sums over the pairs place the stretches (a returning atomic per
// pair on one counter serialises: 9.3 ms at 396,000 pairs); PHASE 1 fills the pair's stretch.
template <int NT, int PHASE>
__global__ __launch_bounds__(NT) void pair_chains_kernel(PairChainArgs A) {
  constexpr int U = 4;
  __shared__ uint32_t ws[NT / 64 + 1];
  __shared__ uint64_t ws64[NT / 64 + 1];
  __shared__ uint32_t sh_base;
  const int tid = threadIdx.x;
  if (A.C->flags & PF_FALLBACK) return;
  const uint32_t rk = A.list[blockIdx.x];
  const PairRun run = A.runs[rk];
  const PairInfo pi = A.info[rk];
  const uint32_t a = run.a, m = pi.m, m_plus = pi.m_plus;
  if (PHASE == 0) {
    uint32_t c0 = 0, c1 = 0, mx = 0;
    for (uint32_t p0 = 0; p0 < m; p0 += NT * U) {
      uint8_t ok[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const uint32_t p = p0 + (uint32_t)u * NT + (uint32_t)tid;
        ok[u] = p < m ? A.ok_head[a + p] : (uint8_t)0;
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const uint32_t p = p0 + (uint32_t)u * NT + (uint32_t)tid;
        if (ok[u]) {
          const HeadRec hr = A.rec[a + p];
          if (p < m_plus) ++c0; else ++c1;
          const uint32_t e = hr.qe > hr.te ? hr.qe : hr.te;
          mx = e > mx ? e : mx;
        }
      }
    }
    const uint32_t nP = block_sum<NT>(c0, ws), nM = block_sum<NT>(c1, ws);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const uint32_t t = __shfl_xor(mx, o, 64);
      mx = t > mx ? t : mx;
    }
    // (the largest coordinate only ever grows: a look first keeps all but a few atomics off the counter)
    if ((tid & 63) == 0 && (unsigned long long)mx > __hip_atomic_load(&A.totals[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
      atomicMax(&A.totals[1], (unsigned long long)mx);
    if (tid == 0) {
      A.np[2 * rk] = nP;
      A.np[2 * rk + 1] = nM;
      A.cnt[rk] = nP + nM;
      A.has[rk] = nP + nM ? 1u : 0u;
    }
    return;
  }
  const uint32_t nP = A.np[2 * rk], nM = A.np[2 * rk + 1];
  if (nP + nM == 0) return;
  const uint32_t cb = A.chain_base[rk];
  if (tid == 0) {  // the pair's stretch of the table: a run of the sweep's input (its begins are sorted segment by segment)
    PairRun tr;
    tr.a = cb;
    tr.n = nP + nM;
    A.T_runs[A.stretch[rk]] = tr;
  }
  (void)sh_base;
  const bool plus_first = pi.first_mem[0] < pi.first_mem[1];
  uint32_t before = 0;
  for (uint32_t p0 = 0; p0 < m; p0 += NT * U) {
    uint8_t ok[U];
    uint64_t packed = 0;
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const uint32_t p = p0 + (uint32_t)u * NT + (uint32_t)tid;
      ok[u] = p < m ? A.ok_head[a + p] : (uint8_t)0;
      packed |= (uint64_t)(ok[u] ? 1u : 0u) << (16 * u);
    }
    uint64_t inc = packed;
    const int lane = tid & 63, w = tid >> 6;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const uint64_t t = __shfl_up(inc, d, 64);
      if (lane >= d) inc += t;
    }
    uint64_t off = 0, tot = 0;
    if (NT == 64) {
      tot = __shfl(inc, 63, 64);
    } else {
      __syncthreads();
      if (lane == 63) ws64[w] = inc;
      __syncthreads();
#pragma unroll
      for (int k = 0; k < NT / 64; ++k) {
        const uint64_t x = ws64[k];
        off += k < w ? x : 0ull;
        tot += x;
      }
    }
    const uint64_t ex = off + inc - packed;
    uint32_t row_base = before;
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const uint32_t p = p0 + (uint32_t)u * NT + (uint32_t)tid;
      if (ok[u]) {
        const uint32_t c_pos = row_base + (uint32_t)((ex >> (16 * u)) & 0xffffu);  // rank in position order ('+' first)
        const uint32_t c_all = plus_first ? c_pos : (p >= m_plus ? c_pos - nP : c_pos + nM);
        const HeadRec hr = A.rec[a + p];
        const uint32_t c = cb + c_all;
        A.T_qs[c] = hr.qs;
        A.T_qe[c] = hr.qe;
        A.T_ts[c] = hr.ts;
        A.T_te[c] = hr.te;
        A.T_wid[c] = hr.wid;
        A.T_seg[c] = rk;
      }
      row_base += (uint32_t)((tot >> (16 * u)) & 0xffffu);
    }
    before = row_base;
  }
}

// plane_sweep_both with no limit on either axis (plane_sweep_exact.rs:268-461 with mappings_to_keep = usize::MAX): a sweep
// over at most one interval returns it; otherwise an interval survives iff it is ever in the tree at a mark_good call, i.e.
// iff start < end.  The target sweep runs over the query sweep's survivors.  So a chain with both spans positive is always
// kept, and only a pair that holds a chain with an empty span needs the counts: the sweep below first ranks the chains under
// "both spans positive", and runs once more under the exact rule if it met such a chain.
// KP: the kept '+' chains of a pair that are staged in LDS for the inversion capture (a longer list is searched in memory).
template <int NT, int KP>
__global__ __launch_bounds__(NT, NT >= 512 ? 4 : 1) void pair_finish_kernel(PairFinishArgs A) {  // (large class: two work-groups per CU)
  constexpr int U = 4;  // positions per thread and round: their loads are requested together
  __shared__ uint32_t ws[NT / 64 + 1];
  __shared__ uint64_t ws64[NT / 64 + 1];
  __shared__ uint64_t ws64n[2 * (NT / 64 + 1)];  // (the ranking's scan of two packed words)
  __shared__ uint32_t l_all[4 * KP];  // the kept '+' chains for the inversion capture, then the rescue's staged members
  uint32_t* const l_qs = l_all;
  uint32_t* const l_qe = l_all + KP;
  uint32_t* const l_ts = l_all + 2 * KP;
  uint32_t* const l_pm = l_all + 3 * KP;
  const int tid = threadIdx.x;
#ifdef SWG_PAIR_TIMING
  unsigned long long pt_last = wall_clock64();
#endif
  // a pair_sort work-group gave the call up (a pair too dense for the LDS batches ...): nothing written from here on is used
  // (the host sees the same flag and runs the global-sort stage)
  if (A.C->flags & PF_FALLBACK) return;
  const uint32_t rk = A.list[blockIdx.x];
  const PairRun run = A.runs[rk];
  const PairInfo pi = A.info[rk];
  const uint32_t a = run.a, m = pi.m, m_plus = pi.m_plus, M = pi.M;
  PairSum sm;
  sm.n_pass = sm.n_kept = 0;
  sm.minmem = NONE;
  sm.base = 0;
  if (m == 0) {
    if (tid == 0) A.sum[rk] = sm;
    return;
  }
  // ---- the chains that pass the span / identity filter, ranked in position order ('+' chains first); the kept '+' chains
  // listed for the inversion capture.  exact: 0 = a chain is kept iff both its spans are positive (counts the others);
  // 1 = the rule above with the counts of the first sweep.
  uint32_t n_pass0 = 0, n_pass1 = 0, n_q = 0, n_deg = 0, n_kept = 0, kP = 0;
  bool all_q = false, all_t = false;
  const bool given = A.kept_in != nullptr;  // the kept flags come from the sweep over the chain table
  const uint32_t g_nP = given ? A.np[2 * rk] : 0u, g_nM = given ? A.np[2 * rk + 1] : 0u, g_cb = given ? A.chain_base[rk] : 0u;
  const bool plus_first0 = pi.first_mem[0] < pi.first_mem[1];
  // exclusive prefix of four 16-bit counters (one per row of the round) over the threads, and the rows' totals
  auto row_scan = [&](uint64_t packed, uint64_t* tot_out) -> uint64_t {
    uint64_t inc = packed;
    const int lane = tid & 63, w = tid >> 6;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const uint64_t t = __shfl_up(inc, d, 64);
      if (lane >= d) inc += t;
    }
    uint64_t off = 0, tot = 0;
    if (NT == 64) {
      tot = __shfl(inc, 63, 64);
    } else {
      lds_barrier();
      if (lane == 63) ws64[w] = inc;
      lds_barrier();
#pragma unroll
      for (int k = 0; k < NT / 64; ++k) {
        const uint64_t x = ws64[k];
        off += k < w ? x : 0ull;
        tot += x;
      }
    }
    *tot_out = tot;
    return off + inc - packed;
  };
  // the same for G packed words at once (4 G rows of a round: one pair of barriers whatever G is)
  auto row_scan_n = [&](auto g_c, const uint64_t (&packed)[decltype(g_c)::value], uint64_t (&ex)[decltype(g_c)::value],
                        uint64_t (&tot)[decltype(g_c)::value]) {
    constexpr int G = decltype(g_c)::value;
    uint64_t inc[G];
    const int lane = tid & 63, w = tid >> 6;
#pragma unroll
    for (int g = 0; g < G; ++g) inc[g] = packed[g];
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
#pragma unroll
      for (int g = 0; g < G; ++g) {
        const uint64_t t = __shfl_up(inc[g], d, 64);
        if (lane >= d) inc[g] += t;
      }
    }
    if (NT == 64) {
#pragma unroll
      for (int g = 0; g < G; ++g) {
        tot[g] = __shfl(inc[g], 63, 64);
        ex[g] = inc[g] - packed[g];
      }
      return;
    }
    lds_barrier();
    if (lane == 63) {
#pragma unroll
      for (int g = 0; g < G; ++g) ws64n[g * (NT / 64 + 1) + w] = inc[g];
    }
    lds_barrier();
#pragma unroll
    for (int g = 0; g < G; ++g) {
      uint64_t off = 0, t = 0;
#pragma unroll
      for (int k = 0; k < NT / 64; ++k) {
        const uint64_t x = ws64n[g * (NT / 64 + 1) + k];
        off += k < w ? x : 0ull;
        t += x;
      }
      tot[g] = t;
      ex[g] = off + inc[g] - packed[g];
    }
  };
  // Round 6: the large class ranks 8 rows of NT positions per round instead of 4 (half the rounds, half the barriers, twice the
  // loads in flight: a pair of 10,000 members was five rounds of dependent flag load -> scan -> store)
  constexpr int US = NT >= 512 ? 8 : 4;  // rows per round
  constexpr int GS = US / 4;             // packed words per scan
  auto sweep = [&](const bool exact) {
    uint32_t kept_before = 0, ok_before = 0, c0 = 0, c1 = 0, cq = 0, cd = 0, cp = 0;
    for (uint32_t p0 = 0; p0 < m; p0 += NT * US) {
      uint8_t ok[US];
      uint32_t h_qs[US], h_qe[US], h_ts[US];
#pragma unroll
      for (int u = 0; u < US; ++u) {
        const uint32_t p = p0 + (uint32_t)u * NT + (uint32_t)tid;
        ok[u] = p < m ? A.ok_head[a + p] : (uint8_t)0;
      }
      bool kept[US];
      if (given) {  // a chain's flag sits at its place in the pair's stretch of the chain table: its rank among the passing chains
        uint64_t okp[GS], ex_ok[GS], tot_ok[GS];
#pragma unroll
        for (int g = 0; g < GS; ++g) okp[g] = 0;
#pragma unroll
        for (int u = 0; u < US; ++u) okp[u / 4] |= (uint64_t)(ok[u] ? 1u : 0u) << (16 * (u % 4));
        row_scan_n(std::integral_constant<int, GS>{}, okp, ex_ok, tot_ok);
        uint32_t rb = ok_before;
#pragma unroll
        for (int u = 0; u < US; ++u) {
          const uint32_t p = p0 + (uint32_t)u * NT + (uint32_t)tid;
          kept[u] = false;
          if (ok[u]) {
            const uint32_t c_pos = rb + (uint32_t)((ex_ok[u / 4] >> (16 * (u % 4))) & 0xffffu);
            const uint32_t c_all = plus_first0 ? c_pos : (p >= m_plus ? c_pos - g_nP : c_pos + g_nM);
            kept[u] = A.kept_in[g_cb + c_all] != 0;
          }
          rb += (uint32_t)((tot_ok[u / 4] >> (16 * (u % 4))) & 0xffffu);
        }
        ok_before = rb;
      }
      uint64_t packed[GS], ex[GS], tot[GS];  // 16-bit counters: kept chains of row u among the threads before this one
#pragma unroll
      for (int g = 0; g < GS; ++g) packed[g] = 0;
#pragma unroll
      for (int u = 0; u < US; ++u) {
        const uint32_t p = p0 + (uint32_t)u * NT + (uint32_t)tid;
        if (!given) kept[u] = false;
        if (ok[u]) {
          const bool hq = (ok[u] & 2u) != 0, ht = (ok[u] & 4u) != 0;  // (the labelling left the two span tests in the flag byte)
          if (!given) kept[u] = exact ? (all_q || hq) && (all_t || ht) : hq && ht;
          if (p < m_plus) ++c0; else ++c1;
          cq += hq ? 1u : 0u;
          cd += (given || (hq && ht)) ? 0u : 1u;
          cp += (kept[u] && p < m_plus) ? 1u : 0u;
        }
        packed[u / 4] |= (uint64_t)(kept[u] ? 1u : 0u) << (16 * (u % 4));
      }
      // the records of the kept '+' chains (the list of the inversion capture): requested here, used behind the scan
#pragma unroll
      for (int u = 0; u < US; ++u) {
        const uint32_t p = p0 + (uint32_t)u * NT + (uint32_t)tid;
        h_qs[u] = h_qe[u] = h_ts[u] = 0;
        if (kept[u] && p < m_plus) {
          const HeadRec* hp = A.rec + a + p;
          h_qs[u] = hp->qs;
          h_qe[u] = hp->qe;
          h_ts[u] = hp->ts;
        }
      }
      // one scan for the rows (NT <= 1024 < 2^16)
      row_scan_n(std::integral_constant<int, GS>{}, packed, ex, tot);
      uint32_t row_base = kept_before;
#pragma unroll
      for (int u = 0; u < US; ++u) {
        const uint32_t p = p0 + (uint32_t)u * NT + (uint32_t)tid;
        const uint32_t r = row_base + (uint32_t)((ex[u / 4] >> (16 * (u % 4))) & 0xffffu);
        if (p < m) A.head_num[a + p] = kept[u] ? r : (ok[u] ? NEVER : NONE);  // (every position: the members look their head's entry up)
        if (kept[u] && p < m_plus) {  // '+' chains come first: r is the chain's slot in the list
          A.f_qs[a + r] = h_qs[u];
          A.f_qe[a + r] = h_qe[u];
          A.f_ts[a + r] = h_ts[u];
          if (r < (uint32_t)KP) {
            l_qs[r] = h_qs[u];
            l_qe[r] = h_qe[u];
            l_ts[r] = h_ts[u];
          }
        }
        row_base += (uint32_t)((tot[u / 4] >> (16 * (u % 4))) & 0xffffu);
      }
      kept_before = row_base;
    }
    n_kept = kept_before;
    n_pass0 = block_sum<NT>(c0, ws);
    n_pass1 = block_sum<NT>(c1, ws);
    n_q = block_sum<NT>(cq, ws);
    n_deg = block_sum<NT>(cd, ws);
    kP = block_sum<NT>(cp, ws);
  };
  sweep(false);
  const uint32_t n_ch = n_pass0 + n_pass1;
  if (n_ch == 0) {
    if (tid == 0) A.sum[rk] = sm;
    return;
  }
  if (n_deg) {  // a chain with an empty span: the rule needs the counts
    all_q = n_ch <= 1;
    all_t = (all_q ? n_ch : n_q) <= 1;
    __syncthreads();
    sweep(true);
  }
  const uint32_t kM = n_kept - kP;
  // the reference's all_chains order inside the pair: the (query, target, strand) group that appears first in the metadata
  const bool plus_first = pi.first_mem[0] < pi.first_mem[1];
  sm.n_pass = n_ch;
  sm.n_kept = n_kept;
  sm.minmem = n_pass0 && n_pass1 ? (pi.first_mem[0] < pi.first_mem[1] ? pi.first_mem[0] : pi.first_mem[1]) : (n_pass0 ? pi.first_mem[0] : pi.first_mem[1]);
  if (tid == 0) A.sum[rk] = sm;
  if (n_kept == 0) return;
  __syncthreads();  // head_num / the chain list of the whole pair are written
  auto local_number = [&](uint32_t r, bool minus) -> uint32_t {  // 1-based, in the pair's all_chains order
    if (plus_first) return r + 1;
    return minus ? r - kP + 1 : r + kM + 1;
  };
  FT_STAMP(0);
  // ---- anchors: the members of kept chains (paf_filter.rs:517-528)
  uint32_t out = 0;
  for (uint32_t p0 = 0; p0 < m; p0 += NT * US) {
    uint32_t h[US], idx[US], r[US];
#pragma unroll
    for (int u = 0; u < US; ++u) {
      const uint32_t p = p0 + (uint32_t)u * NT + (uint32_t)tid;
      h[u] = p < m ? A.hd[a + p] : a;
      idx[u] = (p < m && !A.fin) ? A.s_idx[a + p] : 0u;
    }
#pragma unroll
    for (int u = 0; u < US; ++u) r[u] = A.head_num[h[u]];
#pragma unroll
    for (int u = 0; u < US; ++u) {
      const uint32_t p = p0 + (uint32_t)u * NT + (uint32_t)tid;
      uint32_t an = 0;
      if (p < m && r[u] < NEVER) {
        an = local_number(r[u], h[u] - a >= m_plus);
        if (!A.fin) {
          A.chain[idx[u]] = an;
          A.status[idx[u]] = SWG_ST_SCAFFOLD;
        }
        ++out;
      } else if (r[u] == NEVER) {
        an = NEVER;
      }
      if (A.fin && p < m) A.fin[a + p] = (an != 0u && an != NEVER) ? ((uint32_t)SWG_ST_SCAFFOLD << 30) | an : 0u;
      if (A.rescue_d && p < m) A.anum[a + p] = an;
    }
  }
  // the alive non-members: rescue candidates unless the inversion capture takes them
  for (uint32_t p = m + (uint32_t)tid; p < M; p += NT) {
    if (A.rescue_d) A.anum[a + p] = 0u;
    if (A.fin) A.fin[a + p] = 0u;
  }
  FT_STAMP(1);
  // ---- inversion capture (paf_filter.rs:535-597): a '-' record that is not an anchor joins the first kept '+' chain of its
  // pair whose window and diagonal it sits on
  if (!A.scaffolds_only && kP > 0 && M > m_plus) {
    const bool in_lds = kP <= (uint32_t)KP;
    // running maximum of the chains' ends
    uint64_t carry = 0;
    for (uint32_t c0b = 0; c0b < kP; c0b += NT) {
      const uint32_t c = c0b + tid;
      const uint64_t v = c < kP ? (uint64_t)(in_lds ? l_qe[c] : A.f_qe[a + c]) : 0ull;
      uint64_t tot;
      uint64_t ex = block_excl_max<NT>(v, ws64, &tot);
      ex = ex > carry ? ex : carry;
      if (c < kP) {
        const uint32_t pm = (uint32_t)(v > ex ? v : ex);
        if (in_lds) l_pm[c] = pm; else A.f_pm[a + c] = pm;
      }
      carry = tot > carry ? tot : carry;
    }
    __syncthreads();  // the running maxima, and the anchors' chain numbers
    const uint32_t* __restrict__ c_qs = in_lds ? l_qs : A.f_qs + a;
    const uint32_t* __restrict__ c_qe = in_lds ? l_qe : A.f_qe + a;
    const uint32_t* __restrict__ c_ts = in_lds ? l_ts : A.f_ts + a;
    const uint32_t* __restrict__ c_pm = in_lds ? l_pm : A.f_pm + a;
    const uint64_t gap = A.gap, max_dev = A.max_dev;
    for (uint32_t p = m_plus + tid; p < M; p += NT) {
      const uint32_t iw = A.s_idx[a + p];
      if (p >= m && (iw >> 31) == 0) continue;  // an alive non-member on the '+' strand
      const uint32_t i = iw & 0x7fffffffu;
      const uint64_t qs = A.s_qs[a + p], qe = A.s_qe[a + p], ts = A.s_ts[a + p], te = A.s_te[a + p];
      if (A.fin ? A.fin[a + p] != 0u : A.chain[i] != 0u) continue;  // already an anchor
      const uint64_t qc = (qs + qe) / 2, tc = (ts + te) / 2;
      const uint64_t lim = qe > ~0ull - gap ? ~0ull : qe + gap;  // chain.query_start.saturating_sub(gap) <= qe
      uint32_t l = 0, r = kP;  // first chain with q_start > lim
      while (l < r) {
        const uint32_t mid = l + ((r - l) >> 1);
        if ((uint64_t)c_qs[mid] <= lim) l = mid + 1; else r = mid;
      }
      uint32_t lo = 0, hi = l;  // first slot whose running maximum of ends reaches the record
      while (lo < hi) {
        const uint32_t mid = lo + ((hi - lo) >> 1);
        const uint64_t pm = c_pm[mid];
        if ((pm > ~0ull - gap ? ~0ull : pm + gap) < qs) lo = mid + 1; else hi = mid;
      }
      uint32_t best = NONE;
      for (uint32_t c = lo; c < l; ++c) {
        const uint64_t cqe = c_qe[c];
        if ((cqe > ~0ull - gap ? ~0ull : cqe + gap) < qs) continue;
        const int64_t diag = (int64_t)c_ts[c] - (int64_t)c_qs[c];
        const int64_t dev = (int64_t)tc - (int64_t)qc - diag;
        const uint64_t deviation = dev < 0 ? (uint64_t)0 - (uint64_t)dev : (uint64_t)dev;
        if (deviation <= max_dev) {
          best = c;
          break;
        }
      }
      if (best != NONE) {
        const uint32_t num = local_number(best, false);
        if (A.fin) {
          A.fin[a + p] = ((uint32_t)SWG_ST_SCAFFOLD << 30) | num;
        } else {
          A.chain[i] = num;
          A.status[i] = SWG_ST_SCAFFOLD;
        }
        if (A.rescue_d) A.anum[a + p] = num;  // an anchor from here on, whatever it was
        ++out;
      }
    }
  }
  FT_STAMP(2);
  // ---- rescue (paf_filter.rs:599-732): a record that is neither an anchor nor a member of a swept-away chain is kept when an
  // anchor of its pair lies within rescue_d of it -- |q centre difference| <= D and floor(sqrt(dq^2 + dt^2)) <= D -- and takes
  // the chain of the lowest-index such anchor (the reference iterates a HashSet: any in-range anchor; the oracle fixes the
  // same instance).  The ANCHORS are what is resident: members of kept chains and whatever the inversion capture took, in
  // batches of KA (one batch is the rule), binned in LDS by the cell of width D + 1 that their query centre falls in (cell
  // modulo NB: a counting sort -- count, scan, place).  An anchor within D of a candidate lies in the candidate's cell or a
  // neighbouring one, so every candidate of the pair -- members in any order, alive non-members behind a mapping sweep --
  // reads three bins and applies the exact test to what it finds there.  A candidate's best anchor so far survives between
  // batches in its anum word (bit 31 | record index; never equal to NEVER).
  if (!A.scaffolds_only && A.rescue_d) {
    __syncthreads();  // every anchor's number is in anum (marks, inversion capture); the chain list in l_all is done with
    const uint64_t D = A.rescue_d, max_s2 = A.max_s2;
    // cell width: the power of two from D + 1 up -- a shift, and two centres within D of each other still lie in the same cell
    // or in neighbouring ones (32: one cell holds every 32-bit coordinate)
    int cell_sh = 0;
    while (cell_sh < 32 && (1ull << cell_sh) <= D) ++cell_sh;
    // (bin starts and cursors are 16-bit: KA < 2^16; the cursors are bumped by 32-bit atomics on the word that holds two)
    constexpr uint32_t NB = (uint32_t)KP / 2u, KA = (4u * (uint32_t)KP - NB - 1u) / 3u, BT = NB / NT;
    static_assert(KA >= (uint32_t)NT * U && KA < 65536u && NB % (2 * NT) == 0, "one round of anchors fits a batch; whole words of bins per thread");
    uint32_t* const g_qc = l_all;
    uint32_t* const g_tc = l_all + KA;
    uint32_t* const g_ix = l_all + 2 * KA;
    uint16_t* const b_start = reinterpret_cast<uint16_t*>(l_all + 3 * KA);  // NB + 1
    uint32_t* const b_cur = l_all + 3 * KA + (NB + 2) / 2;                   // NB / 2 words
    auto bump = [&](uint32_t b) -> uint32_t {  // the bin's cursor before the bump
      const uint32_t sh = 16u * (b & 1u);
      return (atomicAdd(&b_cur[b >> 1], 1u << sh) >> sh) & 0xffffu;
    };
    uint32_t* const c_num = A.f_qe + a;        // (a chain list of the inversion capture: free from here on)
    auto is_anchor = [](uint32_t an) { return an != 0u && an < 0x80000000u; };
    static_assert((NB & (NB - 1u)) == 0u && NB >= 4u, "bins by the cell's low bits");
    auto cell_of = [&](uint32_t qc) -> uint32_t { return cell_sh < 32 ? qc >> cell_sh : 0u; };
    auto bin_of = [&](uint32_t qc) -> uint32_t { return cell_of(qc) & (NB - 1u); };
    uint32_t pb = 0, r0 = 0;
    while (pb < M) {  // (uniform)
      for (uint32_t b = tid; b < NB / 2; b += NT) b_cur[b] = 0u;
      __syncthreads();
      // the batch: whole rounds of positions from pb on while their anchors fit, counted per bin
      uint32_t cnt = 0, pe = pb;
      for (uint32_t p0 = pb; p0 < M; p0 += NT * U) {
        uint32_t an[U], qs[U], qe[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const uint32_t p = p0 + (uint32_t)u * NT + (uint32_t)tid;
          const uint32_t j = a + (p < M ? p : 0u);
          an[u] = p < M ? A.anum[j] : 0u;
          qs[u] = A.s_qs[j];
          qe[u] = A.s_qe[j];
        }
        uint32_t mine = 0;
#pragma unroll
        for (int u = 0; u < U; ++u) mine += is_anchor(an[u]) ? 1u : 0u;
        const uint32_t rt = block_sum<NT>(mine, ws);
        if (cnt + rt > KA) break;
#pragma unroll
        for (int u = 0; u < U; ++u)
          if (is_anchor(an[u])) (void)bump(bin_of((uint32_t)(((uint64_t)qs[u] + qe[u]) / 2)));
        cnt += rt;
        pe = p0 + NT * U < M ? p0 + NT * U : M;
      }
      __syncthreads();
      {  // bin starts: BT consecutive bins (BT / 2 words) per thread
        uint32_t c[BT], sum = 0;
#pragma unroll
        for (uint32_t k = 0; k < BT; k += 2) {
          const uint32_t w = b_cur[((uint32_t)tid * BT + k) >> 1];
          c[k] = w & 0xffffu;
          c[k + 1] = w >> 16;
          sum += c[k] + c[k + 1];
        }
        uint32_t tot;
        uint32_t run = block_excl_sum<NT>(sum, ws, &tot);
#pragma unroll
        for (uint32_t k = 0; k < BT; k += 2) {
          const uint32_t s0 = run, s1 = run + c[k];
          b_start[(uint32_t)tid * BT + k] = (uint16_t)s0;
          b_start[(uint32_t)tid * BT + k + 1] = (uint16_t)s1;
          b_cur[((uint32_t)tid * BT + k) >> 1] = s0 | (s1 << 16);
          run = s1 + c[k + 1];
        }
        if (tid == NT - 1) b_start[NB] = (uint16_t)run;
      }
      __syncthreads();
      for (uint32_t p0 = pb; p0 < pe; p0 += NT * U) {
        uint32_t an[U], qs[U], qe[U], ts[U], te[U], ix[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const uint32_t p = p0 + (uint32_t)u * NT + (uint32_t)tid;
          const uint32_t j = a + (p < pe ? p : 0u);
          an[u] = p < pe ? A.anum[j] : 0u;
          qs[u] = A.s_qs[j];
          qe[u] = A.s_qe[j];
          ts[u] = A.s_ts[j];
          te[u] = A.s_te[j];
          ix[u] = A.s_idx[j];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
          if (!is_anchor(an[u])) continue;
          const uint32_t qc = (uint32_t)(((uint64_t)qs[u] + qe[u]) / 2);
          const uint32_t e = bump(bin_of(qc));
          g_qc[e] = qc;
          g_tc[e] = (uint32_t)(((uint64_t)ts[u] + te[u]) / 2);
          g_ix[e] = ix[u] & 0x7fffffffu;
          c_num[r0 + e] = an[u];
        }
      }
      __syncthreads();
      FT_STAMP(3);
      // every candidate of the pair against the batch
      for (uint32_t p0 = 0; cnt && p0 < M; p0 += NT * U) {
        uint32_t an[U], qs[U], qe[U], ts[U], te[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const uint32_t p = p0 + (uint32_t)u * NT + (uint32_t)tid;
          const uint32_t j = a + (p < M ? p : 0u);
          an[u] = A.anum[j];
          qs[u] = A.s_qs[j];
          qe[u] = A.s_qe[j];
          ts[u] = A.s_ts[j];
          te[u] = A.s_te[j];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const uint32_t p = p0 + (uint32_t)u * NT + (uint32_t)tid;
          const uint32_t st = an[u];
          if (p >= M || !(st == 0u || (st >= 0x80000000u && st != NEVER))) continue;
          const uint32_t qc32 = (uint32_t)(((uint64_t)qs[u] + qe[u]) / 2);
          const uint64_t qc = qc32, tc = ((uint64_t)ts[u] + te[u]) / 2;
          uint32_t best_idx = st ? st & 0x7fffffffu : NONE, best_e = NONE;
          auto scan = [&](uint32_t k0, uint32_t k1) {
            for (uint32_t k = k0; k < k1; ++k) {
              const uint64_t aq = g_qc[k];
              const uint64_t q_diff = qc > aq ? qc - aq : aq - qc;
              if (q_diff > D) continue;
              const uint64_t at = g_tc[k];
              const uint64_t t_diff = tc > at ? tc - at : at - tc;
              if (q_diff * q_diff + t_diff * t_diff > max_s2) continue;  // (wrapping sum, as the reference's)
              const uint32_t ix = g_ix[k];
              if (ix < best_idx) {
                best_idx = ix;
                best_e = k;
              }
            }
          };
          // the cell's bin and its two neighbours are consecutive bins, i.e. ONE stretch of the binned anchors -- unless the bin
          // is the first or the last one, where the neighbour wraps around (a bin read for a cell that does not exist is harmless)
          const uint32_t b = bin_of(qc32);
          if (b != 0u && b != NB - 1u) {
            scan(b_start[b - 1], b_start[b + 2]);
          } else if (b == 0u) {
            scan(b_start[0], b_start[2]);
            scan(b_start[NB - 1], b_start[NB]);
          } else {
            scan(b_start[NB - 2], b_start[NB]);
            scan(b_start[0], b_start[1]);
          }
          if (best_e != NONE) {
            if (A.fin) {
              A.fin[a + p] = ((uint32_t)SWG_ST_RESCUED << 30) | c_num[r0 + best_e];
            } else {
              const uint32_t i = A.s_idx[a + p] & 0x7fffffffu;
              A.status[i] = SWG_ST_RESCUED;
              A.chain[i] = c_num[r0 + best_e];
            }
            A.anum[a + p] = 0x80000000u | best_idx;
            if (st == 0u) ++out;
          }
        }
      }
      FT_STAMP(4);
      __syncthreads();  // (the next batch overwrites the bins)
      r0 += cnt;
      pb = pe;
    }
  }
  FT_STAMP(5);
  out = block_sum<NT>(out, ws);
  if (tid == 0) A.n_out_pair[rk] = out;  // (summed by pair_totals_kernel, with the pairs' kept chains)
}

// ---- chain_N bases ----------------------------------------------------------------------------------------------------
// plane_sweep_scaffolds returns the kept chains genome pair by genome pair (first two '#' parts, first appearance among the
// chains that pass the filter), chromosome pair by chromosome pair inside, all_chains order inside that
// (plane_sweep_scaffold.rs:116-130, 204-251).  First appearance in all_chains order = the (query, target, strand) group's
// place in the metadata order: genome pair (prefix up to the last '#') by its first alive record, then the group's first
// member (paf_filter.rs:1037-1046, 1110-1120, 761-770).
__global__ __launch_bounds__(EW) void pair_key1_kernel(uint32_t n_runs, const PairInfo* __restrict__ info, const PairSum* __restrict__ sum,
                                                       PairTable gl_first, const uint32_t* __restrict__ seq_genome_last,
                                                       uint64_t* __restrict__ key, uint32_t* __restrict__ val,
                                                       const PairCounters* __restrict__ C) {
  const uint32_t k = blockIdx.x * EW + threadIdx.x;
  if (k >= n_runs) return;
  uint64_t x = ~0ull;
  if (!(C->flags & PF_FALLBACK) && sum[k].n_pass) {
    const uint32_t g = pair_get(gl_first, seq_genome_last[info[k].q], seq_genome_last[info[k].t]);
    x = ((uint64_t)g << 32) | sum[k].minmem;
  }
  key[k] = x;
  val[k] = k;
}
__global__ __launch_bounds__(EW) void pair_rank1_kernel(uint32_t n_runs, const uint32_t* __restrict__ order, const PairInfo* __restrict__ info,
                                                        const PairSum* __restrict__ sum, const uint32_t* __restrict__ seq_genome_two,
                                                        PairTable gp2_first, uint32_t* __restrict__ rank1,
                                                        const PairCounters* __restrict__ C) {
  const uint32_t r = blockIdx.x * EW + threadIdx.x;
  if (r >= n_runs) return;
  const uint32_t k = order[r];
  rank1[k] = r;
  if (!(C->flags & PF_FALLBACK) && sum[k].n_pass) atomicMin(pair_slot(gp2_first, seq_genome_two[info[k].q], seq_genome_two[info[k].t]), r);
}
__global__ __launch_bounds__(EW) void pair_key2_kernel(uint32_t n_runs, const PairInfo* __restrict__ info, const PairSum* __restrict__ sum,
                                                       const uint32_t* __restrict__ rank1, const uint32_t* __restrict__ seq_genome_two,
                                                       PairTable gp2_first, uint64_t* __restrict__ key, uint32_t* __restrict__ val,
                                                       const PairCounters* __restrict__ C) {
  const uint32_t k = blockIdx.x * EW + threadIdx.x;
  if (k >= n_runs) return;
  uint64_t x = ~0ull;
  if (!(C->flags & PF_FALLBACK) && sum[k].n_pass) x = ((uint64_t)pair_get(gp2_first, seq_genome_two[info[k].q], seq_genome_two[info[k].t]) << 32) | rank1[k];
  key[k] = x;
  val[k] = k;
}
__global__ __launch_bounds__(EW) void pair_sizes_kernel(uint32_t n_runs, const uint32_t* __restrict__ order, const PairSum* __restrict__ sum,
                                                        uint32_t* __restrict__ sizes) {
  const uint32_t r = blockIdx.x * EW + threadIdx.x;
  if (r < n_runs) sizes[r] = sum[order[r]].n_kept;
}
__global__ __launch_bounds__(EW) void pair_base_kernel(uint32_t n_runs, const uint32_t* __restrict__ order, const uint32_t* __restrict__ bases,
                                                       PairSum* __restrict__ sum) {
  const uint32_t r = blockIdx.x * EW + threadIdx.x;
  if (r < n_runs) sum[order[r]].base = bases[r];
}
// A moderate number of pairs: the same by counting, one wavefront per pair (its lanes stride over the keys; O(pairs^2)
// compares, no sort: 10^4 pairs are 10^8 compares).
__global__ __launch_bounds__(EW) void pair_rank_count_kernel(uint32_t n_runs, const uint64_t* __restrict__ key, const PairInfo* __restrict__ info,
                                                             const uint32_t* __restrict__ seq_genome_two, PairTable gp2_first,
                                                             uint32_t* __restrict__ rank1) {
  const uint32_t k = blockIdx.x * (EW / 64) + (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  if (k >= n_runs) return;
  const int lane = threadIdx.x & 63;
  const uint64_t mine = key[k];
  uint32_t r = 0;
  if (mine != ~0ull)
    for (uint32_t j = lane; j < n_runs; j += 64) r += key[j] < mine ? 1u : 0u;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) r += __shfl_xor(r, o, 64);
  if (lane) return;
  rank1[k] = r;
  if (mine != ~0ull) atomicMin(pair_slot(gp2_first, seq_genome_two[info[k].q], seq_genome_two[info[k].t]), r);
}
__global__ __launch_bounds__(EW) void pair_base_count_kernel(uint32_t n_runs, const uint64_t* __restrict__ key, PairSum* sum) {
  const uint32_t k = blockIdx.x * (EW / 64) + (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  if (k >= n_runs) return;
  const int lane = threadIdx.x & 63;
  const uint64_t mine = key[k];
  uint32_t b = 0;
  if (mine != ~0ull)
    for (uint32_t j = lane; j < n_runs; j += 64) b += key[j] < mine ? sum[j].n_kept : 0u;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) b += __shfl_xor(b, o, 64);
  if (lane == 0) sum[k].base = b;  // (a word of its own: the other wavefronts read n_kept)
}
// Few pairs (the usual case for a small input): no sort at all -- a pair's place among the keys is a count, and its base is
// the sum of the kept chains of the pairs whose key is smaller; one work-group, O(pairs^2) compares from LDS.
constexpr int NUMBER_SMALL = 512;
__global__ __launch_bounds__(512) void pair_number_small_kernel(uint32_t n_runs, const PairInfo* __restrict__ info, PairSum* __restrict__ sum,
                                                                 PairTable gl_first, const uint32_t* __restrict__ seq_genome_last,
                                                                 PairTable gp2_first, const uint32_t* __restrict__ seq_genome_two,
                                                                 const PairCounters* __restrict__ C) {
  __shared__ uint64_t key[NUMBER_SMALL];
  __shared__ uint32_t kept[NUMBER_SMALL];
  const int tid = threadIdx.x;
  if (C->flags & PF_FALLBACK) return;  // (the sums were not written)
  uint64_t k1[1];
  uint32_t r1[1];
#pragma unroll
  for (int u = 0; u < 1; ++u) {
    const uint32_t k = (uint32_t)tid + (uint32_t)u * 512u;
    uint64_t x = ~0ull;
    if (k < n_runs && sum[k].n_pass)
      x = ((uint64_t)pair_get(gl_first, seq_genome_last[info[k].q], seq_genome_last[info[k].t]) << 32) | sum[k].minmem;
    k1[u] = x;
    if (k < n_runs) {
      key[k] = x;
      kept[k] = x != ~0ull ? sum[k].n_kept : 0u;
    }
  }
  __syncthreads();
#pragma unroll
  for (int u = 0; u < 1; ++u) {
    const uint32_t k = (uint32_t)tid + (uint32_t)u * 512u;
    uint32_t r = 0;
    if (k < n_runs && k1[u] != ~0ull) {
#pragma unroll 8
      for (uint32_t j = 0; j < n_runs; ++j) r += key[j] < k1[u] ? 1u : 0u;  // (the keys of pairs with chains are distinct)
      atomicMin(pair_slot(gp2_first, seq_genome_two[info[k].q], seq_genome_two[info[k].t]), r);
    }
    r1[u] = r;
  }
  __threadfence();
  __syncthreads();
#pragma unroll
  for (int u = 0; u < 1; ++u) {
    const uint32_t k = (uint32_t)tid + (uint32_t)u * 512u;
    if (k < n_runs && k1[u] != ~0ull) {
      uint32_t* slot = pair_slot(gp2_first, seq_genome_two[info[k].q], seq_genome_two[info[k].t]);
      k1[u] = ((uint64_t)__hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) << 32) | r1[u];
    }
  }
  __syncthreads();
#pragma unroll
  for (int u = 0; u < 1; ++u) {
    const uint32_t k = (uint32_t)tid + (uint32_t)u * 512u;
    if (k < n_runs) key[k] = k1[u];
  }
  __syncthreads();
#pragma unroll
  for (int u = 0; u < 1; ++u) {
    const uint32_t k = (uint32_t)tid + (uint32_t)u * 512u;
    if (k < n_runs) {
      uint32_t b = 0;
      if (k1[u] != ~0ull)
#pragma unroll 8
        for (uint32_t j = 0; j < n_runs; ++j) b += key[j] < k1[u] ? kept[j] : 0u;
      sum[k].base = b;
    }
  }
}
// chain numbers: pair-local -> global (records of pairs without kept chains hold zeros)
__global__ __launch_bounds__(EW) void pair_renumber_kernel(uint32_t n_runs, const PairRun* __restrict__ runs, const PairSum* __restrict__ sum,
                                                           uint32_t* __restrict__ chain, const PairCounters* __restrict__ C,
                                                           const uint32_t* __restrict__ perm) {
  if (C->flags & PF_FALLBACK) return;
  for (uint32_t k = blockIdx.x; k < n_runs; k += gridDim.x) {
    const uint32_t base = sum[k].base;
    if (base == 0 || sum[k].n_kept == 0) continue;
    const uint32_t a = runs[k].a, n = runs[k].n;
    for (uint32_t l0 = 0; l0 < n; l0 += EW * 4) {
      uint32_t i[4], c[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const uint32_t li = l0 + (uint32_t)u * EW + threadIdx.x;
        i[u] = li < n ? (perm ? perm[a + li] : a + li) : NONE;
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) c[u] = i[u] != NONE ? chain[i[u]] : 0u;
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (c[u]) chain[i[u]] = c[u] + base;
    }
  }
}

// The output columns of an input grouped by pair: every record of a run takes its status and its chain number (the pair-local
// one from pair_finish + the pair's base) from its place in the pair's sorted order, brought back to input order through LDS
// -- the records of a pair stand anywhere in its run, and 10^8 scattered 1- and 4-byte stores cost a 32-byte sector each.
// Every record of every run is written (zeros for the dropped and the dead ones): the columns need no clearing beforehand.
// One launch per size class (work-groups of 64 / 256 / 1,024 threads, tiles of 1,024 / 4,096 / 16,384 records): millions of
// tiny pairs must not each occupy a 1,024-thread work-group.
template <int NT, uint32_t TILE>
__global__ __launch_bounds__(NT) void pair_out_kernel(uint32_t n_first, const uint32_t* __restrict__ first, uint32_t n_list,
                                                      const uint32_t* __restrict__ list, const PairRun* __restrict__ runs,
                                                      const PairInfo* __restrict__ info, const PairSum* __restrict__ sum,
                                                      const uint32_t* __restrict__ s_idx, const uint32_t* __restrict__ fin,
                                                      uint32_t* __restrict__ chain, uint8_t* __restrict__ status,
                                                      const PairCounters* __restrict__ C) {
  __shared__ uint32_t tile[TILE];
  if (C->flags & PF_FALLBACK) return;
  for (uint32_t kk = blockIdx.x; kk < n_first + n_list; kk += gridDim.x) {  // (`first`: the longest runs of the launch)
    const uint32_t k = kk < n_first ? first[kk] : list[kk - n_first];
    const uint32_t a = runs[k].a, n = runs[k].n;
    const uint32_t kept = sum[k].n_kept, base = sum[k].base, M = kept ? info[k].M : 0u;
    for (uint32_t t0 = 0; t0 < n; t0 += TILE) {
      const uint32_t tn = n - t0 < TILE ? n - t0 : TILE;
      if (M) {
        __syncthreads();  // (the previous tile's readers)
        for (uint32_t j = threadIdx.x; j < tn; j += NT) tile[j] = 0u;
        __syncthreads();
        for (uint32_t p0 = 0; p0 < M; p0 += NT * 4) {
          uint32_t ix[4], v[4];
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const uint32_t p = p0 + (uint32_t)u * NT + threadIdx.x;
            ix[u] = p < M ? s_idx[a + p] & 0x7fffffffu : NONE;
            v[u] = p < M ? fin[a + p] : 0u;
          }
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const uint32_t o = ix[u] - a - t0;  // (wraps for a record outside the tile)
            if (ix[u] != NONE && o < tn && v[u]) tile[o] = v[u];
          }
        }
        __syncthreads();
      }
      for (uint32_t j = threadIdx.x; j < tn; j += NT) {
        const uint32_t v = M ? tile[j] : 0u;
        const uint32_t num = v & 0x3fffffffu;
        chain[a + t0 + j] = num ? num + base : 0u;
        status[a + t0 + j] = (uint8_t)(v >> 30);
      }
    }
  }
}

// The inversion capture's floating-point test as an integer threshold: (u64)(deviation as f64 / SQRT_2) <= gap is monotone in the
// deviation, so it holds exactly up to a largest one, found by bisection with the same IEEE operations (one conversion, one
// correctly rounded division, one truncation: the host's are the device's; swg_scaffold_internal.h has the device version).
uint64_t pair_max_deviation(uint64_t gap) {
  auto ok = [&](uint64_t deviation) {
    const double pd = (double)deviation / 1.4142135623730951;
    const uint64_t perp = pd >= 18446744073709551616.0 ? ~0ull : (uint64_t)pd;
    return perp <= gap;
  };
  uint64_t lo = 0, hi = ~0ull;
  if (ok(hi)) return hi;
  while (hi - lo > 1) {
    const uint64_t mid = lo + ((hi - lo) >> 1);
    if (ok(mid)) lo = mid; else hi = mid;
  }
  return lo;
}

// ... and the rescue's: (u64)sqrt((q^2 + t^2) as f64) <= D holds exactly up to a largest sum (conversion, correctly rounded square
// root and truncation never decrease)
uint64_t pair_max_dist2(uint64_t D) {
  auto ok = [&](uint64_t s2) {
    const double dd = std::sqrt((double)s2);
    const uint64_t dist = dd >= 18446744073709551616.0 ? ~0ull : (uint64_t)dd;
    return dist <= D;
  };
  uint64_t lo = 0, hi = ~0ull;
  if (ok(hi)) return hi;
  while (hi - lo > 1) {
    const uint64_t mid = lo + ((hi - lo) >> 1);
    if (ok(mid)) lo = mid; else hi = mid;
  }
  return lo;
}

bool pair_path_wanted() {
  static const int knob = getenv("SWG_GROUP_FUSED") ? atoi(getenv("SWG_GROUP_FUSED")) : -1;
  return knob != 0;
}

// ---- large inputs that are not grouped by pair: grouped on the device (round 6) -------------------------------------------------
// The reference groups records in whatever order they come (IndexMap by (query, target, strand), src/paf_filter.rs:761-770).  The
// pair-resident stage wants a pair's records side by side; what an aligner like wfmash writes is one query after the other with
// the targets mixed.  So: key = q_id * n_seq + t_id per record, a stable radix sort of (key, index) -- inside a pair the records
// keep their order, which every tie-break of the stage relies on --, ONE gather of the ten columns into a pair-major copy, the
// stage over the copy, and the results scattered back.  What orders PAIRS by first appearance (the chain numbers) is reported
// in the caller's indices (PairSortArgs.orig).
__global__ __launch_bounds__(256) void pair_group_key_kernel(uint32_t n, const uint32_t* __restrict__ q_id, const uint32_t* __restrict__ t_id,
                                                             uint32_t n_seq, uint64_t* __restrict__ key, uint32_t* __restrict__ val) {
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  if (i >= n) return;
  key[i] = (uint64_t)q_id[i] * n_seq + t_id[i];
  val[i] = i;
}
__global__ __launch_bounds__(256) void pair_group_gather_kernel(uint32_t n, const uint32_t* __restrict__ perm, const uint32_t* __restrict__ q_id,
                                                                const uint32_t* __restrict__ t_id, const uint32_t* __restrict__ qs,
                                                                const uint32_t* __restrict__ qe, const uint32_t* __restrict__ ts,
                                                                const uint32_t* __restrict__ te, const uint32_t* __restrict__ m,
                                                                const uint32_t* __restrict__ b, const double* __restrict__ id,
                                                                const uint8_t* __restrict__ strand, uint32_t* __restrict__ o_q,
                                                                uint32_t* __restrict__ o_t, uint32_t* __restrict__ o_qs, uint32_t* __restrict__ o_qe,
                                                                uint32_t* __restrict__ o_ts, uint32_t* __restrict__ o_te, uint32_t* __restrict__ o_m,
                                                                uint32_t* __restrict__ o_b, double* __restrict__ o_id, uint8_t* __restrict__ o_st) {
  const uint32_t j = swg_xcd_block(blockIdx.x, gridDim.x) * 256u + threadIdx.x;  // (neighbouring blocks read neighbouring records: one L2)
  if (j >= n) return;
  const uint32_t i = perm[j];
  o_q[j] = q_id[i];
  o_t[j] = t_id[i];
  o_qs[j] = qs[i];
  o_qe[j] = qe[i];
  o_ts[j] = ts[i];
  o_te[j] = te[i];
  o_m[j] = m[i];
  o_b[j] = b[i];
  if (id) o_id[j] = id[i];
  o_st[j] = strand[i];
}
__global__ __launch_bounds__(256) void pair_ungroup_kernel(uint32_t n, const uint32_t* __restrict__ perm, const uint8_t* __restrict__ st,
                                                           const uint32_t* __restrict__ ch, uint8_t* __restrict__ status_out,
                                                           uint32_t* __restrict__ chain_out) {
  const uint32_t j = swg_xcd_block(blockIdx.x, gridDim.x) * 256u + threadIdx.x;
  if (j >= n) return;
  const uint32_t i = perm[j];
  status_out[i] = st[j];
  chain_out[i] = ch[j];
}

}  // namespace

int pair_group_records(swg_ctx* ctx, const swg_records* r, swg_records* copy, uint32_t** perm_out, int* ok) {
  *ok = 0;
  static const bool off = getenv("SWG_PAIR_GROUP") && atoi(getenv("SWG_PAIR_GROUP")) == 0;
  const uint64_t n64 = r->n;
  // Where it pays (measured, round 6, tools/order_shapes.py, default flags, grouped against the global-sort stage): by query in
  // query order 4*10^6 records 1.45 against 1.89 ms, 1.6*10^7 3.37 against 3.38, 10^8 29.2 against 18.6; shuffled 2.13 / 1.93,
  // 5.5 / 3.6, 39.6 / 20.4.  The gather is ten scattered 4-byte reads per record: while the columns sit in the last-level cache
  // it is cheap, beyond it every read moves a sector (13.5 ms of the 29.2).  So: up to 2^23 records (SWG_PAIR_GROUP=2: any size).
  static const bool any_size = getenv("SWG_PAIR_GROUP") && atoi(getenv("SWG_PAIR_GROUP")) == 2;
  if (off || !pair_path_wanted() || n64 < 2 || n64 >= (uint64_t(1) << 31) || (n64 > (uint64_t(1) << 23) && !any_size)) return SWG_OK;
  const int key_bits = swg_bits_for((uint64_t)r->n_seq * r->n_seq - 1) ? swg_bits_for((uint64_t)r->n_seq * r->n_seq - 1) : 1;
  if (key_bits > 24) return SWG_OK;  // (three 12-byte passes at most: beyond that the global-sort stage is the cheaper way)
  const uint32_t n = (uint32_t)n64;
  hipStream_t st = ctx->stream;
  uint64_t* key = swg_alloc<uint64_t>(ctx, n);
  uint64_t* key2 = swg_alloc<uint64_t>(ctx, n);
  uint32_t* val = swg_alloc<uint32_t>(ctx, n);
  uint32_t* val2 = swg_alloc<uint32_t>(ctx, n);
  uint32_t* c4[8];
  for (auto& p : c4) p = swg_alloc<uint32_t>(ctx, n);
  double* c_id = r->identity ? swg_alloc<double>(ctx, n) : nullptr;
  uint8_t* c_st = swg_alloc<uint8_t>(ctx, n);
  SWG_CHECK_ARENA(ctx);
  SWG_LAUNCH(ctx, "pair_group_key", pair_group_key_kernel<<<(n + 255) / 256, 256, 0, st>>>(n, r->q_id, r->t_id, r->n_seq, key, val));
  SWG_KERNEL_CHECK(ctx);
  SWG_TRY(swg_radix_sort_pairs(ctx, &key, &val, &key2, &val2, n, 0, key_bits));
  SWG_LAUNCH(ctx, "pair_group_gather", pair_group_gather_kernel<<<(n + 255) / 256, 256, 0, st>>>(n, val, r->q_id, r->t_id, r->q_start, r->q_end, r->t_start, r->t_end,
                                                                                       r->matches, r->block_len, r->identity, r->strand, c4[0], c4[1], c4[2],
                                                                                       c4[3], c4[4], c4[5], c4[6], c4[7], c_id, c_st));
  SWG_KERNEL_CHECK(ctx);
  *copy = *r;
  copy->q_id = c4[0]; copy->t_id = c4[1]; copy->q_start = c4[2]; copy->q_end = c4[3]; copy->t_start = c4[4]; copy->t_end = c4[5];
  copy->matches = c4[6]; copy->block_len = c4[7]; copy->identity = c_id; copy->strand = c_st;
  *perm_out = val;
  *ok = 1;
  return SWG_OK;
}
int pair_ungroup_results(swg_ctx* ctx, uint64_t n, const uint32_t* perm, const uint8_t* st, const uint32_t* ch, uint8_t* status_out, uint32_t* chain_out) {
  SWG_LAUNCH(ctx, "pair_ungroup", pair_ungroup_kernel<<<(unsigned)((n + 255) / 256), 256, 0, ctx->stream>>>((uint32_t)n, perm, st, ch, status_out, chain_out));
  SWG_KERNEL_CHECK(ctx);
  return SWG_OK;
}

namespace {

}  // namespace

// The pairs of the input: runs of equal (q_id, t_id) (large inputs, grouped by pair as an aligner writes them), or through a
// hash table (small inputs, grouped or not).  valid = 0: the pair-resident stage does not apply.
int pair_plan(swg_ctx* ctx, const swg_records* r, const swg_config* cfg, PairPlan* plan) {
  *plan = PairPlan{};
  if (!pair_path_wanted()) return SWG_OK;
  const uint64_t n64 = r->n;
  if (n64 < 2 || n64 >= (uint64_t(1) << 31)) return SWG_OK;
  const uint32_t n = (uint32_t)n64;
  hipStream_t st = ctx->stream;
  static const bool dbg = getenv("SWG_DEBUG") != nullptr;
  const swg_arena_mark mark0 = swg_arena_save(ctx);
  const bool by_hash = n <= PAIR_HASH_MAX;
  // How many pairs the path takes.  Its cost per pair -- a work-group per kernel -- is nothing next to a pair of thousands of
  // records and everything next to a pair of ten.  Measured on 10^8 records of 100 genomes x C chromosomes (round 6,
  // tools/order_shapes.py <n> <flags> <C,...>, pair path against the global-sort stage).  CLI defaults: C = 20 (198,000 pairs of
  // 505 records) 8.5 against 17.7 ms, C = 40 (252) 10.7 / 17.2, C = 60 (168) 13.1 / 18.4, C = 100 (101) 17.7 / 19.4, C = 200 (50)
  // 29.4 / 22.1.  With a scaffold filter that has limits, a rescue or a mapping sweep in front (their per-pair kernels: the
  // segment sorts, pair_chains, the rescue's bins): C = 60 24.5 / 27.1, C = 100 31.4 / 27.5 (c5), 47.2 / 39.4 (full).  So: pairs of
  // 96 records or more on average under the plain flags, 192 otherwise, or any 8,192 pairs.  (Round 5 drew the line at 1,536: its
  // pair kernels bumped five statistics counters of ONE cache line per pair -- now summed afterwards, pair_totals_kernel -- and
  // took a returning atomic per pair for the chunk list -- now slots that follow from the pair's place, emit_chunks.)
  // SWG_PAIR_MIN_AVG overrides the average (experiments).
  bool plain = cfg != nullptr;
  if (cfg) {
    const bool sf_limited = cfg->scaffold_filter_mode == SWG_MODE_ONE_TO_ONE || cfg->scaffold_filter_mode == SWG_MODE_ONE_TO_MANY ||
                            cfg->scaffold_max_per_query != 0 || cfg->scaffold_max_per_target != 0;
    const bool mf_limited = cfg->mapping_filter_mode == SWG_MODE_ONE_TO_ONE || cfg->mapping_filter_mode == SWG_MODE_ONE_TO_MANY ||
                            cfg->mapping_max_per_query != 0 || cfg->mapping_max_per_target != 0;
    plain = !sf_limited && !mf_limited && (cfg->scaffolds_only || cfg->scaffold_max_deviation == 0);
  }
  static const uint32_t per_pair_env = getenv("SWG_PAIR_MIN_AVG") && atoi(getenv("SWG_PAIR_MIN_AVG")) > 0 ? (uint32_t)atoi(getenv("SWG_PAIR_MIN_AVG")) : 0u;
  const uint32_t per_pair = per_pair_env ? per_pair_env : (plain ? 96u : 192u);
  const uint32_t cap = by_hash ? (n < 8192u ? n : 8192u) : (n / per_pair > 8192u ? n / per_pair : 8192u);
  uint32_t tsize = 1;
  while (tsize < 2 * (by_hash ? n : cap)) tsize <<= 1;  // (the hash grouping enters every record's pair: room for n of them)
  PairCounters* C = nullptr;
  PairRun* runs = swg_alloc<PairRun>(ctx, cap);
  uint32_t* class_list = swg_alloc<uint32_t>(ctx, (size_t)4 * cap);
  uint32_t* perm = nullptr;
  if (by_hash) {
    // one zeroed block: the table's keys, the counts per slot, the counters
    char* z = static_cast<char*>(swg_arena_alloc(ctx, (size_t)tsize * 12 + sizeof(PairCounters) + 8));
    uint32_t* slot_of = swg_alloc<uint32_t>(ctx, n);
    uint32_t* start = swg_alloc<uint32_t>(ctx, tsize);
    perm = swg_alloc<uint32_t>(ctx, n);
    SWG_CHECK_ARENA(ctx);
    unsigned long long* table = reinterpret_cast<unsigned long long*>(z);
    uint32_t* count = reinterpret_cast<uint32_t*>(z + (size_t)tsize * 8);
    C = reinterpret_cast<PairCounters*>(z + (size_t)tsize * 12);
    uint32_t* cursor = reinterpret_cast<uint32_t*>(z + (size_t)tsize * 12 + sizeof(PairCounters));
    SWG_HIP(ctx, hipMemsetAsync(z, 0, (size_t)tsize * 12 + sizeof(PairCounters) + 8, st));
    SWG_LAUNCH(ctx, "pair_hash", pair_hash_kernel<<<(n + 255) / 256, 256, 0, st>>>(n, r->q_id, r->t_id, table, tsize - 1, count, slot_of));
    SWG_LAUNCH(ctx, "pair_slots", pair_slots_kernel<<<(tsize + 255) / 256, 256, 0, st>>>(tsize, count, start, runs, class_list, cap, C, cursor));
    SWG_LAUNCH(ctx, "pair_perm", pair_perm_kernel<<<(n + 255) / 256, 256, 0, st>>>(n, slot_of, start, perm));
    SWG_KERNEL_CHECK(ctx);
  } else {
    C = swg_alloc<PairCounters>(ctx, 1);
    const uint32_t n_words = (n + 63) / 64;
    unsigned long long* bitmap = swg_alloc<unsigned long long>(ctx, n_words + 1);
    uint32_t* wcnt = swg_alloc<uint32_t>(ctx, n_words + 1);
    uint32_t* wpre = swg_alloc<uint32_t>(ctx, n_words + 1);
    uint32_t* starts = swg_alloc<uint32_t>(ctx, (size_t)cap + 2);
    unsigned long long* table = swg_alloc<unsigned long long>(ctx, tsize);
    uint64_t* d_nr = swg_alloc<uint64_t>(ctx, 1);
    SWG_CHECK_ARENA(ctx);
    SWG_HIP(ctx, hipMemsetAsync(C, 0, sizeof(PairCounters), st));
    SWG_HIP(ctx, hipMemsetAsync(table, 0xff, (size_t)tsize * 8, st));
    SWG_LAUNCH(ctx, "pair_boundary", pair_boundary_kernel<<<(n + 1023) / 1024, 256, 0, st>>>(n, r->q_id, r->t_id, bitmap, wcnt));
    SWG_KERNEL_CHECK(ctx);
    SWG_TRY(swg_exclusive_scan_u32(ctx, wcnt, wpre, n_words, d_nr));
    SWG_LAUNCH(ctx, "pair_runs", pair_starts_kernel<<<(n_words + 255) / 256, 256, 0, st>>>(n, n_words, bitmap, wpre, d_nr, cap, starts));
    SWG_KERNEL_CHECK(ctx);
    SWG_LAUNCH(ctx, "pair_runs", pair_runs_kernel<<<(cap + 255) / 256, 256, 0, st>>>(n, cap, d_nr, starts, r->q_id, r->t_id, table, tsize - 1, runs, class_list, C));
    SWG_KERNEL_CHECK(ctx);
  }
  uint64_t h[4];
  static_assert(sizeof(PairCounters) >= 32, "the first four words are read back");
  SWG_TRY(swg_read_scalars(ctx, reinterpret_cast<const uint64_t*>(C), h, 3));
  const uint32_t n_runs = (uint32_t)h[0], flags = (uint32_t)(h[0] >> 32);
  if (flags || n_runs == 0) {
    if (dbg) fprintf(stderr, "[swg] pair path: not applicable (%u runs, flags %u)\n", n_runs, flags);
    plan->not_grouped = !by_hash && (flags & (PF_NOT_GROUPED | PF_RUN_OVERFLOW)) && !(flags & PF_TOO_LONG);
    swg_arena_restore(ctx, mark0);
    return SWG_OK;
  }
  // a few very long pairs (the average beyond the third size class): one work-group per pair would leave most of the chip idle,
  // and pairs of that size are deep more often than not -- the global-sort stage's case
  if ((uint64_t)n / n_runs > PAIR_L_MAX) {
    if (dbg) fprintf(stderr, "[swg] pair path: %u pairs of %llu records on average: left to the global-sort stage\n", n_runs, (unsigned long long)n / n_runs);
    swg_arena_restore(ctx, mark0);
    return SWG_OK;
  }
  plan->valid = 1;
  plan->by_hash = by_hash;
  plan->n_runs = n_runs;
  plan->ncls[0] = (uint32_t)h[1];
  plan->ncls[1] = (uint32_t)(h[1] >> 32);
  plan->ncls[2] = (uint32_t)h[2];
  plan->ncls[3] = (uint32_t)(h[2] >> 32);
  plan->cap = cap;
  plan->counters = C;
  plan->runs = runs;
  plan->class_list = class_list;
  plan->perm = perm;
  if (dbg)
    fprintf(stderr, "[swg] pair path: %u pairs (%u / %u / %u / %u by size class)%s\n", n_runs, plan->ncls[0], plan->ncls[1], plan->ncls[2],
            plan->ncls[3], by_hash ? ", found through the hash table" : "");
  return SWG_OK;
}

// The scaffold stage for records grouped by chromosome pair.  *taken = 0: not applicable (not grouped, a pair too long, or a
// condition found on the device) -- nothing the caller cannot overwrite was done, and it runs the global-sort path.
// alive / member: nullptr = step-1 retain evaluated here / members == alive records.
int scaffold_stage_pairs(swg_ctx* ctx, const swg_records* r, const swg_config* cfg, const uint8_t* alive, const uint8_t* member,
                         bool sweep_assumed_identity, uint8_t* status_out, uint32_t* chain_out, swg_stats* stats, int* taken,
                         const PairPlan* plan_in) {
  *taken = 0;
  PairPlan own;
  if (!plan_in) {
    SWG_TRY(pair_plan(ctx, r, cfg, &own));
    plan_in = &own;
  }
  if (!plan_in->valid) return SWG_OK;
  uint64_t kq, kt;
  if (cfg->scaffold_filter_mode == SWG_MODE_ONE_TO_ONE) {
    kq = kt = 1;
  } else {
    kq = cfg->scaffold_max_per_query ? cfg->scaffold_max_per_query : SWG_K_INF;
    kt = cfg->scaffold_max_per_target ? cfg->scaffold_max_per_target : SWG_K_INF;
  }
  const bool limited = kq != SWG_K_INF || kt != SWG_K_INF;  // the scaffold sweep has limits: it runs over a chain table
  const bool rescue = !cfg->scaffolds_only && cfg->scaffold_max_deviation != 0;
  // a chain's weighted identity (sums of matches and block lengths over its members) is read by the identity floor and by the
  // scores of a scaffold sweep with limits; without either the two columns are neither sorted nor summed (an identity is
  // never negative or NaN, paf_filter.rs:896-913: a floor of zero or less passes every chain)
  const bool need_wid = limited || !(cfg->min_scaffold_identity <= 0.0);
  const uint32_t n = (uint32_t)r->n;
  hipStream_t st = ctx->stream;
  static const bool dbg = getenv("SWG_DEBUG") != nullptr;
  const swg_arena_mark mark0 = swg_arena_save(ctx);
  const bool by_hash = plan_in->by_hash;
  const uint32_t cap = plan_in->cap, n_runs = plan_in->n_runs;
  const uint32_t ncls[4] = {plan_in->ncls[0], plan_in->ncls[1], plan_in->ncls[2], plan_in->ncls[3]};
  PairCounters* C = static_cast<PairCounters*>(plan_in->counters);
  PairRun* runs = static_cast<PairRun*>(plan_in->runs);
  uint32_t* class_list = static_cast<uint32_t*>(plan_in->class_list);
  uint32_t* perm = static_cast<uint32_t*>(plan_in->perm);
  // (a plan may be used twice -- an unlimited mapping sweep taken as the identity, then the real one: everything but the
  // pairs' own counts starts from zero)
  SWG_HIP(ctx, hipMemsetAsync(reinterpret_cast<char*>(C) + offsetof(PairCounters, flags), 0, sizeof(uint32_t), st));
  SWG_HIP(ctx, hipMemsetAsync(reinterpret_cast<char*>(C) + offsetof(PairCounters, n_chunks), 0, sizeof(PairCounters) - offsetof(PairCounters, n_chunks), st));
  // ---- scratch, addressed by the pair's offset in the input
  uint8_t* code = swg_alloc<uint8_t>(ctx, n);
  uint32_t* s_qs = swg_alloc<uint32_t>(ctx, n);
  uint32_t* s_qe = swg_alloc<uint32_t>(ctx, n);
  uint32_t* s_ts = swg_alloc<uint32_t>(ctx, n);
  uint32_t* s_te = swg_alloc<uint32_t>(ctx, n);
  uint32_t* s_m = swg_alloc<uint32_t>(ctx, n);
  uint32_t* s_b = swg_alloc<uint32_t>(ctx, n);
  uint32_t* s_idx = swg_alloc<uint32_t>(ctx, n);
  uint32_t* pred = swg_alloc<uint32_t>(ctx, n);
  uint32_t* hd = swg_alloc<uint32_t>(ctx, n);
  uint8_t* ok_head = swg_alloc<uint8_t>(ctx, n);
  unsigned long long* bps = swg_alloc<unsigned long long>(ctx, n);
  HeadRec* head_rec = swg_alloc<HeadRec>(ctx, n);
  uint32_t* anum = rescue ? swg_alloc<uint32_t>(ctx, n) : nullptr;
  PairInfo* info = swg_alloc<PairInfo>(ctx, n_runs);
  PairSum* sum = swg_alloc<PairSum>(ctx, n_runs);
  uint32_t* n_out_pair = swg_alloc<uint32_t>(ctx, n_runs);
  const uint32_t cap_chunks = n / PAIR_CELL + 2 * n_runs + 16;  // (emit_chunks: a pair's stretch of the list follows from its place)
  SpecBlock* chunks = swg_alloc<SpecBlock>(ctx, cap_chunks);
  const uint32_t cap_long = n / LABEL_CAP_ELEMS + 1;
  uint32_t* long_list = swg_alloc<uint32_t>(ctx, cap_long);
  SWG_CHECK_ARENA(ctx);
  PairTable gl_first, gp2_first;
  SWG_TRY(pair_table_make(ctx, r->n_genome_last, n_runs, &gl_first));
  SWG_TRY(pair_table_make(ctx, r->n_genome_two, n_runs, &gp2_first));
  // (large inputs: pair_out_kernel writes every record of both columns; small ones are written in place, record by record)
  uint32_t* fin = by_hash ? nullptr : swg_alloc<uint32_t>(ctx, n);
  SWG_CHECK_ARENA(ctx);
  if (by_hash) {
    SWG_HIP(ctx, hipMemsetAsync(chain_out, 0, (size_t)n * sizeof(uint32_t), st));
    SWG_HIP(ctx, hipMemsetAsync(status_out, 0, n, st));
  }
  SWG_HIP(ctx, hipMemsetAsync(n_out_pair, 0, (size_t)n_runs * sizeof(uint32_t), st));
  if (!by_hash) {  // (emit_chunks: fixed slots, the list's length is its capacity)
    SWG_HIP(ctx, hipMemsetAsync(chunks, 0, (size_t)cap_chunks * sizeof(SpecBlock), st));
    SWG_HIP(ctx, hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(&C->n_chunks), (int)cap_chunks, 1, st));
  }
  PairSortArgs SA{};
  SA.q_id = r->q_id; SA.t_id = r->t_id; SA.q_start = r->q_start; SA.q_end = r->q_end; SA.t_start = r->t_start; SA.t_end = r->t_end;
  SA.matches = r->matches; SA.block_len = r->block_len; SA.identity = r->identity; SA.strand = r->strand;
  SA.alive_in = alive; SA.member_in = member;
  SA.min_block = cfg->min_block_length; SA.keep_self = cfg->keep_self; SA.min_identity = cfg->min_identity;
  SA.check_degenerate = sweep_assumed_identity ? 1 : 0;
  SA.max_gap = cfg->scaffold_gap;
  SA.runs = runs;
  SA.perm = perm;
  SA.orig = plan_in->orig;
  SA.code = code; SA.s_qs = s_qs; SA.s_qe = s_qe; SA.s_ts = s_ts; SA.s_te = s_te; SA.s_m = need_wid ? s_m : nullptr; SA.s_b = need_wid ? s_b : nullptr; SA.s_idx = s_idx; SA.pred = pred;
  SA.info = info; SA.chunks = chunks; SA.cap_chunks = cap_chunks; SA.chunks_by_place = by_hash ? 0 : 1; SA.long_list = long_list; SA.cap_long = cap_long; SA.C = C; SA.gl_first = gl_first; SA.seq_genome_last = r->seq_genome_last;
  if (by_hash) {
    for (int c = 2; c >= 0; --c) {
      if (!ncls[c]) continue;
      SA.list = class_list + (size_t)c * cap;
      if (c == 0)
        SWG_LAUNCH_N(ctx, "pair_sort_ps", 0, pair_sort_kernel<64, 16, 16, 256, 64, true><<<ncls[c], 64, 0, st>>>(SA));
      else if (c == 1)
        SWG_LAUNCH_N(ctx, "pair_sort_pm", 0, pair_sort_kernel<256, 16, 16, 1024, 64, true><<<ncls[c], 256, 0, st>>>(SA));
      else
        SWG_LAUNCH_N(ctx, "pair_sort_pl", 0, pair_sort_kernel<1024, 8, 32, 2048, 1024, true><<<ncls[c], 1024, 0, st>>>(SA));
      SWG_KERNEL_CHECK(ctx);
    }
  } else {
    if (ncls[2] + ncls[3]) {  // the longest pairs first: their chunks open the list the walk's work-groups draw from
      SA.list = class_list + (size_t)2 * cap;
      SWG_LAUNCH_N(ctx, "pair_sort_big", 0, pair_sort_big_kernel<<<ncls[2] + ncls[3], 1024, 0, st>>>(SA, class_list + (size_t)3 * cap, ncls[3]));
      SWG_KERNEL_CHECK(ctx);
    }
    for (int c = 1; c >= 0; --c) {
      if (!ncls[c]) continue;
      SA.list = class_list + (size_t)c * cap;
      if (c == 0)
        SWG_LAUNCH_N(ctx, "pair_sort_s", 0, pair_sort_kernel<64, 16, 16, 256, 64, false><<<ncls[c], 64, 0, st>>>(SA));
      else
        SWG_LAUNCH_N(ctx, "pair_sort_m", 0, pair_sort_kernel<256, 16, 16, 1024, 64, false><<<ncls[c], 256, 0, st>>>(SA));
      SWG_KERNEL_CHECK(ctx);
    }
  }
  if (!by_hash) {  // (large inputs only: what a small one walks for nothing is not worth a launch)
    SWG_LAUNCH(ctx, "pair_gate", pair_gate_kernel<<<1, 64, 0, st>>>(C));
    SWG_KERNEL_CHECK(ctx);
  }
  SWG_TRY(pair_walk_launch(ctx, cap_chunks, &C->n_chunks, chunks, s_qs, s_qe, s_ts, s_te, cfg->scaffold_gap, bps, pred));
  const bool long_possible = ncls[2] + ncls[3] > 0;  // (a chunk of LABEL_CAP_ELEMS members needs a pair of at least as many)
  if (long_possible) {
    SWG_TRY(pair_walk_long_launch(ctx, cap_long, &C->n_long, long_list, chunks, n, s_qs, s_qe, s_ts, s_te, cfg->scaffold_gap, bps, pred,
                                  &C->flags, PF_FALLBACK));
    if (!by_hash) {
      SWG_LAUNCH(ctx, "pair_gate", pair_gate_kernel<<<1, 64, 0, st>>>(C));
      SWG_KERNEL_CHECK(ctx);
    }
  }
  SWG_TRY(pair_label_launch(ctx, cap_chunks, &C->n_chunks, chunks, pred, s_qs, s_qe, s_ts, s_te, need_wid ? s_m : nullptr, need_wid ? s_b : nullptr, cfg->min_scaffold_length,
                            cfg->min_scaffold_identity, hd, ok_head, head_rec, &C->n_heads, long_possible ? cap_long : 0u, &C->n_long, long_list));
  // ---- a scaffold filter with limits: plane_sweep_both over the chain table of the whole input
  uint8_t* kept_flags = nullptr;
  uint32_t *chain_base = nullptr, *np_arr = nullptr;
  if (limited) {
    PairChainArgs CA{};
    CA.runs = runs; CA.info = info; CA.ok_head = ok_head; CA.rec = head_rec; CA.C = C;
    CA.T_qs = swg_alloc<uint32_t>(ctx, n);
    CA.T_qe = swg_alloc<uint32_t>(ctx, n);
    CA.T_ts = swg_alloc<uint32_t>(ctx, n);
    CA.T_te = swg_alloc<uint32_t>(ctx, n);
    CA.T_wid = swg_alloc<double>(ctx, n);
    CA.T_seg = swg_alloc<uint64_t>(ctx, n);
    CA.chain_base = chain_base = swg_alloc<uint32_t>(ctx, n_runs);
    CA.np = np_arr = swg_alloc<uint32_t>(ctx, (size_t)2 * n_runs);
    CA.totals = swg_alloc<unsigned long long>(ctx, 3);
    CA.T_runs = swg_alloc<PairRun>(ctx, n_runs);
    CA.cnt = swg_alloc<uint32_t>(ctx, n_runs);
    CA.has = swg_alloc<uint32_t>(ctx, n_runs);
    CA.stretch = swg_alloc<uint32_t>(ctx, n_runs);
    kept_flags = swg_alloc<uint8_t>(ctx, n);
    SWG_CHECK_ARENA(ctx);
    SWG_HIP(ctx, hipMemsetAsync(CA.totals, 0, 24, st));
    SWG_HIP(ctx, hipMemsetAsync(CA.cnt, 0, (size_t)n_runs * sizeof(uint32_t), st));   // (a call given up on the device leaves them unwritten)
    SWG_HIP(ctx, hipMemsetAsync(CA.has, 0, (size_t)n_runs * sizeof(uint32_t), st));
    for (int phase = 0; phase < 2; ++phase) {
      for (int c = 0; c < 4; ++c) {
        if (!ncls[c]) continue;
        CA.list = class_list + (size_t)c * cap;
        if (phase == 0) {
          switch (c) {
            case 0: SWG_LAUNCH(ctx, "pair_chains_s", pair_chains_kernel<64, 0><<<ncls[c], 64, 0, st>>>(CA)); break;
            case 1: SWG_LAUNCH(ctx, "pair_chains_m", pair_chains_kernel<256, 0><<<ncls[c], 256, 0, st>>>(CA)); break;
            default: SWG_LAUNCH(ctx, "pair_chains", pair_chains_kernel<512, 0><<<ncls[c], 512, 0, st>>>(CA)); break;
          }
        } else {
          switch (c) {
            case 0: SWG_LAUNCH(ctx, "pair_chains_s", pair_chains_kernel<64, 1><<<ncls[c], 64, 0, st>>>(CA)); break;
            case 1: SWG_LAUNCH(ctx, "pair_chains_m", pair_chains_kernel<256, 1><<<ncls[c], 256, 0, st>>>(CA)); break;
            default: SWG_LAUNCH(ctx, "pair_chains", pair_chains_kernel<512, 1><<<ncls[c], 512, 0, st>>>(CA)); break;
          }
        }
        SWG_KERNEL_CHECK(ctx);
      }
      if (phase == 0) {  // where every pair's stretch of the table begins, and which stretch it is
        SWG_TRY(swg_exclusive_scan_u32(ctx, CA.cnt, chain_base, n_runs, reinterpret_cast<uint64_t*>(CA.totals)));
        SWG_TRY(swg_exclusive_scan_u32(ctx, CA.has, CA.stretch, n_runs, reinterpret_cast<uint64_t*>(CA.totals + 2)));
      }
    }
    uint64_t ht[3];
    SWG_TRY(swg_read_scalars(ctx, reinterpret_cast<const uint64_t*>(CA.totals), ht, 3));
    const uint64_t n_table = ht[0];
    if (n_table) {
      const int seg_bits = swg_bits_for(n_runs) ? swg_bits_for(n_runs) : 1, pos_bits = swg_bits_for(ht[1]) ? swg_bits_for(ht[1]) : 1;
      SWG_TRY(scaffold_sweep_segments(ctx, n_table, CA.T_seg, seg_bits, CA.T_qs, CA.T_qe, CA.T_ts, CA.T_te, CA.T_wid, kq, kt,
                                      cfg->scaffold_overlap_threshold, cfg->scoring_function, pos_bits, kept_flags, CA.T_runs, (uint32_t)ht[2]));
    }
  }
  PairFinishArgs FA{};
  FA.kept_in = kept_flags; FA.chain_base = chain_base; FA.np = np_arr;
  FA.runs = runs; FA.info = info; FA.sum = sum;
  FA.s_qs = s_qs; FA.s_qe = s_qe; FA.s_ts = s_ts; FA.s_te = s_te; FA.s_idx = s_idx; FA.hd = hd; FA.ok_head = ok_head; FA.rec = head_rec;
  FA.head_num = pred;
  FA.f_qs = s_m; FA.f_qe = s_b; FA.f_ts = reinterpret_cast<uint32_t*>(bps); FA.f_pm = reinterpret_cast<uint32_t*>(bps) + n;
  FA.status = status_out; FA.chain = chain_out; FA.scaffolds_only = cfg->scaffolds_only; FA.gap = cfg->scaffold_gap; FA.max_dev = pair_max_deviation(cfg->scaffold_gap); FA.C = C;
  FA.rescue_d = rescue ? cfg->scaffold_max_deviation : 0;
  FA.max_s2 = rescue ? pair_max_dist2(cfg->scaffold_max_deviation) : 0;
  FA.anum = anum;
  FA.n_out_pair = n_out_pair;
  FA.fin = fin;
  for (int c = 0; c < 4; ++c) {
    if (!ncls[c]) continue;
    FA.list = class_list + (size_t)c * cap;
    switch (c) {
      case 0: SWG_LAUNCH(ctx, "pair_finish_s", pair_finish_kernel<64, 256><<<ncls[c], 64, 0, st>>>(FA)); break;
      case 1: SWG_LAUNCH(ctx, "pair_finish_m", pair_finish_kernel<256, 1024><<<ncls[c], 256, 0, st>>>(FA)); break;
      default: SWG_LAUNCH(ctx, "pair_finish", pair_finish_kernel<512, 4096><<<ncls[c], 512, 0, st>>>(FA)); break;
    }
    SWG_KERNEL_CHECK(ctx);
  }
  // ---- chain_N bases
  if (n_runs <= (uint32_t)NUMBER_SMALL) {
    SWG_LAUNCH(ctx, "pair_number", pair_number_small_kernel<<<1, 512, 0, st>>>(n_runs, info, sum, gl_first, r->seq_genome_last, gp2_first,
                                                                     r->seq_genome_two, C));
    SWG_KERNEL_CHECK(ctx);
  } else if (n_runs <= 12288u) {  // (a wavefront per key counts the smaller keys: quadratic -- 0.19 ms at 9,900 pairs, 1.45 at 32,000)
    uint64_t* key = swg_alloc<uint64_t>(ctx, n_runs);
    uint32_t* val = swg_alloc<uint32_t>(ctx, n_runs);
    uint32_t* rank1 = swg_alloc<uint32_t>(ctx, n_runs);
    SWG_CHECK_ARENA(ctx);
    const unsigned rb = nblk(n_runs), wb = (n_runs + EW / 64 - 1) / (EW / 64);
    SWG_LAUNCH(ctx, "pair_number", pair_key1_kernel<<<rb, EW, 0, st>>>(n_runs, info, sum, gl_first, r->seq_genome_last, key, val, C));
    SWG_LAUNCH(ctx, "pair_number", pair_rank_count_kernel<<<wb, EW, 0, st>>>(n_runs, key, info, r->seq_genome_two, gp2_first, rank1));
    SWG_LAUNCH(ctx, "pair_number", pair_key2_kernel<<<rb, EW, 0, st>>>(n_runs, info, sum, rank1, r->seq_genome_two, gp2_first, key, val, C));
    SWG_LAUNCH(ctx, "pair_number", pair_base_count_kernel<<<wb, EW, 0, st>>>(n_runs, key, sum));
    SWG_KERNEL_CHECK(ctx);
  } else {
    uint64_t* key = swg_alloc<uint64_t>(ctx, n_runs);
    uint64_t* key_tmp = swg_alloc<uint64_t>(ctx, n_runs);
    uint32_t* val = swg_alloc<uint32_t>(ctx, n_runs);
    uint32_t* val_tmp = swg_alloc<uint32_t>(ctx, n_runs);
    uint32_t* rank1 = swg_alloc<uint32_t>(ctx, n_runs);
    uint32_t* sizes = swg_alloc<uint32_t>(ctx, n_runs);
    SWG_CHECK_ARENA(ctx);
    const unsigned rb = nblk(n_runs);
    SWG_LAUNCH(ctx, "pair_number", pair_key1_kernel<<<rb, EW, 0, st>>>(n_runs, info, sum, gl_first, r->seq_genome_last, key, val, C));
    SWG_KERNEL_CHECK(ctx);
    SWG_TRY(swg_radix_sort_pairs(ctx, &key, &val, &key_tmp, &val_tmp, n_runs, 0, 64));
    SWG_LAUNCH(ctx, "pair_number", pair_rank1_kernel<<<rb, EW, 0, st>>>(n_runs, val, info, sum, r->seq_genome_two, gp2_first, rank1, C));
    SWG_KERNEL_CHECK(ctx);
    SWG_LAUNCH(ctx, "pair_number", pair_key2_kernel<<<rb, EW, 0, st>>>(n_runs, info, sum, rank1, r->seq_genome_two, gp2_first, key, val, C));
    SWG_KERNEL_CHECK(ctx);
    SWG_TRY(swg_radix_sort_pairs(ctx, &key, &val, &key_tmp, &val_tmp, n_runs, 0, 64));
    SWG_LAUNCH(ctx, "pair_number", pair_sizes_kernel<<<rb, EW, 0, st>>>(n_runs, val, sum, sizes));
    SWG_KERNEL_CHECK(ctx);
    SWG_TRY(swg_exclusive_scan_u32(ctx, sizes, sizes, n_runs, nullptr));
    SWG_LAUNCH(ctx, "pair_number", pair_base_kernel<<<rb, EW, 0, st>>>(n_runs, val, sizes, sum));
    SWG_KERNEL_CHECK(ctx);
  }
  {
    const unsigned gb = n_runs < (unsigned)ctx->num_cu * 16 ? n_runs : (unsigned)ctx->num_cu * 16;
    if (fin) {  // (one label for all: the chain numbers' last step)
      const unsigned cu = (unsigned)ctx->num_cu;
      if (ncls[2] + ncls[3]) {
        const uint32_t nb = ncls[2] + ncls[3];
        SWG_LAUNCH(ctx, "pair_renumber", pair_out_kernel<1024, 16384><<<nb < cu * 8 ? nb : cu * 8, 1024, 0, st>>>(
                                             ncls[3], class_list + (size_t)3 * cap, ncls[2], class_list + (size_t)2 * cap, runs, info, sum, s_idx, fin,
                                             chain_out, status_out, C));
        SWG_KERNEL_CHECK(ctx);
      }
      if (ncls[1]) {
        SWG_LAUNCH(ctx, "pair_renumber", pair_out_kernel<256, 4096><<<ncls[1] < cu * 16 ? ncls[1] : cu * 16, 256, 0, st>>>(
                                             0u, nullptr, ncls[1], class_list + (size_t)1 * cap, runs, info, sum, s_idx, fin, chain_out, status_out, C));
        SWG_KERNEL_CHECK(ctx);
      }
      if (ncls[0]) {
        SWG_LAUNCH(ctx, "pair_renumber", pair_out_kernel<64, 1024><<<ncls[0] < cu * 64 ? ncls[0] : cu * 64, 64, 0, st>>>(
                                             0u, nullptr, ncls[0], class_list, runs, info, sum, s_idx, fin, chain_out, status_out, C));
        SWG_KERNEL_CHECK(ctx);
      }
    } else {
      SWG_LAUNCH(ctx, "pair_renumber", pair_renumber_kernel<<<gb, EW, 0, st>>>(n_runs, runs, sum, chain_out, C, perm));
      SWG_KERNEL_CHECK(ctx);
    }
  }
  // ---- the flags found on the device, and the statistics
  SWG_LAUNCH(ctx, "pair_totals", pair_totals_kernel<<<(n_runs + 4095) / 4096 < 64u ? (n_runs + 4095) / 4096 : 64u, 1024, 0, st>>>(n_runs, info, sum, n_out_pair, C));
  SWG_KERNEL_CHECK(ctx);
  uint64_t hc[9];
  static_assert(sizeof(PairCounters) == 72, "PairCounters is read back as nine words");
  SWG_TRY(swg_read_scalars(ctx, reinterpret_cast<const uint64_t*>(C), hc, 9));
  const uint32_t flags2 = (uint32_t)(hc[0] >> 32);
  if (flags2) {
    if (dbg) fprintf(stderr, "[swg] pair path: left on the device's word (flags %u): the global-sort path takes the call\n", flags2);
    swg_arena_restore(ctx, mark0);
    return SWG_OK;
  }
#ifdef SWG_PAIR_TIMING
  {
    unsigned long long ht[16];
    (void)hipMemcpyFromSymbol(ht, HIP_SYMBOL(g_pair_t), sizeof ht);
    fprintf(stderr, "[swg] pair_sort phases (100 MHz ticks summed over work-groups):");
    for (int k = 0; k < 13; ++k) fprintf(stderr, " %d:%llu", k, ht[k]);
    fprintf(stderr, "\n");
    unsigned long long z[16] = {0};
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_pair_t), z, sizeof z);
    (void)hipMemcpyFromSymbol(ht, HIP_SYMBOL(g_fin_t), sizeof ht);
    fprintf(stderr, "[swg] pair_finish phases:");
    for (int k = 0; k < 8; ++k) fprintf(stderr, " %d:%llu", k, ht[k]);
    fprintf(stderr, "\n");
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_fin_t), z, sizeof z);
  }
#endif
  if (stats) {
    stats->n_retained = hc[4];
    stats->n_swept = hc[5];
    stats->n_chains = hc[6];
    stats->n_chains_kept = hc[7];
    stats->n_out = hc[8];
  }
  *taken = 1;
  return SWG_OK;
}

}  // namespace swg_scaf
