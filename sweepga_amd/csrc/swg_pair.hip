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
#include "swg_scaffold_internal.h"

namespace swg_scaf {
namespace {

constexpr uint32_t PAIR_CELL = WALK_CHUNK;        // a chunk = the units that begin in one cell of this many members
constexpr uint32_t PAIR_S_MAX = 1024, PAIR_M_MAX = 4096, PAIR_L_MAX = 16384, PAIR_XL_MAX = uint32_t(1) << 18;
constexpr uint32_t PAIR_XL_CELLS = PAIR_XL_MAX / PAIR_CELL;
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
  uint32_t pad;
  unsigned long long n_alive, n_members, n_heads, n_kept, n_out;
};

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
  __syncthreads();
  if (lane == 63) ws[w] = inc;
  __syncthreads();
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
  __syncthreads();
  if (lane == 63) ws[w] = inc;
  __syncthreads();
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
  __syncthreads();
  if ((threadIdx.x & 63) == 0) ws[threadIdx.x >> 6] = v;
  __syncthreads();
  uint32_t tot = 0;
#pragma unroll
  for (int k = 0; k < NT / 64; ++k) tot += ws[k];
  return tot;
}

// ---- the runs of equal (q_id, t_id) ---------------------------------------------------------------------------------
// One bit per record (a run begins here), the run starts as an unordered list, and every pair entered into a hash set: a
// pair that is entered twice has two runs -- the input is not grouped by chromosome pair.
__global__ __launch_bounds__(256) void pair_boundary_kernel(uint32_t n, const uint32_t* __restrict__ q_id,
                                                            const uint32_t* __restrict__ t_id, unsigned long long* __restrict__ bitmap,
                                                            uint32_t* __restrict__ run_start, uint32_t cap,
                                                            unsigned long long* __restrict__ table, uint32_t tmask,
                                                            PairCounters* __restrict__ C) {
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  const int lane = threadIdx.x & 63;
  uint32_t q = 0, t = 0;
  bool start = false;
  if (i < n) {
    q = q_id[i];
    t = t_id[i];
    start = i == 0 || q_id[i - 1] != q || t_id[i - 1] != t;
  }
  const unsigned long long mask = __ballot(start);
  if (lane == 0 && i < n) bitmap[i >> 6] = mask;
  if (mask == 0) return;
  uint32_t base = 0;
  if (lane == 0) base = atomicAdd(&C->n_runs, (uint32_t)__popcll(mask));
  base = (uint32_t)__shfl((int)base, 0, 64);
  if (!start) return;
  const uint32_t k = base + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull));
  if (k >= cap) {
    atomicOr(&C->flags, PF_RUN_OVERFLOW);
    return;
  }
  run_start[k] = i;
  const unsigned long long key = ((unsigned long long)q << 32) | t;
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
// Every run's end (the next set bit), its size class, and the per-class lists the work-groups of the later kernels index.
__global__ __launch_bounds__(256) void pair_runs_kernel(uint32_t n, uint32_t cap, const uint32_t* __restrict__ run_start,
                                                        const unsigned long long* __restrict__ bitmap, PairRun* __restrict__ runs,
                                                        uint32_t* __restrict__ class_list, PairCounters* __restrict__ C) {
  const uint32_t k = blockIdx.x * 256u + threadIdx.x;
  const uint32_t nr = C->n_runs < cap ? C->n_runs : cap;
  if (k >= nr) return;
  const uint32_t a = run_start[k];
  const uint32_t n_words = (n + 63) >> 6;
  uint32_t end = n;
  bool too_long = false;
  if (a + 1 < n) {
    uint32_t w = (a + 1) >> 6;
    unsigned long long x = bitmap[w] & (~0ull << ((a + 1) & 63));
    const uint32_t w_stop = w + PAIR_XL_MAX / 64 + 2;
    for (;;) {
      if (x) {
        end = (w << 6) + (uint32_t)__builtin_ctzll(x);
        break;
      }
      if (++w >= n_words) break;
      if (w >= w_stop) {
        too_long = true;
        break;
      }
      x = bitmap[w];
    }
  }
  const uint32_t len = end - a;
  PairRun r;
  r.a = a;
  r.n = len;
  runs[k] = r;
  if (too_long || len > PAIR_XL_MAX) {
    atomicOr(&C->flags, PF_TOO_LONG);
    return;
  }
  const int cls = len <= PAIR_S_MAX ? 0 : (len <= PAIR_M_MAX ? 1 : (len <= PAIR_L_MAX ? 2 : 3));
  const uint32_t j = atomicAdd(&C->n_class[cls], 1u);
  class_list[(size_t)cls * cap + j] = k;
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
}
__device__ __forceinline__ uint32_t bucket_of(const BucketMap& B, uint32_t st, uint32_t k) {
  const float f = (float)(k - B.kmin[st]) * B.scale[st];
  uint32_t b = (uint32_t)f;
  const uint32_t top = B.nb[st] - 1u;
  b = b < top ? b : top;
  return B.off[st] + b;
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
  uint8_t* code;
  uint32_t *s_qs, *s_qe, *s_ts, *s_te, *s_m, *s_b, *s_idx, *pred;
  PairInfo* info;
  SpecBlock* chunks;
  uint32_t cap_chunks;
  PairCounters* C;
  PairTable gl_first;
  const uint32_t* seq_genome_last;
};

// NT threads, CAP = NT * E members per LDS batch, NBK buckets, IT = index type inside the pair (u16 while a run has at most
// 65,535 records), XL: runs longer than a batch (several batches over coarse bins of the key range, columns gathered).
template <int NT, int E, int NBK, typename IT, bool XL>
__global__ __launch_bounds__(NT) void pair_sort_kernel(PairSortArgs A) {
  constexpr int CAP = NT * E;
  constexpr int NCELL = XL ? (int)PAIR_XL_CELLS : (CAP + (int)PAIR_CELL - 1) / (int)PAIR_CELL;
  constexpr int NBIN = XL ? 4096 : 1;       // XL: coarse bins that the batches are made of
  constexpr int MAXB = XL ? 128 : 1;        // XL: batches per pair at most (2 * PAIR_XL_MAX / CAP + 1 would do)
  __shared__ uint32_t K[CAP];
  __shared__ IT I[CAP];
  __shared__ uint32_t cnt[NBK];
  __shared__ uint16_t R[XL ? 1 : CAP];
  __shared__ uint32_t QE[XL ? CAP : 1];
  __shared__ uint32_t bins[NBIN];
  __shared__ uint32_t b_lo[MAXB + 1];
  __shared__ uint32_t cellmin[NCELL];
  __shared__ uint64_t ws64[NT / 64 + 1];
  __shared__ uint32_t ws[NT / 64 + 1];
  __shared__ uint32_t sh_cnt[4], sh_kmin[2], sh_kmax[2], sh_first[3], sh_nb, sh_bad;
  const int tid = threadIdx.x;
  const PairRun run = A.runs[A.list[blockIdx.x]];
  const uint32_t a = run.a, n = run.n;
  if (tid < 4) sh_cnt[tid] = 0;
  if (tid < 2) {
    sh_kmin[tid] = 0xffffffffu;
    sh_kmax[tid] = 0;
  }
  if (tid < 3) sh_first[tid] = NONE;
  if (tid == 0) sh_bad = 0;
  for (int c = tid; c < NCELL; c += NT) cellmin[c] = NONE;
  __syncthreads();
  // ---- step-1 retain, members, the key range per strand
  const uint32_t q0 = A.q_id[a], t0 = A.t_id[a];
  const bool self_ok = A.keep_self || q0 != t0;
  uint32_t member_mask = 0;  // (!XL) bit e: record tid + e * NT is a member
  {
    uint32_t c_m[2] = {0, 0}, c_x = 0, kmin[2] = {0xffffffffu, 0xffffffffu}, kmax[2] = {0, 0}, fst[3] = {NONE, NONE, NONE};
    int e = 0;
    for (uint32_t li = tid; li < n; li += NT, ++e) {
      const uint32_t i = a + li;
      bool alive;
      if (A.alive_in) {
        alive = A.alive_in[i] != 0;
      } else {
        double id;
        if (A.identity) {
          id = A.identity[i];
        } else {
          const uint32_t bl = A.block_len[i];
          id = __ddiv_rn((double)A.matches[i], (double)(bl > 1u ? bl : 1u));
        }
        alive = self_ok && (A.min_block == 0 || (uint64_t)A.block_len[i] >= A.min_block) && id >= A.min_identity;
      }
      const bool member = alive && (A.member_in ? A.member_in[i] != 0 : true);
      const uint32_t st = A.strand[i] ? 1u : 0u;
      A.code[i] = alive ? (uint8_t)((member ? 1u : 2u) | (st << 2)) : (uint8_t)0;
      if (alive) {
        if (fst[2] == NONE) fst[2] = i;
        if (member) {
          const uint32_t qs = A.q_start[i];
          ++c_m[st];
          kmin[st] = qs < kmin[st] ? qs : kmin[st];
          kmax[st] = qs > kmax[st] ? qs : kmax[st];
          if (fst[st] == NONE) fst[st] = i;
          if (!XL) member_mask |= 1u << e;
        } else {
          ++c_x;
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
  if (tid == 0) {
    PairInfo pi;
    pi.m = m;
    pi.m_plus = m_plus;
    pi.M = M;
    pi.first_alive = sh_first[2];
    pi.first_mem[0] = sh_first[0];
    pi.first_mem[1] = sh_first[1];
    pi.q = q0;
    pi.t = t0;
    A.info[A.list[blockIdx.x]] = pi;
    if (sh_first[2] != NONE) {  // the genome pair's first alive record (apply_plane_sweep_to_mappings' group order, :1037-1046)
      uint32_t* slot = pair_slot(A.gl_first, A.seq_genome_last[q0], A.seq_genome_last[t0]);
      if (*slot > sh_first[2]) atomicMin(slot, sh_first[2]);
    }
    if (M) {
      atomicAdd(&A.C->n_alive, (unsigned long long)M);
      atomicAdd(&A.C->n_members, (unsigned long long)m);
    }
  }
  if (M == 0) return;
  bool degenerate = false;
  // ---- the alive records that are not members (behind a mapping sweep): kept with the pair for the inversion capture and
  // the rescue, behind the members, in input order; bit 31 of their index carries the strand
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
  const uint32_t kmn[2] = {sh_kmin[0], sh_kmin[1]}, kmx[2] = {sh_kmax[0], sh_kmax[1]};
  // ---- batches (XL: runs of coarse bins of at most CAP members each; otherwise the whole pair is one batch)
  uint32_t n_batches = 1;
  BucketMap BM;
  if (XL) {
    bucket_map_make(BM, NBIN, m_plus, m - m_plus, kmn, kmx);
    for (int b = tid; b < NBIN; b += NT) bins[b] = 0;
    __syncthreads();
    for (uint32_t li = tid; li < n; li += NT) {
      const uint8_t code = A.code[a + li];
      if ((code & 3u) == 1u) atomicAdd(&bins[bucket_of(BM, code >> 2, A.q_start[a + li])], 1u);
    }
    __syncthreads();
    if (tid == 0) {  // greedy: a batch is closed when the next bin would not fit
      uint32_t nb = 0, acc = 0;
      b_lo[0] = 0;
      for (uint32_t b = 0; b < (uint32_t)NBIN; ++b) {
        const uint32_t c = bins[b];
        if (c > (uint32_t)CAP) sh_bad = 1;  // one bin denser than a batch: not for this path
        if (acc + c > (uint32_t)CAP && nb + 1 < (uint32_t)MAXB) {
          b_lo[++nb] = b;
          acc = 0;
        } else if (acc + c > (uint32_t)CAP) {
          sh_bad = 1;
        }
        acc += c;
      }
      b_lo[++nb] = NBIN;
      sh_nb = nb;
    }
    __syncthreads();
    if (sh_bad) {
      if (tid == 0) atomicOr(&A.C->flags, PF_FALLBACK);
      return;
    }
    n_batches = sh_nb;
  }
  uint64_t carry_max = 0;  // running maximum of ((strand << 32) | q_end) over the positions before the batch
  uint32_t base = 0;       // members before the batch
  for (uint32_t bt = 0; bt < n_batches; ++bt) {
    const uint32_t bin_lo = XL ? b_lo[bt] : 0u, bin_hi = XL ? b_lo[bt + 1] : 1u;
    // fine buckets of the batch: XL -- the coarse map refined by a power of two, relative to the batch's first bin
    int xl_shift = 0;      // fine id = (uint32)(f * 2^12) >> xl_shift, minus the batch's first
    uint32_t xl_first = 0;
    BucketMap FM;
    if (XL) {
      // f < NBIN = 2^12 has 12 integer bits, so f * 2^12 is exact in the integer part's relation to f: (uint32)(f * 4096) >> 12
      // == (uint32)f (multiplying a float by a power of two is exact), and the refinement stays monotone
      const uint32_t span = (bin_hi - bin_lo) << 12;
      while ((span >> xl_shift) > (uint32_t)NBK) ++xl_shift;
      xl_first = (bin_lo << 12) >> xl_shift;
    } else {
      bucket_map_make(FM, NBK, m_plus, m - m_plus, kmn, kmx);
    }
    auto coarse_of = [&](uint32_t st, uint32_t k) -> uint32_t { return bucket_of(BM, st, k); };
    auto fine_of = [&](uint32_t st, uint32_t k) -> uint32_t {
      if (XL) {
        const float f = (float)(k - BM.kmin[st]) * BM.scale[st];
        const uint32_t top = BM.nb[st] - 1u;
        uint32_t g = (uint32_t)(f * 4096.0f);
        const uint32_t cb = g >> 12;
        if (cb > top) g = (top << 12) | 0xfffu;  // (the clamp of bucket_of, in fine units)
        g += BM.off[st] << 12;
        const uint32_t b = (g >> xl_shift) - xl_first;
        return b < (uint32_t)NBK ? b : (uint32_t)NBK - 1u;
      }
      return bucket_of(FM, st, k);
    };
    for (int b = tid; b < NBK; b += NT) cnt[b] = 0;
    __syncthreads();
    // count
    if (XL) {
      for (uint32_t li = tid; li < n; li += NT) {
        const uint8_t code = A.code[a + li];
        if ((code & 3u) != 1u) continue;
        const uint32_t k = A.q_start[a + li], st = code >> 2, cb = coarse_of(st, k);
        if (cb >= bin_lo && cb < bin_hi) atomicAdd(&cnt[fine_of(st, k)], 1u);
      }
    } else {
      int e = 0;
      for (uint32_t li = tid; li < n; li += NT, ++e)
        if (member_mask & (1u << e)) atomicAdd(&cnt[fine_of(A.strand[a + li] ? 1u : 0u, A.q_start[a + li])], 1u);
    }
    __syncthreads();
    // exclusive scan of the bucket counts (NBK / NT consecutive counters per thread)
    uint32_t mb;
    {
      constexpr int PER = NBK / NT;
      static_assert(NBK % NT == 0 && PER >= 1, "bucket counters per thread");
      uint32_t c[PER], s = 0;
#pragma unroll
      for (int j = 0; j < PER; ++j) {
        c[j] = cnt[tid * PER + j];
        s += c[j];
      }
      uint32_t off = block_excl_sum<NT>(s, ws, &mb);
#pragma unroll
      for (int j = 0; j < PER; ++j) {
        cnt[tid * PER + j] = off;
        off += c[j];
      }
    }
    __syncthreads();
    // scatter (unordered inside a bucket; cnt[b] ends as the bucket's end)
    if (XL) {
      for (uint32_t li = tid; li < n; li += NT) {
        const uint8_t code = A.code[a + li];
        if ((code & 3u) != 1u) continue;
        const uint32_t k = A.q_start[a + li], st = code >> 2, cb = coarse_of(st, k);
        if (cb >= bin_lo && cb < bin_hi) {
          const uint32_t pos = atomicAdd(&cnt[fine_of(st, k)], 1u);
          K[pos] = k;
          I[pos] = (IT)li;
        }
      }
    } else {
      int e = 0;
      for (uint32_t li = tid; li < n; li += NT, ++e)
        if (member_mask & (1u << e)) {
          const uint32_t k = A.q_start[a + li];
          const uint32_t pos = atomicAdd(&cnt[fine_of(A.strand[a + li] ? 1u : 0u, k)], 1u);
          K[pos] = k;
          I[pos] = (IT)li;
        }
    }
    __syncthreads();
    // order inside the buckets: final position = bucket begin + the bucket's elements that order before by (key, index)
    const uint32_t plus_here = m_plus > base ? (m_plus - base < mb ? m_plus - base : mb) : 0u;  // '+' members of the batch
    {
      uint32_t rk[E], rl[E], rr[E];
#pragma unroll
      for (int e = 0; e < E; ++e) {
        const uint32_t pos = (uint32_t)tid + (uint32_t)e * NT;
        rr[e] = NONE;
        if (pos < mb) {
          const uint32_t k = K[pos], li = (uint32_t)I[pos];
          const uint32_t b = fine_of(pos >= plus_here ? 1u : 0u, k);
          const uint32_t hi = cnt[b], lo = b ? cnt[b - 1] : 0u;
          uint32_t r = lo;
          for (uint32_t x = lo; x < hi; ++x) {
            const uint32_t kx = K[x], lx = (uint32_t)I[x];
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
          I[rr[e]] = (IT)rl[e];
        }
    }
    __syncthreads();
    // ---- the sorted q_start and record index out; the other columns in that order
    for (uint32_t p = tid; p < mb; p += NT) {
      const uint32_t li = (uint32_t)I[p];
      A.s_qs[a + base + p] = K[p];
      A.s_idx[a + base + p] = a + li;
      A.pred[a + base + p] = NONE;
      if (!XL) R[li] = (uint16_t)p;
    }
    uint32_t qs_r[E], qe_r[E];  // the thread's own E consecutive positions, for the cuts
#pragma unroll
    for (int e = 0; e < E; ++e) {
      const uint32_t p = (uint32_t)tid * E + e;
      qs_r[e] = p < mb ? K[p] : 0u;
    }
    __syncthreads();
    if (XL) {
      for (uint32_t p = tid; p < mb; p += NT) {
        const uint32_t i = a + (uint32_t)I[p];
        const uint32_t qe = A.q_end[i], ts = A.t_start[i], te = A.t_end[i];
        QE[p] = qe;
        A.s_qe[a + base + p] = qe;
        A.s_ts[a + base + p] = ts;
        A.s_te[a + base + p] = te;
        A.s_m[a + base + p] = A.matches[i];
        A.s_b[a + base + p] = A.block_len[i];
        degenerate |= K[p] >= qe || ts >= te;
      }
      __syncthreads();
#pragma unroll
      for (int e = 0; e < E; ++e) {
        const uint32_t p = (uint32_t)tid * E + e;
        qe_r[e] = p < mb ? QE[p] : 0u;
      }
    } else {
      // transposition through LDS: coalesced reads in input order land at their sorted position, coalesced writes follow
      auto column = [&](const uint32_t* __restrict__ src, uint32_t* __restrict__ dst, const uint32_t* __restrict__ lower, bool keep) {
        int e = 0;
        for (uint32_t li = tid; li < n; li += NT, ++e)
          if (member_mask & (1u << e)) {
            const uint32_t v = src[a + li];
            K[R[li]] = v;
            if (lower) degenerate |= lower[a + li] >= v;  // start >= end (the start was loaded a moment ago: L1)
          }
        __syncthreads();
        for (uint32_t p = tid; p < mb; p += NT) dst[a + p] = K[p];
        if (keep) {
#pragma unroll
          for (int e2 = 0; e2 < E; ++e2) {
            const uint32_t p = (uint32_t)tid * E + e2;
            qe_r[e2] = p < mb ? K[p] : 0u;
          }
        }
        __syncthreads();
      };
      column(A.q_end, A.s_qe, A.q_start, true);
      column(A.t_start, A.s_ts, nullptr, false);
      column(A.t_end, A.s_te, A.t_start, false);
      column(A.matches, A.s_m, nullptr, false);
      column(A.block_len, A.s_b, nullptr, false);
    }
    // ---- units: position p opens one when its q_start lies beyond every earlier q_end of its strand by more than the gap
    // (no window of paf_filter.rs:786-796 can straddle it); a chunk = the units that begin in one cell
    {
      uint64_t pre[E], tmax = 0;
#pragma unroll
      for (int e = 0; e < E; ++e) {
        const uint32_t p = (uint32_t)tid * E + e;
        pre[e] = tmax;
        if (p < mb) {
          const uint64_t c = ((uint64_t)(base + p >= m_plus ? 1u : 0u) << 32) | qe_r[e];
          tmax = c > tmax ? c : tmax;
        }
      }
      uint64_t tot;
      uint64_t before = block_excl_max<NT>(tmax, ws64, &tot);
      before = before > carry_max ? before : carry_max;
#pragma unroll
      for (int e = 0; e < E; ++e) {
        const uint32_t p = (uint32_t)tid * E + e;
        if (p >= mb) continue;
        const uint32_t pa = base + p;
        const uint64_t prev = pre[e] > before ? pre[e] : before;
        uint64_t lim = (prev & 0xffffffffull) + A.max_gap;
        if (lim < A.max_gap) lim = ~0ull;  // saturate
        const bool unit = pa == 0 || pa == m_plus || (uint64_t)qs_r[e] > lim;
        if (unit) atomicMin(&cellmin[pa / PAIR_CELL], pa);
      }
      carry_max = tot > carry_max ? tot : carry_max;
    }
    base += mb;
    __syncthreads();
  }
  if (A.check_degenerate && __any(degenerate) && (tid & 63) == 0) atomicOr(&A.C->flags, PF_FALLBACK);
  // ---- the chunk list: the first unit of every cell opens a chunk, and so does the first '-' member
  if (tid == 0) {
    uint32_t prev = NONE, count = 0;
    bool mp_pending = m_plus > 0 && m_plus < m;
    // two passes over the starts: count, reserve, write
    auto for_starts = [&](auto&& f) {
      bool pend = mp_pending;
      for (int c = 0; c < NCELL; ++c) {
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
    for_starts([&](uint32_t) { ++count; });
    const uint32_t slot = atomicAdd(&A.C->n_chunks, count);
    if (slot + count > A.cap_chunks) {
      atomicOr(&A.C->flags, PF_FALLBACK);
    } else {
      uint32_t k = 0;
      bool bad = false;
      auto emit = [&](uint32_t b, uint32_t e) {
        SpecBlock d;
        d.bb = a + b;
        d.be = a + e;
        d.ue = a + e;
        d.pad = b >= m_plus ? 1u : 0u;
        A.chunks[slot + k++] = d;
        if (e - b >= LABEL_CAP_ELEMS) bad = true;  // a unit too long for the per-chunk labelling
      };
      for_starts([&](uint32_t v) {
        if (prev != NONE) emit(prev, v);
        prev = v;
      });
      if (prev != NONE) emit(prev, m);
      if (bad) atomicOr(&A.C->flags, PF_FALLBACK);
    }
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
  const uint64_t* fp_thr;
  PairCounters* C;
};

// plane_sweep_both with no limit on either axis (plane_sweep_exact.rs:268-461 with mappings_to_keep = usize::MAX): a sweep
// over at most one interval returns it; otherwise an interval survives iff it is ever in the tree at a mark_good call, i.e.
// iff start < end.  The target sweep runs over the query sweep's survivors.
template <int NT>
__global__ __launch_bounds__(NT) void pair_finish_kernel(PairFinishArgs A) {
  __shared__ uint32_t ws[NT / 64 + 1];
  __shared__ uint64_t ws64[NT / 64 + 1];
  const int tid = threadIdx.x;
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
  // ---- the chains that pass the span / identity filter
  uint32_t c0 = 0, c1 = 0, cq = 0;
  for (uint32_t p = tid; p < m; p += NT)
    if (A.ok_head[a + p]) {
      const HeadRec hr = A.rec[a + p];
      if (p < m_plus) ++c0; else ++c1;
      cq += hr.qs < hr.qe ? 1u : 0u;
    }
  const uint32_t np0 = block_sum<NT>(c0, ws), np1 = block_sum<NT>(c1, ws), nq = block_sum<NT>(cq, ws);
  const uint32_t n_ch = np0 + np1;
  if (n_ch == 0) {
    if (tid == 0) A.sum[rk] = sm;
    return;
  }
  const bool all_q = n_ch <= 1;
  const uint32_t nq_eff = all_q ? n_ch : nq;
  const bool all_t = nq_eff <= 1;
  // ---- kept chains ranked in position order ('+' chains first); the kept '+' chains listed for the inversion capture
  uint32_t kept_before = 0, kept_plus = 0;
  for (uint32_t p0 = 0; p0 < m; p0 += NT) {
    const uint32_t p = p0 + tid;
    bool kept = false;
    HeadRec hr{};
    if (p < m && A.ok_head[a + p]) {
      hr = A.rec[a + p];
      kept = (all_q || hr.qs < hr.qe) && (all_t || hr.ts < hr.te);
    }
    uint32_t tot;
    const uint32_t r = kept_before + block_excl_sum<NT>(kept ? 1u : 0u, ws, &tot);
    if (p < m && A.ok_head[a + p]) A.head_num[a + p] = kept ? r : NONE;
    if (kept && p < m_plus) {  // r < kept_plus_total: its slot in the list
      A.f_qs[a + r] = hr.qs;
      A.f_qe[a + r] = hr.qe;
      A.f_ts[a + r] = hr.ts;
    }
    kept_before += tot;
    if (p0 < m_plus) {  // (block-uniform) kept '+' chains so far
      const uint32_t plus_here = block_sum<NT>((kept && p < m_plus) ? 1u : 0u, ws);
      kept_plus += plus_here;
    }
  }
  const uint32_t n_kept = kept_before, kP = kept_plus, kM = n_kept - kP;
  // the reference's all_chains order inside the pair: the (query, target, strand) group that appears first in the metadata
  const bool plus_first = pi.first_mem[0] < pi.first_mem[1];
  sm.n_pass = n_ch;
  sm.n_kept = n_kept;
  sm.minmem = np0 && np1 ? (pi.first_mem[0] < pi.first_mem[1] ? pi.first_mem[0] : pi.first_mem[1]) : (np0 ? pi.first_mem[0] : pi.first_mem[1]);
  if (tid == 0) A.sum[rk] = sm;
  if (n_kept == 0) return;
  __syncthreads();  // head_num / f_* of the whole pair are written
  auto local_number = [&](uint32_t r, bool minus) -> uint32_t {  // 1-based, in the pair's all_chains order
    if (plus_first) return r + 1;
    return minus ? r - kP + 1 : r + kM + 1;
  };
  // ---- anchors: the members of kept chains (paf_filter.rs:517-528)
  uint32_t out = 0;
  for (uint32_t p = tid; p < m; p += NT) {
    const uint32_t h = A.hd[a + p];
    if (!A.ok_head[h]) continue;
    const uint32_t r = A.head_num[h];
    if (r == NONE) continue;
    const uint32_t i = A.s_idx[a + p];
    A.chain[i] = local_number(r, h - a >= m_plus);
    A.status[i] = SWG_ST_SCAFFOLD;
    ++out;
  }
  // ---- inversion capture (paf_filter.rs:535-597): a '-' record that is not an anchor joins the first kept '+' chain of its
  // pair whose window and diagonal it sits on
  if (!A.scaffolds_only && kP > 0 && M > m_plus) {
    // running maximum of the chains' ends
    uint64_t carry = 0;
    for (uint32_t c0b = 0; c0b < kP; c0b += NT) {
      const uint32_t c = c0b + tid;
      const uint64_t v = c < kP ? (uint64_t)A.f_qe[a + c] : 0ull;
      uint64_t tot;
      uint64_t ex = block_excl_max<NT>(v, ws64, &tot);
      ex = ex > carry ? ex : carry;
      if (c < kP) A.f_pm[a + c] = (uint32_t)(v > ex ? v : ex);
      carry = tot > carry ? tot : carry;
    }
    __syncthreads();  // f_pm, and the anchors' chain numbers
    const uint64_t gap = A.gap, max_dev = A.fp_thr[0];
    for (uint32_t p = m_plus + tid; p < M; p += NT) {
      const uint32_t iw = A.s_idx[a + p];
      if (p >= m && (iw >> 31) == 0) continue;  // an alive non-member on the '+' strand
      const uint32_t i = iw & 0x7fffffffu;
      if (A.chain[i]) continue;  // already an anchor
      const uint64_t qs = A.s_qs[a + p], qe = A.s_qe[a + p], ts = A.s_ts[a + p], te = A.s_te[a + p];
      const uint64_t qc = (qs + qe) / 2, tc = (ts + te) / 2;
      const uint64_t lim = qe > ~0ull - gap ? ~0ull : qe + gap;  // chain.query_start.saturating_sub(gap) <= qe
      uint32_t l = 0, r = kP;  // first chain with q_start > lim
      while (l < r) {
        const uint32_t mid = l + ((r - l) >> 1);
        if ((uint64_t)A.f_qs[a + mid] <= lim) l = mid + 1; else r = mid;
      }
      uint32_t lo = 0, hi = l;  // first slot whose running maximum of ends reaches the record
      while (lo < hi) {
        const uint32_t mid = lo + ((hi - lo) >> 1);
        const uint64_t pm = A.f_pm[a + mid];
        if ((pm > ~0ull - gap ? ~0ull : pm + gap) < qs) lo = mid + 1; else hi = mid;
      }
      uint32_t best = NONE;
      for (uint32_t c = lo; c < l; ++c) {
        const uint64_t cqe = A.f_qe[a + c];
        if ((cqe > ~0ull - gap ? ~0ull : cqe + gap) < qs) continue;
        const int64_t diag = (int64_t)A.f_ts[a + c] - (int64_t)A.f_qs[a + c];
        const int64_t dev = (int64_t)tc - (int64_t)qc - diag;
        const uint64_t deviation = dev < 0 ? (uint64_t)0 - (uint64_t)dev : (uint64_t)dev;
        if (deviation <= max_dev) {
          best = c;
          break;
        }
      }
      if (best != NONE) {
        A.chain[i] = local_number(best, false);
        A.status[i] = SWG_ST_SCAFFOLD;
        ++out;
      }
    }
  }
  out = block_sum<NT>(out, ws);
  if (tid == 0) {
    atomicAdd(&A.C->n_kept, (unsigned long long)n_kept);
    if (out) atomicAdd(&A.C->n_out, (unsigned long long)out);
  }
}

// ---- chain_N bases ----------------------------------------------------------------------------------------------------
// plane_sweep_scaffolds returns the kept chains genome pair by genome pair (first two '#' parts, first appearance among the
// chains that pass the filter), chromosome pair by chromosome pair inside, all_chains order inside that
// (plane_sweep_scaffold.rs:116-130, 204-251).  First appearance in all_chains order = the (query, target, strand) group's
// place in the metadata order: genome pair (prefix up to the last '#') by its first alive record, then the group's first
// member (paf_filter.rs:1037-1046, 1110-1120, 761-770).
__global__ __launch_bounds__(EW) void pair_key1_kernel(uint32_t n_runs, const PairInfo* __restrict__ info, const PairSum* __restrict__ sum,
                                                       PairTable gl_first, const uint32_t* __restrict__ seq_genome_last,
                                                       uint64_t* __restrict__ key, uint32_t* __restrict__ val) {
  const uint32_t k = blockIdx.x * EW + threadIdx.x;
  if (k >= n_runs) return;
  uint64_t x = ~0ull;
  if (sum[k].n_pass) {
    const uint32_t g = pair_get(gl_first, seq_genome_last[info[k].q], seq_genome_last[info[k].t]);
    x = ((uint64_t)g << 32) | sum[k].minmem;
  }
  key[k] = x;
  val[k] = k;
}
__global__ __launch_bounds__(EW) void pair_rank1_kernel(uint32_t n_runs, const uint32_t* __restrict__ order, const PairInfo* __restrict__ info,
                                                        const PairSum* __restrict__ sum, const uint32_t* __restrict__ seq_genome_two,
                                                        PairTable gp2_first, uint32_t* __restrict__ rank1) {
  const uint32_t r = blockIdx.x * EW + threadIdx.x;
  if (r >= n_runs) return;
  const uint32_t k = order[r];
  rank1[k] = r;
  if (sum[k].n_pass) atomicMin(pair_slot(gp2_first, seq_genome_two[info[k].q], seq_genome_two[info[k].t]), r);
}
__global__ __launch_bounds__(EW) void pair_key2_kernel(uint32_t n_runs, const PairInfo* __restrict__ info, const PairSum* __restrict__ sum,
                                                       const uint32_t* __restrict__ rank1, const uint32_t* __restrict__ seq_genome_two,
                                                       PairTable gp2_first, uint64_t* __restrict__ key, uint32_t* __restrict__ val) {
  const uint32_t k = blockIdx.x * EW + threadIdx.x;
  if (k >= n_runs) return;
  uint64_t x = ~0ull;
  if (sum[k].n_pass) x = ((uint64_t)pair_get(gp2_first, seq_genome_two[info[k].q], seq_genome_two[info[k].t]) << 32) | rank1[k];
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
// Few pairs (the usual case for a small input): no sort at all -- a pair's place among the keys is a count, and its base is
// the sum of the kept chains of the pairs whose key is smaller; one work-group, O(pairs^2) compares from LDS.
constexpr int NUMBER_SMALL = 2048;
__global__ __launch_bounds__(1024) void pair_number_small_kernel(uint32_t n_runs, const PairInfo* __restrict__ info, PairSum* __restrict__ sum,
                                                                 PairTable gl_first, const uint32_t* __restrict__ seq_genome_last,
                                                                 PairTable gp2_first, const uint32_t* __restrict__ seq_genome_two) {
  __shared__ uint64_t key[NUMBER_SMALL];
  __shared__ uint32_t kept[NUMBER_SMALL];
  const int tid = threadIdx.x;
  uint64_t k1[NUMBER_SMALL / 1024];
  uint32_t r1[NUMBER_SMALL / 1024];
#pragma unroll
  for (int u = 0; u < NUMBER_SMALL / 1024; ++u) {
    const uint32_t k = (uint32_t)tid + (uint32_t)u * 1024u;
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
  for (int u = 0; u < NUMBER_SMALL / 1024; ++u) {
    const uint32_t k = (uint32_t)tid + (uint32_t)u * 1024u;
    uint32_t r = 0;
    if (k < n_runs && k1[u] != ~0ull) {
      for (uint32_t j = 0; j < n_runs; ++j) r += key[j] < k1[u] ? 1u : 0u;  // (the keys of pairs with chains are distinct)
      atomicMin(pair_slot(gp2_first, seq_genome_two[info[k].q], seq_genome_two[info[k].t]), r);
    }
    r1[u] = r;
  }
  __threadfence();
  __syncthreads();
#pragma unroll
  for (int u = 0; u < NUMBER_SMALL / 1024; ++u) {
    const uint32_t k = (uint32_t)tid + (uint32_t)u * 1024u;
    if (k < n_runs && k1[u] != ~0ull) {
      uint32_t* slot = pair_slot(gp2_first, seq_genome_two[info[k].q], seq_genome_two[info[k].t]);
      k1[u] = ((uint64_t)__hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) << 32) | r1[u];
    }
  }
  __syncthreads();
#pragma unroll
  for (int u = 0; u < NUMBER_SMALL / 1024; ++u) {
    const uint32_t k = (uint32_t)tid + (uint32_t)u * 1024u;
    if (k < n_runs) key[k] = k1[u];
  }
  __syncthreads();
#pragma unroll
  for (int u = 0; u < NUMBER_SMALL / 1024; ++u) {
    const uint32_t k = (uint32_t)tid + (uint32_t)u * 1024u;
    if (k < n_runs) {
      uint32_t b = 0;
      if (k1[u] != ~0ull)
        for (uint32_t j = 0; j < n_runs; ++j) b += key[j] < k1[u] ? kept[j] : 0u;
      sum[k].base = b;
    }
  }
}
// chain numbers: pair-local -> global (records of pairs without kept chains hold zeros)
__global__ __launch_bounds__(EW) void pair_renumber_kernel(uint32_t n_runs, const PairRun* __restrict__ runs, const PairSum* __restrict__ sum,
                                                           uint32_t* __restrict__ chain) {
  for (uint32_t k = blockIdx.x; k < n_runs; k += gridDim.x) {
    const uint32_t base = sum[k].base;
    if (base == 0 || sum[k].n_kept == 0) continue;
    const uint32_t a = runs[k].a, n = runs[k].n;
    for (uint32_t li = threadIdx.x; li < n; li += EW) {
      const uint32_t c = chain[a + li];
      if (c) chain[a + li] = c + base;
    }
  }
}

bool pair_path_wanted() {
  static const int knob = getenv("SWG_GROUP_FUSED") ? atoi(getenv("SWG_GROUP_FUSED")) : -1;
  return knob != 0;
}

}  // namespace

// The scaffold stage for records grouped by chromosome pair.  *taken = 0: not applicable (not grouped, a pair too long, a
// configuration this path does not cover, or a condition found on the device) -- nothing the caller cannot overwrite was
// done, and it runs the global-sort path.  alive / member: nullptr = step-1 retain evaluated here / members == alive records.
int scaffold_stage_pairs(swg_ctx* ctx, const swg_records* r, const swg_config* cfg, const uint8_t* alive, const uint8_t* member,
                         bool sweep_assumed_identity, uint8_t* status_out, uint32_t* chain_out, swg_stats* stats, int* taken) {
  *taken = 0;
  if (!pair_path_wanted()) return SWG_OK;
  const uint64_t n64 = r->n;
  if (n64 < 2 || n64 >= (uint64_t(1) << 31)) return SWG_OK;
  // covered here: no limit on either axis of the scaffold sweep (the CLI default), no rescue
  uint64_t kq, kt;
  if (cfg->scaffold_filter_mode == SWG_MODE_ONE_TO_ONE) {
    kq = kt = 1;
  } else {
    kq = cfg->scaffold_max_per_query ? cfg->scaffold_max_per_query : SWG_K_INF;
    kt = cfg->scaffold_max_per_target ? cfg->scaffold_max_per_target : SWG_K_INF;
  }
  if (kq != SWG_K_INF || kt != SWG_K_INF) return SWG_OK;
  if (!cfg->scaffolds_only && cfg->scaffold_max_deviation != 0) return SWG_OK;
  const uint32_t n = (uint32_t)n64;
  hipStream_t st = ctx->stream;
  static const bool dbg = getenv("SWG_DEBUG") != nullptr;
  const swg_arena_mark mark0 = swg_arena_save(ctx);
  // ---- runs
  const uint32_t cap = n < 65536u ? n : (n / 16 > 65536u ? n / 16 : 65536u);
  uint32_t tsize = 1;
  while (tsize < 2 * cap) tsize <<= 1;
  PairCounters* C = swg_alloc<PairCounters>(ctx, 1);
  unsigned long long* bitmap = swg_alloc<unsigned long long>(ctx, (n + 63) / 64 + 1);
  uint32_t* run_start = swg_alloc<uint32_t>(ctx, cap);
  unsigned long long* table = swg_alloc<unsigned long long>(ctx, tsize);
  PairRun* runs = swg_alloc<PairRun>(ctx, cap);
  uint32_t* class_list = swg_alloc<uint32_t>(ctx, (size_t)4 * cap);
  SWG_CHECK_ARENA(ctx);
  SWG_HIP(ctx, hipMemsetAsync(C, 0, sizeof(PairCounters), st));
  SWG_HIP(ctx, hipMemsetAsync(table, 0xff, (size_t)tsize * 8, st));
  SWG_LAUNCH(ctx, "pair_boundary", pair_boundary_kernel<<<(n + 255) / 256, 256, 0, st>>>(n, r->q_id, r->t_id, bitmap, run_start, cap, table, tsize - 1, C));
  SWG_KERNEL_CHECK(ctx);
  SWG_LAUNCH(ctx, "pair_runs", pair_runs_kernel<<<(cap + 255) / 256, 256, 0, st>>>(n, cap, run_start, bitmap, runs, class_list, C));
  SWG_KERNEL_CHECK(ctx);
  uint64_t h[4];
  static_assert(sizeof(PairCounters) >= 32, "the first four words are read back");
  SWG_TRY(swg_read_scalars(ctx, reinterpret_cast<const uint64_t*>(C), h, 3));
  const uint32_t n_runs = (uint32_t)h[0], flags = (uint32_t)(h[0] >> 32);
  const uint32_t ncls[4] = {(uint32_t)h[1], (uint32_t)(h[1] >> 32), (uint32_t)h[2], (uint32_t)(h[2] >> 32)};
  if (flags || n_runs == 0) {
    if (dbg) fprintf(stderr, "[swg] pair path: not applicable (%u runs, flags %u)\n", n_runs, flags);
    swg_arena_restore(ctx, mark0);
    return SWG_OK;
  }
  if (dbg) fprintf(stderr, "[swg] pair path: %u pairs (%u / %u / %u / %u by size class)\n", n_runs, ncls[0], ncls[1], ncls[2], ncls[3]);
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
  PairInfo* info = swg_alloc<PairInfo>(ctx, n_runs);
  PairSum* sum = swg_alloc<PairSum>(ctx, n_runs);
  const uint32_t cap_chunks = n / PAIR_CELL + 2 * n_runs + 16;
  SpecBlock* chunks = swg_alloc<SpecBlock>(ctx, cap_chunks);
  uint64_t* fp_thr = swg_alloc<uint64_t>(ctx, 2);
  SWG_CHECK_ARENA(ctx);
  PairTable gl_first, gp2_first;
  SWG_TRY(pair_table_make(ctx, r->n_genome_last, n_runs, &gl_first));
  SWG_TRY(pair_table_make(ctx, r->n_genome_two, n_runs, &gp2_first));
  SWG_HIP(ctx, hipMemsetAsync(chain_out, 0, (size_t)n * sizeof(uint32_t), st));
  SWG_HIP(ctx, hipMemsetAsync(status_out, 0, n, st));
  SWG_LAUNCH(ctx, "fp_thresholds", fp_thresholds_kernel<<<1, 64, 0, st>>>(cfg->scaffold_gap, cfg->scaffold_max_deviation, fp_thr));
  SWG_KERNEL_CHECK(ctx);
  PairSortArgs SA{};
  SA.q_id = r->q_id; SA.t_id = r->t_id; SA.q_start = r->q_start; SA.q_end = r->q_end; SA.t_start = r->t_start; SA.t_end = r->t_end;
  SA.matches = r->matches; SA.block_len = r->block_len; SA.identity = r->identity; SA.strand = r->strand;
  SA.alive_in = alive; SA.member_in = member;
  SA.min_block = cfg->min_block_length; SA.keep_self = cfg->keep_self; SA.min_identity = cfg->min_identity;
  SA.check_degenerate = sweep_assumed_identity ? 1 : 0;
  SA.max_gap = cfg->scaffold_gap;
  SA.runs = runs;
  SA.code = code; SA.s_qs = s_qs; SA.s_qe = s_qe; SA.s_ts = s_ts; SA.s_te = s_te; SA.s_m = s_m; SA.s_b = s_b; SA.s_idx = s_idx; SA.pred = pred;
  SA.info = info; SA.chunks = chunks; SA.cap_chunks = cap_chunks; SA.C = C; SA.gl_first = gl_first; SA.seq_genome_last = r->seq_genome_last;
  for (int c = 0; c < 4; ++c) {
    if (!ncls[c]) continue;
    SA.list = class_list + (size_t)c * cap;
    switch (c) {
      case 0: SWG_LAUNCH_N(ctx, "pair_sort", 0, pair_sort_kernel<64, 16, 256, uint16_t, false><<<ncls[c], 64, 0, st>>>(SA)); break;
      case 1: SWG_LAUNCH_N(ctx, "pair_sort", 0, pair_sort_kernel<256, 16, 1024, uint16_t, false><<<ncls[c], 256, 0, st>>>(SA)); break;
      case 2: SWG_LAUNCH_N(ctx, "pair_sort", 0, pair_sort_kernel<1024, 16, 4096, uint16_t, false><<<ncls[c], 1024, 0, st>>>(SA)); break;
      default: SWG_LAUNCH_N(ctx, "pair_sort_xl", 0, pair_sort_kernel<1024, 8, 4096, uint32_t, true><<<ncls[c], 1024, 0, st>>>(SA)); break;
    }
    SWG_KERNEL_CHECK(ctx);
  }
  SWG_TRY(pair_walk_launch(ctx, cap_chunks, &C->n_chunks, chunks, s_qs, s_qe, s_ts, s_te, cfg->scaffold_gap, bps, pred));
  SWG_TRY(pair_label_launch(ctx, cap_chunks, &C->n_chunks, chunks, pred, s_qs, s_qe, s_ts, s_te, s_m, s_b, cfg->min_scaffold_length,
                            cfg->min_scaffold_identity, hd, ok_head, head_rec, &C->n_heads));
  PairFinishArgs FA{};
  FA.runs = runs; FA.info = info; FA.sum = sum;
  FA.s_qs = s_qs; FA.s_qe = s_qe; FA.s_ts = s_ts; FA.s_te = s_te; FA.s_idx = s_idx; FA.hd = hd; FA.ok_head = ok_head; FA.rec = head_rec;
  FA.head_num = pred;
  FA.f_qs = s_m; FA.f_qe = s_b; FA.f_ts = reinterpret_cast<uint32_t*>(bps); FA.f_pm = reinterpret_cast<uint32_t*>(bps) + n;
  FA.status = status_out; FA.chain = chain_out; FA.scaffolds_only = cfg->scaffolds_only; FA.gap = cfg->scaffold_gap; FA.fp_thr = fp_thr; FA.C = C;
  for (int c = 0; c < 4; ++c) {
    if (!ncls[c]) continue;
    FA.list = class_list + (size_t)c * cap;
    switch (c) {
      case 0: SWG_LAUNCH(ctx, "pair_finish", pair_finish_kernel<64><<<ncls[c], 64, 0, st>>>(FA)); break;
      case 1: SWG_LAUNCH(ctx, "pair_finish", pair_finish_kernel<256><<<ncls[c], 256, 0, st>>>(FA)); break;
      default: SWG_LAUNCH(ctx, "pair_finish", pair_finish_kernel<1024><<<ncls[c], 1024, 0, st>>>(FA)); break;
    }
    SWG_KERNEL_CHECK(ctx);
  }
  // ---- chain_N bases
  if (n_runs <= (uint32_t)NUMBER_SMALL) {
    SWG_LAUNCH(ctx, "pair_number", pair_number_small_kernel<<<1, 1024, 0, st>>>(n_runs, info, sum, gl_first, r->seq_genome_last, gp2_first,
                                                                     r->seq_genome_two));
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
    SWG_LAUNCH(ctx, "pair_number", pair_key1_kernel<<<rb, EW, 0, st>>>(n_runs, info, sum, gl_first, r->seq_genome_last, key, val));
    SWG_KERNEL_CHECK(ctx);
    SWG_TRY(swg_radix_sort_pairs(ctx, &key, &val, &key_tmp, &val_tmp, n_runs, 0, 64));
    SWG_LAUNCH(ctx, "pair_number", pair_rank1_kernel<<<rb, EW, 0, st>>>(n_runs, val, info, sum, r->seq_genome_two, gp2_first, rank1));
    SWG_KERNEL_CHECK(ctx);
    SWG_LAUNCH(ctx, "pair_number", pair_key2_kernel<<<rb, EW, 0, st>>>(n_runs, info, sum, rank1, r->seq_genome_two, gp2_first, key, val));
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
    SWG_LAUNCH(ctx, "pair_renumber", pair_renumber_kernel<<<gb, EW, 0, st>>>(n_runs, runs, sum, chain_out));
    SWG_KERNEL_CHECK(ctx);
  }
  // ---- the flags found on the device, and the statistics
  uint64_t hc[9];
  static_assert(sizeof(PairCounters) == 72, "PairCounters is read back as nine words");
  SWG_TRY(swg_read_scalars(ctx, reinterpret_cast<const uint64_t*>(C), hc, 9));
  const uint32_t flags2 = (uint32_t)(hc[0] >> 32);
  if (flags2) {
    if (dbg) fprintf(stderr, "[swg] pair path: left on the device's word (flags %u): the global-sort path takes the call\n", flags2);
    swg_arena_restore(ctx, mark0);
    return SWG_OK;
  }
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
