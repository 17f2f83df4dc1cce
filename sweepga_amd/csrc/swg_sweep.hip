// Plane sweep (src/plane_sweep_exact.rs:268-433) for gfx950, one axis, all segments at once.
//
// Reference semantics (closed form): inside a segment, with priority
// (score desc, axis start asc, index asc), for every event coordinate x let A(x) be the
// intervals with s <= x < e and T(x) the first min(k,|A|) of A(x) by priority.  Then
//     keep_i = (exists x in [s_i,e_i): i in T(x))                               "ever top-k"
//              and not (thr < 1 and exists x: i in A(x)\T(x), some t in T(x), ovl(i,t) > thr)
// and a segment of size <= 1 is returned whole.  The `overlapped` flag is sticky
// (plane_sweep_exact.rs:251-254).  The sequential BTreeSet sweep is replaced by an evaluation
// that is independent per event coordinate:
//
//   1. every live interval gets the composite start key X = ((segment+1) << pos_bits) | start;
//      ONE radix sort of the n begins orders all segments of the axis (dead intervals get X = 0
//      and fall out in front).  Ends are never sorted.  (Round 4: the sort runs on X without its low
//      bits where that saves a pass; the gather behind it orders the short runs of equal truncated
//      keys in LDS -- begin_gather_words_kernel.)
//   2. the sorted begins are cut into tiles of TB = 256 (k = 1: 128, 256 or 512 begins, chosen on the device from an
//      estimate of the carry-in volume -- sparse, ordinary, deep data).  With X_b the first key of tile b:
//        - an interval that begins before tile b and ends after X_b is a *carry-in* of b;
//        - an interval's end coordinate E is an evaluation point of the last tile with X_b < E
//          (unless E coincides with the next tile's first key, which is evaluated as a start);
//          that tile holds the interval either as one of its own begins or as a carry-in, so
//          end points need no list of their own.
//      Carry-in lists are built once (exponential + binary search over the tile-start array,
//      count, scan, fill); entries carry {start, end, score key, index} inline.
//   3. one work-group per tile.  Evaluation points are processed in batches of 256 (one per
//      thread): the tile's own start coordinates, the ends of its own begins, then the ends of
//      its carry-ins chunk by chunk (only ends before the next tile's first key).  For a point x
//      the candidates are the tile's own begins at or before x (backward scan in LDS from the
//      last begin <= x, cut short by a prefix maximum of interval ends) and the carry-ins
//      (streamed through LDS in chunks).  Pass 1 finds T(x), pass 2 marks `ever-top` for its
//      members and `overlapped` for every other active interval whose overlap fraction with a
//      member exceeds thr (f64 division, as the reference).  k == 1 keeps T(x) in registers;
//      2 <= k < inf walks the priority order k times (successive minima) so any k works without
//      per-thread storage; k == inf needs no sweep at all.
//   4. flags are combined: keep = single-in-segment | (ever_top & !overlapped).
//
// Equal coordinates: the reference applies all events at one position before marking
// (plane_sweep_exact.rs:306-334), so a start coordinate is evaluated once, by the last begin of its
// run, and a run that continues into the next tile is left to that tile (whose carry-ins then
// contain every interval that began at that coordinate earlier).  Evaluating a coordinate twice
// (an end that equals some start) is harmless: marks are idempotent.
#include "swg_internal.h"
#include "swg_log.h"

namespace {

constexpr int TB = 256;  // begins per tile == threads per workgroup
constexpr int CC = 256;  // carry-in entries staged per LDS chunk
constexpr int EW_THREADS = 256;
constexpr uint64_t DEEP_CARRY_PER_TILE = 64;  // average carry-ins per 256-begin tile from which k = 1 uses 512-begin tiles
constexpr uint64_t SPARSE_CARRY_PER_TILE = 4;  // ... and below which it uses 128-begin tiles (two-wavefront work-groups: the
                                               // tile kernel 5.9 -> 5.5 ms on S-pan, routing +0.07; 13 vs 8.3 ms on the deep pair)
constexpr int TBF = 128;                       // the gathers publish tile-start keys at this granularity; coarser tilings take
                                               // every 2nd / 4th of them

__device__ __forceinline__ bool prio_less(uint64_t ak, uint64_t as, uint32_t ai, uint64_t bk, uint64_t bs,
                                          uint32_t bi) {
  if (ak != bk) return ak < bk;
  if (as != bs) return as < bs;
  return ai < bi;
}

// ---- score keys ---------------------------------------------------------------------------
// src/plane_sweep_exact.rs:29-86: length = q_end - q_start (query span on BOTH axes);
// -inf when length <= 0 or (for the identity-using scores) identity <= 0.
__device__ __forceinline__ uint64_t sortable_desc(double score) {
  if (score == 0.0) score = 0.0;  // -0 -> +0 (partial_cmp treats them equal)
  uint64_t b = (uint64_t)__double_as_longlong(score);
  if (score != score) return ~0ull;  // NaN: worst (the reference's order is not total here)
  uint64_t asc = (b >> 63) ? ~b : (b | 0x8000000000000000ull);
  return ~asc;
}

__device__ __forceinline__ uint64_t score_key_of(uint32_t qs, uint32_t qe, double id, int scoring) {
  const double NEG_INF = -__longlong_as_double(0x7ff0000000000000ll);
  const uint64_t len_u = (uint64_t)qe - (uint64_t)qs;  // u64 wrapping subtraction as in the reference
  const double length = (double)len_u;
  double score;
  switch (scoring) {
    case SWG_SCORE_IDENTITY: score = id <= 0.0 ? NEG_INF : id; break;
    case SWG_SCORE_LENGTH: score = length <= 0.0 ? NEG_INF : length; break;
    case SWG_SCORE_LENGTH_IDENTITY:
    case SWG_SCORE_MATCHES: score = (length <= 0.0 || id <= 0.0) ? NEG_INF : __dmul_rn(length, id); break;
    default: score = (length <= 0.0 || id <= 0.0) ? NEG_INF : __dmul_rn(id, swg_log_glibc(length)); break;
  }
  return sortable_desc(score);
}

__global__ __launch_bounds__(EW_THREADS) void score_key_kernel(uint64_t n, const uint32_t* __restrict__ qs,
                                                               const uint32_t* __restrict__ qe,
                                                               const double* __restrict__ identity, int scoring,
                                                               uint64_t* __restrict__ key) {
  uint64_t i = (uint64_t)blockIdx.x * EW_THREADS + threadIdx.x;
  if (i < n) key[i] = score_key_of(qs[i], qe[i], identity[i], scoring);
}

// Is the input grouped by sequence pair (as an aligner writes it)?  16,384 sampled records are compared with their successors:
// in pair-major order nearly all of them share the pair, in random order nearly none.  *flag = 1: not grouped -- the column
// gathers behind the scaffold stage's first sort then have no locality in L2 (S-pan shuffled: gather_all_words 13.5 ms
// instead of 3.4) and the record slots are worth writing.  The decision stays on the device: prepare and the gathers read it.
__global__ __launch_bounds__(256) void input_order_probe_kernel(uint64_t n, const uint32_t* __restrict__ q_id,
                                                                const uint32_t* __restrict__ t_id, uint32_t* __restrict__ flag) {
  __shared__ uint32_t same_w[4];
  uint32_t same = 0;
  constexpr uint32_t PER_THREAD = 64;
  if (n >= 2) {
    uint64_t x = 0x9e3779b97f4a7c15ull * (threadIdx.x + 1);
    for (uint32_t k = 0; k < PER_THREAD; ++k) {
      x = x * 6364136223846793005ull + 1442695040888963407ull;
      const uint64_t i = (x >> 11) % (n - 1);
      same += (q_id[i] == q_id[i + 1] && t_id[i] == t_id[i + 1]) ? 1u : 0u;
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) same += __shfl_down(same, o, 64);
  if ((threadIdx.x & 63) == 0) same_w[threadIdx.x >> 6] = same;
  __syncthreads();
  if (threadIdx.x == 0) *flag = (n >= 2 && 2 * (same_w[0] + same_w[1] + same_w[2] + same_w[3]) < 256 * PER_THREAD) ? 1u : 0u;
}

// retain + score key + max coordinate + retained count in one pass (grid-stride, one atomic pair per wave)
__global__ __launch_bounds__(EW_THREADS) void prepare_kernel(uint64_t n, const uint32_t* __restrict__ q_id,
                                                             const uint32_t* __restrict__ t_id,
                                                             const uint32_t* __restrict__ block_len,
                                                             const uint32_t* __restrict__ matches,
                                                             const double* __restrict__ identity,
                                                             const uint32_t* __restrict__ qs, const uint32_t* __restrict__ qe,
                                                             const uint32_t* __restrict__ ts, const uint32_t* __restrict__ te,
                                                             uint64_t min_block, int keep_self, double min_identity,
                                                             int scoring, uint8_t* __restrict__ alive,
                                                             swg_key_ends* __restrict__ key_ends, int slot_payload,
                                                             int with_keys, unsigned long long* __restrict__ scalars,
                                                             const uint8_t* __restrict__ strand, uint32_t n_seq,
                                                             uint32_t* __restrict__ group32,
                                                             const uint32_t* __restrict__ probe_flag,
                                                             uint64_t* __restrict__ score_out) {
  if (probe_flag && *probe_flag == 0) key_ends = nullptr;  // (wave-uniform) grouped input: the slots are not worth their traffic
  uint32_t mx = 0, cnt = 0, zero = 0;
  for (uint64_t i = (uint64_t)blockIdx.x * EW_THREADS + threadIdx.x; i < n; i += (uint64_t)gridDim.x * EW_THREADS) {
    // identity == nullptr: RecordMeta.identity as extract_metadata derives it without a dv:f: override -- matches over
    // max(block length, 1), one IEEE division (src/paf_filter.rs:322; the host then sends 8 bytes less per record)
    double id;
    if (identity) {
      id = identity[i];
    } else {
      const uint32_t bl = block_len[i];
      id = __ddiv_rn((double)matches[i], (double)(bl > 1u ? bl : 1u));
    }
    const uint32_t a = qs[i], b = qe[i], c = ts[i], d = te[i];
    const bool ok = (min_block == 0 || (uint64_t)block_len[i] >= min_block) && (keep_self || q_id[i] != t_id[i]) && id >= min_identity;
    alive[i] = ok ? 1 : 0;
    // the (query, target, strand) group as one 4-byte value: sort A behind a mapping sweep then gathers one column, not three
    if (group32) group32[i] = (q_id[i] * n_seq + t_id[i]) * 2u + (strand[i] ? 1u : 0u);
    if (score_out) score_out[i] = score_key_of(a, b, id, scoring);  // (the segment-resident sort reads plain columns: swg_segsort.hip)
    if (key_ends) {  // nullptr: neither a mapping-level sweep nor a scaffold stage will read the record slots
      swg_key_ends ke;
      ke.key = with_keys ? score_key_of(a, b, id, scoring) : 0ull;  // (no sweep: nobody reads the scores)
      ke.start[0] = a;
      ke.start[1] = c;
      ke.end[0] = b;
      ke.end[1] = d;
      // the slot's spare 8 bytes: matches and block length when a scaffold stage follows -- its gathers after sort A then take
      // everything they need of a record from this one sector
      ke.pad[0] = slot_payload ? matches[i] : 0u;
      ke.pad[1] = slot_payload ? block_len[i] : 0u;
      key_ends[i] = ke;
    }
    const uint32_t m1 = a > b ? a : b, m2 = c > d ? c : d;
    const uint32_t m = m1 > m2 ? m1 : m2;
    if (m > mx) mx = m;
    cnt += ok ? 1u : 0u;
    // retained records that are degenerate on an axis: zero length (an unlimited sweep drops them), and reversed (malformed
    // PAF, not modelled -- DESIGN.md section 4 -- but counted here with the SAME test kinf_mark uses, start >= end, so that
    // whether the identity shortcut is taken never depends on which other records share the input with such a record)
    zero += (ok && (a >= b || c >= d)) ? 1u : 0u;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const uint32_t t = __shfl_down(mx, o, 64);
    if (t > mx) mx = t;
    cnt += __shfl_down(cnt, o, 64);
    zero += __shfl_down(zero, o, 64);
  }
  // one set of atomics per work-group (the three scalars share a cache line: per wavefront these were 49,152 serialised
  // atomics at the end of a kernel that streams 4 GB)
  __shared__ uint32_t w_mx[EW_THREADS / 64], w_cnt[EW_THREADS / 64], w_zero[EW_THREADS / 64];
  if ((threadIdx.x & 63) == 0) {
    w_mx[threadIdx.x >> 6] = mx;
    w_cnt[threadIdx.x >> 6] = cnt;
    w_zero[threadIdx.x >> 6] = zero;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
#pragma unroll
    for (int w = 1; w < EW_THREADS / 64; ++w) {
      mx = w_mx[w] > mx ? w_mx[w] : mx;
      cnt += w_cnt[w];
      zero += w_zero[w];
    }
    atomicMax(&scalars[0], (unsigned long long)mx);
    atomicAdd(&scalars[1], (unsigned long long)cnt);
    if (zero) atomicAdd(&scalars[2], (unsigned long long)zero);
  }
}

// ---- begins ---------------------------------------------------------------------------------
__global__ __launch_bounds__(EW_THREADS) void begin_build_kernel(uint64_t n, const uint64_t* __restrict__ seg,
                                                                 const uint32_t* __restrict__ seg_a,
                                                                 const uint32_t* __restrict__ seg_b,
                                                                 const uint32_t* __restrict__ seg_table, uint32_t seg_mul,
                                                                 const uint32_t* __restrict__ start,
                                                                 const uint8_t* __restrict__ alive, int pos_bits,
                                                                 uint64_t* __restrict__ key,
                                                                 uint32_t* __restrict__ val, swg_radix_plan plan,
                                                                 uint32_t* __restrict__ ghist, int drop = 0, int idx_bits = 0) {
  // drop > 0: sort on the truncated key -- key[i] becomes the WORD ((X >> drop) << idx_bits) | i (swg_radix_sort_words) and
  // the histograms are those of X >> drop.
  // grid-stride over whole work-groups (the trip count is block-uniform); with `ghist` the digit histograms of the sort
  // that follows are accumulated here, while the key is in a register (the sort then skips its own pass over the keys)
  __shared__ uint32_t h[SWG_RADIX_MAX_PASSES][SWG_RADIX_BINS];
  const int npasses = plan.npasses;
  if (ghist) {
    swg_radix_hist_zero(h, npasses);
    __syncthreads();
  }
  for (uint64_t base = (uint64_t)blockIdx.x * EW_THREADS; base < n; base += (uint64_t)gridDim.x * EW_THREADS) {
    const uint64_t i = base + threadIdx.x;
    const bool in = i < n;
    uint64_t k = 0;
    if (in) {
      const bool live = alive ? alive[i] != 0 : true;
      if (live) {
        uint64_t sg;
        if (seg) {
          sg = seg[i];
        } else {
          const uint32_t b = seg_b[i];
          sg = (uint64_t)seg_a[i] * seg_mul + (seg_table ? seg_table[b] : b);
        }
        k = ((sg + 1) << pos_bits) | start[i];
      }
      if (drop) {
        k >>= drop;
        key[i] = (k << idx_bits) | i;
      } else {
        key[i] = k;
        if (val) val[i] = (uint32_t)i;  // (nullptr: the packed sort takes the identity as read)
      }
    }
    if (ghist) swg_radix_hist_add(h, k, in, plan);
  }
  if (ghist) {
    __syncthreads();
    swg_radix_hist_flush(h, npasses, ghist);
  }
}

// After the sort: pull each begin's end coordinate and score key next to it (so the tile kernel reads
// only coalesced streams), publish tile-start keys, and flag segments with exactly one live interval
// (returned whole by the reference, plane_sweep_exact.rs:274-276, zero-length or not).
__global__ __launch_bounds__(EW_THREADS) void begin_gather_kernel(uint64_t n, const uint64_t* __restrict__ S,
                                                                  const uint32_t* __restrict__ I,
                                                                  const uint32_t* __restrict__ end,
                                                                  const uint64_t* __restrict__ score_key,
                                                                  const swg_key_ends* __restrict__ packed, int packed_end,
                                                                  int pos_bits, uint32_t* __restrict__ E,
                                                                  uint64_t* __restrict__ KEY,
                                                                  uint64_t* __restrict__ tile_x,
                                                                  uint8_t* __restrict__ single) {
  // E holds the END COORDINATE only (4 bytes): the composite end is the begin's segment part | end (swg_comp_end)
  uint64_t p = (uint64_t)swg_xcd_block(blockIdx.x, gridDim.x) * EW_THREADS + threadIdx.x;
  if (p >= n) return;
  const uint64_t s = S[p];
  uint32_t e = 0;
  uint64_t k = 0;
  if (s != 0) {
    const uint32_t id = I[p];
    if (packed) {
      const swg_key_ends ke = packed[id];
      e = ke.end[packed_end];
      k = ke.key;
    } else {
      e = end[id];
      k = score_key ? score_key[id] : 0;  // no scores: the k = inf path only wants the `single` flags
    }
    const uint64_t sg = s >> pos_bits;
    const bool prev_same = p > 0 && (S[p - 1] >> pos_bits) == sg;
    const bool next_same = p + 1 < n && (S[p + 1] >> pos_bits) == sg;
    if (!prev_same && !next_same) single[id] = 1;
  }
  E[p] = e;
  KEY[p] = k;
  if ((p % TBF) == 0) tile_x[p / TBF] = s;
}

// The same after the packed sort (swg_radix_sort_packed): P[p] = ((X >> 8) << idx_bits) | record index.  The begin's full
// key X (its low 8 bits are the low 8 bits of the start: pos_bits >= 8 here) and the index are written out for the routing
// and tile kernels; start, end and score key come from the record's 32-byte slot, one sector.
__global__ __launch_bounds__(EW_THREADS) void begin_gather_packed_kernel(uint64_t n, const uint64_t* __restrict__ P, int idx_bits,
                                                                         const swg_key_ends* __restrict__ packed, int axis,
                                                                         int pos_bits, uint64_t* __restrict__ S,
                                                                         uint32_t* __restrict__ I, uint32_t* __restrict__ E,
                                                                         uint64_t* __restrict__ KEY,
                                                                         uint64_t* __restrict__ tile_x,
                                                                         uint8_t* __restrict__ single) {
  uint64_t p = (uint64_t)swg_xcd_block(blockIdx.x, gridDim.x) * EW_THREADS + threadIdx.x;
  if (p >= n) return;
  const uint64_t w = P[p];
  const uint64_t hi = w >> idx_bits;  // X >> 8; 0 = a dead record (live keys are >= 2^pos_bits >= 256)
  const uint32_t id = (uint32_t)(w & ((uint64_t(1) << idx_bits) - 1));
  uint64_t s = 0, k = 0;
  uint32_t e = 0;
  if (hi != 0) {
    const swg_key_ends ke = packed[id];
    s = (hi << 8) | (ke.start[axis] & 0xffu);
    e = ke.end[axis];
    k = ke.key;
    const uint64_t sg = hi >> (pos_bits - 8);
    const bool prev_same = p > 0 && ((P[p - 1] >> idx_bits) >> (pos_bits - 8)) == sg;
    const bool next_same = p + 1 < n && ((P[p + 1] >> idx_bits) >> (pos_bits - 8)) == sg;
    if (!prev_same && !next_same) single[id] = 1;
  }
  S[p] = s;
  I[p] = id;
  E[p] = e;
  KEY[p] = k;
  if ((p % TBF) == 0) tile_x[p / TBF] = s;
}

// The same after a sort on the TRUNCATED key (swg_radix_sort_words): P[p] = ((X >> drop) << idx_bits) | record index, in
// (X >> drop, index) order.  Begins whose starts differ only in the low `drop` bits form short runs; each member finds the
// other members of its run in LDS (the work-group's 256 words plus SWG_RUN_HALO on either side, with the low bits of their
// starts from the record slots), counts those that order before it by (low bits, index) and writes its begin to
// run start + that rank: the full (X, index) order, one radix pass cheaper.  A run that reaches beyond the halo cannot be
// ordered here: *long_run is raised (the positions written are then meaningless but in range) and the caller sorts again,
// the ordinary way.
__global__ __launch_bounds__(EW_THREADS) void begin_gather_words_kernel(uint64_t n, const uint64_t* __restrict__ P, int idx_bits,
                                                                        int drop, const swg_key_ends* __restrict__ packed,
                                                                        int axis, int pos_bits, uint64_t* __restrict__ S,
                                                                        uint32_t* __restrict__ I, uint32_t* __restrict__ E,
                                                                        uint64_t* __restrict__ KEY,
                                                                        uint64_t* __restrict__ tile_x,
                                                                        uint8_t* __restrict__ single,
                                                                        uint32_t* __restrict__ long_run) {
  constexpr int H = SWG_RUN_HALO, W = EW_THREADS + 2 * H;
  __shared__ uint64_t l_hi[W];   // X >> drop (0: dead record, or no element at this position)
  __shared__ uint64_t l_ord[W];  // (low bits of the start << 32) | record index: the order inside a run
  const uint64_t p0 = (uint64_t)swg_xcd_block(blockIdx.x, gridDim.x) * EW_THREADS;
  const int t = threadIdx.x;
  const uint64_t idx_mask = (uint64_t(1) << idx_bits) - 1;
  const uint32_t low_mask = (1u << drop) - 1u;
  // own element
  const uint64_t p = p0 + t;
  uint64_t w = 0, hi = 0;
  uint32_t id = 0, e = 0, low = 0;
  uint64_t k = 0;
  if (p < n) {
    w = P[p];
    hi = w >> idx_bits;
    id = (uint32_t)(w & idx_mask);
    if (hi != 0) {
      const swg_key_ends ke = packed[id];
      low = ke.start[axis] & low_mask;
      e = ke.end[axis];
      k = ke.key;
    }
  }
  l_hi[H + t] = hi;
  l_ord[H + t] = ((uint64_t)low << 32) | id;
  // halo words (their record slots are read below, and only for the elements of a run that crosses the block's edge)
  if (t < 2 * H) {
    const bool left = t < H;
    const int64_t q = left ? (int64_t)p0 - H + t : (int64_t)p0 + EW_THREADS + (t - H);
    const int li = left ? t : EW_THREADS + t;  // H + EW_THREADS + (t - H)
    l_hi[li] = (q >= 0 && (uint64_t)q < n) ? (P[q] >> idx_bits) : 0ull;
    l_ord[li] = (q >= 0 && (uint64_t)q < n) ? (P[q] & idx_mask) : 0ull;  // (index only so far)
  }
  __syncthreads();
  if (t < 2 * H) {
    const bool left = t < H;
    const int li = left ? t : EW_THREADS + t;
    const uint64_t edge = left ? l_hi[H] : l_hi[H + EW_THREADS - 1];
    if (edge != 0 && l_hi[li] == edge) {
      const uint32_t hid = (uint32_t)l_ord[li];
      l_ord[li] = ((uint64_t)(packed[hid].start[axis] & low_mask) << 32) | hid;
    }
  }
  __syncthreads();
  if (p >= n) return;
  uint64_t np = p;
  if (hi != 0) {
    const uint64_t mine = l_ord[H + t];
    uint32_t before = 0, rank = 0;
    int j = H + t - 1;
    for (; j >= 0 && l_hi[j] == hi; --j) {
      ++before;
      rank += l_ord[j] < mine ? 1u : 0u;
    }
    bool too_long = j < 0 && p0 > (uint64_t)H;  // ran off the left halo with elements still before it
    int j2 = H + t + 1;
    for (; j2 < W && l_hi[j2] == hi; ++j2) rank += l_ord[j2] < mine ? 1u : 0u;
    if (j2 >= W && p0 + EW_THREADS + H < n) too_long = true;  // ran off the right halo with elements still after it
    if (too_long) {
      *long_run = 1u;
    } else {
      np = p - before + rank;
    }
  }
  const uint64_t s = hi ? ((hi << drop) | low) : 0ull;
  S[np] = s;
  I[np] = id;
  E[np] = e;
  KEY[np] = k;
  if ((np % TBF) == 0) tile_x[np / TBF] = s;
  if (hi != 0) {
    const uint64_t sg = hi >> (pos_bits - drop);
    const bool prev_same = p > 0 && (l_hi[H + t - 1] >> (pos_bits - drop)) == sg;
    const bool next_same = p + 1 < n && (l_hi[H + t + 1] >> (pos_bits - drop)) == sg;
    if (!prev_same && !next_same) single[id] = 1;
  }
}

// composite end of a begin with composite start s (0 = dead) and end coordinate e
__device__ __forceinline__ uint64_t swg_comp_end(uint64_t s, uint32_t e, int pos_bits) {
  return s ? ((s >> pos_bits) << pos_bits) | e : 0ull;
}

// first keys of tiles of `tile` begins, straight from the sorted begins (SWG_TILE_SMALL experiment)
__global__ __launch_bounds__(EW_THREADS) void tile_x_stride_kernel(uint32_t nt, const uint64_t* __restrict__ S, uint32_t tile,
                                                                   uint64_t* __restrict__ tile_x) {
  const uint32_t b = blockIdx.x * EW_THREADS + threadIdx.x;
  if (b < nt) tile_x[b] = S[(uint64_t)b * tile];
}
// first keys of the 512-begin tiles = every second one of the 256-begin tiles
__global__ __launch_bounds__(EW_THREADS) void tile_x_pairs_kernel(uint32_t ntiles2, const uint64_t* __restrict__ tile_x,
                                                                  uint64_t* __restrict__ tile_x2, uint32_t every = 2) {
  const uint32_t b = blockIdx.x * EW_THREADS + threadIdx.x;
  if (b < ntiles2) tile_x2[b] = tile_x[(size_t)every * b];
}

// ---- routing: carry-ins and end points ----------------------------------------------------------
// te = last tile b with X_b < E (>= the interval's own tile).  Exponential then binary search.
__device__ __forceinline__ uint32_t last_tile_below(const uint64_t* __restrict__ tile_x, uint32_t ntiles, uint32_t tb,
                                                    uint64_t e) {
  uint32_t lo = tb;  // invariant: tile_x[lo] < e
  uint32_t hi = tb + 1, step = 1;
  while (hi < ntiles && tile_x[hi] < e) {
    lo = hi;
    step <<= 1;
    hi = (uint64_t)lo + step < ntiles ? lo + step : ntiles;
  }
  // first tile in (lo, hi) with tile_x >= e; everything in (lo, hi) is unknown, tile_x[hi] >= e or hi == ntiles
  uint32_t l = lo + 1, r = hi;
  while (l < r) {
    const uint32_t mid = l + ((r - l) >> 1);
    if (tile_x[mid] < e)
      l = mid + 1;
    else
      r = mid;
  }
  return l - 1;
}

// mode 0: 256-begin tiles; 1: 512-begin tiles; 2: chosen here from the estimate (route_estimate_kernel) -- the host learns the
// choice from the same read-back that tells it the carry-in total
__global__ __launch_bounds__(EW_THREADS) void route_count_kernel(uint64_t n, const uint64_t* __restrict__ S,
                                                                 const uint32_t* __restrict__ E, int pos_bits,
                                                                 const uint64_t* __restrict__ tile_x, uint32_t ntiles,
                                                                 const uint64_t* __restrict__ tile_x2, uint32_t ntiles2, int mode,
                                                                 const unsigned long long* __restrict__ reach, uint32_t stride,
                                                                 uint32_t* __restrict__ te_out,
                                                                 uint32_t* __restrict__ carry_cnt, uint32_t small_tile = 0,
                                                                 const uint64_t* __restrict__ tile_xf = nullptr, uint32_t ntilesf = 0) {
  // four consecutive begins per thread (one tile: tile sizes are multiples of four): 16-byte loads and one 16-byte store
  const uint64_t p = ((uint64_t)blockIdx.x * EW_THREADS + threadIdx.x) * 4;
  if (p >= n) return;
  // mode 3 (small_tile != 0): tiles of `small_tile` begins, tile_x2 / ntiles2 describe them (knob).  mode 2: 512-begin tiles
  // (tile_x2) for deep data, 128-begin tiles (tile_xf) for sparse data, 256 otherwise -- by the estimate
  // (avg = reach * stride / ntiles, compared without the 64-bit division; every tile size is a power of two: shifts)
  const uint64_t rs = mode == 2 ? *reach * stride : 0;
  const bool wide = mode == 1 || mode == 3 || (mode == 2 && rs >= (uint64_t)DEEP_CARRY_PER_TILE * ntiles);
  const bool fine = mode == 2 && rs < (uint64_t)SPARSE_CARRY_PER_TILE * ntiles;
  const uint64_t* tx = fine ? tile_xf : wide ? tile_x2 : tile_x;
  const uint32_t nt = fine ? ntilesf : wide ? ntiles2 : ntiles;
  const uint32_t tb = (uint32_t)(p >> (31 - __clz((int)(mode == 3 ? small_tile : fine ? (uint32_t)TBF : (wide ? 2 * TB : TB)))));
  uint64_t sv[4];
  uint32_t ev[4], tev[4];
  const bool whole = p + 4 <= n;
  if (whole) {
    const ulonglong2 s01 = *reinterpret_cast<const ulonglong2*>(S + p), s23 = *reinterpret_cast<const ulonglong2*>(S + p + 2);
    const uint4 e4 = *reinterpret_cast<const uint4*>(E + p);
    sv[0] = s01.x; sv[1] = s01.y; sv[2] = s23.x; sv[3] = s23.y;
    ev[0] = e4.x; ev[1] = e4.y; ev[2] = e4.z; ev[3] = e4.w;
  } else {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      sv[u] = p + u < n ? S[p + u] : 0ull;
      ev[u] = p + u < n ? E[p + u] : 0u;
    }
  }
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const uint64_t s = sv[u], e = swg_comp_end(s, ev[u], pos_bits);
    uint32_t te = tb;
    if (s != 0 && e > s) {  // live and not zero-length
      te = last_tile_below(tx, nt, tb, e);
      for (uint32_t b = tb + 1; b <= te; ++b) atomicAdd(&carry_cnt[b], 1u);
    }
    tev[u] = te;
  }
  if (whole) {
    *reinterpret_cast<uint4*>(te_out + p) = make_uint4(tev[0], tev[1], tev[2], tev[3]);
  } else {
#pragma unroll
    for (int u = 0; u < 4; ++u)
      if (p + u < n) te_out[p + u] = tev[u];
  }
}

// Estimate of the carry-in volume (sum over intervals of the tiles they reach into beyond their own) from every
// `stride`-th begin: searches only, no per-tile counting.  Chooses the tile size before the real routing pass.
__global__ __launch_bounds__(EW_THREADS) void route_estimate_kernel(uint64_t n, uint32_t stride, const uint64_t* __restrict__ S,
                                                                    const uint32_t* __restrict__ E, int pos_bits,
                                                                    const uint64_t* __restrict__ tile_x, uint32_t ntiles,
                                                                    unsigned long long* __restrict__ out) {
  const uint64_t p = ((uint64_t)blockIdx.x * EW_THREADS + threadIdx.x) * stride;
  uint32_t reach = 0;
  if (p < n) {
    const uint64_t s = S[p], e = swg_comp_end(s, E[p], pos_bits);
    if (s != 0 && e > s) {
      const uint32_t tb = (uint32_t)(p / TB);
      reach = last_tile_below(tile_x, ntiles, tb, e) - tb;
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) reach += __shfl_down(reach, o, 64);
  if ((threadIdx.x & 63) == 0 && reach) atomicAdd(out, (unsigned long long)reach);
}

__global__ __launch_bounds__(EW_THREADS) void route_fill_kernel(
    uint64_t n, const uint64_t* __restrict__ S, const uint32_t* __restrict__ E, int pos_bits, const uint64_t* __restrict__ KEY,
    const uint32_t* __restrict__ I, uint32_t tile_size, const uint32_t* __restrict__ te_in, const uint32_t* __restrict__ carry_off,
    uint32_t* __restrict__ carry_cur, uint64_t* __restrict__ c_s, uint64_t* __restrict__ c_e,
    uint64_t* __restrict__ c_key, uint32_t* __restrict__ c_id) {
  uint64_t p = (uint64_t)blockIdx.x * EW_THREADS + threadIdx.x;
  if (p >= n) return;
  const uint32_t te = te_in[p];
  const uint32_t tb = (uint32_t)(p >> (31 - __clz((int)tile_size)));  // (a power of two)
  if (te <= tb) return;
  const uint64_t s = S[p], e = swg_comp_end(s, E[p], pos_bits), k = KEY[p];
  const uint32_t id = I[p];
  for (uint32_t b = tb + 1; b <= te; ++b) {
    const uint32_t slot = carry_off[b] + atomicAdd(&carry_cur[b], 1u);
    c_s[slot] = s;
    c_e[slot] = e;
    c_key[slot] = k;
    c_id[slot] = id;
  }
}

// ---- the tile kernel ----------------------------------------------------------------------------
struct TileArgs {
  uint64_t n;
  const uint64_t* S;    // sorted begin keys
  const uint32_t* E;    // end coordinates (composite end = swg_comp_end)
  int pos_bits;
  const uint64_t* KEY;  // score keys
  const uint32_t* I;    // interval ids
  const uint64_t* tile_x;
  uint32_t ntiles;
  const uint32_t* carry_off;  // [ntiles + 1]
  const uint64_t* c_s;
  const uint64_t* c_e;
  const uint64_t* c_key;
  const uint32_t* c_id;
  uint64_t k;
  double thr;
  uint8_t* top;
  uint8_t* ovl;
  uint8_t* tile_done;  // [ntiles] 2 <= k < inf: written by the pruned kernel (1 = tile finished there), read by the plain one
};

__device__ __forceinline__ bool overlap_exceeds(uint64_t as, uint64_t ae, uint64_t bs, uint64_t be, double thr) {
  // query_overlap / target_overlap, plane_sweep_exact.rs:113-144 (composite coords share the segment part)
  const uint64_t os = as > bs ? as : bs;
  const uint64_t oe = ae < be ? ae : be;
  const double ol = oe > os ? (double)(oe - os) : 0.0;
  const uint64_t la = ae - as, lb = be - bs;
  const double ml = (double)(la < lb ? la : lb);
  if (!(ml > 0.0)) return false;
  // (a multiply-and-compare pre-filter around the division was measured: slower than the division it saves)
  return __ddiv_rn(ol, ml) > thr;
}

// ---- k == 1 -----------------------------------------------------------------------------------------------------
// With T(x) the best active interval at x:
//   * T changes at an event coordinate x only if T(x) begins at x, or if the previous top ends at x;
//   * an active i != T(x) has been tested against this T already unless T changed at x or i begins at x
//     (induction over the event coordinates), so the O(active) pass over everything runs only at points where the top
//     changes (and at the tile's first coordinate); elsewhere only the begins AT x are tested; and an END coordinate
//     needs no work at all unless the interval that ends there was the top just before it;
//   * inside a tile, let S* be the best carry-in that spans the tile's whole coordinate range: it is active at every point
//     of the tile, so an interval that ranks below it is never T there.  Candidates for T are S* and the intervals that
//     rank above it, and only THEIR ends can change T, so the other intervals' ends are not evaluation points at all.
// On deep data (S-big1: ~165 active intervals everywhere) a tile keeps a handful of candidates instead of ~420
// intervals x ~600 points x 2 passes; on sparse data there is no spanning interval, every interval is a candidate and
// the kernel degenerates to the plain per-point evaluation with the cheaper second pass.
// tools/model_sweep_k1.py is the executable model of this kernel (checked against the oracle).
constexpr int CCAP = 128;     // candidate carry-ins kept in LDS (17 KB per work-group in all: 8 resident per CU)
constexpr int STAR_MIN = 32;  // fewer carry-ins than this: no pruning (nothing to gain on sparse data)

#ifdef SWG_TILE_TIMING  // (a build knob, SWG_DEFINES of sweepga_amd/build.py) the phases of sweep_tile_k1 in 100 MHz ticks, summed over tiles
__device__ unsigned long long g_tile_t[8];
#define TT_STAMP(k) do { __syncthreads(); if (threadIdx.x == 0) { const unsigned long long t_ = wall_clock64(); atomicAdd(&g_tile_t[k], t_ - tt_last); tt_last = t_; } } while (0)
#else
#define TT_STAMP(k) do { } while (0)
#endif
template <int TBT>
__global__ __launch_bounds__(TBT) void sweep_tile_k1_kernel(TileArgs a) {
  __shared__ uint64_t sx[TBT];    // composite start of begin q
  __shared__ uint64_t se2[2 * TBT];   // [0, TBT): its composite end; [TBT, 2 TBT): the same for candidates, 0 for the others
  __shared__ uint64_t spm2[2 * TBT];  // prefix maxima of the two halves of se2
  __shared__ uint64_t skey[TBT];      // its score key
  __shared__ uint32_t sid[TBT];   // its interval index
  __shared__ uint64_t wmax[TBT / 64];
  constexpr int CC = CCAP < TBT ? CCAP : TBT;  // candidate carry-ins kept in LDS: one per thread at most (small tiles)
  __shared__ uint64_t ls[CC], le[CC], lkey[CC];  // candidate carry-ins
  __shared__ uint32_t lid[CC];
  __shared__ uint32_t l_count;
  __shared__ uint64_t r_k[TBT / 64], r_s[TBT / 64], r_e[TBT / 64];
  __shared__ uint32_t r_i[TBT / 64], r_have[TBT / 64];

  const uint32_t tile_id = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
#ifdef SWG_TILE_TIMING
  unsigned long long tt_last = wall_clock64();
#endif
  const uint64_t p = (uint64_t)tile_id * TBT + tid;
  const bool valid = p < a.n;
  uint64_t X = ~0ull, EE = 0, KEY = 0;
  uint32_t ID = 0;
  if (valid) {
    X = a.S[p];
    EE = swg_comp_end(X, a.E[p], a.pos_bits);
    KEY = a.KEY[p];
    ID = a.I[p];
  }
  sx[tid] = X;
  se2[tid] = EE;
  skey[tid] = KEY;
  sid[tid] = ID;
  if (tid == 0) l_count = 0;
  auto block_prefix_max = [&](uint64_t v, uint64_t* dst) {  // dst[tid] = max(v of threads 0..tid); two barriers
    uint64_t m = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const uint64_t t = __shfl_up(m, d, 64);
      if (lane >= d && t > m) m = t;
    }
    if (lane == 63) wmax[wave] = m;
    __syncthreads();
#pragma unroll
    for (int w = 0; w < TBT / 64; ++w)
      if (w < wave && wmax[w] > m) m = wmax[w];
    dst[tid] = m;
    __syncthreads();
  };
  block_prefix_max(EE, spm2);
  const uint64_t x_b = sx[0];
  const uint64_t x_next = (tile_id + 1 < a.ntiles) ? a.tile_x[tile_id + 1] : ~0ull;
  const uint32_t c_begin = a.carry_off[tile_id], c_end = a.carry_off[tile_id + 1];

  // ---- S*: the best carry-in that is active over the whole range of the tile, and the candidate carry-ins
  bool have_star = false;
  uint64_t star_k = 0, star_s = 0, star_e = 0;
  uint32_t star_i = 0;
  const uint32_t n_carry = c_end - c_begin;
  uint32_t n_cc = 0;
  bool cc_in_lds = true;
  // wave-level then block-level arg-min of (key, start, index) over the lanes with hv set; every thread gets the result
  auto reduce_star = [&](bool hv, uint64_t bk, uint64_t bs, uint64_t be, uint32_t bi) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const uint64_t ok = __shfl_down(bk, o, 64), os = __shfl_down(bs, o, 64), oe = __shfl_down(be, o, 64);
      const uint32_t oi = __shfl_down(bi, o, 64);
      const bool oh = __shfl_down((int)hv, o, 64) != 0;
      if (oh && (!hv || prio_less(ok, os, oi, bk, bs, bi))) {
        bk = ok;
        bs = os;
        be = oe;
        bi = oi;
        hv = true;
      }
    }
    if (lane == 0) {
      r_k[wave] = bk;
      r_s[wave] = bs;
      r_e[wave] = be;
      r_i[wave] = bi;
      r_have[wave] = hv ? 1u : 0u;
    }
    __syncthreads();
#pragma unroll
    for (int w = 0; w < TBT / 64; ++w)
      if (r_have[w] && (!have_star || prio_less(r_k[w], r_s[w], r_i[w], star_k, star_s, star_i))) {
        star_k = r_k[w];
        star_s = r_s[w];
        star_e = r_e[w];
        star_i = r_i[w];
        have_star = true;
      }
  };
  if (n_carry != 0 && n_carry <= (uint32_t)CC) {  // block-uniform.  The usual case: the whole list goes to LDS as it is
    const bool mine = (uint32_t)tid < n_carry;
    uint64_t cs_ = 0, ce_ = 0, ck_ = 0;
    uint32_t ci_ = 0;
    if (mine) {
      cs_ = a.c_s[c_begin + tid];
      ce_ = a.c_e[c_begin + tid];
      ck_ = a.c_key[c_begin + tid];
      ci_ = a.c_id[c_begin + tid];
    }
    if (n_carry >= STAR_MIN) {  // pruning only pays on deep data
      reduce_star(mine && ce_ >= x_next, ck_, cs_, ce_, ci_);
      // entries S* dominates leave the list (end 0 = never active); S* itself stays in registers
      if (have_star && !(ce_ < x_next && prio_less(ck_, cs_, ci_, star_k, star_s, star_i))) ce_ = 0;
    }
    if (mine) {
      ls[tid] = cs_;
      le[tid] = ce_;
      lkey[tid] = ck_;
      lid[tid] = ci_;
    }
    n_cc = n_carry;
  } else if (n_carry != 0) {  // long list: two passes over global memory
    bool hv = false;
    uint64_t bk = 0, bs = 0, be = 0;
    uint32_t bi = 0;
    for (uint32_t c = c_begin + tid; c < c_end; c += TBT) {
      const uint64_t e = a.c_e[c];
      if (e >= x_next) {
        const uint64_t k = a.c_key[c], s = a.c_s[c];
        const uint32_t id = a.c_id[c];
        if (!hv || prio_less(k, s, id, bk, bs, bi)) {
          bk = k;
          bs = s;
          be = e;
          bi = id;
          hv = true;
        }
      }
    }
    reduce_star(hv, bk, bs, be, bi);
    // candidates: end inside the tile's range and (no S* or better than S*); without S* every carry-in ends inside
    for (uint32_t c = c_begin + tid; c < c_end; c += TBT) {
      const uint64_t e = a.c_e[c];
      if (e < x_next) {
        const uint64_t k = a.c_key[c], s = a.c_s[c];
        const uint32_t id = a.c_id[c];
        if (!have_star || prio_less(k, s, id, star_k, star_s, star_i)) {
          const uint32_t slot = atomicAdd(&l_count, 1u);
          if (slot < (uint32_t)CC) {
            ls[slot] = s;
            le[slot] = e;
            lkey[slot] = k;
            lid[slot] = id;
          }
        }
      }
    }
  }
  // candidates among the tile's own begins
  const uint64_t* se = se2;
  const uint64_t* spm = spm2;
  int o1 = 0;  // offset of the candidate view in se2 / spm2 (no S*: every begin is a candidate)
  if (have_star) {  // block-uniform
    const bool cand = valid && X != 0 && prio_less(KEY, X, ID, star_k, star_s, star_i);
    se2[TBT + tid] = cand ? EE : 0;
    block_prefix_max(cand ? EE : 0, spm2 + TBT);
    o1 = TBT;
  } else if (n_carry != 0) {
    __syncthreads();
  }
  if (n_carry > (uint32_t)CC) {
    n_cc = l_count;
    cc_in_lds = n_cc <= (uint32_t)CC;  // else: every carry-in is scanned from global memory (a superset is harmless)
  }
  const bool cc_complete = cc_in_lds && !have_star;  // the LDS list holds every carry-in of the tile
  const uint32_t n_batches = 2 + (cc_in_lds ? (n_cc ? 1u : 0u) : (c_end - c_begin + TBT - 1) / TBT);
  const bool pass2 = a.thr < 1.0;
  TT_STAMP(0);  // loads, prefix maxima, carry-ins

  for (uint32_t batch = 0; batch < n_batches; ++batch) {
    if (batch) TT_STAMP(batch < 3 ? batch : 3);  // 1: the start points, 2: the tile's own ends, 3: the carry-ins' ends
    // ---- this thread's evaluation point: coordinate PX, last own begin at or before it Q0
    bool eval;
    uint64_t PX;
    int Q0;
    uint64_t PK = KEY, PS = X;  // priority of the interval whose end is the point (batches >= 1)
    uint32_t PI = ID;
    if (batch == 0) {  // start coordinates: the last begin of each run, unless the run continues in the next tile
      eval = valid && X != 0 && (tid == TBT - 1 || sx[tid + 1] != X) && X != x_next;
      PX = X;
      Q0 = tid;
    } else {
      // end coordinates of candidates that fall inside this tile's range (X_b < PX < X_{b+1}); an end equal to the next
      // tile's first key is evaluated there as a start coordinate
      if (batch == 1) {
        eval = valid && X != 0 && EE > X && EE < x_next && se2[o1 + tid] != 0;
        PX = EE;
      } else if (cc_in_lds) {
        PX = (uint32_t)tid < n_cc ? le[tid] : 0;
        eval = PX != 0 && PX < x_next;
        if (eval) {
          PK = lkey[tid];
          PS = ls[tid];
          PI = lid[tid];
        }
      } else {
        const uint32_t ci = c_begin + (batch - 2) * TBT + tid;
        eval = ci < c_end;
        PX = eval ? a.c_e[ci] : 0;
        eval = eval && PX < x_next;
        if (eval) {
          PK = a.c_key[ci];
          PS = a.c_s[ci];
          PI = a.c_id[ci];
        }
      }
      int l = 0, r = TBT;  // upper_bound(sx, PX) - 1; sx is ~0 past the end of a short last tile
      while (l < r) {
        const int mid = (l + r) >> 1;
        if (sx[mid] <= PX)
          l = mid + 1;
        else
          r = mid;
      }
      Q0 = l - 1;
    }
    if (!eval) continue;  // no barrier below this line

    // ---- an end coordinate changes nothing unless the interval that ends here was the top just before x, i.e. unless no
    // candidate j with s_j < x <= e_j ranks above it (S* ranks below every candidate; other intervals that end or begin
    // at x have their own threads)
    if (batch != 0) {
      bool was_top = true;
      for (int q = Q0; q >= 0 && was_top; --q) {
        if (spm2[o1 + q] < PX) break;  // no earlier candidate reaches PX
        if (se2[o1 + q] >= PX) {
          const uint64_t sq = sx[q];
          if (sq < PX && prio_less(skey[q], sq, sid[q], PK, PS, PI)) was_top = false;
        }
      }
      if (cc_in_lds) {
        for (uint32_t c = 0; c < n_cc && was_top; ++c)
          if (le[c] >= PX && prio_less(lkey[c], ls[c], lid[c], PK, PS, PI)) was_top = false;
      } else {
        for (uint32_t c = c_begin; c < c_end && was_top; ++c)
          if (a.c_e[c] >= PX && prio_less(a.c_key[c], a.c_s[c], a.c_id[c], PK, PS, PI)) was_top = false;
      }
      if (!was_top) continue;
    }
    // ---- pass 1: T(x) = best of {s <= x < e} over S* and the candidates
    uint64_t tk = star_k, ts = star_s, te = star_e;
    uint32_t ti = star_i;
    bool have_t = have_star;
    auto take = [&](uint64_t s, uint64_t e, uint64_t key, uint32_t id) {  // caller: s <= PX < e
      if (!have_t || prio_less(key, s, id, tk, ts, ti)) {
        tk = key;
        ts = s;
        te = e;
        ti = id;
        have_t = true;
      }
    };
    for (int q = Q0; q >= 0; --q) {
      if (spm2[o1 + q] <= PX) break;  // no earlier candidate covers PX
      const uint64_t ee = se2[o1 + q];
      if (ee > PX) take(sx[q], ee, skey[q], sid[q]);
    }
    if (cc_in_lds) {
      for (uint32_t c = 0; c < n_cc; ++c) {
        const uint64_t e = le[c];
        if (e > PX) take(ls[c], e, lkey[c], lid[c]);
      }
    } else {
      for (uint32_t c = c_begin; c < c_end; ++c) {
        const uint64_t e = a.c_e[c];
        if (e > PX) take(a.c_s[c], e, a.c_key[c], a.c_id[c]);
      }
    }
    if (!have_t) continue;
    a.top[ti] = 1;
    if (!pass2) continue;
    // ---- pass 2: active intervals that overlap T(x) too much.  The top is new at x when the old one ended here (an end
    // point whose interval was the top) or when T(x) itself begins at x; otherwise every interval that was active before x
    // has met this top already and only the begins AT x are tested.  The tile's first coordinate always takes the full
    // pass (carry-ins that begin exactly there are not among the tile's own begins).
    const bool full = batch != 0 || ts == PX || PX == x_b;
    if (full) {  // the top changed here: everything that is active
      for (int q = Q0; q >= 0; --q) {
        if (spm[q] <= PX) break;
        const uint64_t ee = se[q];
        if (ee > PX) {
          const uint32_t id = sid[q];
          if (id != ti && overlap_exceeds(sx[q], ee, ts, te, a.thr)) a.ovl[id] = 1;
        }
      }
      if (cc_complete) {
        for (uint32_t c = 0; c < n_cc; ++c) {
          const uint64_t e = le[c];
          if (e > PX) {
            const uint32_t id = lid[c];
            if (id != ti && overlap_exceeds(ls[c], e, ts, te, a.thr)) a.ovl[id] = 1;
          }
        }
      } else {
        for (uint32_t c = c_begin; c < c_end; ++c) {
          const uint64_t e = a.c_e[c];
          if (e > PX) {
            const uint32_t id = a.c_id[c];
            if (id != ti && overlap_exceeds(a.c_s[c], e, ts, te, a.thr)) a.ovl[id] = 1;
          }
        }
      }
    } else {  // same top as just before x: only the intervals that begin at x have not met it yet
      for (int q = Q0; q >= 0 && sx[q] == PX; --q) {
        const uint64_t ee = se[q];
        if (ee > PX) {
          const uint32_t id = sid[q];
          if (id != ti && overlap_exceeds(PX, ee, ts, te, a.thr)) a.ovl[id] = 1;
        }
      }
    }
  }
  TT_STAMP(4);  // the last batch
}

// ---- 2 <= k < inf ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(TB) void sweep_tile_kn_kernel(TileArgs a) {
  __shared__ uint64_t sx[TB];    // composite start of begin q
  __shared__ uint64_t se[TB];    // its composite end
  __shared__ uint64_t skey[TB];  // its score key
  __shared__ uint64_t spm[TB];   // prefix maximum of se
  __shared__ uint32_t sid[TB];   // its interval index
  __shared__ uint64_t wmax[TB / 64];

  const uint32_t tile_id = blockIdx.x;
  if (a.tile_done && a.tile_done[tile_id]) return;  // block-uniform: sweep_tile_kp_kernel has done this tile
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const uint64_t p0 = (uint64_t)tile_id * TB;
  const uint64_t p = p0 + tid;
  const bool valid = p < a.n;
  uint64_t X = ~0ull, EE = 0, KEY = 0;
  uint32_t ID = 0;
  if (valid) {
    X = a.S[p];
    EE = swg_comp_end(X, a.E[p], a.pos_bits);
    KEY = a.KEY[p];
    ID = a.I[p];
  }
  sx[tid] = X;
  se[tid] = EE;
  skey[tid] = KEY;
  sid[tid] = ID;
  uint64_t m = EE;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    uint64_t t = __shfl_up(m, d, 64);
    if (lane >= d && t > m) m = t;
  }
  if (lane == 63) wmax[wave] = m;
  __syncthreads();
#pragma unroll
  for (int w = 0; w < TB / 64; ++w)
    if (w < wave && wmax[w] > m) m = wmax[w];
  spm[tid] = m;
  const uint64_t x_next = (tile_id + 1 < a.ntiles) ? a.tile_x[tile_id + 1] : ~0ull;
  __syncthreads();

  const uint32_t c_begin = a.carry_off[tile_id], c_end = a.carry_off[tile_id + 1];
  const uint32_t n_chunks = (c_end - c_begin + CC - 1) / CC;
  const uint32_t n_batches = 2 + n_chunks;

  for (uint32_t batch = 0; batch < n_batches; ++batch) {
    // ---- this thread's evaluation point: coordinate PX, last own begin at or before it Q0
    bool eval;
    uint64_t PX;
    int Q0;
    if (batch == 0) {  // start coordinates: the last begin of each run, unless the run continues in the next tile
      eval = valid && X != 0 && (tid == TB - 1 || sx[tid + 1] != X) && X != x_next;
      PX = X;
      Q0 = tid;
    } else {
      // end coordinates that fall inside this tile's range (X_b < PX < X_{b+1}): batch 1 = ends of the
      // tile's own begins, batch 2+j = ends of the carry-ins of chunk j.  An end equal to the next
      // tile's first key is evaluated there as a start coordinate.
      if (batch == 1) {
        eval = valid && X != 0 && EE > X && EE < x_next;
        PX = EE;
      } else {
        const uint32_t ci = c_begin + (batch - 2) * CC + tid;
        eval = ci < c_end;
        PX = eval ? a.c_e[ci] : 0;
        eval = eval && PX < x_next;
      }
      int l = 0, r = TB;  // upper_bound(sx, PX) - 1; sx is ~0 past the end of a short last tile
      while (l < r) {
        const int mid = (l + r) >> 1;
        if (sx[mid] <= PX)
          l = mid + 1;
        else
          r = mid;
      }
      Q0 = l - 1;
    }

    // own-tile begins: scan backwards from Q0, stop once no earlier interval can reach PX
    auto own_tile = [&](auto&& f) {
      for (int q = Q0; q >= 0; --q) {
        if (spm[q] <= PX) break;
        const uint64_t ee = se[q];
        if (ee > PX) f(sx[q], ee, skey[q], sid[q]);
      }
    };

    if (eval) {
      // ---- general k: walk the priority order by successive minima; no per-thread storage.
      // Carry-ins are read straight from global memory here (every lane reads the same entry, so
      // the loads coalesce to one request); this path is for 2 <= k < inf.
      auto all_active = [&](auto&& f) {
        own_tile(f);
        for (uint32_t c = c_begin; c < c_end; ++c) {
          const uint64_t ee = a.c_e[c];
          if (ee > PX) f(a.c_s[c], ee, a.c_key[c], a.c_id[c]);
        }
      };
      // next_after(prev): smallest priority strictly greater than prev among the active set
      auto next_after = [&](bool have_prev, uint64_t pk, uint64_t ps, uint32_t pi, uint64_t* nk, uint64_t* ns,
                            uint64_t* ne, uint32_t* ni) -> bool {
        bool found = false;
        uint64_t fk = 0, fs = 0, fe = 0;
        uint32_t fi = 0;
        all_active([&](uint64_t s, uint64_t e, uint64_t key, uint32_t id) {
          if (have_prev && !prio_less(pk, ps, pi, key, s, id)) return;  // not after prev
          if (!found || prio_less(key, s, id, fk, fs, fi)) {
            fk = key;
            fs = s;
            fe = e;
            fi = id;
            found = true;
          }
        });
        *nk = fk;
        *ns = fs;
        *ne = fe;
        *ni = fi;
        return found;
      };
      // phase 1: the k-th best priority (tau); fewer than k actives -> everyone is in T(x)
      uint64_t tk = 0, ts = 0;
      uint32_t ti = 0;
      bool have_prev = false, exhausted = false;
      for (uint64_t r = 0; r < a.k; ++r) {
        uint64_t nk, ns, ne;
        uint32_t ni;
        if (!next_after(have_prev, tk, ts, ti, &nk, &ns, &ne, &ni)) {
          exhausted = true;
          break;
        }
        tk = nk;
        ts = ns;
        ti = ni;
        have_prev = true;
      }
      if (have_prev) {
        // phase 2: members of T(x) are `ever top`
        all_active([&](uint64_t s, uint64_t e, uint64_t key, uint32_t id) {
          (void)e;
          if (exhausted || !prio_less(tk, ts, ti, key, s, id)) a.top[id] = 1;
        });
        if (!exhausted && a.thr < 1.0) {
          // phase 3: every non-member against every member
          uint64_t mk = 0, ms = 0, me = 0;
          uint32_t mi = 0;
          bool mprev = false;
          for (uint64_t r = 0; r < a.k; ++r) {
            uint64_t nk, ns, ne;
            uint32_t ni;
            if (!next_after(mprev, mk, ms, mi, &nk, &ns, &ne, &ni)) break;
            mk = nk;
            ms = ns;
            me = ne;
            mi = ni;
            mprev = true;
            all_active([&](uint64_t s, uint64_t e, uint64_t key, uint32_t id) {
              if (prio_less(tk, ts, ti, key, s, id) && overlap_exceeds(s, e, ms, me, a.thr)) a.ovl[id] = 1;
            });
          }
        }
      }
    }
  }
}

// ---- 2 <= k <= KSTAR_MAX on deep data: the k = 1 kernel's pruning carried over ---------------------------------------
// T(x) is the set of the k best active intervals ("members").  With S*_1..S*_k the k best carry-ins that are active over the
// whole range of the tile, nothing ranked below S*_k is ever a member inside the tile, so only "candidates" (ranked above
// S*_k) take part in finding T(x), and only THEIR end points are evaluated -- and those only when the ending interval was a
// member just before x (fewer than k stars / candidates active there rank above it).  Non-members are flagged when they
// overlap a member too much; a (non-member, member) pair first meets either where the member enters T (it begins at x, or a
// member ended at x: every active interval is tested against every member there, as at the tile's first coordinate) or
// where the non-member begins (only the begins at x are tested).  With fewer than k spanning carry-ins (sparse data) there is
// no threshold: every interval is a candidate, and what remains is the saving on the end points and on the overlap passes.
// Tiles with more candidates than the LDS list holds, and k > KSTAR_MAX, are left to sweep_tile_kn_kernel.
constexpr int KSTAR_MAX = 16;
constexpr int KP_CAP = CCAP;  // candidate carry-ins kept in LDS (256 was measured: slower, and the tiles that overflow have no threshold at all)
__global__ __launch_bounds__(TB) void sweep_tile_kp_kernel(TileArgs a) {
  __shared__ uint64_t sx[TB];
  __shared__ uint64_t se2[2 * TB];   // [0, TB): ends; [TB, 2 TB): ends of the candidates, 0 for the others
  __shared__ uint64_t spm2[2 * TB];  // prefix maxima of the two halves
  __shared__ uint64_t skey[TB];
  __shared__ uint32_t sid[TB];
  __shared__ uint64_t wmax[TB / 64];
  __shared__ uint64_t ls[KP_CAP], le[KP_CAP], lkey[KP_CAP];  // candidate carry-ins
  __shared__ uint32_t lid[KP_CAP];
  __shared__ uint32_t l_count;
  __shared__ uint64_t r_k[TB / 64], r_s[TB / 64], r_e[TB / 64];
  __shared__ uint32_t r_i[TB / 64], r_have[TB / 64];
  __shared__ uint64_t st_k[KSTAR_MAX], st_s[KSTAR_MAX], st_e[KSTAR_MAX];  // the stars, best first
  __shared__ uint32_t st_i[KSTAR_MAX];

  const uint32_t tile_id = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const uint32_t c_begin = a.carry_off[tile_id], c_end = a.carry_off[tile_id + 1];
  const uint32_t n_carry = c_end - c_begin;
  const int K = (int)a.k;
  if (a.k > (uint64_t)KSTAR_MAX) {  // block-uniform
    if (tid == 0) a.tile_done[tile_id] = 0;
    return;
  }
  const uint64_t p = (uint64_t)tile_id * TB + tid;
  const bool valid = p < a.n;
  uint64_t X = ~0ull, EE = 0, KEY = 0;
  uint32_t ID = 0;
  if (valid) {
    X = a.S[p];
    EE = swg_comp_end(X, a.E[p], a.pos_bits);
    KEY = a.KEY[p];
    ID = a.I[p];
  }
  sx[tid] = X;
  se2[tid] = EE;
  skey[tid] = KEY;
  sid[tid] = ID;
  if (tid == 0) l_count = 0;
  auto block_prefix_max = [&](uint64_t v, uint64_t* dst) {  // dst[tid] = max(v of threads 0..tid); two barriers
    uint64_t m = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const uint64_t t = __shfl_up(m, d, 64);
      if (lane >= d && t > m) m = t;
    }
    if (lane == 63) wmax[wave] = m;
    __syncthreads();
#pragma unroll
    for (int w = 0; w < TB / 64; ++w)
      if (w < wave && wmax[w] > m) m = wmax[w];
    dst[tid] = m;
    __syncthreads();
  };
  block_prefix_max(EE, spm2);
  const uint64_t x_b = sx[0];
  const uint64_t x_next = (tile_id + 1 < a.ntiles) ? a.tile_x[tile_id + 1] : ~0ull;

  // ---- the stars: K rounds of a block-wide arg-min over the spanning carry-ins ranked after the previous star
  int n_star = 0;
  uint64_t pk = 0, ps = 0;
  uint32_t pi = 0;
  for (int r = 0; r < K; ++r) {
    bool hv = false;
    uint64_t bk = 0, bs = 0, be = 0;
    uint32_t bi = 0;
    for (uint32_t c = c_begin + tid; c < c_end; c += TB) {
      const uint64_t e = a.c_e[c];
      if (e >= x_next) {
        const uint64_t k = a.c_key[c], s_ = a.c_s[c];
        const uint32_t id = a.c_id[c];
        if (r > 0 && !prio_less(pk, ps, pi, k, s_, id)) continue;  // not after the previous star
        if (!hv || prio_less(k, s_, id, bk, bs, bi)) {
          bk = k;
          bs = s_;
          be = e;
          bi = id;
          hv = true;
        }
      }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const uint64_t ok = __shfl_down(bk, o, 64), os = __shfl_down(bs, o, 64), oe = __shfl_down(be, o, 64);
      const uint32_t oi = __shfl_down(bi, o, 64);
      const bool oh = __shfl_down((int)hv, o, 64) != 0;
      if (oh && (!hv || prio_less(ok, os, oi, bk, bs, bi))) {
        bk = ok;
        bs = os;
        be = oe;
        bi = oi;
        hv = true;
      }
    }
    if (lane == 0) {
      r_k[wave] = bk;
      r_s[wave] = bs;
      r_e[wave] = be;
      r_i[wave] = bi;
      r_have[wave] = hv ? 1u : 0u;
    }
    __syncthreads();
    bool got = false;
    uint64_t gk = 0, gs = 0, ge = 0;
    uint32_t gi = 0;
#pragma unroll
    for (int w = 0; w < TB / 64; ++w)
      if (r_have[w] && (!got || prio_less(r_k[w], r_s[w], r_i[w], gk, gs, gi))) {
        gk = r_k[w];
        gs = r_s[w];
        ge = r_e[w];
        gi = r_i[w];
        got = true;
      }
    __syncthreads();  // r_* are rewritten by the next round
    if (!got) break;  // block-uniform
    if (tid == 0) {
      st_k[r] = gk;
      st_s[r] = gs;
      st_e[r] = ge;
      st_i[r] = gi;
    }
    pk = gk;
    ps = gs;
    pi = gi;
    ++n_star;
  }
  // fewer than k carry-ins span the tile (the usual case on sparse data): no threshold, every interval is a candidate and
  // the stars are simply the spanning carry-ins; else (pk, ps, pi) is S*_k
  const bool have_thr = n_star == K;
  // ---- candidate carry-ins: end inside the tile's range and ranked above S*_k
  for (uint32_t c = c_begin + tid; c < c_end; c += TB) {
    const uint64_t e = a.c_e[c];
    if (e < x_next) {
      const uint64_t k = a.c_key[c], s_ = a.c_s[c];
      const uint32_t id = a.c_id[c];
      if (!have_thr || prio_less(k, s_, id, pk, ps, pi)) {
        const uint32_t slot = atomicAdd(&l_count, 1u);
        if (slot < (uint32_t)KP_CAP) {
          ls[slot] = s_;
          le[slot] = e;
          lkey[slot] = k;
          lid[slot] = id;
        }
      }
    }
  }
  // candidates among the tile's own begins
  const bool own_cand = valid && X != 0 && (!have_thr || prio_less(KEY, X, ID, pk, ps, pi));
  se2[TB + tid] = own_cand ? EE : 0;
  block_prefix_max(own_cand ? EE : 0, spm2 + TB);  // two barriers: l_count, the LDS list and the stars are visible after it
  const uint32_t n_cc = l_count;
  if (n_cc > (uint32_t)KP_CAP) {  // block-uniform
    if (tid == 0) a.tile_done[tile_id] = 0;
    return;
  }
  if (tid == 0) a.tile_done[tile_id] = 1;
  const uint64_t* se = se2;
  const uint64_t* spm = spm2;
  const uint64_t* sec = se2 + TB;   // candidate view
  const uint64_t* spmc = spm2 + TB;
  const bool pass2 = a.thr < 1.0;
  const uint32_t n_batches = 2 + (n_cc ? 1u : 0u);

  for (uint32_t batch = 0; batch < n_batches; ++batch) {
    bool eval;
    uint64_t PX;
    int Q0;
    uint64_t PK = KEY, PS = X;  // priority of the interval whose end is the point (batches >= 1)
    uint32_t PI = ID;
    if (batch == 0) {  // start coordinates: the last begin of each run, unless the run continues in the next tile
      eval = valid && X != 0 && (tid == TB - 1 || sx[tid + 1] != X) && X != x_next;
      PX = X;
      Q0 = tid;
    } else {
      if (batch == 1) {  // ends of the tile's own candidates inside its range
        eval = valid && X != 0 && EE > X && EE < x_next && sec[tid] != 0;
        PX = EE;
      } else {
        PX = (uint32_t)tid < n_cc ? le[tid] : 0;
        eval = PX != 0 && PX < x_next;
        if (eval) {
          PK = lkey[tid];
          PS = ls[tid];
          PI = lid[tid];
        }
      }
      int l = 0, r = TB;  // upper_bound(sx, PX) - 1
      while (l < r) {
        const int mid = (l + r) >> 1;
        if (sx[mid] <= PX)
          l = mid + 1;
        else
          r = mid;
      }
      Q0 = l - 1;
    }
    if (!eval) continue;  // no barrier below this line

    if (batch != 0) {
      // was the ending interval a member just before x?  Count what ranks above it among the stars (all active) and the
      // candidates with s < x <= e.
      int above = 0;
      for (int r = 0; r < n_star; ++r) above += prio_less(st_k[r], st_s[r], st_i[r], PK, PS, PI) ? 1 : 0;
      for (int q = Q0; q >= 0 && above < K; --q) {
        if (spmc[q] < PX) break;  // no earlier candidate reaches PX
        if (sec[q] >= PX) {
          const uint64_t sq = sx[q];
          if (sq < PX && prio_less(skey[q], sq, sid[q], PK, PS, PI)) ++above;
        }
      }
      for (uint32_t c = 0; c < n_cc && above < K; ++c)
        if (le[c] >= PX && prio_less(lkey[c], ls[c], lid[c], PK, PS, PI)) ++above;
      if (above >= K) continue;
    }
    // the stars and the candidates active at x (s <= x < e)
    auto active_cands = [&](auto&& f) {
      for (int r = 0; r < n_star; ++r) f(st_s[r], st_e[r], st_k[r], st_i[r]);
      for (int q = Q0; q >= 0; --q) {
        if (spmc[q] <= PX) break;
        const uint64_t ee = sec[q];
        if (ee > PX) f(sx[q], ee, skey[q], sid[q]);
      }
      for (uint32_t c = 0; c < n_cc; ++c) {
        const uint64_t e = le[c];
        if (e > PX) f(ls[c], e, lkey[c], lid[c]);
      }
    };
    // next member after (mk, ms, mi) in priority order; false when the active set is exhausted
    auto next_member = [&](bool have_prev, uint64_t mk, uint64_t ms, uint32_t mi, uint64_t* nk, uint64_t* ns, uint64_t* ne,
                           uint32_t* ni) -> bool {
      bool found = false;
      uint64_t fk = 0, fs = 0, fe = 0;
      uint32_t fi = 0;
      active_cands([&](uint64_t s_, uint64_t e, uint64_t key, uint32_t id) {
        if (have_prev && !prio_less(mk, ms, mi, key, s_, id)) return;
        if (!found || prio_less(key, s_, id, fk, fs, fi)) {
          fk = key;
          fs = s_;
          fe = e;
          fi = id;
          found = true;
        }
      });
      if (found) {
        *nk = fk;
        *ns = fs;
        *ne = fe;
        *ni = fi;
      }
      return found;
    };
    // ---- members: ever top; does one of them begin here?
    uint64_t tk = 0, ts = 0, te = 0;
    uint32_t ti = 0;
    bool member_begins = false;
    int n_mem = 0;
    for (int r = 0; r < K; ++r) {
      if (!next_member(r > 0, tk, ts, ti, &tk, &ts, &te, &ti)) break;
      ++n_mem;
      a.top[ti] = 1;
      member_begins = member_begins || ts == PX;
    }
    // with K members (tk, ts, ti) is the K-th: everything ranked after it is a non-member; with fewer, everything active is a
    // member and there is nobody to flag (plane_sweep_exact.rs: the overlap pass needs more than k actives)
    if (!pass2 || n_mem < K) continue;
    const bool full = batch != 0 || member_begins || PX == x_b;
    uint64_t mk = 0, ms = 0, me = 0;
    uint32_t mi = 0;
    for (int r = 0; r < K; ++r) {
      (void)next_member(r > 0, mk, ms, mi, &mk, &ms, &me, &mi);
      if (full) {  // the member set changed here: every active non-member against this member
        for (int q = Q0; q >= 0; --q) {
          if (spm[q] <= PX) break;
          const uint64_t ee = se[q];
          if (ee > PX) {
            const uint32_t id = sid[q];
            const uint64_t sq = sx[q];
            if (prio_less(tk, ts, ti, skey[q], sq, id) && overlap_exceeds(sq, ee, ms, me, a.thr)) a.ovl[id] = 1;
          }
        }
        for (uint32_t c = c_begin; c < c_end; ++c) {
          const uint64_t e = a.c_e[c];
          if (e > PX) {
            const uint64_t s_ = a.c_s[c], key = a.c_key[c];
            const uint32_t id = a.c_id[c];
            if (prio_less(tk, ts, ti, key, s_, id) && overlap_exceeds(s_, e, ms, me, a.thr)) a.ovl[id] = 1;
          }
        }
      } else {  // same members as just before x: only the intervals that begin at x have not met them yet
        for (int q = Q0; q >= 0 && sx[q] == PX; --q) {
          const uint64_t ee = se[q];
          if (ee > PX) {
            const uint32_t id = sid[q];
            if (prio_less(tk, ts, ti, skey[q], PX, id) && overlap_exceeds(PX, ee, ms, me, a.thr)) a.ovl[id] = 1;
          }
        }
      }
    }
  }
}

// k == inf: every active interval is always in T(x) (plane_sweep_exact.rs:219-228 with
// usize::MAX), so keep = single-in-segment | (start < end): no sweep.  Only zero-length
// intervals need the segment size; when any exist the begins are sorted once to get the
// `single` flags, otherwise nothing else is needed.
__global__ __launch_bounds__(EW_THREADS) void kinf_mark_kernel(uint64_t n, const uint32_t* __restrict__ start,
                                                               const uint32_t* __restrict__ end,
                                                               const uint8_t* __restrict__ alive,
                                                               const uint8_t* __restrict__ and_with,
                                                               uint8_t* __restrict__ keep,
                                                               uint32_t* __restrict__ n_zero) {
  uint64_t i = (uint64_t)blockIdx.x * EW_THREADS + threadIdx.x;
  if (i >= n) return;
  const bool live = alive ? alive[i] != 0 : true;
  uint8_t kf = 0;
  if (live) {
    if (start[i] < end[i])
      kf = (!and_with || and_with[i]) ? 1 : 0;
    else
      atomicAdd(n_zero, 1u);
  }
  keep[i] = kf;
}
// Both axes of a sweep with no limit at all (the CLI's default --num-mappings): keep = alive and both spans non-empty; the
// zero-length intervals are only counted -- if there are any, the caller takes the two per-axis passes instead (they need
// the segment sizes).
__global__ __launch_bounds__(EW_THREADS) void kinf_mark_both_kernel(uint64_t n, const uint32_t* __restrict__ qs,
                                                                    const uint32_t* __restrict__ qe,
                                                                    const uint32_t* __restrict__ ts,
                                                                    const uint32_t* __restrict__ te,
                                                                    const uint8_t* __restrict__ alive,
                                                                    uint8_t* __restrict__ keep, uint32_t* __restrict__ n_zero) {
  uint64_t i = (uint64_t)blockIdx.x * EW_THREADS + threadIdx.x;
  if (i >= n) return;
  const bool live = alive ? alive[i] != 0 : true;
  uint8_t kf = 0;
  if (live) {
    if (qs[i] < qe[i] && ts[i] < te[i])
      kf = 1;
    else
      atomicAdd(n_zero, 1u);
  }
  keep[i] = kf;
}
__global__ __launch_bounds__(EW_THREADS) void kinf_single_kernel(uint64_t n, const uint8_t* __restrict__ single,
                                                                 const uint8_t* __restrict__ and_with,
                                                                 uint8_t* __restrict__ keep) {
  uint64_t i = (uint64_t)blockIdx.x * EW_THREADS + threadIdx.x;
  if (i < n && single[i] && (!and_with || and_with[i])) keep[i] = 1;
}

// keep = live & (single | (top & ~ovl)) [& and_with], 16 records per thread with 16-byte loads when the arrays are
// 16-byte aligned (byte-per-lane loads made this a 1.2 TB/s kernel), any non-zero byte counting as set.
__device__ __forceinline__ uint32_t bytes_nonzero01(uint32_t w) {  // 0x01 in every byte of w that is != 0
  return ((((w & 0x7f7f7f7fu) + 0x7f7f7f7fu) | w) >> 7) & 0x01010101u;
}
__global__ __launch_bounds__(EW_THREADS) void combine_kernel(uint64_t n, const uint8_t* __restrict__ alive,
                                                             const uint8_t* __restrict__ single,
                                                             const uint8_t* __restrict__ top,
                                                             const uint8_t* __restrict__ ovl,
                                                             const uint8_t* __restrict__ and_with,
                                                             uint8_t* __restrict__ keep, int aligned) {
  const uint64_t i0 = ((uint64_t)blockIdx.x * EW_THREADS + threadIdx.x) * 16;
  if (i0 >= n) return;
  if (aligned && i0 + 16 <= n) {
    const uint4 s4 = *reinterpret_cast<const uint4*>(single + i0);
    const uint4 t4 = *reinterpret_cast<const uint4*>(top + i0);
    const uint4 o4 = *reinterpret_cast<const uint4*>(ovl + i0);
    uint4 a4 = make_uint4(0x01010101u, 0x01010101u, 0x01010101u, 0x01010101u), w4 = a4;
    if (alive) a4 = *reinterpret_cast<const uint4*>(alive + i0);
    if (and_with) w4 = *reinterpret_cast<const uint4*>(and_with + i0);
    auto f = [](uint32_t a, uint32_t sg, uint32_t t, uint32_t o, uint32_t w) {
      return bytes_nonzero01(a) & (bytes_nonzero01(sg) | (bytes_nonzero01(t) & ~bytes_nonzero01(o))) & bytes_nonzero01(w);
    };
    *reinterpret_cast<uint4*>(keep + i0) =
        make_uint4(f(a4.x, s4.x, t4.x, o4.x, w4.x), f(a4.y, s4.y, t4.y, o4.y, w4.y), f(a4.z, s4.z, t4.z, o4.z, w4.z),
                   f(a4.w, s4.w, t4.w, o4.w, w4.w));
    return;
  }
  for (uint64_t i = i0; i < i0 + 16 && i < n; ++i) {
    const bool live = alive ? alive[i] != 0 : true;
    keep[i] = (live && (single[i] || (top[i] && !ovl[i])) && (!and_with || and_with[i])) ? 1 : 0;
  }
}

// the begins at the listed sorted positions, side by side again
__global__ __launch_bounds__(EW_THREADS) void begin_compact_kernel(uint64_t m, const uint32_t* __restrict__ list, const uint64_t* __restrict__ S,
                                                                   const uint32_t* __restrict__ I, const uint32_t* __restrict__ E,
                                                                   const uint64_t* __restrict__ KEY, uint64_t* __restrict__ S2,
                                                                   uint32_t* __restrict__ I2, uint32_t* __restrict__ E2, uint64_t* __restrict__ KEY2) {
  const uint64_t j = (uint64_t)blockIdx.x * EW_THREADS + threadIdx.x;
  if (j >= m) return;
  const uint32_t p = list[j];
  S2[j] = S[p];
  I2[j] = I[p];
  E2[j] = E[p];
  KEY2[j] = KEY[p];
}
// the same over the begins of a compact list (the segment-resident sweep answered every other record): an interval of a segment
// of more than one live record is kept iff it was the top somewhere and never overlapped one
__global__ __launch_bounds__(EW_THREADS) void combine_begins_kernel(uint64_t nb, const uint32_t* __restrict__ I, const uint8_t* __restrict__ top,
                                                                    const uint8_t* __restrict__ ovl, const uint8_t* __restrict__ and_with,
                                                                    uint8_t* __restrict__ keep) {
  const uint64_t p = (uint64_t)blockIdx.x * EW_THREADS + threadIdx.x;
  if (p >= nb) return;
  const uint32_t i = I[p];
  keep[i] = (top[i] && !ovl[i] && (!and_with || and_with[i])) ? 1 : 0;
}

inline unsigned blocks_for(uint64_t n, int threads) { return (unsigned)((n + threads - 1) / threads); }

}  // namespace

int swg_kinf_both(swg_ctx* ctx, uint64_t n, const uint32_t* qs, const uint32_t* qe, const uint32_t* ts, const uint32_t* te,
                  const uint8_t* alive, uint8_t* keep, int* done) {
  *done = 0;
  if (n == 0) {
    *done = 1;
    return SWG_OK;
  }
  swg_arena_mark mark = swg_arena_save(ctx);
  uint32_t* n_zero = swg_alloc<uint32_t>(ctx, 2);
  SWG_CHECK_ARENA(ctx);
  SWG_HIP(ctx, hipMemsetAsync(n_zero, 0, 8, ctx->stream));
  SWG_LAUNCH(ctx, "kinf_mark_both", kinf_mark_both_kernel<<<(unsigned)((n + EW_THREADS - 1) / EW_THREADS), EW_THREADS, 0, ctx->stream>>>(
                                   n, qs, qe, ts, te, alive, keep, n_zero));
  SWG_KERNEL_CHECK(ctx);
  uint64_t h = 0;
  SWG_TRY(swg_read_scalars(ctx, reinterpret_cast<uint64_t*>(n_zero), &h, 1));
  swg_arena_restore(ctx, mark);
  *done = (uint32_t)h == 0;
  return SWG_OK;
}

int swg_score_keys(swg_ctx* ctx, uint64_t n, const uint32_t* q_start, const uint32_t* q_end,
                   const double* identity, int scoring, uint64_t* key_out) {
  if (n == 0) return SWG_OK;
  SWG_LAUNCH(ctx, "score_key", score_key_kernel<<<blocks_for(n, EW_THREADS), EW_THREADS, 0, ctx->stream>>>(
                                   n, q_start, q_end, identity, scoring, key_out));
  SWG_KERNEL_CHECK(ctx);
  return SWG_OK;
}

int swg_prepare(swg_ctx* ctx, const swg_records* r, const swg_config* cfg, uint8_t* alive, swg_key_ends* key_ends, bool with_keys,
                unsigned long long* scalars, uint32_t* group32, uint32_t* probe_flag, uint64_t* score_out) {
  if (r->n == 0) return SWG_OK;
  if (probe_flag) {
    SWG_LAUNCH(ctx, "input_order_probe", input_order_probe_kernel<<<1, 256, 0, ctx->stream>>>(r->n, r->q_id, r->t_id, probe_flag));
    SWG_KERNEL_CHECK(ctx);
  }
  SWG_LAUNCH(ctx, "prepare", prepare_kernel<<<ctx->num_cu * 16, EW_THREADS, 0, ctx->stream>>>(
                                 r->n, r->q_id, r->t_id, r->block_len, r->matches, r->identity, r->q_start, r->q_end, r->t_start, r->t_end,
                                 cfg->min_block_length, cfg->keep_self, cfg->min_identity, cfg->scoring_function, alive, key_ends,
                                 cfg->scaffold_gap != 0 ? 1 : 0, with_keys ? 1 : 0, scalars, r->strand, r->n_seq, group32, probe_flag, score_out));
  SWG_KERNEL_CHECK(ctx);
  return SWG_OK;
}

int swg_sweep_axis(swg_ctx* ctx, const swg_axis_input& in, uint64_t k, double thr, uint8_t* keep) {
  const uint64_t n = in.n;
  if (in.sorted_idx_valid) *in.sorted_idx_valid = 0;
  if (n == 0) return SWG_OK;
  if (n >= (uint64_t(1) << 31)) return swg_set_error(ctx, SWG_ERR_RANGE, "sweep: n >= 2^31 intervals");
  hipStream_t st = ctx->stream;
  swg_arena_mark mark = swg_arena_save(ctx);
  const int key_bits = in.seg_bits + in.pos_bits;  // seg_bits must cover (max segment id + 1)
  if (key_bits > 64)
    return swg_set_error(ctx, SWG_ERR_RANGE, "sweep: segment id (%d bits) + coordinate (%d bits) exceed 64 bits",
                         in.seg_bits, in.pos_bits);
  uint32_t ntiles = (uint32_t)((n + TB - 1) / TB);

  // sorted begins + gathered columns + `single` flags (shared by the k = inf and the general path)
  uint64_t* S = nullptr;
  uint32_t* I = nullptr;
  uint32_t* E = nullptr;  // end coordinates, in one of the sort's u64 scratch buffers
  uint64_t *KEY = nullptr, *tile_x = nullptr, *tile_xf = nullptr;  // tile-start keys of the 256-begin / TBF-begin tilings
  uint32_t ntilesf = (uint32_t)((n + TBF - 1) / TBF);
  uint8_t* single = nullptr;
  swg_seg_plan_view seg_view;  // k = 1 over the plan's runs: the segments of the sorted begins, for the streaming sweep
  uint8_t* tile_flag = nullptr;  // the segment sorts settled the lone intervals: the sorted positions that still need the tile kernels
  auto sort_begins = [&]() -> int {
    S = swg_alloc<uint64_t>(ctx, n);
    I = swg_alloc<uint32_t>(ctx, n);
    uint64_t* S2 = swg_alloc<uint64_t>(ctx, n);
    uint32_t* I2 = swg_alloc<uint32_t>(ctx, n);
    KEY = swg_alloc<uint64_t>(ctx, n);
    tile_x = swg_alloc<uint64_t>(ctx, (size_t)ntiles + 1);
    tile_xf = swg_alloc<uint64_t>(ctx, (size_t)ntilesf + 1);
    single = swg_alloc<uint8_t>(ctx, n);
    SWG_CHECK_ARENA(ctx);
    if (in.seg_runs && !in.sorted_idx_out) {
      // records grouped by (seg_a, seg_b) pair: the begins sorted segment by segment in LDS (swg_segsort.hip); E and KEY in
      // buffers of their own (the radix path takes them from its scratch)
      const swg_arena_mark seg_mark = swg_arena_save(ctx);
      uint32_t* E2 = swg_alloc<uint32_t>(ctx, n);
      SWG_CHECK_ARENA(ctx);
      SWG_HIP(ctx, hipMemsetAsync(single, 0, n, st));
      // SWG_SEG_LONE=1 (opt-in, read at every call), finite k: the sort also settles the intervals that overlap nobody in their
      // segment -- they are kept whatever k is -- and only the others are compacted for the routing and the tile kernels.
      // Measured on S-pan (round 6, profiles/README.md): 56 % of the begins are settled that way and the tile kernel's time falls
      // from 5.6 to 3.2 ms, but the classification inside the sort (+1.2 ms) and the compaction (+1.7 ms) take it back: 13.8
      // against 13.6 ms.  Exact either way (tests/test_gpu_segsweep.py).
      const char* lone_s = getenv("SWG_SEG_LONE");
      const char* stream_s = getenv("SWG_SEG_STREAM");
      if (k != SWG_K_INF && lone_s && atoi(lone_s) == 1 && !(stream_s && atoi(stream_s) >= 1)) {
        tile_flag = swg_alloc<uint8_t>(ctx, n);
        SWG_CHECK_ARENA(ctx);
        SWG_HIP(ctx, hipMemsetAsync(tile_flag, 0, n, st));
      }
      int done = 0;
      SWG_TRY(swg_seg_sort_begins(ctx, in, S, I, E2, KEY, tile_xf, ntilesf, single, &done, k == 1 ? &seg_view : nullptr, tile_flag));
      if (!done) tile_flag = nullptr;
      if (done) {
        E = E2;
        SWG_LAUNCH(ctx, "tile_x_pairs", tile_x_pairs_kernel<<<blocks_for(ntiles, EW_THREADS), EW_THREADS, 0, st>>>(ntiles, tile_xf, tile_x, TB / TBF));
        SWG_KERNEL_CHECK(ctx);
        return SWG_OK;
      }
      swg_arena_restore(ctx, seg_mark);
    }
    // the sort's digit histograms come out of begin_build (when the onesweep path will run: up to 8 passes)
    static const bool sort_fallback = getenv("SWG_SORT_FALLBACK") != nullptr;
    uint32_t* prehist = nullptr;
    if (key_bits <= 8 * SWG_RADIX_MAX_PASSES && !sort_fallback && n > 1) {
      prehist = swg_alloc<uint32_t>(ctx, (size_t)SWG_RADIX_MAX_PASSES * SWG_RADIX_BINS);
      SWG_CHECK_ARENA(ctx);
      SWG_HIP(ctx, hipMemsetAsync(prehist, 0, sizeof(uint32_t) * SWG_RADIX_MAX_PASSES * SWG_RADIX_BINS, st));
    }
    const int idx_bits = swg_bits_for(n - 1) ? swg_bits_for(n - 1) : 1;
    const bool packed_sort = in.packed && in.pos_bits >= 8 && swg_radix_sort_packed_applies(n, key_bits, idx_bits);
    // One radix pass fewer: sort on the key without the low `drop` bits of the start coordinate, order the short runs of equal
    // truncated keys in the gather (begin_gather_words_kernel).  A run longer than the gather can see (dense data, heavy ties)
    // raises a flag: the context stops trying (for good, after a second failure with fewer bits) and this sort runs again
    // the ordinary way.
    const int drop = (packed_sort && prehist) ? swg_radix_drop_bits(n, key_bits, in.pos_bits, idx_bits, ctx->sort_drop_level) : 0;
    if (drop) {
      const unsigned full = blocks_for(n, EW_THREADS), cap = (unsigned)ctx->num_cu * 16;
      SWG_LAUNCH(ctx, "begin_build", begin_build_kernel<<<full > cap ? cap : full, EW_THREADS, 0, st>>>(
                                         n, in.seg, in.seg_a, in.seg_b, in.seg_table, in.seg_mul, in.start, in.alive, in.pos_bits, S, nullptr,
                                         swg_radix_plan_words(key_bits - drop), prehist, drop, idx_bits));
      SWG_KERNEL_CHECK(ctx);
      uint64_t* P = nullptr;
      const int wrc = swg_radix_sort_words(ctx, S, S2, n, key_bits - drop, idx_bits, prehist, &P);
      if (wrc != SWG_OK) return wrc == SWG_ERR_UNSUPPORTED ? swg_set_error(ctx, SWG_ERR_HIP, "word sort declined a shape it accepted") : wrc;
      uint64_t* other = P == S ? S2 : S;
      const swg_arena_mark third_mark = swg_arena_save(ctx);
      uint64_t* third = swg_alloc<uint64_t>(ctx, n);
      uint64_t* d_flag = swg_alloc<uint64_t>(ctx, 1);
      SWG_CHECK_ARENA(ctx);
      SWG_HIP(ctx, hipMemsetAsync(single, 0, n, st));
      SWG_HIP(ctx, hipMemsetAsync(d_flag, 0, 8, st));
      SWG_LAUNCH(ctx, "begin_gather_words", begin_gather_words_kernel<<<blocks_for(n, EW_THREADS), EW_THREADS, 0, st>>>(
                                         n, P, idx_bits, drop, in.packed, in.packed_end, in.pos_bits, other, I, reinterpret_cast<uint32_t*>(third), KEY,
                                         tile_xf, single, reinterpret_cast<uint32_t*>(d_flag)));
      SWG_KERNEL_CHECK(ctx);
      uint64_t long_run = 0;
      SWG_TRY(swg_read_scalars(ctx, d_flag, &long_run, 1));
      if ((uint32_t)long_run == 0) {
        S = other;
        E = reinterpret_cast<uint32_t*>(third);
        SWG_LAUNCH(ctx, "tile_x_pairs", tile_x_pairs_kernel<<<blocks_for(ntiles, EW_THREADS), EW_THREADS, 0, st>>>(ntiles, tile_xf, tile_x, TB / TBF));
        SWG_KERNEL_CHECK(ctx);
        if (in.sorted_idx_out) {
          SWG_HIP(ctx, hipMemcpyAsync(in.sorted_idx_out, I, n * sizeof(uint32_t), hipMemcpyDeviceToDevice, st));
          if (in.sorted_idx_valid) *in.sorted_idx_valid = 1;
        }
        return SWG_OK;
      }
      ++ctx->sort_drop_level;
      static const bool dbg = getenv("SWG_DEBUG") != nullptr;
      if (dbg) fprintf(stderr, "[swg] sweep: runs of equal starts >> %d longer than %d: sorting again on the whole key\n", drop, SWG_RUN_HALO);
      swg_arena_restore(ctx, third_mark);
      SWG_HIP(ctx, hipMemsetAsync(prehist, 0, sizeof(uint32_t) * SWG_RADIX_MAX_PASSES * SWG_RADIX_BINS, st));
    }
    {
      const unsigned full = blocks_for(n, EW_THREADS), cap = (unsigned)ctx->num_cu * 16;
      SWG_LAUNCH(ctx, "begin_build", begin_build_kernel<<<(prehist && full > cap) ? cap : full, EW_THREADS, 0, st>>>(
                                         n, in.seg, in.seg_a, in.seg_b, in.seg_table, in.seg_mul, in.start, in.alive, in.pos_bits, S,
                                         packed_sort ? nullptr : I, packed_sort ? swg_radix_plan_packed(key_bits) : swg_radix_plan_pairs(0, key_bits),
                                         prehist));
    }
    SWG_KERNEL_CHECK(ctx);
    uint64_t* P = nullptr;
    int prc = SWG_ERR_UNSUPPORTED;
    if (packed_sort) prc = swg_radix_sort_packed(ctx, S, nullptr, S2, n, key_bits, idx_bits, prehist, &P);
    if (packed_sort && prc == SWG_ERR_UNSUPPORTED) return swg_set_error(ctx, SWG_ERR_HIP, "packed sort declined a shape it accepted");
    SWG_HIP(ctx, hipMemsetAsync(single, 0, n, st));
    if (prc == SWG_OK) {
      // 8-byte passes after the first; the sorted packed words sit in S or S2, the other one and a third buffer take S and E
      uint64_t* other = P == S ? S2 : S;
      uint64_t* third = swg_alloc<uint64_t>(ctx, n);
      SWG_CHECK_ARENA(ctx);
      S = other;
      E = reinterpret_cast<uint32_t*>(third);
      SWG_LAUNCH(ctx, "begin_gather_packed", begin_gather_packed_kernel<<<blocks_for(n, EW_THREADS), EW_THREADS, 0, st>>>(
                                          n, P, idx_bits, in.packed, in.packed_end, in.pos_bits, S, I, E, KEY, tile_xf, single));
      SWG_KERNEL_CHECK(ctx);
    } else if (prc != SWG_ERR_UNSUPPORTED) {
      return prc;
    } else {
      SWG_TRY(swg_radix_sort_pairs(ctx, &S, &I, &S2, &I2, n, 0, key_bits, prehist));
      E = reinterpret_cast<uint32_t*>(S2);  // the sort's scratch key buffer is free again
      SWG_LAUNCH(ctx, "begin_gather", begin_gather_kernel<<<blocks_for(n, EW_THREADS), EW_THREADS, 0, st>>>(
                                          n, S, I, in.end, in.score_key, in.packed, in.packed_end, in.pos_bits, E, KEY, tile_xf, single));
      SWG_KERNEL_CHECK(ctx);
    }
    SWG_LAUNCH(ctx, "tile_x_pairs", tile_x_pairs_kernel<<<blocks_for(ntiles, EW_THREADS), EW_THREADS, 0, st>>>(ntiles, tile_xf, tile_x, TB / TBF));
    SWG_KERNEL_CHECK(ctx);
    if (in.sorted_idx_out) {
      SWG_HIP(ctx, hipMemcpyAsync(in.sorted_idx_out, I, n * sizeof(uint32_t), hipMemcpyDeviceToDevice, st));
      if (in.sorted_idx_valid) *in.sorted_idx_valid = 1;
    }
    return SWG_OK;
  };

  if (k == SWG_K_INF) {
    uint32_t* n_zero = swg_alloc<uint32_t>(ctx, 2);
    SWG_CHECK_ARENA(ctx);
    SWG_HIP(ctx, hipMemsetAsync(n_zero, 0, 8, st));
    SWG_LAUNCH(ctx, "kinf_mark", kinf_mark_kernel<<<blocks_for(n, EW_THREADS), EW_THREADS, 0, st>>>(n, in.start, in.end, in.alive, in.and_with,
                                                                                        keep, n_zero));
    SWG_KERNEL_CHECK(ctx);
    uint64_t h = 0;
    SWG_TRY(swg_read_scalars(ctx, reinterpret_cast<uint64_t*>(n_zero), &h, 1));
    if ((uint32_t)h != 0) {  // zero-length intervals exist: need segment sizes -> sort the begins once
      SWG_TRY(sort_begins());
      SWG_LAUNCH(ctx, "kinf_single", kinf_single_kernel<<<blocks_for(n, EW_THREADS), EW_THREADS, 0, st>>>(n, single, in.and_with, keep));
      SWG_KERNEL_CHECK(ctx);
    }
    swg_arena_restore(ctx, mark);
    return SWG_OK;
  }

  // k = 1 over a pair-grouped input: the sweep of every segment that fits runs in the LDS residency of its sort
  // (swg_seg_sweep_k1) and answers into `keep`; what is left for the kernels below are the begins of the longest segments.
  uint64_t nbg = n;      // begins in S / I / E / KEY
  bool compact = false;  // ... those of the longest segments only, nothing in front of them
  bool lone_mode = false;  // ... those that overlap another interval of their segment (the others are settled: `single`)
  if (k == 1 && in.seg_runs && !in.sorted_idx_out) {
    S = swg_alloc<uint64_t>(ctx, n);
    I = swg_alloc<uint32_t>(ctx, n);
    uint32_t* E2 = swg_alloc<uint32_t>(ctx, n);
    KEY = swg_alloc<uint64_t>(ctx, n);
    tile_x = swg_alloc<uint64_t>(ctx, (size_t)ntiles + 1);
    tile_xf = swg_alloc<uint64_t>(ctx, (size_t)ntilesf + 1);
    single = swg_alloc<uint8_t>(ctx, n);
    SWG_CHECK_ARENA(ctx);
    SWG_HIP(ctx, hipMemsetAsync(keep, 0, n, st));
    int outcome = 0;
    uint64_t left = 0;
    SWG_TRY(swg_seg_sweep_k1(ctx, in, thr, keep, S, I, E2, KEY, tile_xf, single, &left, &outcome));
    if (outcome == 1) {
      swg_arena_restore(ctx, mark);
      return SWG_OK;
    }
    if (outcome == 2) {
      E = E2;
      nbg = left;
      compact = true;
      ntiles = (uint32_t)((nbg + TB - 1) / TB);
      ntilesf = (uint32_t)((nbg + TBF - 1) / TBF);
      SWG_LAUNCH(ctx, "tile_x_pairs", tile_x_pairs_kernel<<<blocks_for(ntiles, EW_THREADS), EW_THREADS, 0, st>>>(ntiles, tile_xf, tile_x, TB / TBF));
      SWG_KERNEL_CHECK(ctx);
    } else {
      swg_arena_restore(ctx, mark);
      S = nullptr;
      I = nullptr;
      KEY = tile_x = tile_xf = nullptr;
      single = nullptr;
    }
  }
  if (!compact) SWG_TRY(sort_begins());
  if (tile_flag && !compact) {
    // the begins that overlap another interval of their segment, compacted (order kept): all the routing and the tile kernels see
    swg_flag_scan fs;
    uint64_t* d_left = swg_alloc<uint64_t>(ctx, 1);
    SWG_CHECK_ARENA(ctx);
    SWG_HIP(ctx, hipMemsetAsync(d_left, 0, sizeof(uint64_t), st));
    SWG_TRY(swg_flags_count(ctx, tile_flag, n, &fs, d_left));
    uint64_t left = 0;
    SWG_TRY(swg_read_scalars(ctx, d_left, &left, 1));
    static const bool dbg = getenv("SWG_DEBUG") != nullptr;
    if (dbg) fprintf(stderr, "[swg] sweep: %llu of %llu begins overlap another interval of their segment and go to the tile kernels\n", (unsigned long long)left, (unsigned long long)n);
    lone_mode = true;  // (the kept lone intervals are in `single`: the full combine below merges them with the tile kernels' flags)
    uint32_t* list = swg_alloc<uint32_t>(ctx, left + 1);
    uint64_t* S2 = swg_alloc<uint64_t>(ctx, left);
    uint32_t* I2 = swg_alloc<uint32_t>(ctx, left);
    uint32_t* E3 = swg_alloc<uint32_t>(ctx, left);
    uint64_t* KEY2 = swg_alloc<uint64_t>(ctx, left);
    SWG_CHECK_ARENA(ctx);
    if (left == 0) {  // nothing overlaps anything: the kept records are exactly the marked ones
      const size_t n_pad0 = ((size_t)n + 255) & ~size_t(255);
      uint8_t* zero = swg_alloc<uint8_t>(ctx, 2 * n_pad0);
      SWG_CHECK_ARENA(ctx);
      SWG_HIP(ctx, hipMemsetAsync(zero, 0, 2 * n_pad0, st));
      const uintptr_t ptrs0 = reinterpret_cast<uintptr_t>(in.alive) | reinterpret_cast<uintptr_t>(single) | reinterpret_cast<uintptr_t>(zero) |
                              reinterpret_cast<uintptr_t>(in.and_with) | reinterpret_cast<uintptr_t>(keep);
      SWG_LAUNCH(ctx, "combine", combine_kernel<<<blocks_for((n + 15) / 16, EW_THREADS), EW_THREADS, 0, st>>>(n, in.alive, single, zero, zero + n_pad0, in.and_with,
                                                                                                   keep, (ptrs0 & 15) == 0));
      SWG_KERNEL_CHECK(ctx);
      swg_arena_restore(ctx, mark);
      return SWG_OK;
    }
    SWG_TRY(swg_flags_compact(ctx, fs, list));
    SWG_LAUNCH(ctx, "begin_compact", begin_compact_kernel<<<blocks_for(left, EW_THREADS), EW_THREADS, 0, st>>>(left, list, S, I, E, KEY, S2, I2, E3, KEY2));
    SWG_KERNEL_CHECK(ctx);
    S = S2;
    I = I2;
    E = E3;
    KEY = KEY2;
    nbg = left;
    compact = true;
    seg_view.valid = 0;  // (the streaming sweep reads the segments' own stretches: not these)
    ntiles = (uint32_t)((nbg + TB - 1) / TB);
    ntilesf = (uint32_t)((nbg + TBF - 1) / TBF);
    SWG_LAUNCH(ctx, "tile_x_stride", tile_x_stride_kernel<<<blocks_for(ntilesf, EW_THREADS), EW_THREADS, 0, st>>>(ntilesf, S, TBF, tile_xf));
    SWG_KERNEL_CHECK(ctx);
    SWG_LAUNCH(ctx, "tile_x_pairs", tile_x_pairs_kernel<<<blocks_for(ntiles, EW_THREADS), EW_THREADS, 0, st>>>(ntiles, tile_xf, tile_x, TB / TBF));
    SWG_KERNEL_CHECK(ctx);
  }
  if (k == 1 && seg_view.valid && !compact) {
    // the begins were sorted segment by segment: every segment's sweep streams through LDS (swg_seg_stream_sweep_k1) and answers
    // into `keep`; on deep data it declines and the tile kernels below take the axis over the same arrays
    SWG_HIP(ctx, hipMemsetAsync(keep, 0, n, st));
    int swept = 0;
    SWG_TRY(swg_seg_stream_sweep_k1(ctx, seg_view, S, I, E, KEY, in.pos_bits, thr, in.and_with, keep, n, &swept));
    if (swept) {
      swg_arena_restore(ctx, mark);
      return SWG_OK;
    }
  }
  uint32_t* te = swg_alloc<uint32_t>(ctx, nbg);
  const size_t n_pad = ((size_t)n + 255) & ~size_t(255);  // keeps `ovl` 16-byte aligned for combine's vector loads
  uint8_t* flags = swg_alloc<uint8_t>(ctx, 2 * n_pad);  // top | ovl
  // SWG_TILE_SMALL=64|128 (experiment knob): k = 1 tiles of that many begins -- a work-group of one or two wavefronts
  static const int small_knob = getenv("SWG_TILE_SMALL") ? atoi(getenv("SWG_TILE_SMALL")) : 0;
  const uint32_t small_tile = (k == 1 && ntiles > 1 && (small_knob == 64 || small_knob == 128)) ? (uint32_t)small_knob : 0u;
  static const bool force512_ = getenv("SWG_TILE_512") != nullptr, force256_ = getenv("SWG_TILE_256") != nullptr;
  const bool auto_tile = k == 1 && ntiles > 1 && !small_tile && !force512_ && !force256_;  // chosen on the device: 128 / 256 / 512
  const uint32_t nt_max = small_tile ? (uint32_t)((nbg + small_tile - 1) / small_tile) : auto_tile ? ntilesf : ntiles;  // finest tiling in use
  uint32_t* cnts = swg_alloc<uint32_t>(ctx, 2 * ((size_t)nt_max + 1));  // carry_cnt | carry_cur
  uint64_t* d_total = swg_alloc<uint64_t>(ctx, 2);
  SWG_CHECK_ARENA(ctx);
  uint8_t* top = flags;
  uint8_t* ovl = flags + n_pad;
  uint32_t* carry_cnt = cnts;
  uint32_t* carry_cur = cnts + ((size_t)nt_max + 1);
  SWG_HIP(ctx, hipMemsetAsync(flags, 0, 2 * n_pad, st));
  SWG_HIP(ctx, hipMemsetAsync(cnts, 0, sizeof(uint32_t) * 2 * ((size_t)nt_max + 1), st));
  // Deep data (a tile's carry-in list is long: one chromosome pair at depth 165 has ~150 per 256-begin tile): tiles of 512
  // begins -- half as many carry-in entries to route and stage, half as many tiles -- are faster there for k = 1 (S-big1 sweep
  // 9.6 -> 8.8 ms) and slower on sparse data (S-pan's tile kernel 3.0 -> 3.7 ms), so the tiling is chosen by an estimate of
  // the average list length from every 1021st begin.  The choice is made on the device (the routing pass reads the estimate)
  // and comes back with the carry-in total: no extra synchronisation.  SWG_TILE_512=1 / SWG_TILE_256=1 (test knobs) force
  // either.
  constexpr uint32_t EST_STRIDE = 1021;  // (prime: the samples fall on every position inside a tile)
  uint32_t tile_size = TB;
  int mode = 0;
  uint64_t* tile_x2 = tile_x;
  uint32_t ntiles2 = ntiles;
  SWG_HIP(ctx, hipMemsetAsync(d_total, 0, 2 * sizeof(uint64_t), st));
  if (small_tile) {
    mode = 3;
    ntiles2 = nt_max;
    tile_x2 = swg_alloc<uint64_t>(ctx, (size_t)ntiles2 + 1);
    SWG_CHECK_ARENA(ctx);
    SWG_LAUNCH(ctx, "tile_x_stride", tile_x_stride_kernel<<<blocks_for(ntiles2, EW_THREADS), EW_THREADS, 0, st>>>(ntiles2, S, small_tile, tile_x2));
    SWG_KERNEL_CHECK(ctx);
  } else if (k == 1 && ntiles > 1) {
    static const bool force512 = getenv("SWG_TILE_512") != nullptr, force256 = getenv("SWG_TILE_256") != nullptr;
    mode = force512 ? 1 : force256 ? 0 : 2;
    if (mode) {
      ntiles2 = (uint32_t)((nbg + 2 * TB - 1) / (2 * TB));
      tile_x2 = swg_alloc<uint64_t>(ctx, (size_t)ntiles2 + 1);
      SWG_CHECK_ARENA(ctx);
      SWG_LAUNCH(ctx, "tile_x_pairs", tile_x_pairs_kernel<<<blocks_for(ntiles2, EW_THREADS), EW_THREADS, 0, st>>>(ntiles2, tile_x, tile_x2));
      SWG_KERNEL_CHECK(ctx);
    }
    if (mode == 2) {
      SWG_LAUNCH(ctx, "route_estimate", route_estimate_kernel<<<blocks_for((nbg + EST_STRIDE - 1) / EST_STRIDE, EW_THREADS), EW_THREADS, 0, st>>>(
                                            nbg, EST_STRIDE, S, E, in.pos_bits, tile_x, ntiles, reinterpret_cast<unsigned long long*>(d_total + 1)));
      SWG_KERNEL_CHECK(ctx);
    }
  }
  SWG_LAUNCH(ctx, "route_count", route_count_kernel<<<blocks_for((nbg + 3) / 4, EW_THREADS), EW_THREADS, 0, st>>>(
                                     nbg, S, E, in.pos_bits, tile_x, ntiles, tile_x2, ntiles2, mode, reinterpret_cast<unsigned long long*>(d_total + 1),
                                     EST_STRIDE, te, carry_cnt, small_tile, tile_xf, ntilesf));
  SWG_KERNEL_CHECK(ctx);
  // (scanned over the finest tiling's length either way: the entries past a coarser tiling's end are zero)
  SWG_TRY(swg_exclusive_scan_u32(ctx, carry_cnt, carry_cnt, (uint64_t)nt_max + 1, d_total));
  uint64_t h2[2] = {0, 0};
  SWG_TRY(swg_read_scalars(ctx, d_total, h2, 2));
  const uint64_t n_carry = h2[0];
  if (mode == 3) {
    tile_size = small_tile;
    tile_x = tile_x2;
    ntiles = ntiles2;
  } else if (mode == 2 && h2[1] * EST_STRIDE / ntiles < SPARSE_CARRY_PER_TILE) {  // the kernel's rule (sparse)
    tile_size = TBF;
    tile_x = tile_xf;
    ntiles = ntilesf;
  } else if (mode == 1 || (mode == 2 && h2[1] * EST_STRIDE / ntiles >= DEEP_CARRY_PER_TILE)) {  // the kernel's rule (deep)
    tile_size = 2 * TB;
    tile_x = tile_x2;
    ntiles = ntiles2;
  }
  {
    static const bool dbg = getenv("SWG_DEBUG") != nullptr;
    if (dbg && mode == 2)
      fprintf(stderr, "[swg] sweep: n %llu, estimated carry-ins per 256-begin tile %.2f -> tiles of %u begins, %llu carry-ins\n",
              (unsigned long long)nbg, (double)h2[1] * EST_STRIDE / (double)((nbg + TB - 1) / TB), tile_size, (unsigned long long)n_carry);
  }
  uint64_t* c_s = swg_alloc<uint64_t>(ctx, n_carry + 1);
  uint64_t* c_e = swg_alloc<uint64_t>(ctx, n_carry + 1);
  uint64_t* c_key = swg_alloc<uint64_t>(ctx, n_carry + 1);
  uint32_t* c_id = swg_alloc<uint32_t>(ctx, n_carry + 1);
  SWG_CHECK_ARENA(ctx);
  if (n_carry) {
    SWG_LAUNCH(ctx, "route_fill", route_fill_kernel<<<blocks_for(nbg, EW_THREADS), EW_THREADS, 0, st>>>(
                                      nbg, S, E, in.pos_bits, KEY, I, tile_size, te, carry_cnt, carry_cur, c_s, c_e, c_key, c_id));
    SWG_KERNEL_CHECK(ctx);
  }
  TileArgs ta;
  ta.n = nbg;
  ta.S = S;
  ta.E = E;
  ta.pos_bits = in.pos_bits;
  ta.KEY = KEY;
  ta.I = I;
  ta.tile_x = tile_x;
  ta.ntiles = ntiles;
  ta.carry_off = carry_cnt;
  ta.c_s = c_s;
  ta.c_e = c_e;
  ta.c_key = c_key;
  ta.c_id = c_id;
  ta.k = k;
  ta.thr = thr;
  ta.top = top;
  ta.ovl = ovl;
  ta.tile_done = nullptr;
  if (k == 1) {
    if (tile_size == 64u)
      SWG_LAUNCH(ctx, "sweep_tile_k1", sweep_tile_k1_kernel<64><<<ntiles, 64, 0, st>>>(ta));
    else if (tile_size == 128u)
      SWG_LAUNCH(ctx, "sweep_tile_k1", sweep_tile_k1_kernel<128><<<ntiles, 128, 0, st>>>(ta));
    else if (tile_size == (uint32_t)TB)
      SWG_LAUNCH(ctx, "sweep_tile_k1", sweep_tile_k1_kernel<TB><<<ntiles, TB, 0, st>>>(ta));
    else
      SWG_LAUNCH(ctx, "sweep_tile_k1", sweep_tile_k1_kernel<2 * TB><<<ntiles, 2 * TB, 0, st>>>(ta));
#ifdef SWG_TILE_TIMING
    {
      (void)hipStreamSynchronize(st);
      unsigned long long ht[8], z[8] = {0};
      (void)hipMemcpyFromSymbol(ht, HIP_SYMBOL(g_tile_t), sizeof ht);
      fprintf(stderr, "[swg] sweep_tile_k1 phases (%u tiles of %u; 100 MHz ticks summed over tiles): loads+carry %llu, starts %llu, own ends %llu, carried ends %llu, last %llu\n",
              (unsigned)ntiles, (unsigned)tile_size, ht[0], ht[1], ht[2], ht[3], ht[4]);
      (void)hipMemcpyToSymbol(HIP_SYMBOL(g_tile_t), z, sizeof z);
    }
#endif
  } else {
    static const bool no_prune = getenv("SWG_KN_PLAIN") != nullptr;  // test knob: every tile through the plain kernel
    if (k <= (uint64_t)KSTAR_MAX && !no_prune) {
      ta.tile_done = swg_alloc<uint8_t>(ctx, ntiles);
      SWG_CHECK_ARENA(ctx);
      SWG_LAUNCH(ctx, "sweep_tile_kp", sweep_tile_kp_kernel<<<ntiles, TB, 0, st>>>(ta));
      SWG_KERNEL_CHECK(ctx);
    }
    SWG_LAUNCH(ctx, "sweep_tile_kn", sweep_tile_kn_kernel<<<ntiles, TB, 0, st>>>(ta));
  }
  SWG_KERNEL_CHECK(ctx);
  if (compact && !lone_mode) {  // (the other records' answers are in `keep` already)
    SWG_LAUNCH(ctx, "combine_begins", combine_begins_kernel<<<blocks_for(nbg, EW_THREADS), EW_THREADS, 0, st>>>(nbg, I, top, ovl, in.and_with, keep));
  } else {
    const uintptr_t ptrs = reinterpret_cast<uintptr_t>(in.alive) | reinterpret_cast<uintptr_t>(single) | reinterpret_cast<uintptr_t>(top) |
                           reinterpret_cast<uintptr_t>(ovl) | reinterpret_cast<uintptr_t>(in.and_with) | reinterpret_cast<uintptr_t>(keep);
    SWG_LAUNCH(ctx, "combine", combine_kernel<<<blocks_for((n + 15) / 16, EW_THREADS), EW_THREADS, 0, st>>>(n, in.alive, single, top, ovl,
                                                                                                   in.and_with, keep, (ptrs & 15) == 0));
  }
  SWG_KERNEL_CHECK(ctx);
  swg_arena_restore(ctx, mark);
  return SWG_OK;
}
