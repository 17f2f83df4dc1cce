// Scaffold stage, part 1 (src/paf_filter.rs:750-851): sort A and the best-buddy predecessor selection.
//
//   sort A      all step-1 survivors by (query seq, target seq, strand, q_start), stable
//               (groups of merge_mappings_into_chains, paf_filter.rs:761-777; ties fall to input order)
//   chaining    best-buddy predecessor selection (paf_filter.rs:784-851): units cut where no window can straddle,
//               parallel candidate lists, then the reference's sequential greedy per unit (one lane, one wavefront or
//               block-speculative, by unit length)
#include <cmath>
#include <type_traits>

#include "swg_scaffold_internal.h"

namespace swg_scaf {
namespace {

// ---- sort A -----------------------------------------------------------------------------------------
// key = (((q * n_seq + t) * 2 + strand) << pos_bits) | q_start      value = original index
__global__ __launch_bounds__(EW) void sortA_keys_kernel(uint64_t M, const uint32_t* __restrict__ a_idx,
                                                        const uint32_t* __restrict__ q_id,
                                                        const uint32_t* __restrict__ t_id,
                                                        const uint8_t* __restrict__ strand,
                                                        const uint32_t* __restrict__ q_start, uint32_t n_seq,
                                                        int pos_bits, uint64_t* __restrict__ key) {
  // a_idx is either ascending (coalesced reads) or the query axis' sorted order (gathers: neighbouring blocks on one XCD)
  uint64_t a = (uint64_t)swg_xcd_block(blockIdx.x, gridDim.x) * EW + threadIdx.x;
  if (a >= M) return;
  const uint32_t i = a_idx[a];
  const uint64_t g = ((uint64_t)q_id[i] * n_seq + t_id[i]) * 2 + (strand[i] ? 1 : 0);
  key[a] = (g << pos_bits) | q_start[i];
}

// Sort A behind the mapping sweep, as words (round 4): the query axis' order already has every (query, target, strand) group
// in q_start order, so only the group bits are sorted -- stably -- and q_start is not needed for that: word = (group <<
// idx_bits) | record index, three gathers instead of four, 8-byte passes instead of 12-byte (key, index) pairs.  The full
// key is put together again by gatherA_slots_words_kernel, which reads the record's slot anyway.
__global__ __launch_bounds__(EW) void sortA_words_kernel(uint64_t M, const uint32_t* __restrict__ a_idx,
                                                         const uint32_t* __restrict__ q_id, const uint32_t* __restrict__ t_id,
                                                         const uint8_t* __restrict__ strand, uint32_t n_seq, int idx_bits,
                                                         uint64_t* __restrict__ words,
                                                         const uint32_t* __restrict__ group32) {  // (optional: the same value, from prepare)
  uint64_t a = (uint64_t)swg_xcd_block(blockIdx.x, gridDim.x) * EW + threadIdx.x;
  if (a >= M) return;
  const uint32_t i = a_idx[a];
  const uint64_t g = group32 ? (uint64_t)group32[i] : ((uint64_t)q_id[i] * n_seq + t_id[i]) * 2 + (strand[i] ? 1 : 0);
  words[a] = (g << idx_bits) | i;
}

// The same for an ascending (or identity: a_idx == nullptr, then idx_out receives it) index list, with the digit histograms
// of the sort that follows accumulated on the way (the sort then skips its own pass over the keys, as in the sweep's
// begin_build).  Grid-stride over whole work-groups.
__global__ __launch_bounds__(EW) void sortA_keys_hist_kernel(uint64_t M, const uint32_t* __restrict__ a_idx,
                                                             uint32_t* __restrict__ idx_out,
                                                             const uint32_t* __restrict__ q_id,
                                                             const uint32_t* __restrict__ t_id,
                                                             const uint8_t* __restrict__ strand,
                                                             const uint32_t* __restrict__ q_start, uint32_t n_seq,
                                                             int pos_bits, uint64_t* __restrict__ key, swg_radix_plan plan,
                                                             uint32_t* __restrict__ ghist, int drop = 0, int idx_bits = 0) {
  // drop > 0: sort on the truncated key -- key[a] becomes the WORD ((key >> drop) << idx_bits) | record index
  // (swg_radix_sort_words; gather_all_words_kernel orders the runs of equal truncated keys), histograms of key >> drop
  __shared__ uint32_t h[SWG_RADIX_MAX_PASSES][SWG_RADIX_BINS];
  const int npasses = plan.npasses;
  swg_radix_hist_zero(h, npasses);
  __syncthreads();
  for (uint64_t base = (uint64_t)blockIdx.x * EW; base < M; base += (uint64_t)gridDim.x * EW) {
    const uint64_t a = base + threadIdx.x;
    const bool in = a < M;
    uint64_t k = 0;
    if (in) {
      const uint32_t i = a_idx ? a_idx[a] : (uint32_t)a;
      const uint64_t g = ((uint64_t)q_id[i] * n_seq + t_id[i]) * 2 + (strand[i] ? 1 : 0);
      k = (g << pos_bits) | q_start[i];
      if (drop) {
        k >>= drop;
        key[a] = (k << idx_bits) | i;
      } else {
        key[a] = k;
        if (idx_out) idx_out[a] = i;
      }
    }
    swg_radix_hist_add(h, k, in, plan);
  }
  __syncthreads();
  swg_radix_hist_flush(h, npasses, ghist);
}

// After sort A: per A-position columns + pair boundaries.
__global__ __launch_bounds__(EW) void gatherA_kernel(uint64_t M, const uint64_t* __restrict__ keyA,
                                                     const uint32_t* __restrict__ idxA,
                                                     const uint32_t* __restrict__ q_end,
                                                     const uint32_t* __restrict__ t_start,
                                                     const uint32_t* __restrict__ t_end,
                                                     const uint8_t* __restrict__ keep1, int pos_bits,
                                                     uint32_t* __restrict__ a_qe, uint32_t* __restrict__ a_ts,
                                                     uint32_t* __restrict__ a_te, uint8_t* __restrict__ a_keep,
                                                     uint32_t* __restrict__ pair_flag) {
  uint64_t a = (uint64_t)swg_xcd_block(blockIdx.x, gridDim.x) * EW + threadIdx.x;
  if (a >= M) return;
  const uint32_t i = idxA[a];
  a_qe[a] = q_end[i];
  a_ts[a] = t_start[i];
  a_te[a] = t_end[i];
  a_keep[a] = keep1[i] ? 1 : 0;
  const uint64_t pair = keyA[a] >> (pos_bits + 1);
  pair_flag[a] = (a == 0 || (keyA[a - 1] >> (pos_bits + 1)) != pair) ? 1u : 0u;
}

// The same from the 32-byte record slots prepare wrote for the mapping sweep (swg_key_ends: both ends, both starts, matches
// and block length in one sector): one gather per record instead of three, and matches / block length come along for gatherS.
__global__ __launch_bounds__(EW) void gatherA_slots_kernel(uint64_t M, const uint64_t* __restrict__ keyA,
                                                           const uint32_t* __restrict__ idxA,
                                                           const swg_key_ends* __restrict__ slots,
                                                           const uint8_t* __restrict__ keep1, int pos_bits,
                                                           uint32_t* __restrict__ a_qe, uint32_t* __restrict__ a_ts,
                                                           uint32_t* __restrict__ a_te, uint32_t* __restrict__ a_m,
                                                           uint32_t* __restrict__ a_b, uint8_t* __restrict__ a_keep,
                                                           uint32_t* __restrict__ pair_flag) {
  uint64_t a = (uint64_t)swg_xcd_block(blockIdx.x, gridDim.x) * EW + threadIdx.x;
  if (a >= M) return;
  const uint32_t i = idxA[a];
  const swg_key_ends ke = slots[i];
  a_qe[a] = ke.end[0];
  a_ts[a] = ke.start[1];
  a_te[a] = ke.end[1];
  a_m[a] = ke.pad[0];
  a_b[a] = ke.pad[1];
  a_keep[a] = keep1[i] ? 1 : 0;
  const uint64_t pair = keyA[a] >> (pos_bits + 1);
  pair_flag[a] = (a == 0 || (keyA[a - 1] >> (pos_bits + 1)) != pair) ? 1u : 0u;
}

// ... from sort A's words (sortA_words_kernel): keyA and idxA are written here
__global__ __launch_bounds__(EW) void gatherA_slots_words_kernel(uint64_t M, const uint64_t* __restrict__ words, int idx_bits,
                                                                 const swg_key_ends* __restrict__ slots,
                                                                 const uint8_t* __restrict__ keep1, int pos_bits,
                                                                 uint64_t* __restrict__ keyA, uint32_t* __restrict__ idxA,
                                                                 uint32_t* __restrict__ a_qe, uint32_t* __restrict__ a_ts,
                                                                 uint32_t* __restrict__ a_te, uint32_t* __restrict__ a_m,
                                                                 uint32_t* __restrict__ a_b, uint8_t* __restrict__ a_keep,
                                                                 uint32_t* __restrict__ pair_flag) {
  uint64_t a = (uint64_t)swg_xcd_block(blockIdx.x, gridDim.x) * EW + threadIdx.x;
  if (a >= M) return;
  const uint64_t w = words[a];
  const uint32_t i = (uint32_t)(w & ((uint64_t(1) << idx_bits) - 1));
  const uint64_t g = w >> idx_bits;  // ((query * n_seq + target) << 1) | strand
  const swg_key_ends ke = slots[i];
  keyA[a] = (g << pos_bits) | ke.start[0];
  idxA[a] = i;
  a_qe[a] = ke.end[0];
  a_ts[a] = ke.start[1];
  a_te[a] = ke.end[1];
  a_m[a] = ke.pad[0];
  a_b[a] = ke.pad[1];
  a_keep[a] = keep1[i] ? 1 : 0;
  pair_flag[a] = (a == 0 || ((words[a - 1] >> idx_bits) >> 1) != (g >> 1)) ? 1u : 0u;
}

// dense pair id of every A position: inclusive count of pair heads - 1 (scan result is exclusive)
__global__ __launch_bounds__(EW) void dense_from_scan_kernel(uint64_t M, const uint32_t* __restrict__ excl,
                                                             const uint32_t* __restrict__ flag,
                                                             uint32_t* __restrict__ dense) {
  uint64_t a = (uint64_t)blockIdx.x * EW + threadIdx.x;
  if (a < M) dense[a] = excl[a] + flag[a] - 1;
}

// ---- survivors (mapping-sweep survivors in A order) -----------------------------------------------------
// The same when every record of sort A is a member (no mapping-level filter ran, or it kept everything): positions
// coincide, so the ends / targets / indices of sort A are used as they are and only what sort A does not hold is produced.
__global__ __launch_bounds__(EW) void gatherS_all_kernel(uint64_t m, const uint64_t* __restrict__ keyA,
                                                         const uint32_t* __restrict__ idxA,
                                                         const uint32_t* __restrict__ matches,
                                                         const uint32_t* __restrict__ block_len, int pos_bits,
                                                         uint32_t* __restrict__ s_qs, uint32_t* __restrict__ s_m,
                                                         uint32_t* __restrict__ s_b, uint64_t* __restrict__ s_grp,
                                                         uint32_t* __restrict__ head_flag) {
  uint64_t p = (uint64_t)swg_xcd_block(blockIdx.x, gridDim.x) * EW + threadIdx.x;
  if (p >= m) return;
  const uint64_t k = keyA[p];
  const uint32_t i = idxA[p];
  s_qs[p] = (uint32_t)(k & ((uint64_t(1) << pos_bits) - 1));
  s_m[p] = matches[i];
  s_b[p] = block_len[i];
  const uint64_t g = k >> pos_bits;
  s_grp[p] = g;
  head_flag[p] = (p == 0 || (keyA[p - 1] >> pos_bits) != g) ? 1u : 0u;
}
__global__ __launch_bounds__(EW) void gatherS_kernel(uint64_t m, const uint32_t* __restrict__ s_a,
                                                     const uint64_t* __restrict__ keyA,
                                                     const uint32_t* __restrict__ idxA,
                                                     const uint32_t* __restrict__ a_qe,
                                                     const uint32_t* __restrict__ a_ts,
                                                     const uint32_t* __restrict__ a_te,
                                                     const uint32_t* __restrict__ matches,
                                                     const uint32_t* __restrict__ block_len,
                                                     const uint32_t* __restrict__ a_m, const uint32_t* __restrict__ a_b,
                                                     int pos_bits,
                                                     uint32_t* __restrict__ s_qs, uint32_t* __restrict__ s_qe,
                                                     uint32_t* __restrict__ s_ts, uint32_t* __restrict__ s_te,
                                                     uint32_t* __restrict__ s_m, uint32_t* __restrict__ s_b,
                                                     uint32_t* __restrict__ s_idx, uint64_t* __restrict__ s_grp,
                                                     uint32_t* __restrict__ head_flag) {
  uint64_t p = (uint64_t)swg_xcd_block(blockIdx.x, gridDim.x) * EW + threadIdx.x;
  if (p >= m) return;
  const uint32_t a = s_a[p];
  const uint64_t k = keyA[a];
  const uint32_t i = idxA[a];
  s_qs[p] = (uint32_t)(k & ((uint64_t(1) << pos_bits) - 1));
  s_qe[p] = a_qe[a];
  s_ts[p] = a_ts[a];
  s_te[p] = a_te[a];
  s_m[p] = a_m ? a_m[a] : matches[i];  // a_m / a_b: by A position, from the record slots (gatherA_slots_kernel); else gathered here
  s_b[p] = a_b ? a_b[a] : block_len[i];
  s_idx[p] = i;
  const uint64_t g = k >> pos_bits;
  s_grp[p] = g;
  bool head = p == 0;
  if (!head) head = (keyA[s_a[p - 1]] >> pos_bits) != g;
  head_flag[p] = head ? 1u : 0u;
}

// Both gathers in one pass when every record of sort A is a member of the chaining (the mapping-level sweep removed
// nothing: the CLI defaults): positions coincide, nothing is compacted, and the two boundary flags -- (query, target) pair
// heads in the high word, (query, target, strand) group heads in the low word -- go through ONE u64 sum scan.
// Boundaries of a 256-element block, counted: (pair boundaries << 32) | group boundaries.  The dense pair and group numbers
// are prefix counts of these flags; per-block counts + a scan over the blocks + the flags recomputed from the sorted keys
// (group_pair_kernel) replace an element-wise 8-byte flag column and its 10^8-element scan.
__device__ __forceinline__ void block_boundary_count(bool pf, bool gf, uint64_t* out) {
  __shared__ uint32_t wc[2][EW / 64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const uint32_t np = (uint32_t)__popcll(__ballot(pf)), ng = (uint32_t)__popcll(__ballot(gf));
  if (lane == 0) {
    wc[0][wave] = np;
    wc[1][wave] = ng;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    uint32_t p = 0, g = 0;
#pragma unroll
    for (int w = 0; w < EW / 64; ++w) {
      p += wc[0][w];
      g += wc[1][w];
    }
    *out = ((uint64_t)p << 32) | g;
  }
}

__global__ __launch_bounds__(EW) void gather_all_kernel(uint64_t M, const uint64_t* __restrict__ keyA,
                                                        const uint32_t* __restrict__ idxA,
                                                        const uint32_t* __restrict__ q_end,
                                                        const uint32_t* __restrict__ t_start,
                                                        const uint32_t* __restrict__ t_end,
                                                        const uint32_t* __restrict__ matches,
                                                        const uint32_t* __restrict__ block_len, int pos_bits,
                                                        uint32_t* __restrict__ s_qs, uint32_t* __restrict__ s_qe,
                                                        uint32_t* __restrict__ s_ts, uint32_t* __restrict__ s_te,
                                                        uint32_t* __restrict__ s_m, uint32_t* __restrict__ s_b,
                                                        uint64_t* __restrict__ s_grp, uint64_t* __restrict__ blk_cnt) {
  const uint32_t lb = swg_xcd_block(blockIdx.x, gridDim.x);
  const uint64_t a = (uint64_t)lb * EW + threadIdx.x;
  bool pf = false, gf = false;  // a (query, target) pair / a (query, target, strand) group begins here
  if (a < M) {
    const uint64_t k = keyA[a];
    const uint32_t i = idxA[a];
    s_qs[a] = (uint32_t)(k & ((uint64_t(1) << pos_bits) - 1));
    s_qe[a] = q_end[i];
    s_ts[a] = t_start[i];
    s_te[a] = t_end[i];
    s_m[a] = matches[i];
    s_b[a] = block_len[i];
    const uint64_t g = k >> pos_bits;
    s_grp[a] = g;
    const uint64_t gp = a ? keyA[a - 1] >> pos_bits : ~0ull;
    pf = (gp >> 1) != (g >> 1);
    gf = gp != g;
  }
  block_boundary_count(pf, gf, blk_cnt + lb);
}
// The same after the packed sort (swg_radix_sort_packed): P[a] = ((key >> 8) << idx_bits) | record index.  The key's low
// 8 bits are the low 8 bits of the start coordinate (pos_bits >= 8 here), which comes from the record like the other
// columns; the full key and the index are written out for the later stages (inversion capture, rescue).
__global__ __launch_bounds__(EW) void gather_all_packed_kernel(uint64_t M, const uint64_t* __restrict__ P, int idx_bits,
                                                               const uint32_t* __restrict__ q_start,
                                                               const uint32_t* __restrict__ q_end,
                                                               const uint32_t* __restrict__ t_start,
                                                               const uint32_t* __restrict__ t_end,
                                                               const uint32_t* __restrict__ matches,
                                                               const uint32_t* __restrict__ block_len, int pos_bits,
                                                               uint64_t* __restrict__ keyA, uint32_t* __restrict__ idxA,
                                                               uint32_t* __restrict__ s_qs, uint32_t* __restrict__ s_qe,
                                                               uint32_t* __restrict__ s_ts, uint32_t* __restrict__ s_te,
                                                               uint32_t* __restrict__ s_m, uint32_t* __restrict__ s_b,
                                                               uint64_t* __restrict__ s_grp, uint64_t* __restrict__ blk_cnt,
                                                               const swg_key_ends* __restrict__ slots,
                                                               const uint32_t* __restrict__ slot_flag) {
  if (slot_flag && *slot_flag == 0) slots = nullptr;  // (wave-uniform; see gather_all_words_kernel)
  const uint32_t lb = swg_xcd_block(blockIdx.x, gridDim.x);
  const uint64_t a = (uint64_t)lb * EW + threadIdx.x;
  bool pf = false, gf = false;
  if (a < M) {
    const uint64_t w = P[a];
    const uint32_t i = (uint32_t)(w & ((uint64_t(1) << idx_bits) - 1));
    const uint64_t hi = w >> idx_bits;  // key >> 8
    uint32_t qs, qe, ts, te, mt, bl;
    if (slots) {  // the record's 32-byte slot (prepare_kernel): one line from L2 instead of six
      const swg_key_ends ke = slots[i];
      qs = ke.start[0];
      ts = ke.start[1];
      qe = ke.end[0];
      te = ke.end[1];
      mt = ke.pad[0];
      bl = ke.pad[1];
    } else {
      qs = q_start[i];
      qe = q_end[i];
      ts = t_start[i];
      te = t_end[i];
      mt = matches[i];
      bl = block_len[i];
    }
    keyA[a] = (hi << 8) | (uint64_t)(qs & 0xffu);
    idxA[a] = i;
    s_qs[a] = qs;
    s_qe[a] = qe;
    s_ts[a] = ts;
    s_te[a] = te;
    s_m[a] = mt;
    s_b[a] = bl;
    const uint64_t g = hi >> (pos_bits - 8);
    s_grp[a] = g;
    const uint64_t gp = a ? (P[a - 1] >> idx_bits) >> (pos_bits - 8) : ~0ull;
    pf = (gp >> 1) != (g >> 1);
    gf = gp != g;
  }
  block_boundary_count(pf, gf, blk_cnt + lb);
}
// The same after a sort on the TRUNCATED key (swg_radix_sort_words): P[a] = ((key >> drop) << idx_bits) | record index, in
// (key >> drop, index) order.  Records whose q_start differ only in the low `drop` bits form short runs; each member counts,
// in LDS (the work-group's 256 words + SWG_RUN_HALO on either side, the halo's low bits gathered only for a run that crosses
// the block's edge), the members that order before it by (low bits, index) and writes its columns at run start + that rank
// (begin_gather_words_kernel in swg_sweep.hip is the same idea).  The boundary counts are taken at the positions BEFORE the
// reordering: a pair / group begins where a run begins, and a run's first position does not move.  A run that reaches beyond
// the halo raises *long_run (the caller sorts again on the whole key).
__global__ __launch_bounds__(EW) void gather_all_words_kernel(uint64_t M, const uint64_t* __restrict__ P, int idx_bits, int drop,
                                                              const uint32_t* __restrict__ q_start,
                                                              const uint32_t* __restrict__ q_end,
                                                              const uint32_t* __restrict__ t_start,
                                                              const uint32_t* __restrict__ t_end,
                                                              const uint32_t* __restrict__ matches,
                                                              const uint32_t* __restrict__ block_len, int pos_bits,
                                                              uint64_t* __restrict__ keyA, uint32_t* __restrict__ idxA,
                                                              uint32_t* __restrict__ s_qs, uint32_t* __restrict__ s_qe,
                                                              uint32_t* __restrict__ s_ts, uint32_t* __restrict__ s_te,
                                                              uint32_t* __restrict__ s_m, uint32_t* __restrict__ s_b,
                                                              uint64_t* __restrict__ s_grp, uint64_t* __restrict__ blk_cnt,
                                                              unsigned long long* __restrict__ long_run,
                                                              const swg_key_ends* __restrict__ slots,
                                                              const uint32_t* __restrict__ slot_flag) {
  // slots != nullptr: the six columns of a record from its 32-byte slot (prepare_kernel) -- one line from L2 per record
  // instead of six.  slot_flag: the slots were only written if *slot_flag != 0 (the input-order probe, swg_filter.hip).
  if (slot_flag && *slot_flag == 0) slots = nullptr;  // (wave-uniform)
  constexpr int H = SWG_RUN_HALO, W = EW + 2 * H;
  __shared__ uint64_t l_hi[W];   // key >> drop, + 1 (0: no element at this position)
  __shared__ uint64_t l_ord[W];  // (low bits of q_start << 32) | record index
  const uint32_t lb = swg_xcd_block(blockIdx.x, gridDim.x);
  const uint64_t p0 = (uint64_t)lb * EW;
  const int t = threadIdx.x;
  const uint64_t idx_mask = (uint64_t(1) << idx_bits) - 1;
  const uint32_t low_mask = (1u << drop) - 1u;
  const uint64_t a = p0 + t;
  uint64_t hi = 0;
  uint32_t i = 0, qs = 0, qe = 0, ts = 0, te = 0, mt = 0, bl = 0;
  if (a < M) {
    const uint64_t w = P[a];
    hi = (w >> idx_bits) + 1;  // (+1: a key of 0 is a real key here -- sequence 0 onto itself at coordinate < 2^drop)
    i = (uint32_t)(w & idx_mask);
    if (slots) {
      const swg_key_ends ke = slots[i];
      qs = ke.start[0];
      ts = ke.start[1];
      qe = ke.end[0];
      te = ke.end[1];
      mt = ke.pad[0];
      bl = ke.pad[1];
    } else {
      qs = q_start[i];
      qe = q_end[i];
      ts = t_start[i];
      te = t_end[i];
      mt = matches[i];
      bl = block_len[i];
    }
  }
  l_hi[H + t] = hi;
  l_ord[H + t] = ((uint64_t)(qs & low_mask) << 32) | i;
  if (t < 2 * H) {
    const bool left = t < H;
    const int64_t q = left ? (int64_t)p0 - H + t : (int64_t)p0 + EW + (t - H);
    const int li = left ? t : EW + t;
    const bool there = q >= 0 && (uint64_t)q < M;
    l_hi[li] = there ? (P[q] >> idx_bits) + 1 : 0ull;
    l_ord[li] = there ? (P[q] & idx_mask) : 0ull;  // (index only so far)
  }
  __syncthreads();
  if (t < 2 * H) {
    const bool left = t < H;
    const int li = left ? t : EW + t;
    const uint64_t edge = left ? l_hi[H] : l_hi[H + EW - 1];
    if (edge != 0 && l_hi[li] == edge) {
      const uint32_t hid = (uint32_t)l_ord[li];
      l_ord[li] = ((uint64_t)((slots ? slots[hid].start[0] : q_start[hid]) & low_mask) << 32) | hid;
    }
  }
  __syncthreads();
  bool pf = false, gf = false;
  if (a < M) {
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
    if (j2 >= W && p0 + EW + H < M) too_long = true;
    uint64_t np = a;
    if (too_long)
      *long_run = 1ull;
    else
      np = a - before + rank;
    const uint64_t kt = hi - 1;  // key >> drop
    keyA[np] = (kt << drop) | (uint64_t)(qs & low_mask);
    idxA[np] = i;
    s_qs[np] = qs;
    s_qe[np] = qe;
    s_ts[np] = ts;
    s_te[np] = te;
    s_m[np] = mt;
    s_b[np] = bl;
    const uint64_t g = kt >> (pos_bits - drop);
    s_grp[np] = g;
    const uint64_t hp = l_hi[H + t - 1];  // the element before, at its position before the reordering (0: none)
    const uint64_t gp = (a && hp) ? (hp - 1) >> (pos_bits - drop) : ~0ull;
    pf = (gp >> 1) != (g >> 1);
    gf = gp != g;
  }
  block_boundary_count(pf, gf, blk_cnt + lb);
}

// after the inclusive sum scan of the blocks' counts: dense pair and group ids, group begins.  The flags are recomputed from
// the sorted keys; a block's 256 prefix counts come from two ballots per wavefront and the wavefronts' totals in LDS.
__global__ __launch_bounds__(EW) void group_pair_kernel(uint64_t M, const uint64_t* __restrict__ keyA, int pos_bits,
                                                        const uint64_t* __restrict__ blk_incl,
                                                        uint32_t* __restrict__ a_dpair, uint32_t* __restrict__ s_gidx,
                                                        uint32_t* __restrict__ head_flag, uint32_t* __restrict__ group_begin) {
  __shared__ uint32_t wc[2][EW / 64];
  const uint64_t a = (uint64_t)blockIdx.x * EW + threadIdx.x;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  bool pf = false, gf = false;
  if (a < M) {
    const uint64_t g = keyA[a] >> pos_bits;
    const uint64_t gp = a ? keyA[a - 1] >> pos_bits : ~0ull;
    pf = (gp >> 1) != (g >> 1);
    gf = gp != g;
  }
  const uint64_t mp = __ballot(pf), mg = __ballot(gf);
  if (lane == 0) {
    wc[0][wave] = (uint32_t)__popcll(mp);
    wc[1][wave] = (uint32_t)__popcll(mg);
  }
  __syncthreads();
  if (a >= M) return;
  const uint64_t base = blockIdx.x ? blk_incl[blockIdx.x - 1] : 0ull;
  const uint64_t upto = lane == 63 ? ~0ull : ((2ull << lane) - 1ull);  // lanes 0..lane
  uint32_t np = (uint32_t)(base >> 32) + (uint32_t)__popcll(mp & upto), ng = (uint32_t)base + (uint32_t)__popcll(mg & upto);
  for (int w = 0; w < wave; ++w) {
    np += wc[0][w];
    ng += wc[1][w];
  }
  a_dpair[a] = np - 1;  // (the first element is a boundary of both kinds: counts are >= 1)
  s_gidx[a] = ng - 1;
  if (head_flag) head_flag[a] = gf ? 1u : 0u;  // (nullptr: nobody reads it -- only the scan-based reductions of few, long groups do)
  if (gf) group_begin[ng - 1] = (uint32_t)a;
}

__global__ __launch_bounds__(EW) void group_bounds_kernel(uint64_t m, const uint32_t* __restrict__ head_flag,
                                                          const uint32_t* __restrict__ gidx_excl,
                                                          uint32_t* __restrict__ s_gidx,
                                                          uint32_t* __restrict__ group_begin) {
  uint64_t p = (uint64_t)blockIdx.x * EW + threadIdx.x;
  if (p >= m) return;
  const uint32_t g = gidx_excl[p] + head_flag[p] - 1;
  s_gidx[p] = g;
  if (head_flag[p]) group_begin[g] = (uint32_t)p;
}

// ---- best-buddy chaining (paf_filter.rs:784-851) -----------------------------------------------------------
// The reference's greedy is sequential in i (later i read best_pred_score[] written by earlier i), but the
// expensive part of a step -- the distance d(i, j) to every j of the window -- does not depend on that state.
// So the work is split:
//   chain_candidates : parallel, one thread per i: the KC smallest (d, j) over the valid j of i's window
//                      (both gaps within the limit), sorted by (d, j); plus how many valid j there were.
//   chain_select     : one wavefront per unit (a group cut where no window can straddle, chain_cuts), sequential
//                      in i: i takes the first of its candidates with d < best_pred_score[j] -- which is the
//                      reference's choice, because any valid j outside the list is worse than every listed one.
//                      best_pred_score of the next 128 elements lives in registers (one element per lane and
//                      block); a step is a handful of v_readlane + scalar compares.  Only when all KC listed
//                      candidates are blocked AND the window held more than KC valid j is the window
//                      re-evaluated in full (wave-parallel, global memory).
constexpr int KC = 4;

__device__ __forceinline__ uint32_t readlane_u32(uint32_t v, int l) {
  return (uint32_t)__builtin_amdgcn_readlane((int)v, l);
}
__device__ __forceinline__ uint64_t readlane_u64(uint64_t v, int l) {
  return ((uint64_t)readlane_u32((uint32_t)(v >> 32), l) << 32) | readlane_u32((uint32_t)v, l);
}

// minimum of a u64 over the wavefront (all lanes get it)
__device__ __forceinline__ uint64_t wave_min_u64(uint64_t v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const uint64_t t = __shfl_xor(v, o, 64);
    if (t < v) v = t;
  }
  return v;
}

// d(i, j) of paf_filter.rs:798-836; returns false when a gap exceeds the limit
__device__ __forceinline__ bool chain_dist(bool minus, uint64_t qe_i, uint64_t ts_i, uint64_t te_i, uint64_t qs_j,
                                           uint64_t ts_j, uint64_t te_j, uint64_t max_gap, uint64_t fifth, uint64_t* d) {
  uint64_t q_gap, r_gap;
  if (qs_j >= qe_i) {
    q_gap = qs_j - qe_i;
  } else {
    const uint64_t ov = qe_i - qs_j;
    q_gap = ov <= fifth ? ov : max_gap + 1;
  }
  if (!minus) {
    if (ts_j >= te_i) {
      r_gap = ts_j - te_i;
    } else {
      const uint64_t ov = te_i - ts_j;
      r_gap = ov <= fifth ? ov : max_gap + 1;
    }
  } else if (ts_i >= te_j) {
    r_gap = ts_i - te_j;
  } else {
    const uint64_t ov = te_j - ts_i;
    r_gap = ov <= fifth ? ov : max_gap + 1;
  }
  if (q_gap > max_gap || r_gap > max_gap) return false;
  *d = q_gap * q_gap + r_gap * r_gap;  // wrapping, as release Rust
  return true;
}

__global__ __launch_bounds__(EW) void chain_candidates_kernel(uint64_t m, const uint32_t* __restrict__ s_gidx,
                                                              const uint32_t* __restrict__ group_begin,
                                                              uint32_t n_groups, const uint64_t* __restrict__ s_grp,
                                                              const uint32_t* __restrict__ s_qs,
                                                              const uint32_t* __restrict__ s_qe,
                                                              const uint32_t* __restrict__ s_ts,
                                                              const uint32_t* __restrict__ s_te, uint64_t max_gap,
                                                              unsigned long long* __restrict__ c_d,  // [KC][m]
                                                              uint32_t* __restrict__ c_j,            // [KC][m]
                                                              uint32_t* __restrict__ c_n,            // valid count (saturating)
                                                              uint32_t* __restrict__ c_ext,          // window extent in elements
                                                              const uint8_t* __restrict__ only = nullptr) {  // optional: lists for these elements only
  uint64_t p = (uint64_t)blockIdx.x * EW + threadIdx.x;
  if (p >= m) return;
  if (only && !only[p]) return;
  const uint32_t g = s_gidx[p];
  const uint32_t e = (g + 1 < n_groups) ? group_begin[g + 1] : (uint32_t)m;
  const bool minus = (s_grp[p] & 1ull) != 0;
  const uint64_t qe_i = s_qe[p], ts_i = s_ts[p], te_i = s_te[p];
  const uint64_t bound = qe_i + max_gap, fifth = max_gap / 5;
  uint64_t bd[KC];
  uint32_t bj[KC];
#pragma unroll
  for (int k = 0; k < KC; ++k) {
    bd[k] = ~0ull;
    bj[k] = NONE;
  }
  uint32_t count = 0, ext = 0;
  for (uint32_t j = (uint32_t)p + 1; j < e; ++j) {
    const uint64_t qs_j = s_qs[j];
    if (qs_j > bound) break;  // sorted by q_start (paf_filter.rs:794-796)
    ext = j - (uint32_t)p;
    uint64_t d;
    if (!chain_dist(minus, qe_i, ts_i, te_i, qs_j, s_ts[j], s_te[j], max_gap, fifth, &d)) continue;
    if (count < 0xffffffffu) ++count;
    // insert (d, j) keeping (d asc, j asc): j only grows, so the new entry goes after every entry with d' <= d
    // (strict `<` finds that slot); from there on every entry moves down one slot, equal distances included
    if (d < bd[KC - 1]) {
      uint64_t cd = d;
      uint32_t cj = j;
      bool placed = false;
#pragma unroll
      for (int k = 0; k < KC; ++k) {
        if (placed || cd < bd[k]) {
          placed = true;
          const uint64_t td = bd[k];
          const uint32_t tj = bj[k];
          bd[k] = cd;
          bj[k] = cj;
          cd = td;
          cj = tj;
        }
      }
    }
  }
#pragma unroll
  for (int k = 0; k < KC; ++k) {
    c_d[(uint64_t)k * m + p] = bd[k];
    c_j[(uint64_t)k * m + p] = bj[k];
  }
  c_n[p] = count;
  c_ext[p] = ext;
}

// The KC smallest (d, j) of a wavefront, every lane holding its own list sorted by (d asc, j asc) and every candidate
// living in exactly one lane: butterfly all-reduce, six dependent shuffle rounds.  Merging two sorted KC-lists: the
// element-wise minimum of A[k] and B[KC-1-k] is the KC smallest of the union as a bitonic sequence, which two
// compare-exchange stages sort.  On return every lane holds the wavefront's list.
static_assert(KC == 4, "wave_top_kc is written for four candidates");
__device__ __forceinline__ bool cand_less(uint64_t ad, uint32_t aj, uint64_t bd_, uint32_t bj_) {
  return ad < bd_ || (ad == bd_ && aj < bj_);
}
__device__ __forceinline__ void wave_top_kc(uint64_t bd[KC], uint32_t bj[KC]) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    uint64_t pd[KC];
    uint32_t pj[KC];
#pragma unroll
    for (int k = 0; k < KC; ++k) {
      pd[k] = __shfl_xor(bd[k], o, 64);
      pj[k] = __shfl_xor(bj[k], o, 64);
    }
#pragma unroll
    for (int k = 0; k < KC; ++k)
      if (cand_less(pd[KC - 1 - k], pj[KC - 1 - k], bd[k], bj[k])) {
        bd[k] = pd[KC - 1 - k];
        bj[k] = pj[KC - 1 - k];
      }
    auto cx = [&](int x, int y) {
      if (cand_less(bd[y], bj[y], bd[x], bj[x])) {
        const uint64_t td = bd[x];
        const uint32_t tj = bj[x];
        bd[x] = bd[y];
        bj[x] = bj[y];
        bd[y] = td;
        bj[y] = tj;
      }
    };
    cx(0, 2);
    cx(1, 3);
    cx(0, 1);
    cx(2, 3);
  }
}

// The same candidate lists for deep groups (windows of hundreds to thousands of elements, S-big1): one WAVEFRONT per i.
// A thread per i walks its ~2,000-element window alone and a wavefront waits for its longest window (a 500-kb mapping
// has a window ten times the average); here the 64 lanes take the window 64 elements at a time (coalesced loads, the
// neighbouring i's re-read the same lines from L1/L2) and every lane keeps its own KC best in (d, j) order.
// The KC best of the wavefront are drawn by one butterfly merge of the lanes' lists (wave_top_kc).  32-bit arithmetic
// (coordinates are u32; a gap limit beyond 2^32 cannot bind, so it is clamped).  The lists are identical to
// chain_candidates_kernel's.
// The scan stops early: j runs in q_start order, so past q_end[i] the query gap only grows, and once KC candidates are held
// whose distance is at most gap^2 of the next unseen element no later j can enter the list (d >= gap^2; equal distances keep
// the smaller j).  The test needs no merge of the lanes' lists -- "how many held entries are <= T" is four compares per
// lane and four ballots -- so it runs after every batch (an earlier attempt derived the KC-th best itself per batch, which
// cost more than the batches it saved; profiles/README.md).  After a cut the exact number of valid j is unknown: the count
// is reported as one more than what was seen -- "the window may hold more" -- which at worst lets the selection re-evaluate
// a window that has nothing left to offer (same result); the window extent then comes from a galloping search.
constexpr int CW_PER_WAVE = 16;  // consecutive i handled by one wavefront

// Round 4: the batch loop for gap limits below 2^31 (no wrap-around, the early cut applies), written for the vector unit,
// which is what bounds this kernel (a wave64 instruction occupies the SIMD for four cycles; the generic loop below spends
// ~45 vector instructions and two quarter-rate 64-bit multiply-adds per batch):
//   * the strand is a template argument (two selects per batch gone), so is the width of the distance: a limit of at most
//     46340 keeps q^2 + r^2 below 2^32 -- 24-bit multiplies, one 32-bit compare against the list's last entry;
//   * a gap is |a - b| (one v_sad_u32) and passes when it is at most a fifth of the limit, or at most the limit on the gap
//     side -- the same predicate as paf_filter.rs:798-836 with the value selects folded away (a rejected pair's distance is
//     never read);
//   * loads address `uniform base + lane` with the lane offset clamped to the group's end (no exec-mask branch, no 64-bit
//     address arithmetic per lane); two batches per trip, so the prefetched registers are not moved.
// The wavefront's list stays wave-uniform (scalar registers), inserted into in ascending lane = ascending j order as before.
constexpr size_t CAND_PAD = 8192;  // the scan reads up to three batches (768 bytes per array) past a group's end without clamping
struct CandScan {
  uint64_t d0, d1, d2, d3;
  uint32_t j0, j1, j2, j3;
  uint32_t count, ext, cut_at;
  bool cut;
};
// |a - b|, one instruction (the compiler expands the generic form to min / max / sub)
__device__ __forceinline__ uint32_t absdiff_vv(uint32_t a, uint32_t b) {
  uint32_t r;
  asm("v_sad_u32 %0, %1, %2, 0" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
// |a - b| of a per-lane and a wave-uniform value: one instruction (the compiler expands the generic form to min / max / sub)
__device__ __forceinline__ uint32_t absdiff_vs(uint32_t v, uint32_t s) {
  uint32_t r;
  asm("v_sad_u32 %0, %1, %2, 0" : "=v"(r) : "v"(v), "s"(s));
  return r;
}
template <bool MINUS, bool D32>
__device__ __forceinline__ void cand_scan_fast(uint32_t p, uint32_t e, uint32_t qe_i, uint32_t ts_i, uint32_t te_i, uint32_t bound,
                                               uint32_t gap, uint32_t fifth, const uint32_t* __restrict__ s_qs,
                                               const uint32_t* __restrict__ s_ts, const uint32_t* __restrict__ s_te, uint32_t lane,
                                               CandScan& L) {
  // the list: 32-bit distances with 0xffffffff = empty when they fit (every compare of the insertion on the scalar unit)
  using DT = typename std::conditional<D32, uint32_t, uint64_t>::type;
  constexpr DT EMPTY = (DT)~(DT)0;
  DT sd0 = EMPTY, sd1 = EMPTY, sd2 = EMPTY, sd3 = EMPTY;
  uint32_t sj0 = NONE, sj1 = NONE, sj2 = NONE, sj3 = NONE;
  uint32_t count = 0, ext = 0, cut_at = 0;
  bool cut = false;
  // (plain locals and one copy of the loop body: lambdas that capture the list by reference end up with the list in scratch
  // memory, addressed through selected pointers.  Predicates that feed both a ballot and a select are kept apart: combined as
  // `bool` they are materialised as 0/1 and compared again, two vector instructions each.)
  const uint32_t* __restrict__ s_r = MINUS ? s_te : s_ts;  // the target coordinate of j that enters d(i, j)
  const uint32_t r_i = MINUS ? ts_i : te_i;                // ... and that of i
  // A wave64 vector instruction occupies the SIMD for four cycles and the scalar unit serves a SIMD every fourth cycle: a
  // batch costs 4 x max(vector, scalar) instructions (SQ counters, profiles/README.md).  Both are kept to the loop's own work:
  //   * addresses: uniform base (s_qs + p + 1) + a per-lane byte offset that advances by 256 (one vector add for both arrays);
  //     no clamping -- the arrays are followed by CAND_PAD readable bytes, what lies past the group's end is masked;
  //   * j itself is a per-lane value (j < e is a vector compare); the scan ends when the window mask is not full, which a
  //     batch past the group's end also satisfies -- no uniform batch counter in the loop, the window extent is taken from
  //     the last batch's position;
  //   * a rejected pair gets the distance "empty" by selects; the list test reads that one value;
  //   * the number of valid j saturates at KC + 1 (all its readers ask: how many of the KC entries, and were there more),
  //     so it is counted only until it gets there;
  //   * the early cut is evaluated by every lane for its own element (two vector instructions) and read from lane 63.
  const char* pq = reinterpret_cast<const char*>(s_qs + (p + 1u));
  const char* pr = reinterpret_cast<const char*>(s_r + (p + 1u));
  uint32_t voff = lane * 4u;
  uint32_t vj = p + 1u + lane;
  uint32_t n_qs = *reinterpret_cast<const uint32_t*>(pq + voff);
  uint32_t n_r = *reinterpret_cast<const uint32_t*>(pr + voff);
  // the two limits as per-lane values the compiler cannot re-materialise inside the loop (a select takes one scalar operand:
  // its mask -- both limits would be moved into vector registers again in every batch)
  uint32_t vgap = gap, vfifth = fifth;
  asm volatile("" : "+v"(vgap), "+v"(vfifth));
  if (p + 1u < e) {
    // (one exit, decided by scalar arithmetic, and the two ways out told apart afterwards: two `break`s with their own
    // epilogues come out of the compiler as a chain of flag registers and half a dozen branches per batch)
    uint64_t wmask;
    uint32_t c;
    for (;;) {
      const uint32_t qs_j = n_qs, r_j = n_r;
      voff += 256u;  // the next batch is requested before this one is evaluated
      n_qs = *reinterpret_cast<const uint32_t*>(pq + voff);
      n_r = *reinterpret_cast<const uint32_t*>(pr + voff);
      // sorted by q_start (paf_filter.rs:794-796): the window is a prefix
      const bool in_e = vj < e, in_b = qs_j <= bound;
      wmask = __builtin_amdgcn_ballot_w64(in_e) & __builtin_amdgcn_ballot_w64(in_b);
      const uint32_t aq = absdiff_vs(qs_j, qe_i);
      const bool q_ge = qs_j >= qe_i;
      const uint32_t lim_q = q_ge ? vgap : vfifth;  // a gap may reach the limit, an overlap a fifth of it
      const uint32_t ar = absdiff_vs(r_j, r_i);
      // gap side: plus strand t_start[j] >= t_end[i], minus strand t_start[i] >= t_end[j]
      const uint32_t lim_r = (MINUS ? r_j <= r_i : r_j >= r_i) ? vgap : vfifth;
      DT d;
      uint64_t cutmask;  // (taken before the insertions: they only lower sd3)
      if constexpr (D32) {
        // (an accepted gap is below 2^16; a rejected pair's value is not read.  The cut: this lane's query gap squared
        // reaches the list's last entry -- 65535^2 exceeds every distance and stays below "empty")
        const uint32_t aqc = min(aq, 65535u);
        const uint32_t t1 = __umul24(aqc, aqc);
        d = t1 + __umul24(ar, ar);
        cutmask = __builtin_amdgcn_ballot_w64(q_ge) & __builtin_amdgcn_ballot_w64(t1 >= sd3);
      } else {
        const uint64_t t1 = (uint64_t)aq * aq;
        d = t1 + (uint64_t)ar * ar;
        cutmask = __builtin_amdgcn_ballot_w64(q_ge) & __builtin_amdgcn_ballot_w64(t1 >= sd3);  // (a square is never "empty")
      }
      d = (aq <= lim_q) ? d : EMPTY;
      d = (ar <= lim_r) ? d : EMPTY;
      d = in_e ? d : EMPTY;
      d = in_b ? d : EMPTY;
      if (count <= (uint32_t)KC) count += (uint32_t)__popcll(__builtin_amdgcn_ballot_w64(d != EMPTY));
      uint64_t cmask = __builtin_amdgcn_ballot_w64(d < sd3);
      {  // lane 63's bit (through an opaque move: the compiler otherwise turns it into a 64-bit sign test on the vector unit)
        uint32_t hi = (uint32_t)(cutmask >> 32);
        asm("" : "+s"(hi));
        c = hi >> 31;
      }
      if (cmask) {
        do {
          const int l = __builtin_ctzll(cmask);
          DT dl;
          if constexpr (D32)
            dl = readlane_u32(d, l);
          else
            dl = readlane_u64(d, l);
          const uint32_t jl = readlane_u32(vj, l);
          if (dl < sd0) {
            sd3 = sd2; sj3 = sj2; sd2 = sd1; sj2 = sj1; sd1 = sd0; sj1 = sj0; sd0 = dl; sj0 = jl;
          } else if (dl < sd1) {
            sd3 = sd2; sj3 = sj2; sd2 = sd1; sj2 = sj1; sd1 = dl; sj1 = jl;
          } else if (dl < sd2) {
            sd3 = sd2; sj3 = sj2; sd2 = dl; sj2 = jl;
          } else {
            sd3 = dl; sj3 = jl;
          }
          cmask = __builtin_amdgcn_ballot_w64(d < sd3) & ~((2ull << l) - 1ull);  // lanes above l that still beat the list
        } while (cmask);
        if (!c && sd3 != EMPTY) {  // the list changed: lane 63's test against the list as it stands now
          const uint32_t q_last = readlane_u32(qs_j, 63);
          if (q_last >= qe_i) {
            const uint64_t qg = (uint64_t)q_last - qe_i;
            // (readfirstlane: a 64-bit compare of uniform values is done on the vector unit and would make `c`, and with it
            // the loop's exit test, a per-lane value)
            c = (uint32_t)__builtin_amdgcn_readfirstlane((uint64_t)sd3 <= qg * qg ? 1 : 0);
          }
        }
      }
      // the window ended inside these 64, or the group did (a batch past the group's end is empty); or the cut: every later
      // element starts at or after this batch's last one, so its query gap is at least `qg`, and with KC entries held at
      // distance <= qg^2 no later j can enter the list (d >= qg^2; equal distances keep the smaller j)
      const uint64_t nw = ~wmask;
      if (((uint32_t)nw | (uint32_t)(nw >> 32) | c) != 0u) break;
      vj += 64u;
    }
    const uint32_t j_first = readlane_u32(vj, 0);  // of the last batch
    if (wmask != ~0ull) {
      ext = j_first - (p + 1u) + (uint32_t)__popcll(wmask);
    } else {
      cut_at = j_first + 64u;
      cut = cut_at < e;  // (a full batch that ends the group: nothing was cut, the counts are exact)
      if (!cut) ext = cut_at - (p + 1u);
    }
  }
  auto wide = [](DT v) { return v == EMPTY ? ~0ull : (uint64_t)v; };
  L.d0 = wide(sd0); L.d1 = wide(sd1); L.d2 = wide(sd2); L.d3 = wide(sd3);
  L.j0 = sj0; L.j1 = sj1; L.j2 = sj2; L.j3 = sj3;
  L.count = min(count, (uint32_t)KC + 1u);
  L.ext = ext;
  L.cut_at = cut_at;
  L.cut = cut;
}

// MODE 0: any gap limit (the generic loop); 1: limit < 2^31 (cand_scan_fast, 64-bit distances); 2: limit <= 46340 (32-bit)
template <int MODE>
__global__ __launch_bounds__(EW) void chain_candidates_wave_kernel(uint64_t m, const uint32_t* __restrict__ s_gidx,
                                                                   const uint32_t* __restrict__ group_begin,
                                                                   uint32_t n_groups, const uint64_t* __restrict__ s_grp,
                                                                   const uint32_t* __restrict__ s_qs,
                                                                   const uint32_t* __restrict__ s_qe,
                                                                   const uint32_t* __restrict__ s_ts,
                                                                   const uint32_t* __restrict__ s_te, uint64_t max_gap,
                                                                   unsigned long long* __restrict__ c_d,
                                                                   uint32_t* __restrict__ c_j, uint32_t* __restrict__ c_n,
                                                                   uint32_t* __restrict__ c_ext) {
  const int lane = threadIdx.x & 63;
  // readfirstlane: the compiler cannot see that threadIdx.x >> 6 is the same in all lanes of a wavefront, and would carry the
  // whole loop nest below in divergent form (exec-mask bookkeeping on the scalar unit, loop counters in vector registers)
  const uint64_t wave = (uint64_t)blockIdx.x * (EW / 64) + (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const uint32_t gap = max_gap > 0xffffffffull ? 0xffffffffu : (uint32_t)max_gap;
  const bool wrap = max_gap == ~0ull;  // `max_gap + 1` (= reject) wraps to 0 in release Rust
  const uint32_t fifth = (uint32_t)((max_gap / 5) > 0xffffffffull ? 0xffffffffull : (max_gap / 5));
  const bool can_cut = max_gap < (uint64_t(1) << 31);  // accepted gaps < 2^31: q^2 + r^2 cannot wrap and grows with the query gap
  uint64_t out_d[KC];
  uint32_t out_j[KC];
  uint32_t out_n = 0, out_ext = 0;
#pragma unroll
  for (int k = 0; k < KC; ++k) {
    out_d[k] = ~0ull;
    out_j[k] = NONE;
  }
  const uint64_t p0 = wave * CW_PER_WAVE;
  // the parameters of the wavefront's CW_PER_WAVE elements, one element per lane (read once, up front: per element they would
  // be two dependent memory round trips before its first batch)
  uint32_t l_e = 0, l_qe = 0, l_ts = 0, l_te = 0, l_minus = 0;
  if (lane < CW_PER_WAVE && p0 + lane < m) {
    const uint64_t pl = p0 + lane;
    const uint32_t g = s_gidx[pl];
    l_e = (g + 1 < n_groups) ? group_begin[g + 1] : (uint32_t)m;
    l_minus = (uint32_t)(s_grp[pl] & 1ull);
    l_qe = s_qe[pl];
    l_ts = s_ts[pl];
    l_te = s_te[pl];
  }
  for (uint64_t p = p0; p < p0 + CW_PER_WAVE && p < m; ++p) {  // wave-uniform
    const int pt = (int)(p - p0);
    const uint32_t e = readlane_u32(l_e, pt);
    const bool minus = readlane_u32(l_minus, pt) != 0;
    const uint32_t qe_i = readlane_u32(l_qe, pt), ts_i = readlane_u32(l_ts, pt), te_i = readlane_u32(l_te, pt);
    const uint64_t bound64 = (uint64_t)qe_i + max_gap;  // wrapping, as release Rust
    const uint32_t bound = bound64 > 0xffffffffull ? 0xffffffffu : (uint32_t)bound64;
    // the wavefront's list, (d asc, j asc); wave-uniform values
    uint64_t sd0 = ~0ull, sd1 = ~0ull, sd2 = ~0ull, sd3 = ~0ull;
    uint32_t sj0 = NONE, sj1 = NONE, sj2 = NONE, sj3 = NONE;
    uint32_t count = 0, ext = 0, cut_at = 0;
    bool cut = false;
    if (MODE != 0) {
      CandScan L;
      if (minus)
        cand_scan_fast<true, MODE == 2>((uint32_t)p, e, qe_i, ts_i, te_i, bound, gap, fifth, s_qs, s_ts, s_te, (uint32_t)lane, L);
      else
        cand_scan_fast<false, MODE == 2>((uint32_t)p, e, qe_i, ts_i, te_i, bound, gap, fifth, s_qs, s_ts, s_te, (uint32_t)lane, L);
      sd0 = L.d0; sd1 = L.d1; sd2 = L.d2; sd3 = L.d3;
      sj0 = L.j0; sj1 = L.j1; sj2 = L.j2; sj3 = L.j3;
      count = L.count;
      ext = L.ext;
      cut_at = L.cut_at;
      cut = L.cut;
    } else {
    // software pipeline: the three coordinates of the NEXT batch are requested before the current one is evaluated (one
    // memory round trip per batch, overlapped with the arithmetic, instead of two dependent ones)
    uint32_t n_qs = 0xffffffffu, n_ts = 0, n_te = 0;
    {
      const uint32_t j = (uint32_t)p + 1 + lane;
      if (j < e) {
        n_qs = s_qs[j];
        n_ts = s_ts[j];
        n_te = s_te[j];
      }
    }
    // (loop control on values the compiler can see are wave-uniform -- readfirstlane -- so that it stays on the scalar unit
    // without exec-mask bookkeeping: the kernel is bound by scalar instructions)
    for (uint32_t j0 = (uint32_t)p + 1; j0 < e;) {
      const uint32_t j = j0 + lane;
      const uint32_t qs_j = n_qs, ts_j = n_ts, te_j = n_te;
      {
        const uint32_t jn = j + 64;
        n_qs = 0xffffffffu;
        if (jn < e) {
          n_qs = s_qs[jn];
          n_ts = s_ts[jn];
          n_te = s_te[jn];
        }
      }
      const bool inwin = (j < e) & (qs_j <= bound);  // sorted by q_start (paf_filter.rs:794-796): the window is a prefix
      const uint64_t wmask = __ballot(inwin);
      ext += (uint32_t)__popcll(wmask);
      // d(i, j) of paf_filter.rs:798-836 in 32-bit arithmetic, branch-free (the kernel is bound by vector instructions: every
      // lane evaluates every line, selects instead of divergent branches): a gap is the difference when it is one, the
      // overlap when that is at most a fifth of the limit, and otherwise `limit + 1` = reject (which wraps to 0 in release
      // Rust when the limit is u64::MAX)
      const bool q_ge = qs_j >= qe_i;
      const uint32_t q_ov = qe_i - qs_j;
      const bool q_in = q_ge | (q_ov <= fifth);
      const uint32_t q_gap = q_ge ? qs_j - qe_i : (q_in ? q_ov : 0u);
      const uint32_t ra = minus ? ts_i : ts_j, rb = minus ? te_j : te_i;  // gap = ra - rb, overlap = rb - ra
      const bool r_ge = ra >= rb;
      const uint32_t r_ov = rb - ra;
      const bool r_in = r_ge | (r_ov <= fifth);
      const uint32_t r_gap = r_ge ? ra - rb : (r_in ? r_ov : 0u);
      const bool ok = inwin & (q_in | wrap) & (r_in | wrap) & (q_gap <= gap) & (r_gap <= gap);
      const uint64_t dd = (uint64_t)q_gap * q_gap + (uint64_t)r_gap * r_gap;  // wrapping, as release Rust
      const uint64_t d = ok ? dd : ~0ull;
      count += (uint32_t)__popcll(__ballot(ok));  // cannot exceed 2^32 - 1 here
      // ascending lane = ascending j: an entry goes after every entry with d' <= d (strict `<` finds the slot).  The ballot
      // is taken again after every insertion: the list's last entry only falls, most lanes drop out at once.
      uint64_t cmask = __ballot(ok && d < sd3);
      while (cmask) {
        const int l = __builtin_ctzll(cmask);
        const uint64_t dl = readlane_u64(d, l);
        const uint32_t jl = j0 + (uint32_t)l;
        if (dl < sd0) {
          sd3 = sd2; sj3 = sj2; sd2 = sd1; sj2 = sj1; sd1 = sd0; sj1 = sj0; sd0 = dl; sj0 = jl;
        } else if (dl < sd1) {
          sd3 = sd2; sj3 = sj2; sd2 = sd1; sj2 = sj1; sd1 = dl; sj1 = jl;
        } else if (dl < sd2) {
          sd3 = sd2; sj3 = sj2; sd2 = dl; sj2 = jl;
        } else {
          sd3 = dl; sj3 = jl;
        }
        cmask = __ballot(ok && d < sd3) & ~((2ull << l) - 1ull);  // lanes above l that still beat the list
      }
      int stop = (wmask != ~0ull) | (j0 + 64 >= e);  // the window ended inside these 64, or the group did
      if (!stop && can_cut && sd3 != ~0ull) {
        // every later element starts at or after this batch's last one: its query gap is at least `qg`; with KC entries held
        // at distance <= qg^2 no later j can enter the list (d >= qg^2; equal distances keep the smaller j)
        const uint32_t q_last = readlane_u32(qs_j, 63);
        if (q_last >= qe_i) {
          const uint64_t qg = (uint64_t)q_last - qe_i;
          if (sd3 <= qg * qg) {
            cut = true;
            cut_at = j0 + 64;
            stop = 1;
          }
        }
      }
      if (__builtin_amdgcn_readfirstlane(stop)) break;
      j0 += 64;
    }
    }  // MODE == 0
    if (cut) {
      // Window extent without scanning: the batch before `cut_at` lies inside the window; gallop ahead 64 x 64 elements at
      // a time (one probe per lane), then resolve inside the 64-element block that holds the boundary.
      uint32_t lo = cut_at - 1;  // last element known to be inside the window
      bool found = false;
      while (!found) {
        const uint64_t pj = (uint64_t)lo + 1 + (uint64_t)lane * 64;  // first element of the lane's block
        const bool inside = pj < e && s_qs[pj] <= bound;
        const uint64_t m_in = __ballot(inside);
        if (m_in == ~0ull) {  // all 64 block starts are inside: the boundary is further on
          lo += 1 + 63 * 64;  // the last block start probed (inside)
          if (lo + 1 >= e) found = true;
          continue;
        }
        const int nb_in = __popcll(m_in);  // blocks whose first element is inside (a prefix: sorted)
        if (nb_in == 0) break;             // the very next element is already outside
        const uint64_t blk = (uint64_t)lo + 1 + (uint64_t)(nb_in - 1) * 64;  // the boundary lies in [blk, blk + 64)
        const uint64_t ej = blk + lane;
        const bool in2 = ej < e && s_qs[ej] <= bound;
        lo = (uint32_t)(blk + __popcll(__ballot(in2)) - 1);
        found = true;
      }
      ext = lo - (uint32_t)p;
      ++count;  // "there may be more": the exact number of valid j is unknown after a cut
    }
    if (lane == (int)(p - p0)) {
      out_d[0] = sd0; out_d[1] = sd1; out_d[2] = sd2; out_d[3] = sd3;
      out_j[0] = sj0; out_j[1] = sj1; out_j[2] = sj2; out_j[3] = sj3;
      out_n = count;
      out_ext = ext;
    }
  }
  const uint64_t po = p0 + lane;
  if (lane < CW_PER_WAVE && po < m) {
#pragma unroll
    for (int k = 0; k < KC; ++k) {
      c_d[(uint64_t)k * m + po] = out_d[k];
      c_j[(uint64_t)k * m + po] = out_j[k];
    }
    c_n[po] = out_n;
    c_ext[po] = out_ext;
  }
}

// SWG_CAND_GENERIC=1 (test knob): the generic loop whatever the limit
template <class G>
static void launch_candidates_wave(swg_ctx* ctx, hipStream_t st, G grid, uint64_t m, const uint32_t* s_gidx, const uint32_t* group_begin,
                                   uint32_t n_groups, const uint64_t* s_grp, const uint32_t* s_qs, const uint32_t* s_qe, const uint32_t* s_ts,
                                   const uint32_t* s_te, uint64_t max_gap, unsigned long long* c_d, uint32_t* c_j, uint32_t* c_n, uint32_t* c_ext) {
  static const bool generic = getenv("SWG_CAND_GENERIC") != nullptr;
  // (the fast loops address a window by a 32-bit byte offset from its first element: up to 2^30 elements)
  const int mode = (generic || max_gap >= (uint64_t(1) << 31) || m > (uint64_t(1) << 30)) ? 0 : (max_gap <= 46340 ? 2 : 1);
#define SWG_CAND_LAUNCH(MODE)                                                                                                        \
  SWG_LAUNCH(ctx, "chain_candidates_wave", chain_candidates_wave_kernel<MODE><<<grid, EW, 0, st>>>(m, s_gidx, group_begin, n_groups, s_grp, s_qs, \
                                                                                                    s_qe, s_ts, s_te, max_gap, c_d, c_j, c_n, c_ext))
  if (mode == 0)
    SWG_CAND_LAUNCH(0);
  else if (mode == 1)
    SWG_CAND_LAUNCH(1);
  else
    SWG_CAND_LAUNCH(2);
#undef SWG_CAND_LAUNCH
}

struct SelBlock {
  uint64_t d[KC];
  uint32_t j[KC];
  uint32_t n;
  uint64_t bps;
};

// Units come in three sizes.  Short ones (<= SMALL_UNIT elements: nearly all of them in sparse data) are walked one per
// LANE (chain_select_lanes_kernel); the longest (>= BIG_UNIT) are cut into blocks that run speculatively in parallel
// (spec_round_kernel); the middle ones get one wavefront each (chain_select_kernel), which keeps the scores of the
// next 128 elements in registers.

// best_pred_score[j] of an element beyond the 128 the wavefront keeps in registers; out of line for the same reason as
// spec_view_far (an inlined load would make every step of the walk wait for the stores of the step before)
__device__ __noinline__ uint64_t bps_far(const unsigned long long* bps, uint32_t j) {
  return __hip_atomic_load(&bps[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__global__ __launch_bounds__(256) void chain_select_kernel(uint32_t n_list, const uint32_t* __restrict__ unit_list,
                                                           uint32_t n_units, const uint32_t* __restrict__ unit_begin,
                                                           uint32_t m, const uint64_t* __restrict__ s_grp,
                                                           const uint32_t* __restrict__ s_qs,
                                                           const uint32_t* __restrict__ s_qe,
                                                           const uint32_t* __restrict__ s_ts,
                                                           const uint32_t* __restrict__ s_te, uint64_t max_gap,
                                                           const unsigned long long* __restrict__ c_d,
                                                           const uint32_t* __restrict__ c_j,
                                                           const uint32_t* __restrict__ c_n,
                                                           const uint32_t* __restrict__ s_gidx,
                                                           const uint32_t* __restrict__ group_begin, uint32_t n_groups,
                                                           unsigned long long* bps, uint32_t* __restrict__ pred) {
  const int lane = threadIdx.x & 63;
  const uint32_t wave_global = blockIdx.x * 4 + (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const uint32_t n_waves = (gridDim.x * 256) >> 6;
  const uint64_t INF = ~0ull;
  const uint64_t fifth = max_gap / 5;
  const bool can_cut = max_gap < (uint64_t(1) << 31);  // accepted gaps < 2^31: d cannot wrap and grows with the query gap
  // one wavefront per listed unit (the middle-sized ones: longer than a lane should walk, shorter than BIG_UNIT)
  for (uint32_t k = wave_global; k < n_list; k += n_waves) {
    const uint32_t u = unit_list[k];
    const uint32_t b = unit_begin[u];
    const uint32_t e = (u + 1 < n_units) ? unit_begin[u + 1] : m;
    if (e - b < 2) continue;
    auto load_block = [&](uint32_t pos) {
      SelBlock k;
      const uint32_t p = pos + lane;
      if (p < e) {
#pragma unroll
        for (int c = 0; c < KC; ++c) {
          k.d[c] = c_d[(uint64_t)c * m + p];
          k.j[c] = c_j[(uint64_t)c * m + p];
        }
        k.n = c_n[p];
        k.bps = __hip_atomic_load(&bps[p], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      } else {
#pragma unroll
        for (int c = 0; c < KC; ++c) {
          k.d[c] = INF;
          k.j[c] = NONE;
        }
        k.n = 0;
        k.bps = 0;
      }
      return k;
    };
    uint32_t base = b;
    SelBlock A = load_block(base), B = load_block(base + 64);
    // best_pred_score[j] as the sequential loop sees it now (wave-uniform j)
    auto current = [&](uint32_t j) -> uint64_t {
      const uint32_t lj = j - base;
      if (lj < 64) return readlane_u64(A.bps, (int)lj);
      if (lj < 128) return readlane_u64(B.bps, (int)(lj - 64));
      return bps_far(bps, j);
    };
    // steps in groups of 64; the two register blocks are loaded and touched before the inner loop, so that no step waits
    // for vector memory (see spec_round_kernel)
    for (uint32_t i0 = b; i0 + 1 < e; i0 += 64) {
      if (i0 != b) {
        A = B;
        base += 64;
        B = load_block(base + 64);
      }
      {
        uint64_t t64 = A.bps ^ B.bps;
        uint32_t t32 = A.n ^ B.n;
#pragma unroll
        for (int c = 0; c < KC; ++c) {
          t64 ^= A.d[c] ^ B.d[c];
          t32 ^= A.j[c] ^ B.j[c];
        }
        asm volatile("" ::"v"(t64), "v"(t32));
      }
      const uint32_t i_end = i0 + 64;
      for (uint32_t i = i0; i < i_end && i + 1 < e; ++i) {
      const int li = (int)(i - base);
      const uint32_t nvalid = readlane_u32(A.n, li);
      if (nvalid == 0) continue;
      uint64_t best_d = INF;
      uint32_t best_j = NONE;
#pragma unroll
      for (int c = 0; c < KC; ++c) {
        if (best_j == NONE && (uint32_t)c < nvalid) {
          const uint64_t d = readlane_u64(A.d[c], li);
          const uint32_t j = readlane_u32(A.j[c], li);
          if (d < current(j)) {
            best_d = d;
            best_j = j;
          }
        }
      }
      if (best_j == NONE && nvalid > (uint32_t)KC) {
        // every listed candidate is blocked and the window held more: evaluate it in full (rare)
        const uint64_t qe_i = s_qe[i], ts_i = s_ts[i], te_i = s_te[i];
        const bool minus = (s_grp[i] & 1ull) != 0;
        const uint64_t bound = qe_i + max_gap;
        // the window ends with i's (q, t, strand) group at the latest (the chunk may hold several groups)
        const uint32_t gi = s_gidx[i];
        const uint32_t ge = (gi + 1 < n_groups) ? group_begin[gi + 1] : m;
        uint64_t ld = INF;
        uint32_t lj2 = NONE;
        for (uint32_t j0 = i + 1; j0 < ge; j0 += 64) {
          if (can_cut) {  // the batch starts past q_end[i] and its smallest query gap already reaches the best distance held
            const uint64_t wmin = wave_min_u64(ld);
            const uint64_t q0 = s_qs[j0];
            if (wmin != INF && q0 >= qe_i && (q0 - qe_i) * (q0 - qe_i) >= wmin) break;
          }
          const uint32_t j = j0 + lane;
          bool in = j < ge;
          uint64_t qs_j = 0;
          if (in) {
            qs_j = s_qs[j];
            in = qs_j <= bound;
          }
          // this lane's own view of best_pred_score[j] (j differs per lane here); the cross-lane reads are
          // done by all lanes before any branch
          const uint32_t rel = in ? j - base : 0u;
          const uint64_t va = __shfl(A.bps, (int)(rel & 63), 64), vb = __shfl(B.bps, (int)(rel & 63), 64);
          if (in) {
            uint64_t d;
            if (chain_dist(minus, qe_i, ts_i, te_i, qs_j, s_ts[j], s_te[j], max_gap, fifth, &d)) {
              const uint64_t cur = rel < 64 ? va
                                   : rel < 128 ? vb
                                               : __hip_atomic_load(&bps[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
              if (d < cur && d < ld) {
                ld = d;
                lj2 = j;
              }
            }
          }
          if (!__any(in)) break;
        }
        uint64_t cand = __ballot(lj2 != NONE);
        while (cand) {
          const int l = __builtin_ctzll(cand);
          cand &= cand - 1;
          const uint64_t d = readlane_u64(ld, l);
          const uint32_t j = readlane_u32(lj2, l);
          if (d < best_d || (d == best_d && j < best_j)) {
            best_d = d;
            best_j = j;
          }
        }
      }
      if (best_j == NONE) continue;
      const uint32_t lj = best_j - base;
      if (lj < 64) {
        if ((uint32_t)lane == lj) A.bps = best_d;
      } else if (lj < 128) {
        if ((uint32_t)lane == lj - 64) B.bps = best_d;
      } else {
        if (lane == 0) __hip_atomic_store(&bps[best_j], best_d, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");  // drain before any later read of it
      }
      if (lane == 0) pred[best_j] = i;
      }
    }
  }
}

// Short units -- the bulk of sparse data, where a gap larger than max_gap cuts a group every few dozen elements --
// are the opposite case: a wavefront per chunk spends 64 lanes on one sequential walk.  Here every LANE walks its own
// unit: the reference's greedy unchanged (listed candidates in (d, j) order against best_pred_score, full window when
// all of them are blocked and the window held more), state in global memory but private to the lane, since windows
// never leave a unit.
constexpr uint32_t SMALL_UNIT = 256;

__global__ __launch_bounds__(EW) void chain_select_lanes_kernel(uint32_t n_units, const uint32_t* __restrict__ unit_begin, uint32_t m,
                                                                const uint64_t* __restrict__ s_grp,
                                                                const uint32_t* __restrict__ s_qs,
                                                                const uint32_t* __restrict__ s_qe,
                                                                const uint32_t* __restrict__ s_ts,
                                                                const uint32_t* __restrict__ s_te, uint64_t max_gap,
                                                                const unsigned long long* __restrict__ c_d,
                                                                const uint32_t* __restrict__ c_j,
                                                                const uint32_t* __restrict__ c_n,
                                                                const uint32_t* __restrict__ c_ext, unsigned long long* bps,
                                                                uint32_t* pred) {
  const uint32_t u = blockIdx.x * EW + threadIdx.x;
  if (u >= n_units) return;
  const uint32_t b = unit_begin[u];
  const uint32_t e = (u + 1 < n_units) ? unit_begin[u + 1] : m;
  const uint32_t len = e - b;
  if (len > SMALL_UNIT) return;
  const uint64_t fifth = max_gap / 5;
  const bool can_cut = max_gap < (uint64_t(1) << 31);  // accepted gaps < 2^31: d cannot wrap and grows with the query gap
  for (uint32_t i = b; i < e; ++i) {
    const uint32_t nvalid = c_n[i];
    if (nvalid == 0) continue;
    uint64_t best_d = ~0ull;
    uint32_t best_j = NONE;
#pragma unroll
    for (int c = 0; c < KC; ++c) {
      if (best_j == NONE && (uint32_t)c < nvalid) {
        const uint64_t d = c_d[(uint64_t)c * m + i];
        const uint32_t j = c_j[(uint64_t)c * m + i];
        if (d < bps[j]) {
          best_d = d;
          best_j = j;
        }
      }
    }
    if (best_j == NONE && nvalid > (uint32_t)KC) {  // paf_filter.rs:794-840 over the whole window
      const uint64_t qe_i = s_qe[i], ts_i = s_ts[i], te_i = s_te[i];
      const bool minus = (s_grp[i] & 1ull) != 0;
      const uint32_t last = i + c_ext[i];  // last element with q_start <= q_end[i] + max_gap (inside the unit)
      for (uint32_t j = i + 1; j <= last && j < e; ++j) {
        const uint64_t qs_j = s_qs[j];
        if (can_cut && qs_j >= qe_i && best_j != NONE) {  // past q_end[i] the query gap only grows:
          const uint64_t qg = qs_j - qe_i;                                      // nothing closer can follow
          if (qg * qg >= best_d) break;
        }
        uint64_t d;
        if (!chain_dist(minus, qe_i, ts_i, te_i, qs_j, s_ts[j], s_te[j], max_gap, fifth, &d)) continue;
        if (d < best_d && d < bps[j]) {
          best_d = d;
          best_j = j;
        }
      }
    }
    if (best_j == NONE) continue;
    bps[best_j] = best_d;
    pred[best_j] = i;
  }
}

// Long units (>= BIG_UNIT elements; dense data, or a few per genome pair in sparse data) would pin one wavefront
// for their whole length.  They are cut into blocks of S elements, S >= the longest window of the unit, and all
// blocks run in parallel, round after round, until nothing changes:
//   * a predecessor i of j lies in j's block or in the block before it (windows are shorter than a block), so the
//     only state a block needs from outside is, for each of its own j, the best distance offered by the previous
//     block (`ext[j]`); it starts from the previous round's value, the previous block publishes this round's;
//   * inside a block the reference's sequential greedy runs unchanged (candidate lists, LDS ring of scores);
//   * block 0 of a unit needs nothing from outside, so after round r the first r blocks are final: the loop ends
//     when a round reproduces `ext` (by then every block has run on the inputs its predecessor's final choices
//     imply).  In practice two or three rounds.
// Two views keep the rounds apart: own[j] is what j's block sees (starts at ext[j]), prev[j] what the previous
// block sees (starts at infinity); each has its own predecessor array.


// per long unit: block size S (multiple of 64, >= longest window + 1, >= 512) and number of blocks
__global__ __launch_bounds__(EW) void spec_plan_kernel(uint32_t n_big, const uint32_t* __restrict__ big_list,
                                                       uint32_t n_units, const uint32_t* __restrict__ unit_begin,
                                                       uint32_t m, const uint32_t* __restrict__ c_ext,
                                                       uint32_t* __restrict__ S_out, uint32_t* __restrict__ nblk_out,
                                                       uint32_t* __restrict__ s_max) {
  __shared__ uint32_t wmaxs[EW / 64];
  const uint32_t bi = blockIdx.x;
  if (bi >= n_big) return;
  const uint32_t u = big_list[bi];
  const uint32_t b = unit_begin[u];
  const uint32_t e = (u + 1 < n_units) ? unit_begin[u + 1] : m;
  uint32_t w = 0;
  for (uint32_t p = b + threadIdx.x; p < e; p += EW) {
    const uint32_t x = c_ext[p];
    if (x > w) w = x;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const uint32_t t = __shfl_xor(w, o, 64);
    if (t > w) w = t;
  }
  if ((threadIdx.x & 63) == 0) wmaxs[threadIdx.x >> 6] = w;
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int k = 1; k < EW / 64; ++k)
      if (wmaxs[k] > w) w = wmaxs[k];
    uint32_t S = ((w + 1 + 63) / 64) * 64;
    if (S < 512) S = 512;
    S_out[bi] = S;
    nblk_out[bi] = (e - b + S - 1) / S;
    atomicMax(s_max, S);
  }
}
// The same plan when a unit holds millions of elements (one work-group per unit would walk it alone): every element
// of a long unit contributes its window extent to the unit's maximum (one atomic per wavefront whose lanes share the
// unit), then one thread per long unit derives S and the block count.
__global__ __launch_bounds__(EW) void unit_wmax_kernel(uint64_t m, const uint32_t* __restrict__ unit_flag,
                                                       const uint32_t* __restrict__ unit_excl,
                                                       const uint8_t* __restrict__ is_big, const uint32_t* __restrict__ c_ext,
                                                       uint32_t* __restrict__ wmax_u, const uint8_t* __restrict__ span_big = nullptr) {
  const int lane = threadIdx.x & 63;
  uint32_t cur_u = NONE, cur_w = 0;  // wave-uniform: the unit this wavefront is accumulating and its maximum so far
  for (uint64_t base = (uint64_t)blockIdx.x * EW; base < m; base += (uint64_t)gridDim.x * EW) {  // block-uniform trip count
    if (span_big && !span_big[base >> BIG_SPAN_SHIFT]) continue;  // no member of a long unit in these 256 elements' span
    const uint64_t p = base + threadIdx.x;
    const bool valid = p < m;
    const uint32_t u = valid ? unit_excl[p] + unit_flag[p] - 1 : 0u;
    const bool big = valid && is_big[u] != 0;
    uint32_t w = big ? c_ext[p] : 0u;
    const uint64_t vm = __ballot(valid);
    if (vm == 0) continue;  // wave-uniform
    const uint32_t u0 = (uint32_t)__shfl((int)u, (int)__builtin_ctzll(vm), 64);
    if (__ballot(valid && u != u0) == 0) {  // the wavefront's elements share one unit
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) {
        const uint32_t t = __shfl_xor(w, o, 64);
        if (t > w) w = t;
      }
      if (u0 != cur_u) {  // flush the previous unit (one atomic per wavefront and unit, not per row)
        if (lane == 0 && cur_w) atomicMax(&wmax_u[cur_u], cur_w);
        cur_u = u0;
        cur_w = 0;
      }
      if (w > cur_w) cur_w = w;
    } else if (big && w) {
      atomicMax(&wmax_u[u], w);
    }
  }
  if (lane == 0 && cur_w) atomicMax(&wmax_u[cur_u], cur_w);
}
__global__ __launch_bounds__(EW) void spec_plan_from_wmax_kernel(uint32_t n_big, const uint32_t* __restrict__ big_list,
                                                                 uint32_t n_units, const uint32_t* __restrict__ unit_begin,
                                                                 uint32_t m, const uint32_t* __restrict__ wmax_u,
                                                                 uint32_t* __restrict__ S_out, uint32_t* __restrict__ nblk_out,
                                                                 uint32_t* __restrict__ s_max, int wmax_by_big_index = 0) {
  const uint32_t bi = blockIdx.x * EW + threadIdx.x;
  if (bi >= n_big) return;
  const uint32_t u = big_list[bi];
  const uint32_t b = unit_begin[u];
  const uint32_t e = (u + 1 < n_units) ? unit_begin[u + 1] : m;
  uint32_t S = ((wmax_u[wmax_by_big_index ? bi : u] + 1 + 63) / 64) * 64;
  if (S < 512) S = 512;
  S_out[bi] = S;
  nblk_out[bi] = (e - b + S - 1) / S;
  atomicMax(s_max, S);
}
__global__ __launch_bounds__(EW) void spec_desc_kernel(uint32_t n_big, const uint32_t* __restrict__ big_list,
                                                       uint32_t n_units, const uint32_t* __restrict__ unit_begin,
                                                       uint32_t m, const uint32_t* __restrict__ S_in,
                                                       const uint32_t* __restrict__ nblk_in,
                                                       const uint32_t* __restrict__ blk_off, SpecBlock* __restrict__ desc) {
  // one thread per (unit, block) pair would need a search; units are few, blocks per unit can be many: one
  // work-group per unit, threads stride over its blocks
  const uint32_t bi = blockIdx.x;
  if (bi >= n_big) return;
  const uint32_t u = big_list[bi];
  const uint32_t b = unit_begin[u];
  const uint32_t e = (u + 1 < n_units) ? unit_begin[u + 1] : m;
  const uint32_t S = S_in[bi], nb = nblk_in[bi], off = blk_off[bi];
  for (uint32_t k = threadIdx.x; k < nb; k += EW) {
    SpecBlock d;
    d.ue = e;
    d.bb = b + k * S;
    d.be = d.bb + S < e ? d.bb + S : e;
    d.pad = 0;
    desc[off + k] = d;
  }
}
__global__ __launch_bounds__(EW) void spec_ext_init_kernel(uint32_t n_blocks, const SpecBlock* __restrict__ desc,
                                                           unsigned long long* __restrict__ ext) {
  for (uint32_t bk = blockIdx.x; bk < n_blocks; bk += gridDim.x) {
    const SpecBlock D = desc[bk];
    for (uint32_t p = D.bb + threadIdx.x; p < D.be; p += EW) ext[p] = ~0ull;
  }
}
// The three per-round passes touch only the elements of long units: one work-group per block descriptor.
__global__ __launch_bounds__(EW) void spec_init_kernel(uint32_t n_blocks, const SpecBlock* __restrict__ desc,
                                                       const unsigned long long* __restrict__ ext,
                                                       unsigned long long* __restrict__ own,
                                                       unsigned long long* __restrict__ prev,
                                                       uint32_t* __restrict__ pred_own, uint32_t* __restrict__ pred_prev,
                                                       const uint32_t* __restrict__ n_dev = nullptr,
                                                       const uint32_t* __restrict__ gate = nullptr) {
  // (n_dev: the block list's length lives on the device; gate: a round that follows a round without changes does nothing --
  // the pair-resident path enqueues its rounds without reading anything back, pair_walk_long_launch)
  if (gate && *gate == 0) return;
  if (n_dev) n_blocks = min(n_blocks, *n_dev);
  for (uint32_t bk = blockIdx.x; bk < n_blocks; bk += gridDim.x) {
    const SpecBlock D = desc[bk];
    for (uint32_t p = D.bb + threadIdx.x; p < D.be; p += EW) {
      own[p] = ext[p];
      prev[p] = ~0ull;
      pred_own[p] = NONE;
      pred_prev[p] = NONE;
    }
  }
}
__global__ __launch_bounds__(EW) void spec_check_kernel(uint32_t n_blocks, const SpecBlock* __restrict__ desc,
                                                        const unsigned long long* __restrict__ prev,
                                                        unsigned long long* __restrict__ ext,
                                                        uint32_t* __restrict__ changed,
                                                        const uint32_t* __restrict__ n_dev = nullptr,
                                                        const uint32_t* __restrict__ gate = nullptr) {
  if (gate && *gate == 0) return;
  if (n_dev) n_blocks = min(n_blocks, *n_dev);
  for (uint32_t bk = blockIdx.x; bk < n_blocks; bk += gridDim.x) {
    const SpecBlock D = desc[bk];
    for (uint32_t p = D.bb + threadIdx.x; p < D.be; p += EW) {
      const unsigned long long v = prev[p];
      if (v != ext[p]) {
        ext[p] = v;
        *changed = 1;
      }
    }
  }
}
__global__ __launch_bounds__(EW) void spec_final_kernel(uint32_t n_blocks, const SpecBlock* __restrict__ desc,
                                                        const uint32_t* __restrict__ pred_own,
                                                        const uint32_t* __restrict__ pred_prev,
                                                        uint32_t* __restrict__ pred,
                                                        const uint32_t* __restrict__ n_dev = nullptr) {
  if (n_dev) n_blocks = min(n_blocks, *n_dev);
  for (uint32_t bk = blockIdx.x; bk < n_blocks; bk += gridDim.x) {
    const SpecBlock D = desc[bk];
    for (uint32_t p = D.bb + threadIdx.x; p < D.be; p += EW) {
      const uint32_t a = pred_own[p], b = pred_prev[p];
      // own-block choosers come later in the sequence and had to beat the previous block's offer
      if (a != NONE)
        pred[p] = a;
      else if (b != NONE)
        pred[p] = b;
    }
  }
}

// A position outside the LDS ring, read from the block's view in global memory.  Kept out of line on purpose: inlined, its
// load shares a destination register with the ring's LDS read, and the hazard bookkeeping then makes every step of the walk
// wait for all outstanding vector-memory operations -- i.e. for the write-through stores of the step before.
__device__ __noinline__ uint64_t spec_view_far(const unsigned long long* own, const unsigned long long* prev, uint32_t be,
                                               uint32_t p) {
  return __hip_atomic_load(p < be ? &own[p] : &prev[p], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
#ifdef SWG_SPEC_STATS
__device__ unsigned long long g_spec_stats[8];
#define SPEC_STAT(k, v) do { if (lane == 0) atomicAdd(&g_spec_stats[k], (unsigned long long)(v)); } while (0)
#else
#define SPEC_STAT(k, v) do { } while (0)
#endif
template <int BIGW>
__global__ __launch_bounds__(64) void spec_round_kernel(uint32_t n_blocks, const SpecBlock* __restrict__ desc, uint32_t m,
                                                        const uint64_t* __restrict__ s_grp,
                                                        const uint32_t* __restrict__ s_qs,
                                                        const uint32_t* __restrict__ s_qe,
                                                        const uint32_t* __restrict__ s_ts,
                                                        const uint32_t* __restrict__ s_te, uint64_t max_gap,
                                                        const unsigned long long* __restrict__ c_d,
                                                        const uint32_t* __restrict__ c_j,
                                                        const uint32_t* __restrict__ c_n, unsigned long long* own,
                                                        unsigned long long* prev, uint32_t* __restrict__ pred_own,
                                                        uint32_t* __restrict__ pred_prev) {
  __shared__ unsigned long long ring[BIGW];          // scores of positions [base, base + BIGW) as this block sees them
  __shared__ uint32_t rq[BIGW], rt[BIGW], re[BIGW];  // their q_start, t_start, t_end (for full-window passes)
  const int lane = threadIdx.x;
  const uint64_t INF = ~0ull;
  const uint64_t fifth = max_gap / 5;
  const bool can_cut = max_gap < (uint64_t(1) << 31);  // accepted gaps < 2^31: d cannot wrap and grows with the query gap
  for (uint32_t bk = blockIdx.x; bk < n_blocks; bk += gridDim.x) {
    const SpecBlock D = desc[bk];
    const uint32_t b = D.bb, be = D.be, e = D.ue;  // i runs over [b, be), j may reach into the next block (< e)
    if (be - b < 1 || e - b < 2) continue;
    const bool minus = (s_grp[b] & 1ull) != 0;
    // this block's view of position p: its own elements start from ext (in `own`), later ones from infinity (`prev`)
    auto view_load = [&](uint32_t p) -> uint64_t {
      return __hip_atomic_load(p < be ? &own[p] : &prev[p], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    };
    uint32_t base = b;
    __syncthreads();
    for (uint32_t p = b + lane; p < b + BIGW; p += 64) {
      const bool ok = p < e;
      ring[p % BIGW] = ok ? view_load(p) : INF;
      rq[p % BIGW] = ok ? s_qs[p] : 0xffffffffu;
      rt[p % BIGW] = ok ? s_ts[p] : 0u;
      re[p % BIGW] = ok ? s_te[p] : 0u;
    }
    __syncthreads();
    uint64_t cd[KC];
    uint32_t cj[KC];
    uint32_t cn;
    auto load_cands = [&](uint32_t pos) {
      const uint32_t p = pos + lane;
      if (p < be) {
#pragma unroll
        for (int c = 0; c < KC; ++c) {
          cd[c] = c_d[(uint64_t)c * m + p];
          cj[c] = c_j[(uint64_t)c * m + p];
        }
        cn = c_n[p];
      } else {
#pragma unroll
        for (int c = 0; c < KC; ++c) {
          cd[c] = INF;
          cj[c] = NONE;
        }
        cn = 0;
      }
    };
    load_cands(base);
    auto current = [&](uint32_t j) -> uint64_t {
      if (j - base < (uint32_t)BIGW) return ring[j % BIGW];
      return spec_view_far(own, prev, be, j);
    };
    // The walk is one dependent chain per wavefront, so what a step waits for is what the block costs.  Steps run in groups
    // of 64 (one candidate list per lane); the lists are loaded and TOUCHED before the inner loop, so that the wait for those
    // loads sits in front of it: inside, the only outstanding vector-memory operations are the write-through stores of
    // earlier steps, which no instruction of a step depends on (a wait for the lists placed inside the loop would also
    // drain those stores -- a memory round trip per step, 3/4 of the kernel's time on one deep chromosome pair).
    for (uint32_t i0 = b; i0 < be && i0 + 1 < e; i0 += 64) {
      if (i0 != b) {
        const uint32_t pn = base + BIGW + lane;
        __syncthreads();
        {
          const bool ok = pn < e;
          ring[pn % BIGW] = ok ? view_load(pn) : INF;
          rq[pn % BIGW] = ok ? s_qs[pn] : 0xffffffffu;
          rt[pn % BIGW] = ok ? s_ts[pn] : 0u;
          re[pn % BIGW] = ok ? s_te[pn] : 0u;
        }
        base += 64;
        load_cands(base);
        __syncthreads();
      }
      {
        uint64_t touch = cd[0] ^ cd[1] ^ cd[2] ^ cd[3];
        uint32_t touch32 = cj[0] ^ cj[1] ^ cj[2] ^ cj[3] ^ cn;
        asm volatile("" ::"v"(touch), "v"(touch32));
      }
      const uint32_t i_end = (i0 + 64 < be ? i0 + 64 : be);
      for (uint32_t i = i0; i < i_end && i + 1 < e; ++i) {
      const int li = (int)(i - base);
      const uint32_t nvalid = readlane_u32(cn, li);
      if (nvalid == 0) continue;
      SPEC_STAT(0, 1);
      uint64_t best_d = INF;
      uint32_t best_j = NONE;
#pragma unroll
      for (int c = 0; c < KC; ++c) {
        if (best_j == NONE && (uint32_t)c < nvalid) {
          const uint64_t d = readlane_u64(cd[c], li);
          const uint32_t j = readlane_u32(cj[c], li);
          if (d < current(j)) {
            best_d = d;
            best_j = j;
          }
        }
      }
      if (best_j == NONE && nvalid > (uint32_t)KC) {
        SPEC_STAT(1, 1);
        // every listed candidate is blocked and the window held more: evaluate it in full
        const uint64_t qe_i = s_qe[i], ts_i = s_ts[i], te_i = s_te[i];
        const uint64_t bound = qe_i + max_gap;
        uint64_t ld = INF;
        uint32_t lj2 = NONE;
        for (uint32_t j0 = i + 1; j0 < e; j0 += 64) {
          SPEC_STAT(2, 1);
          if (can_cut) {  // see chain_select_kernel
            const uint64_t wmin = wave_min_u64(ld);
            const uint64_t q0 = (j0 - base) < (uint32_t)BIGW ? (uint64_t)rq[j0 % BIGW] : (uint64_t)s_qs[j0];
            if (wmin != INF && q0 >= qe_i && (q0 - qe_i) * (q0 - qe_i) >= wmin) break;
          }
          const uint32_t j = j0 + lane;
          bool in = j < e;
          const bool inring = in && (j - base) < (uint32_t)BIGW;
          uint64_t qs_j = 0;
          if (in) {
            qs_j = inring ? rq[j % BIGW] : s_qs[j];
            in = qs_j <= bound;
          }
          if (in) {
            uint64_t d;
            const uint64_t ts_j = inring ? rt[j % BIGW] : s_ts[j], te_j = inring ? re[j % BIGW] : s_te[j];
            if (chain_dist(minus, qe_i, ts_i, te_i, qs_j, ts_j, te_j, max_gap, fifth, &d)) {
              const uint64_t cur = current(j);
              if (d < cur && d < ld) {
                ld = d;
                lj2 = j;
              }
            }
          }
          if (!__any(in)) break;
        }
        uint64_t cand = __ballot(lj2 != NONE);
        while (cand) {
          const int l = __builtin_ctzll(cand);
          cand &= cand - 1;
          const uint64_t d = readlane_u64(ld, l);
          const uint32_t j = readlane_u32(lj2, l);
          if (d < best_d || (d == best_d && j < best_j)) {
            best_d = d;
            best_j = j;
          }
        }
      }
      if (best_j == NONE) continue;
      const bool in_ring = best_j - base < (uint32_t)BIGW;
      if (lane == 0) {
        if (in_ring) ring[best_j % BIGW] = best_d;  // read back by this wavefront in program order
        // write-through: the views are what the check / final kernels (and the ring refill) read
        __hip_atomic_store(best_j < be ? &own[best_j] : &prev[best_j], best_d, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        (best_j < be ? pred_own : pred_prev)[best_j] = i;
      }
      if (!in_ring) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");  // drain before a later global read of it
      }
    }
  }
}


// SWG_WALK_PLAIN=1: the fused walk builds its candidate lists with the per-lane loop for every gap limit (A/B knob and the
// other half of the tests of the packed-key loop)
static int walk_plain_knob() {
  const char* e = getenv("SWG_WALK_PLAIN");
  return e && e[0] == '1';
}

// ---- the walk: the reference's greedy, 64 elements per step -----------------------------------------------------------
// One wavefront walks a range [bb, be) of consecutive elements in order (chain_walk_kernel), 64 at a time, lane l holding
// element i0 + l.  The sequential rule (paf_filter.rs:790-831) -- i takes the first j of its (d, j)-ordered valid list
// with d < best_pred_score[j], and best_pred_score[j] = d from then on -- is evaluated for the 64 elements at once:
//
//   * acc(i, c): d_c < score[j_c] as the scores stood BEFORE the batch (one LDS gather per listed candidate);
//   * the lanes below i change what i sees only through proposals made inside the batch: i's candidate (d, j) is blocked
//     iff a lower lane's FINAL choice is (d', j) with d' <= d (a blocked lower lane does not propose to j, but then an even
//     lower lane with d'' <= d' <= d does, so testing against current choices of lower lanes that are final is exact);
//   * every lane starts from its first acc candidate.  That choice stands unless a lower lane's final choice names the same
//     j, so only lanes that share a j with a lower lane (LDS hash filter, a superset) are re-evaluated, in ascending lane
//     order (scalar loop: v_readlane of the candidate, one ballot over the lower lanes' choices); a lane that moves to a new
//     j puts the higher lanes holding that j on the work list.  Lower lanes are final when a lane is evaluated, so the loop
//     reproduces the sequential order exactly;
//   * a lane whose listed candidates are all refused although its window held more (rare) needs the whole window: the lanes
//     below it are committed to the score ring, the wavefront scans the window together (as the old per-step kernels did)
//     and the walk goes on;
//   * commit: scores by atomic minimum (accepted distances to one j strictly decrease in lane order, so the minimum is the
//     last acceptance and its lane is j's predecessor).
//
// Ranges come in two kinds.  Ranges that begin and end at unit boundaries (units shorter than BIG_UNIT, glued into chunks of
// ~CHUNK elements) need nothing from outside: the candidate lists are built by the lanes themselves from the LDS ring
// (FUSED: no candidate arrays in HBM at all).  Blocks of long units run speculatively (see above spec_plan_kernel), lists
// from the candidate kernels.
constexpr int WALK_HASH = 256;

// The walk's work-group IS one wavefront (64 threads): what its barriers have to order are LDS accesses of different lanes,
// and the LDS executes one wavefront's instructions in issue order.  __syncthreads() would do, but it lowers to a
// workgroup-scope fence -- s_waitcnt vmcnt(0) lgkmcnt(0) -- in front of the (elided) s_barrier, i.e. every one of the six
// barriers of a batch also waited for ALL outstanding vector-memory operations: the loads prefetched for the next batch and
// the predecessor stores of this one.  This waits for the LDS queue only and keeps the compiler from moving memory
// operations across it.
__device__ __forceinline__ void wave_lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0); vmcnt and expcnt not waited for (gfx9 encoding)
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// SPEC: a block of a long unit, run speculatively (own / prev views); a template parameter so that the main walk and the
// speculative rounds are different kernels to a profiler (rocprofv3 and the library's own table then name the same launches).
template <int BIGW, bool FUSED, bool SPEC>
__global__ __launch_bounds__(64) void chain_walk_kernel(uint32_t n_blocks, const SpecBlock* __restrict__ desc, uint32_t m,
                                                        const uint64_t* __restrict__ s_grp,
                                                        const uint32_t* __restrict__ s_qs,
                                                        const uint32_t* __restrict__ s_qe,
                                                        const uint32_t* __restrict__ s_ts,
                                                        const uint32_t* __restrict__ s_te,
                                                        const uint32_t* __restrict__ s_gidx,
                                                        const uint32_t* __restrict__ group_begin, uint32_t n_groups,
                                                        uint64_t max_gap, const unsigned long long* __restrict__ c_d,
                                                        const uint32_t* __restrict__ c_j, const uint32_t* __restrict__ c_n,
                                                        unsigned long long* own, unsigned long long* prev,
                                                        uint32_t* __restrict__ pred_own, uint32_t* __restrict__ pred_prev,
                                                        unsigned long long* __restrict__ wstats,
                                                        const uint32_t* __restrict__ n_blocks_dev = nullptr,
                                                        const uint32_t* __restrict__ gate = nullptr,
                                                        int walk_plain_lists = 0) {
  constexpr bool spec = SPEC;
  if (gate && *gate == 0) return;  // (a speculative round behind a round without changes: pair_walk_long_launch)
  // The pair-resident path (swg_pair.hip): the chunk list is made on the device, so its length is read here (n_blocks is
  // then the list's capacity), and a chunk never leaves one (query, target, strand) group -- the strand comes with the
  // descriptor (pad) and the group's end is the chunk's: s_grp / s_gidx / group_begin are not read (nullptr).
  if (n_blocks_dev) n_blocks = min(n_blocks, *n_blocks_dev);
  const bool pair_desc = s_gidx == nullptr;
  __shared__ unsigned long long ring[BIGW];          // scores of positions [base, base + BIGW) as this range sees them
  __shared__ uint32_t rq[BIGW], rt[BIGW], re[BIGW];  // their q_start, t_start, t_end
  __shared__ uint32_t hcnt[WALK_HASH];   // lowest lane naming a j that hashes here
  __shared__ uint32_t hnum[WALK_HASH];   // ... and how many lanes do
  __shared__ unsigned long long l_fd[64];  // the lanes' first choices, for the lanes that share a slot with one other lane
  __shared__ uint32_t l_fj[64];
  const int lane = threadIdx.x;
  const uint64_t INF = ~0ull;
  const uint64_t fifth = max_gap / 5;
  const uint32_t gap32 = max_gap > 0xffffffffull ? 0xffffffffu : (uint32_t)max_gap;  // coordinates are u32: a larger limit cannot bind
  const uint32_t fifth32 = fifth > 0xffffffffull ? 0xffffffffu : (uint32_t)fifth;
  const bool wrap = max_gap == ~0ull;  // `max_gap + 1` (= reject) wraps to 0 in release Rust
  const bool can_cut = max_gap < (uint64_t(1) << 31);  // accepted gaps < 2^31: d cannot wrap and grows with the query gap
  const bool fast_keys = max_gap <= (uint64_t(1) << 22) && !walk_plain_lists;  // the candidates as packed keys (see the batch loop)
  for (int k = lane; k < WALK_HASH; k += 64) {
    hcnt[k] = 0xffffffffu;
    hnum[k] = 0;
  }
  wave_lds_sync();
  for (uint32_t bk = blockIdx.x; bk < n_blocks; bk += gridDim.x) {
    const SpecBlock D = desc[bk];
    const uint32_t b = D.bb, be = D.be, ue = D.ue;  // i runs over [b, be), j may reach into the next block (< ue)
    if (be <= b || ue - b < 2) continue;
    if (!SPEC && pair_desc && be - b >= LABEL_CAP_ELEMS) continue;  // a long unit of the pair-resident path: walked in blocks
    auto view_ptr = [&](uint32_t p) -> unsigned long long* { return (spec && p >= be) ? &prev[p] : &own[p]; };
    auto view_load = [&](uint32_t p) -> uint64_t {
      return __hip_atomic_load(view_ptr(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    };
    uint32_t base = b;
    // A chunk of whole units (!SPEC) starts from "no score anywhere" and nothing outside the chunk ever reads its scores, so
    // they live in the LDS ring alone: the score array in memory (`own`) is neither initialised beforehand nor read -- until
    // the first commit beyond the ring (rare), which first writes "no score" to the part of the chunk that has not entered
    // the ring yet and from then on (far_any) the chunk reads that array like a speculative block does.
    bool far_any = SPEC;
    wave_lds_sync();
    for (uint32_t p = b + lane; p < b + BIGW; p += 64) {
      const bool ok = p < ue;
      ring[p % BIGW] = (ok && SPEC) ? view_load(p) : INF;
      rq[p % BIGW] = ok ? s_qs[p] : 0xffffffffu;
      rt[p % BIGW] = ok ? s_ts[p] : 0u;
      re[p % BIGW] = ok ? s_te[p] : 0u;
    }
    wave_lds_sync();
    auto current = [&](uint32_t j) -> uint64_t {  // the committed score of j
      if (j - base < (uint32_t)BIGW) return ring[j % BIGW];
      return far_any ? view_load(j) : INF;
    };
    // What a batch needs from memory is requested one batch ahead (the walk of a range is one dependent chain per wavefront:
    // what a batch waits for is what the range costs): the 64 positions that enter the ring, and the batch's own elements --
    // their candidate lists (!FUSED) or their coordinates and group ends (FUSED).
    struct Pre {
      uint64_t view;
      uint32_t q, t, e;  // ring refill
      uint64_t bd[KC];
      uint32_t bj[KC];
      uint32_t nv;                    // lists
      uint32_t qe, ts, te, minus, ei; // own element
    };
    auto prefetch = [&](uint32_t i0n, uint32_t base_n, bool refill) {
      Pre P;
      P.view = INF;
      P.q = 0xffffffffu;
      P.t = P.e = 0;
      if (refill) {
        const uint32_t pn = base_n + BIGW - 64 + lane;  // the positions that enter when the ring moves to base_n
        if (pn < ue) {
          if (far_any) P.view = view_load(pn);
          P.q = s_qs[pn];
          P.t = s_ts[pn];
          P.e = s_te[pn];
        }
      }
      const uint32_t i = i0n + lane;
      const bool valid = i < be && i + 1 < ue;
#pragma unroll
      for (int c = 0; c < KC; ++c) {
        P.bd[c] = INF;
        P.bj[c] = NONE;
      }
      P.nv = 0;
      P.qe = P.ts = P.te = P.minus = 0;
      P.ei = ue;
      if (valid) {
        if (!spec && !pair_desc) {
          const uint32_t g = s_gidx[i];
          const uint32_t ge = (g + 1 < n_groups) ? group_begin[g + 1] : m;
          if (ge < P.ei) P.ei = ge;
        }
        if (FUSED) {
          P.qe = s_qe[i];
          P.ts = s_ts[i];
          P.te = s_te[i];
          P.minus = pair_desc ? D.pad : (uint32_t)(s_grp[i] & 1ull);
        } else {
#pragma unroll
          for (int c = 0; c < KC; ++c) {
            P.bd[c] = c_d[(uint64_t)c * m + i];
            P.bj[c] = c_j[(uint64_t)c * m + i];
          }
          P.nv = c_n[i];
        }
      }
      return P;
    };
    Pre cur = prefetch(b, base, false);
    bool far_dirty = false;  // a score beyond the ring was written during the batch: the prefetched refill may be stale
    for (uint32_t i0 = b; i0 < be && i0 + 1 < ue; i0 += 64) {
      if (i0 != b) {
        const uint32_t pn = base + BIGW + lane;
        if (far_dirty && pn < ue) cur.view = view_load(pn);
        far_dirty = false;
        wave_lds_sync();
        ring[pn % BIGW] = cur.view;
        rq[pn % BIGW] = cur.q;
        rt[pn % BIGW] = cur.t;
        re[pn % BIGW] = cur.e;
        base += 64;
        wave_lds_sync();
      }
      const bool more = i0 + 64 < be && i0 + 65 < ue;
      Pre nxt = cur;
      if (more) nxt = prefetch(i0 + 64, base + 64, true);
      const uint32_t i = i0 + lane;
      const bool valid = i < be && i + 1 < ue;
      // ---- the lane's candidate list
      uint64_t bd[KC];
      uint32_t bj[KC];
      uint32_t nv = cur.nv;
#pragma unroll
      for (int c = 0; c < KC; ++c) {
        bd[c] = cur.bd[c];
        bj[c] = cur.bj[c];
      }
      const uint32_t e_i = cur.ei;  // end of the element's (query, target, strand) group inside the range
      bool listed = false;  // (wave-uniform) the part of the windows inside the ring went through the packed-key loop
      uint32_t j_next = i + 1;
      bool open_next = valid;
      if (FUSED && fast_keys) {
        // The windows' part inside the ring, for gap limits up to 2^22 (the CLI's 50,000 among them): ONE loop for the
        // wavefront, as long as its longest window, without a branch in its body.  A candidate is one 64-bit key --
        // bit 62 | d << 16 | (j - i), d = q_gap^2 + r_gap^2 < 2^45 -- and the keys are read as doubles: positive normal
        // numbers order like their bit patterns, so v_min_f64 / v_max_f64 are the 64-bit integer minimum and maximum in one
        // instruction each, and keeping the KC smallest keys in order is a chain of KC min / max pairs (no compare, no
        // select, no "does it enter the list" branch); a rejected pair is +infinity.  Keys are distinct ((j - i) is), so
        // the list is the (d, j)-ordered list of paf_filter.rs:798-836 exactly.  The per-lane loop this replaces spent
        // ~40 scalar and branch instructions per step on exec-mask bookkeeping and ~20 vector ones per insertion.
        const uint32_t qe_i = cur.qe;
        const bool minus = cur.minus != 0;
        const uint64_t bound64 = (uint64_t)qe_i + max_gap;  // (no wrap: max_gap <= 2^22)
        const uint32_t bound = bound64 > 0xffffffffull ? 0xffffffffu : (uint32_t)bound64;
        const uint32_t r_i = minus ? cur.ts : cur.te;
        const uint32_t* r_ring = minus ? re : rt;
        const uint32_t e_ring = min(e_i, base + (uint32_t)BIGW);
        constexpr uint64_t KEY_INF = 0x7ff0000000000000ull, KEY_BIT = 1ull << 62;
        double kb[KC];
#pragma unroll
        for (int c = 0; c < KC; ++c) kb[c] = __longlong_as_double((long long)KEY_INF);
        // (the loop's own variables: `off` = j - i, `sl` = the ring slot's byte offset, both advanced only while the lane is
        // inside its window; the two limits sit in vector registers, the key's bit 62 in the addend's high word)
        uint32_t off = 1, cnt = 0, sl = ((i + 1) % BIGW) * 4u;
        const uint32_t off_end = e_ring > i ? e_ring - i : 0u;  // j < e_ring  <=>  off < off_end
        uint32_t gap_v = gap32, fifth_v = fifth32, key_hi = (uint32_t)(KEY_BIT >> 32);
        asm volatile("" : "+v"(gap_v), "+v"(fifth_v), "+v"(key_hi));
        const char* const rq_b = reinterpret_cast<const char*>(rq);
        const char* const rr_b = reinterpret_cast<const char*>(r_ring);
        bool open = valid;
        for (;;) {
          const bool in = open && off < off_end;
          if (__ballot(in) == 0ull) break;
          uint32_t qs_j = *reinterpret_cast<const uint32_t*>(rq_b + sl), r_j = *reinterpret_cast<const uint32_t*>(rr_b + sl);
          asm volatile("" : "+v"(qs_j), "+v"(r_j));  // both reads issued here, one wait
          const bool inw = in && qs_j <= bound;  // sorted by q_start (paf_filter.rs:794-796)
          open = open && !(in && qs_j > bound);
          const uint32_t q_gap = absdiff_vv(qs_j, qe_i), r_gap = absdiff_vv(r_j, r_i);
          // (at q_gap == 0 / r_gap == 0 either limit passes: the sides need no "or equal")
          const uint32_t lim_q = qs_j < qe_i ? fifth_v : gap_v;
          const uint32_t lim_r = ((r_j > r_i) != minus) ? gap_v : fifth_v;
          const bool ok = inw && q_gap <= lim_q && r_gap <= lim_r;
          const uint32_t qa = q_gap << 8, ra = r_gap << 8;  // (their squares: the gaps' squares << 16; garbage when !ok)
          const uint64_t add = ((uint64_t)key_hi << 32) | off;
          uint64_t key = (uint64_t)qa * qa + ((uint64_t)ra * ra + add);
          key = ok ? key : KEY_INF;
          cnt += ok ? 1u : 0u;
          const uint32_t step = in ? 1u : 0u;
          off += step;
          sl = (sl + (step << 2)) & (BIGW * 4u - 1u);
          double t = __longlong_as_double((long long)key);
#pragma unroll
          for (int c = 0; c < KC; ++c) {  // (the list entry is updated in place: no register copies at the loop's back edge)
            double hi = t;
            if (c + 1 < KC) asm("v_max_f64 %0, %1, %2" : "=v"(hi) : "v"(kb[c]), "v"(t));
            asm("v_min_f64 %0, %0, %1" : "+v"(kb[c]) : "v"(t));
            t = hi;
          }
        }
        const uint32_t j = i + off;
        nv = cnt < (uint32_t)KC + 1u ? cnt : (uint32_t)KC + 1u;
#pragma unroll
        for (int c = 0; c < KC; ++c) {
          const uint64_t bits = (uint64_t)__double_as_longlong(kb[c]);
          const bool none = bits == KEY_INF;
          bd[c] = none ? INF : (bits & ~KEY_BIT) >> 16;
          bj[c] = none ? NONE : i + (uint32_t)(bits & 0xffffu);
        }
        listed = true;
        j_next = j;
        open_next = open;
      }
      if (FUSED && valid && (!listed || (open_next && j_next < e_i))) {
        // d(i, j) of paf_filter.rs:798-836 in 32-bit arithmetic, selects instead of branches (as in chain_candidates_wave_kernel)
        const uint32_t qe_i = cur.qe;
        const bool minus = cur.minus != 0;
        const uint64_t bound64 = (uint64_t)qe_i + max_gap;  // wrapping, as release Rust
        const uint32_t bound = bound64 > 0xffffffffull ? 0xffffffffu : (uint32_t)bound64;
        // the target coordinates that enter d(i, j): plus strand t_start[j] against t_end[i], minus strand t_end[j] against
        // t_start[i] -- chosen once per element (two LDS reads per j instead of three)
        const uint32_t r_i = minus ? cur.ts : cur.te;
        const uint32_t* r_ring = minus ? re : rt;
        const uint32_t* __restrict__ r_mem = minus ? s_te : s_ts;
        // one j: a gap is |a - b|; it passes when it is at most the limit on the gap side, a fifth of the limit on the
        // overlap side (the predicate of paf_filter.rs:798-836; with the limit at u64::MAX both bounds are 2^32 - 1:
        // everything passes)
        auto consider = [&](uint32_t j, uint32_t qs_j, uint32_t r_j) {
          const uint32_t q_gap = absdiff_vv(qs_j, qe_i), r_gap = absdiff_vv(r_j, r_i);
          const uint32_t lim_q = qs_j >= qe_i ? gap32 : fifth32;
          const uint32_t lim_r = (minus ? r_j <= r_i : r_j >= r_i) ? gap32 : fifth32;
          if ((q_gap > lim_q) | (r_gap > lim_r)) return;
          const uint64_t d = (uint64_t)q_gap * q_gap + (uint64_t)r_gap * r_gap;  // wrapping, as release Rust
          if (nv <= (uint32_t)KC) ++nv;
          if (d < bd[KC - 1]) {  // insert keeping (d asc, j asc), see chain_candidates_kernel
            uint64_t cd = d;
            uint32_t cj = j;
            bool placed = false;
#pragma unroll
            for (int c = 0; c < KC; ++c) {
              if (placed || cd < bd[c]) {
                placed = true;
                const uint64_t td = bd[c];
                const uint32_t tj = bj[c];
                bd[c] = cd;
                bj[c] = cj;
                cd = td;
                cj = tj;
              }
            }
          }
        };
        // The window in two stretches: the part inside the ring (nearly always all of it) reads LDS, both coordinates of a
        // step requested together; what lies beyond reads memory.  One loop with `in the ring ? LDS : memory` per load is
        // compiled to flat loads of a selected pointer, two dependent ones per step with a full wait each.
        const uint32_t e_ring = listed ? j_next : min(e_i, base + (uint32_t)BIGW);  // (listed: only what lies beyond the ring is left)
        uint32_t j = j_next;
        bool open = true;  // the window has not ended yet
        for (; j < e_ring; ++j) {
          uint32_t qs_j = rq[j % BIGW], r_j = r_ring[j % BIGW];
          asm volatile("" : "+v"(qs_j), "+v"(r_j));  // both reads issued here, one wait (the second would sink below the test)
          if (qs_j > bound) {  // sorted by q_start (paf_filter.rs:794-796)
            open = false;
            break;
          }
          consider(j, qs_j, r_j);
        }
        if (open)
          for (; j < e_i; ++j) {
            const uint32_t qs_j = s_qs[j];
            if (qs_j > bound) break;
            consider(j, qs_j, r_mem[j]);
          }
      }
      // ---- acc bits and the first acceptable candidate
      uint32_t acc = 0;
#pragma unroll
      for (int c = 0; c < KC; ++c)
        if ((uint32_t)c < nv && bd[c] < current(bj[c])) acc |= 1u << c;
      uint64_t fd = INF;
      uint32_t fj = NONE;
#pragma unroll
      for (int c = KC - 1; c >= 0; --c)
        if (acc & (1u << c)) {
          fd = bd[c];
          fj = bj[c];
        }
      // ---- lanes that share their j with another lane (hash filter; superset), and lanes that need their whole window
      // (the slot keeps the LOWEST lane that names a j hashing there: that lane's choice stands unless a lower lane moves to
      // its j later -- which puts it on the list then --, so only the other lanes of the slot start on the work list; and
      // when the slot has exactly two lanes, the higher one only if the lower one really blocks it: same j, distance not
      // larger than its own.  That second test is worth its LDS traffic in the blocks of long units (S-big1 chain_walk_spec
      // 6.2 -> 5.8 ms: 40 % fewer lanes on the list) and not in the chunks of short ones (S-pan chain_walk 3.2 -> 3.3 ms).)
      constexpr bool PAIR_TEST = SPEC;
      const uint32_t hs = fj & (WALK_HASH - 1);
      if (fj != NONE) {
        atomicMin(&hcnt[hs], (uint32_t)lane);
        if (PAIR_TEST) atomicAdd(&hnum[hs], 1u);
      }
      if (PAIR_TEST) {
        l_fd[lane] = fd;
        l_fj[lane] = fj;
      }
      wave_lds_sync();
      bool shared_j = false;
      if (fj != NONE) {
        const uint32_t low = hcnt[hs];
        if (low != (uint32_t)lane) shared_j = !PAIR_TEST || hnum[hs] > 2u || (l_fj[low] == fj && (uint64_t)l_fd[low] <= fd);
      }
      wave_lds_sync();
      if (fj != NONE) {
        hcnt[hs] = 0xffffffffu;
        if (PAIR_TEST) hnum[hs] = 0;
      }
      uint64_t work = __ballot(shared_j || (fj == NONE && nv > (uint32_t)KC));
      if (wstats && lane == 0) {  // SWG_WALK_STATS: batches, lanes on the work list
        atomicAdd(&wstats[0], 1ull);
        atomicAdd(&wstats[1], (unsigned long long)__popcll(work));
      }
      uint32_t committed = 0;  // lanes below this one are in the ring
      auto commit = [&](uint32_t upto) {  // lanes [committed, upto)
        const bool me = (uint32_t)lane >= committed && (uint32_t)lane < upto && fj != NONE;
        const bool inr = me && (fj - base < (uint32_t)BIGW);
        // in the ring: LDS minimum, then the lane that finds its own distance there is the last acceptance
        if (inr) atomicMin(&ring[fj % BIGW], (unsigned long long)fd);
        wave_lds_sync();
        if (inr && (uint64_t)ring[fj % BIGW] == fd) {
          ((spec && fj >= be) ? pred_prev : pred_own)[fj] = i;
          if (spec && fj >= be)  // the next block's input: must be in memory at the end of the round
            __hip_atomic_store(&prev[fj], fd, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        // beyond the ring (rare): one lane at a time in lane order, a returning atomic minimum at the memory side -- the lane
        // that lowers the score is (so far) the last acceptance; no fence, no cache maintenance
        uint64_t far = __ballot(me && !inr);
        if (far) {
          far_dirty = true;
          if (!far_any) {  // (wave-uniform) the first commit beyond the ring in this chunk: see far_any above
            for (uint32_t p = base + BIGW + lane; p < ue; p += 64) own[p] = INF;
            __threadfence();  // the stores are performed before the atomics below (once per chunk at most)
            far_any = true;
          }
        }
        while (far) {
          const int l = __builtin_ctzll(far);
          far &= far - 1;
          if (lane == l) {
            const unsigned long long old = atomicMin(view_ptr(fj), (unsigned long long)fd);
            if (old > fd) ((spec && fj >= be) ? pred_prev : pred_own)[fj] = i;
          }
        }
        committed = upto;
      };
      while (work) {
        const int l = __builtin_ctzll(work);
        work &= work - 1;
        const uint32_t nv_l = readlane_u32(nv, l), acc_l = readlane_u32(acc, l);
        uint64_t nd = INF;
        uint32_t nj = NONE;
        const uint64_t lower = l ? (~0ull >> (64 - l)) : 0ull;
#pragma unroll
        for (int c = 0; c < KC; ++c) {
          if (nj == NONE && (acc_l & (1u << c))) {
            const uint64_t d = readlane_u64(bd[c], l);
            const uint32_t j = readlane_u32(bj[c], l);
            if ((__ballot(fj == j && fd <= d) & lower) == 0) {
              nd = d;
              nj = j;
            }
          }
        }
        if (nj == NONE && nv_l > (uint32_t)KC) {
          // every listed candidate is refused and the window held more: the whole window (paf_filter.rs:794-826), against
          // the scores as they stand after every lower lane
          commit((uint32_t)l);
          if (wstats && lane == 0) atomicAdd(&wstats[2], 1ull);  // whole-window passes
          const uint32_t ii = i0 + l;
          const uint64_t qe_l = s_qe[ii], ts_l = s_ts[ii], te_l = s_te[ii];
          const bool minus_l = pair_desc ? D.pad != 0 : (s_grp[ii] & 1ull) != 0;
          const uint32_t e_l = readlane_u32(e_i, l);
          const uint64_t bound = qe_l + max_gap;
          uint64_t ld = INF;
          uint32_t lj2 = NONE;
          for (uint32_t j0 = ii + 1; j0 < e_l; j0 += 64) {
            if (wstats && lane == 0) atomicAdd(&wstats[3], 1ull);  // their 64-element batches
            if (can_cut) {  // the batch starts past q_end[i] and its smallest query gap already reaches the best distance held
              const uint64_t wmin = wave_min_u64(ld);
              const uint64_t q0 = (j0 - base) < (uint32_t)BIGW ? (uint64_t)rq[j0 % BIGW] : (uint64_t)s_qs[j0];
              if (wmin != INF && q0 >= qe_l && (q0 - qe_l) * (q0 - qe_l) >= wmin) break;
            }
            const uint32_t j = j0 + lane;
            bool in = j < e_l;
            const bool inring = in && (j - base) < (uint32_t)BIGW;
            uint64_t qs_j = 0;
            if (in) {
              qs_j = inring ? rq[j % BIGW] : s_qs[j];
              in = qs_j <= bound;
            }
            if (in) {
              uint64_t d;
              const uint64_t ts_j = inring ? rt[j % BIGW] : s_ts[j], te_j = inring ? re[j % BIGW] : s_te[j];
              if (chain_dist(minus_l, qe_l, ts_l, te_l, qs_j, ts_j, te_j, max_gap, fifth, &d)) {
                const uint64_t cur = current(j);
                if (d < cur && d < ld) {
                  ld = d;
                  lj2 = j;
                }
              }
            }
            if (!__any(in)) break;
          }
          uint64_t cand = __ballot(lj2 != NONE);
          while (cand) {
            const int c2 = __builtin_ctzll(cand);
            cand &= cand - 1;
            const uint64_t d = readlane_u64(ld, c2);
            const uint32_t j = readlane_u32(lj2, c2);
            if (d < nd || (d == nd && j < nj)) {
              nd = d;
              nj = j;
            }
          }
        }
        const uint32_t old_j = readlane_u32(fj, l);
        if (lane == l) {
          fd = nd;
          fj = nj;
        }
        if (nj != NONE && nj != old_j) {
          const uint64_t woken = __ballot(fj == nj && fd >= nd) & ~lower & ~(1ull << l);  // higher lanes holding the new j that it blocks
          if (wstats && lane == 0) atomicAdd(&wstats[4], (unsigned long long)__popcll(woken & ~work));
          work |= woken;
        }
      }
      commit(64u);
      cur = nxt;
    }
  }
}


// Pair-resident path: the block plan of the long chunks, made on the device.  One work-group per long chunk: the longest window
// (how many later members start within q_end + gap: the members are sorted by q_start, a binary search each), the block size S
// (a multiple of 64, >= longest window + 1, >= 512), the block descriptors appended to one list, `ext` cleared.
constexpr uint32_t PAIR_LONG_WINDOW_MAX = 512;
__global__ __launch_bounds__(EW) void pair_long_plan_kernel(uint32_t cap_long, const uint32_t* __restrict__ n_long_dev,
                                                            const uint32_t* __restrict__ long_list, const SpecBlock* __restrict__ chunks,
                                                            const uint32_t* __restrict__ s_qs, const uint32_t* __restrict__ s_qe,
                                                            uint64_t max_gap, SpecBlock* __restrict__ desc, uint32_t cap_spec,
                                                            uint32_t* __restrict__ counters /* [0] n_spec, [1] overflow */,
                                                            unsigned long long* __restrict__ ext, uint32_t* __restrict__ flags,
                                                            uint32_t fallback_bit) {
  __shared__ uint32_t wmaxs[EW / 64];
  __shared__ uint32_t sh_off;
  const uint32_t n_long = *n_long_dev < cap_long ? *n_long_dev : cap_long;
  for (uint32_t c = blockIdx.x; c < n_long; c += gridDim.x) {
    const SpecBlock D = chunks[long_list[c]];
    const uint32_t b = D.bb, e = D.be;
    // An upper bound of the longest window, one binary search per 64 members instead of one per member: a member's window
    // ends no later than q_start[last of its 64] + (longest member + gap) reaches, because the members are sorted by q_start.
    uint32_t ml = 0;
    for (uint32_t p = b + threadIdx.x; p < e; p += EW) {
      const uint32_t len = s_qe[p] - s_qs[p];
      ml = len > ml ? len : ml;
      ext[p] = ~0ull;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const uint32_t t = __shfl_xor(ml, o, 64);
      if (t > ml) ml = t;
    }
    __syncthreads();
    if ((threadIdx.x & 63) == 0) wmaxs[threadIdx.x >> 6] = ml;
    __syncthreads();
    for (int k = 0; k < EW / 64; ++k)
      if (wmaxs[k] > ml) ml = wmaxs[k];
    const uint64_t reach = (uint64_t)ml + (max_gap > (uint64_t(1) << 33) ? (uint64_t(1) << 33) : max_gap);
    uint32_t w = 0;
    for (uint32_t p0 = b + threadIdx.x * 64u; p0 < e; p0 += EW * 64u) {
      const uint32_t last = p0 + 63 < e ? p0 + 63 : e - 1;
      const uint64_t bound = (uint64_t)s_qs[last] + reach;
      uint32_t lo = last + 1, hi = e;  // first position in (last, e) with q_start > bound
      while (lo < hi) {
        const uint32_t mid = lo + ((hi - lo) >> 1);
        if ((uint64_t)s_qs[mid] <= bound)
          lo = mid + 1;
        else
          hi = mid;
      }
      const uint32_t x = lo - 1 - p0;
      w = x > w ? x : w;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const uint32_t t = __shfl_xor(w, o, 64);
      if (t > w) w = t;
    }
    __syncthreads();
    if ((threadIdx.x & 63) == 0) wmaxs[threadIdx.x >> 6] = w;
    __syncthreads();
    for (int k = 0; k < EW / 64; ++k)
      if (wmaxs[k] > w) w = wmaxs[k];
    // a deep unit (hundreds of members inside the gap limit of one): the blocks of the speculative walk would have to be as long
    // as the windows, and each step a pass over hundreds of candidates -- that is the global-sort stage's case (candidate lists by
    // a wavefront per member): the call is handed over before any round runs
    if (w > PAIR_LONG_WINDOW_MAX) {
      if (threadIdx.x == 0) atomicOr(flags, fallback_bit);
      continue;
    }
    uint32_t S = ((w + 1 + 63) / 64) * 64;
    if (S < 512) S = 512;
    const uint32_t nb = (e - b + S - 1) / S;
    if (threadIdx.x == 0) sh_off = atomicAdd(&counters[0], nb);
    __syncthreads();
    const uint32_t off = sh_off;
    if (off + nb > cap_spec) {
      if (threadIdx.x == 0) counters[1] = 1;
      continue;
    }
    for (uint32_t k = threadIdx.x; k < nb; k += EW) {
      SpecBlock d;
      d.ue = e;
      d.bb = b + k * S;
      d.be = d.bb + S < e ? d.bb + S : e;
      d.pad = D.pad;
      desc[off + k] = d;
    }
  }
}
// ... and what the rounds leave behind: a plan that did not fit, or rounds that did not settle, mark the call (the caller's
// flag word: it then takes the global-sort path)
__global__ void pair_long_verdict_kernel(const uint32_t* __restrict__ counters, const uint32_t* __restrict__ changed_last,
                                         uint32_t* __restrict__ flags, uint32_t bit) {
  if (threadIdx.x == 0 && blockIdx.x == 0 && (counters[1] || *changed_last)) atomicOr(flags, bit);
}

// Chunks of the walk over the units shorter than BIG_UNIT: unit u opens a chunk when it is the first unit to begin in its
// WALK_CHUNK-element cell, or when it or its predecessor is a long unit (long units are chunks of their own, skipped here).
__global__ __launch_bounds__(EW) void chunk_flag_kernel(uint32_t n_units, const uint32_t* __restrict__ unit_begin,
                                                        const uint8_t* __restrict__ is_big, uint8_t* __restrict__ flag) {
  uint32_t u = blockIdx.x * EW + threadIdx.x;
  if (u >= n_units) return;
  bool f = u == 0 || is_big[u] || is_big[u - 1];
  if (!f) f = unit_begin[u] / WALK_CHUNK != unit_begin[u - 1] / WALK_CHUNK;
  flag[u] = f ? 1 : 0;
}
__global__ __launch_bounds__(EW) void chunk_desc_kernel(uint32_t n_chunks, const uint32_t* __restrict__ chunk_unit,
                                                        uint32_t n_units, const uint32_t* __restrict__ unit_begin,
                                                        const uint8_t* __restrict__ is_big, uint32_t m,
                                                        SpecBlock* __restrict__ desc) {
  uint32_t c = blockIdx.x * EW + threadIdx.x;
  if (c >= n_chunks) return;
  const uint32_t u = chunk_unit[c];
  const uint32_t b = unit_begin[u];
  const uint32_t un = c + 1 < n_chunks ? chunk_unit[c + 1] : n_units;
  const uint32_t e = un < n_units ? unit_begin[un] : m;
  SpecBlock d;
  d.bb = b;
  d.be = is_big[u] ? b : e;  // long units: empty range here (they take the block-speculative path)
  d.ue = d.be;
  d.pad = 0;
  desc[c] = d;
}
// The long units as ranges: big_rng[2k] = begin, big_rng[2k + 1] = end of the k-th long unit (ascending).
__global__ __launch_bounds__(EW) void big_ranges_kernel(uint32_t n_big, const uint32_t* __restrict__ big_list, uint32_t n_units,
                                                        const uint32_t* __restrict__ unit_begin, uint32_t m,
                                                        uint32_t* __restrict__ big_rng) {
  uint32_t k = blockIdx.x * EW + threadIdx.x;
  if (k >= n_big) return;
  const uint32_t u = big_list[k];
  big_rng[2 * k] = unit_begin[u];
  big_rng[2 * k + 1] = (u + 1 < n_units) ? unit_begin[u + 1] : m;
}
// index of the long unit that holds p, or NONE (the ranges are disjoint and ascending)
__device__ __forceinline__ uint32_t big_unit_of(const uint32_t* __restrict__ big_rng, uint32_t n_big, uint32_t p) {
  uint32_t l = 0, r = n_big;  // first k with begin > p
  while (l < r) {
    const uint32_t mid = l + ((r - l) >> 1);
    if (big_rng[2 * mid] <= p)
      l = mid + 1;
    else
      r = mid;
  }
  if (l == 0) return NONE;
  return p < big_rng[2 * (l - 1) + 1] ? l - 1 : NONE;
}
// Membership of the long units, element by element and per 1024-element span.  One work-group per span: the long units are
// few, so a span first asks whether any of them touches it at all (two searches) and then writes its 1024 flags -- nothing of
// size m is read (round 3 looked every element's unit up in a u32 scan of the unit flags).
__global__ __launch_bounds__(EW) void big_member_fill_kernel(uint64_t m, uint32_t n_big, const uint32_t* __restrict__ big_rng,
                                                             uint8_t* __restrict__ f, uint8_t* __restrict__ span_big) {
  const uint64_t sp = blockIdx.x;
  const uint64_t base = sp << BIG_SPAN_SHIFT;
  uint64_t top = base + (uint64_t(1) << BIG_SPAN_SHIFT);
  if (top > m) top = m;
  // the long units that can touch [base, top): from the one holding `base` (or the first beginning after it) on
  uint32_t l = 0, r = n_big;  // first k with end > base
  while (l < r) {
    const uint32_t mid = l + ((r - l) >> 1);
    if (big_rng[2 * mid + 1] <= base)
      l = mid + 1;
    else
      r = mid;
  }
  const bool any = l < n_big && big_rng[2 * l] < top;
  if (threadIdx.x == 0) span_big[sp] = any ? 1 : 0;
  for (uint64_t p = base + threadIdx.x; p < top; p += EW) {
    uint8_t b = 0;
    if (any) {
      for (uint32_t k = l; k < n_big && big_rng[2 * k] <= p; ++k)
        if (p < big_rng[2 * k + 1]) {
          b = 1;
          break;
        }
    }
    f[p] = b;
  }
}

// Longest window per long unit (indexed like big_rng) when a unit holds millions of elements: every member contributes its
// window extent to its unit's maximum -- one atomic per wavefront whose members share the unit, which they nearly always do.
__global__ __launch_bounds__(EW) void big_wmax_kernel(uint64_t m, const uint8_t* __restrict__ big_member,
                                                      const uint32_t* __restrict__ big_rng, uint32_t n_big,
                                                      const uint32_t* __restrict__ c_ext, uint32_t* __restrict__ wmax_k,
                                                      const uint8_t* __restrict__ span_big) {
  const int lane = threadIdx.x & 63;
  uint32_t cur_k = NONE, cur_w = 0;  // wave-uniform: the unit this wavefront is accumulating and its maximum so far
  for (uint64_t base = (uint64_t)blockIdx.x * EW; base < m; base += (uint64_t)gridDim.x * EW) {  // block-uniform trip count
    if (!span_big[base >> BIG_SPAN_SHIFT]) continue;
    const uint64_t p = base + threadIdx.x;
    const bool big = p < m && big_member[p] != 0;
    const uint64_t bm = __ballot(big);
    if (bm == 0) continue;  // wave-uniform
    const uint32_t k = big ? big_unit_of(big_rng, n_big, (uint32_t)p) : NONE;
    uint32_t w = big ? c_ext[p] : 0u;
    const uint32_t k0 = (uint32_t)__shfl((int)k, (int)__builtin_ctzll(bm), 64);
    if (__ballot(big && k != k0) == 0) {  // the wavefront's members share one unit
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) {
        const uint32_t t = __shfl_xor(w, o, 64);
        if (t > w) w = t;
      }
      if (k0 != cur_k) {  // flush the previous unit (one atomic per wavefront and unit, not per row)
        if (lane == 0 && cur_w) atomicMax(&wmax_k[cur_k], cur_w);
        cur_k = k0;
        cur_w = 0;
      }
      if (w > cur_w) cur_w = w;
    } else if (big && w) {
      atomicMax(&wmax_k[k], w);
    }
  }
  if (lane == 0 && cur_w) atomicMax(&wmax_k[cur_k], cur_w);
}

// Window extent of the members of long units (how many later elements of the unit start within q_end + gap): what the block
// plan needs (a block must be at least as long as the longest window).  The unit is sorted by q_start: a binary search.
__global__ __launch_bounds__(EW) void window_extent_kernel(uint64_t m, const uint8_t* __restrict__ big_member,
                                                           const uint32_t* __restrict__ big_rng, uint32_t n_big,
                                                           const uint32_t* __restrict__ s_qs, const uint32_t* __restrict__ s_qe,
                                                           uint64_t max_gap, uint32_t* __restrict__ ext,
                                                           const uint8_t* __restrict__ span_big) {
  uint64_t p = (uint64_t)blockIdx.x * EW + threadIdx.x;
  if (p >= m) return;
  if (span_big && !span_big[p >> BIG_SPAN_SHIFT]) return;  // (extents are only read for members of long units)
  uint32_t x = 0;
  if (big_member[p]) {
    const uint32_t e = big_rng[2 * big_unit_of(big_rng, n_big, (uint32_t)p) + 1];  // end of p's unit
    const uint64_t bound = (uint64_t)s_qe[p] + max_gap;  // wrapping, as release Rust
    uint32_t lo = (uint32_t)p + 1, hi = e;  // first position in (p, e) with q_start > bound
    while (lo < hi) {
      const uint32_t mid = lo + ((hi - lo) >> 1);
      if ((uint64_t)s_qs[mid] <= bound)
        lo = mid + 1;
      else
        hi = mid;
    }
    x = lo - 1 - (uint32_t)p;
  }
  ext[p] = x;
}
__global__ __launch_bounds__(EW) void unit_big_flag_kernel(uint32_t n_units, const uint32_t* __restrict__ unit_begin,
                                                           uint32_t m, uint8_t* __restrict__ is_big) {
  uint32_t u = blockIdx.x * EW + threadIdx.x;
  if (u >= n_units) return;
  const uint32_t e = (u + 1 < n_units) ? unit_begin[u + 1] : m;
  const uint8_t f = (e - unit_begin[u]) >= BIG_UNIT ? 1 : 0;
  is_big[u] = f;
}

__global__ __launch_bounds__(EW) void unit_mid_flag_kernel(uint32_t n_units, const uint32_t* __restrict__ unit_begin,
                                                           uint32_t m, uint8_t* __restrict__ is_mid) {
  uint32_t u = blockIdx.x * EW + threadIdx.x;
  if (u >= n_units) return;
  const uint32_t e = (u + 1 < n_units) ? unit_begin[u + 1] : m;
  const uint32_t len = e - unit_begin[u];
  is_mid[u] = (len > SMALL_UNIT && len < BIG_UNIT) ? 1 : 0;
}

// Independent sub-ranges of a group: position p opens a new unit when q_start[p] lies beyond every earlier
// q_end of the group by more than the gap -- no (i, j) pair of the reference's window test
// (`q_start[j] <= q_end[i] + gap`, paf_filter.rs:786-796) can then straddle p, so the greedy on either side is
// independent.  One wavefront per group, 64 elements per step, running maximum carried along.
// (F: uint8_t -- the walk's path: the unit list comes out of a compaction of byte flags -- or uint32_t, the round-2 kernels')
template <class F>
__global__ __launch_bounds__(EW) void chain_cuts_kernel(uint32_t n_groups, const uint32_t* __restrict__ group_begin,
                                                        uint32_t m, const uint32_t* __restrict__ s_qs,
                                                        const uint32_t* __restrict__ s_qe, uint64_t max_gap,
                                                        F* __restrict__ unit_flag) {
  // 512 elements per step: eight coalesced 64-element rows requested together, then one wave scan per row with the carry from
  // the rows before.  (Round 4.  The kernel lasts as long as its LONGEST group, which one wavefront walks step by step, one
  // memory round trip each; four consecutive elements per lane, 256 per step, were 113 steps for S-pan's largest group, and
  // sixteen per lane made every load a 64-byte-stride gather: 0.60 -> 0.82 ms.)
  constexpr int K = 8;
  const int lane = threadIdx.x & 63;
  const uint32_t wave_global = blockIdx.x * (EW / 64) + (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));  // wave-uniform, and visibly so
  const uint32_t n_waves = (gridDim.x * EW) >> 6;
  for (uint32_t g = wave_global; g < n_groups; g += n_waves) {
    const uint32_t b = group_begin[g];
    const uint32_t e = (g + 1 < n_groups) ? group_begin[g + 1] : m;
    uint32_t carry = 0;  // max q_end over [b, row)
    for (uint32_t p0 = b; p0 < e; p0 += 64 * K) {
      uint32_t qe[K], qs[K];
#pragma unroll
      for (int k = 0; k < K; ++k) {
        const uint32_t p = p0 + (uint32_t)k * 64 + (uint32_t)lane;
        qe[k] = p < e ? s_qe[p] : 0u;
        qs[k] = p < e ? s_qs[p] : 0u;
      }
#pragma unroll
      for (int k = 0; k < K; ++k) {
        const uint32_t p = p0 + (uint32_t)k * 64 + (uint32_t)lane;
        uint32_t inc = qe[k];  // inclusive running max over the row
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
          const uint32_t t = __shfl_up(inc, d, 64);
          if (lane >= d && t > inc) inc = t;
        }
        uint32_t before = __shfl_up(inc, 1, 64);  // max over the earlier lanes of the row
        if (lane == 0) before = 0;
        if (carry > before) before = carry;
        uint64_t lim = (uint64_t)before + max_gap;
        if (lim < max_gap) lim = ~0ull;  // saturate
        if (p < e) unit_flag[p] = (F)((p == b || (uint64_t)qs[k] > lim) ? 1 : 0);
        const uint32_t last = __shfl(inc, 63, 64);
        if (last > carry) carry = last;
      }
    }
  }
}
// The same two per-group reductions for inputs with few, very long groups (one wavefront per group would crawl):
// a running maximum over the composite (group index << 32 | value) is a segmented running maximum, because the
// group index never decreases along the survivor order.
__global__ __launch_bounds__(EW) void seg_compose_kernel(uint64_t m, const uint32_t* __restrict__ s_gidx,
                                                         const uint32_t* __restrict__ v, int complement,
                                                         uint64_t* __restrict__ out) {
  uint64_t p = (uint64_t)blockIdx.x * EW + threadIdx.x;
  if (p < m) out[p] = ((uint64_t)s_gidx[p] << 32) | (complement ? 0xffffffffu - v[p] : v[p]);
}
template <class F>
__global__ __launch_bounds__(EW) void cuts_from_scan_kernel(uint64_t m, const uint32_t* __restrict__ head_flag,
                                                            const uint64_t* __restrict__ run_max,
                                                            const uint32_t* __restrict__ s_qs, uint64_t max_gap,
                                                            F* __restrict__ unit_flag) {
  uint64_t p = (uint64_t)blockIdx.x * EW + threadIdx.x;
  if (p >= m) return;
  uint32_t f = 1;
  if (!head_flag[p]) {  // p > 0 and p - 1 is in the same group
    uint64_t lim = (run_max[p - 1] & 0xffffffffull) + max_gap;
    if (lim < max_gap) lim = ~0ull;
    f = (uint64_t)s_qs[p] > lim ? 1u : 0u;
  }
  unit_flag[p] = (F)f;
}
__global__ __launch_bounds__(EW) void unit_begin_kernel(uint64_t m, const uint32_t* __restrict__ unit_flag,
                                                        const uint32_t* __restrict__ unit_excl,
                                                        uint32_t* __restrict__ unit_begin) {
  uint64_t p = (uint64_t)blockIdx.x * EW + threadIdx.x;
  if (p < m && unit_flag[p]) unit_begin[unit_excl[p]] = (uint32_t)p;
}

}  // namespace

int chain_predecessors(swg_ctx* ctx, const swg_records* r, const uint8_t* alive, const uint8_t* member, uint64_t max_gap,
                       int pos_bits, ChainBuild* out, ChainWork* work, const uint32_t* q_order, uint64_t n_alive,
                       const swg_key_ends* slots) {
  const uint64_t n = r->n;
  hipStream_t st = ctx->stream;
  ChainBuild& B = *out;
  const int pair_bits = swg_bits_for((uint64_t)r->n_seq * r->n_seq * 2);
  if (pair_bits + pos_bits > 64)
    return swg_set_error(ctx, SWG_ERR_RANGE, "chain sort key (%d pair bits + %d coordinate bits) exceeds 64 bits",
                         pair_bits, pos_bits);
  // ---- compaction of alive, sort A
  uint64_t* d_tot = swg_alloc<uint64_t>(ctx, 4);
  SWG_CHECK_ARENA(ctx);
  swg_flag_scan alive_scan{};
  uint64_t M = n_alive;  // the caller's count (prepare has it), or ~0: counted here
  if (M == ~0ull || (M != n && !q_order)) SWG_TRY(swg_flags_count(ctx, alive, n, &alive_scan, d_tot));  // (M == n: nothing to compact)
  if (M == ~0ull) SWG_TRY(swg_read_scalars(ctx, d_tot, &M, 1));
  B.M = M;
  if (M == 0) return SWG_OK;
  const bool all_members = member == alive;  // the mapping-level sweep removed nothing (the caller passes the same array)
  B.a_qe = swg_alloc<uint32_t>(ctx, M);
  B.a_ts = swg_alloc<uint32_t>(ctx, M);
  B.a_te = swg_alloc<uint32_t>(ctx, M);
  B.a_dpair = swg_alloc<uint32_t>(ctx, M);
  uint8_t* a_keep = all_members ? nullptr : swg_alloc<uint8_t>(ctx, M);
  uint32_t* pair_flag = all_members ? nullptr : swg_alloc<uint32_t>(ctx, M);
  uint32_t* pair_excl = all_members ? nullptr : swg_alloc<uint32_t>(ctx, M);
  B.keyA = swg_alloc<uint64_t>(ctx, M);
  B.idxA = swg_alloc<uint32_t>(ctx, M);
  // the sort's scratch pair comes last: when the sorted data ends up in the primary buffers (an even number of passes) the
  // scratch goes back to the arena right after the sort (12 of ~300 bytes of scratch per record; every GB of scratch is
  // 30-45 ms of hipMalloc on a cold context)
  const swg_arena_mark sort_mark = swg_arena_save(ctx);
  uint64_t* const keyA0 = B.keyA;
  uint64_t* key_tmp = swg_alloc<uint64_t>(ctx, M);
  uint32_t* idx_tmp = swg_alloc<uint32_t>(ctx, M);
  SWG_CHECK_ARENA(ctx);
  uint64_t* packedA = nullptr;  // sort A's result as packed words (then B.keyA / B.idxA are written by the gather)
  uint64_t* wordsA = nullptr;   // ... as (group, index) words behind the mapping sweep (gatherA_slots_words writes B.keyA / B.idxA)
  int wordsA_idx_bits = 0;
  int packed_idx_bits = 0;
  int dropA = 0;                // low key bits left out of sort A (swg_radix_sort_words; the gather orders the runs)
  // keys + histograms + the packed / word sort of the all-members case; drop = 0: the packed sort over the whole key
  uint32_t* prehistA = nullptr;
  auto sortA_packed = [&](int drop, int key_bits, int idx_bits) -> int {
    const bool identity = M == n;
    const unsigned full = nblk(M), cap = (unsigned)ctx->num_cu * 16;
    SWG_HIP(ctx, hipMemsetAsync(prehistA, 0, sizeof(uint32_t) * SWG_RADIX_MAX_PASSES * SWG_RADIX_BINS, st));
    SWG_LAUNCH(ctx, "sortA_keys_hist", sortA_keys_hist_kernel<<<full > cap ? cap : full, EW, 0, st>>>(
                                      M, identity ? nullptr : B.idxA, nullptr, r->q_id, r->t_id, r->strand, r->q_start, r->n_seq, pos_bits,
                                      keyA0, drop ? swg_radix_plan_words(key_bits - drop) : swg_radix_plan_packed(key_bits), prehistA, drop,
                                      idx_bits));
    SWG_KERNEL_CHECK(ctx);
    int prc;
    if (drop)
      prc = swg_radix_sort_words(ctx, keyA0, key_tmp, M, key_bits - drop, idx_bits, prehistA, &packedA);
    else
      prc = swg_radix_sort_packed(ctx, keyA0, identity ? nullptr : B.idxA, key_tmp, M, key_bits, idx_bits, prehistA, &packedA);
    if (prc == SWG_ERR_UNSUPPORTED) return swg_set_error(ctx, SWG_ERR_HIP, "packed sort declined a shape it accepted");
    if (prc != SWG_OK) return prc;
    packed_idx_bits = idx_bits;
    dropA = drop;
    B.keyA = packedA == keyA0 ? key_tmp : keyA0;  // the buffer the words are not in takes the unpacked keys
    return SWG_OK;
  };
  int sortA_key_bits = 0;
  if (q_order) {
    // The mapping sweep sorted the same alive records by (query sequence, target genome, q_start, index): dead records
    // first, so the last M entries are the alive ones, and inside every (query, target, strand) group they already stand
    // in q_start order with ties in index order -- the order sort A must end in (paf_filter.rs:777).  Stable passes over
    // the group bits alone finish it: 2 radix passes instead of 6 for a 100-genome pangenome.
    static const bool sortA_pairs = getenv("SWG_SORTA_PAIRS") != nullptr;  // test / A-B knob: the (key, index) pairs as before
    const int idx_bits = swg_bits_for(n - 1) ? swg_bits_for(n - 1) : 1;
    if (slots && !all_members && !sortA_pairs && pair_bits + idx_bits <= 64 && M > 1 && M < (uint64_t(1) << 30) &&
        swg_radix_plan_words(pair_bits).npasses > 0) {
      // as words over the group bits alone; the keys come back in gatherA_slots_words (see sortA_words_kernel)
      SWG_LAUNCH(ctx, "sortA_words", sortA_words_kernel<<<nblk(M), EW, 0, st>>>(M, q_order + (n - M), r->q_id, r->t_id, r->strand, r->n_seq,
                                                                    idx_bits, keyA0, ctx->call_group32));
      SWG_KERNEL_CHECK(ctx);
      const int prc = swg_radix_sort_words(ctx, keyA0, key_tmp, M, pair_bits, idx_bits, nullptr, &wordsA);
      if (prc == SWG_ERR_UNSUPPORTED) return swg_set_error(ctx, SWG_ERR_HIP, "word sort declined a shape it accepted");
      if (prc != SWG_OK) return prc;
      wordsA_idx_bits = idx_bits;
      B.keyA = wordsA == keyA0 ? key_tmp : keyA0;  // the buffer the words are not in takes the keys
    } else {
      SWG_HIP(ctx, hipMemcpyAsync(B.idxA, q_order + (n - M), M * sizeof(uint32_t), hipMemcpyDeviceToDevice, st));
      SWG_LAUNCH(ctx, "sortA_keys", sortA_keys_kernel<<<nblk(M), EW, 0, st>>>(M, B.idxA, r->q_id, r->t_id, r->strand, r->q_start,
                                                                  r->n_seq, pos_bits, B.keyA));
      SWG_KERNEL_CHECK(ctx);
      SWG_TRY(swg_radix_sort_pairs(ctx, &B.keyA, &B.idxA, &key_tmp, &idx_tmp, M, pos_bits, pair_bits + pos_bits));
    }
  } else {
    if (M != n) SWG_TRY(swg_flags_compact(ctx, alive_scan, B.idxA));  // M == n: the list is the identity, written below
    const int key_bits = pair_bits + pos_bits;
    static const bool sort_fallback = getenv("SWG_SORT_FALLBACK") != nullptr;
    if (key_bits <= 8 * SWG_RADIX_MAX_PASSES && !sort_fallback && M > 1 && M < (uint64_t(1) << 32)) {
      // keys, (identity) indices and the sort's digit histograms in one pass
      uint32_t* prehist = swg_alloc<uint32_t>(ctx, (size_t)SWG_RADIX_MAX_PASSES * SWG_RADIX_BINS);
      SWG_CHECK_ARENA(ctx);
      SWG_HIP(ctx, hipMemsetAsync(prehist, 0, sizeof(uint32_t) * SWG_RADIX_MAX_PASSES * SWG_RADIX_BINS, st));
      const unsigned full = nblk(M), cap = (unsigned)ctx->num_cu * 16;
      // 8-byte passes when every record is a member (the fused gather below unpacks the words) and the word has the room;
      // with M == n the index list is the identity and is not even written (the packed sort takes it as read)
      const int idx_bits = swg_bits_for(n - 1) ? swg_bits_for(n - 1) : 1;
      const bool packed_sort = all_members && pos_bits >= 8 && swg_radix_sort_packed_applies(M, key_bits, idx_bits);
      const bool identity = M == n;
      if (packed_sort) {
        // a pass fewer when the low bits of q_start can be left to the gather (as in the sweep, swg_sweep.hip)
        prehistA = prehist;
        sortA_key_bits = key_bits;
        // How far the truncation may go follows the density of the keys: with G (query, target, strand) groups at most,
        // 2^d-wide buckets of q_start hold M * 2^d / (G * 2^pos_bits) records each on average, and the gather orders runs of
        // up to SWG_RUN_HALO -- buckets of four records on average leave that far away (S-pan: 16 bits, three passes instead of
        // four; a pair of 10^7 records: 9 bits).  G is only known as an upper bound here (every pair of sequences, both
        // strands), so a real run that is too long still takes the fall-back below, and the context remembers.
        static const bool no_dense_cap = getenv("SWG_SORT_DROP10") != nullptr;  // A/B knob: at most 10 bits as before
        int dcap = 10;
        if (!no_dense_cap) {
          const double buckets4 = 4.0 * 2.0 * (double)r->n_seq * (double)r->n_seq * std::ldexp(1.0, pos_bits) / (double)M;
          dcap = buckets4 >= 65536.0 ? 16 : (buckets4 >= 1024.0 ? (int)std::floor(std::log2(buckets4)) : 10);
        }
        SWG_TRY(sortA_packed(swg_radix_drop_bits(M, key_bits, pos_bits, idx_bits, ctx->sort_drop_level, dcap), key_bits, idx_bits));
      } else {
        SWG_LAUNCH(ctx, "sortA_keys_hist", sortA_keys_hist_kernel<<<full > cap ? cap : full, EW, 0, st>>>(
                                          M, identity ? nullptr : B.idxA, identity ? B.idxA : nullptr, r->q_id, r->t_id, r->strand,
                                          r->q_start, r->n_seq, pos_bits, B.keyA, swg_radix_plan_pairs(0, key_bits), prehist));
        SWG_KERNEL_CHECK(ctx);
        SWG_TRY(swg_radix_sort_pairs(ctx, &B.keyA, &B.idxA, &key_tmp, &idx_tmp, M, 0, key_bits, prehist));
      }
    } else {
      if (M == n) {
        SWG_LAUNCH(ctx, "iota", iota_u32_kernel<<<nblk(M), EW, 0, st>>>(M, B.idxA));
        SWG_KERNEL_CHECK(ctx);
      }
      SWG_LAUNCH(ctx, "sortA_keys", sortA_keys_kernel<<<nblk(M), EW, 0, st>>>(M, B.idxA, r->q_id, r->t_id, r->strand, r->q_start,
                                                                  r->n_seq, pos_bits, B.keyA));
      SWG_KERNEL_CHECK(ctx);
      SWG_TRY(swg_radix_sort_pairs(ctx, &B.keyA, &B.idxA, &key_tmp, &idx_tmp, M, 0, key_bits));
    }
  }
  if (!packedA && !wordsA && B.keyA == keyA0) swg_arena_restore(ctx, sort_mark);  // (idxA swaps together with keyA; with packed words: no release)
  uint64_t m = 0, n_groups = 0;
  uint32_t *s_qs = nullptr, *s_qe = nullptr, *s_ts = nullptr, *s_te = nullptr, *s_m = nullptr, *s_b = nullptr;
  uint64_t* s_grp = nullptr;
  uint32_t *head_flag = nullptr, *s_gidx = nullptr, *group_begin = nullptr;
  unsigned long long* bps = nullptr;
  uint32_t* pred = nullptr;
  B.T.nc = 0;
  if (all_members) {
    m = M;
    B.m = m;
    if (m >= 0xffffffffull) return swg_set_error(ctx, SWG_ERR_RANGE, "too many chain members");
    B.s_a = nullptr;  // positions coincide (nullptr = identity)
    B.s_idx = B.idxA;
    B.s_chain = B.want_s_chain ? swg_alloc<uint32_t>(ctx, m) : nullptr;
    s_qs = swg_alloc<uint32_t>(ctx, m);
    s_qe = B.a_qe;
    s_ts = B.a_ts;
    s_te = B.a_te;
    s_m = swg_alloc<uint32_t>(ctx, m);
    s_b = swg_alloc<uint32_t>(ctx, m);
    s_grp = swg_alloc<uint64_t>(ctx, m);
    s_gidx = swg_alloc<uint32_t>(ctx, m);
    group_begin = swg_alloc<uint32_t>(ctx, m);
    bps = swg_alloc<unsigned long long>(ctx, m);
    pred = swg_alloc<uint32_t>(ctx, m);
    const uint64_t n_blk = nblk(M);
    // per 256-element block: (pair boundaries << 32) | group boundaries; one more word behind them: the words gather's
    // "a run was too long" flag, read back together with the scan's total
    uint64_t* blk_cnt = swg_alloc<uint64_t>(ctx, n_blk + 1);
    SWG_CHECK_ARENA(ctx);
    uint64_t tot = 0;
    for (;;) {
      if (packedA && dropA) {
        SWG_HIP(ctx, hipMemsetAsync(blk_cnt + n_blk, 0, 8, st));
        SWG_LAUNCH(ctx, "gather_all_words", gather_all_words_kernel<<<nblk(M), EW, 0, st>>>(
                                          M, packedA, packed_idx_bits, dropA, r->q_start, r->q_end, r->t_start, r->t_end, r->matches,
                                          r->block_len, pos_bits, B.keyA, B.idxA, s_qs, s_qe, s_ts, s_te, s_m, s_b, s_grp, blk_cnt,
                                          reinterpret_cast<unsigned long long*>(blk_cnt + n_blk), slots ? slots : ctx->call_probe_slots,
                                          slots ? nullptr : ctx->call_probe_flag));
      } else if (packedA) {
        SWG_LAUNCH(ctx, "gather_all_packed", gather_all_packed_kernel<<<nblk(M), EW, 0, st>>>(
                                          M, packedA, packed_idx_bits, r->q_start, r->q_end, r->t_start, r->t_end, r->matches, r->block_len,
                                          pos_bits, B.keyA, B.idxA, s_qs, s_qe, s_ts, s_te, s_m, s_b, s_grp, blk_cnt,
                                          slots ? slots : ctx->call_probe_slots, slots ? nullptr : ctx->call_probe_flag));
      } else {
        SWG_LAUNCH(ctx, "gather_all", gather_all_kernel<<<nblk(M), EW, 0, st>>>(M, B.keyA, B.idxA, r->q_end, r->t_start, r->t_end, r->matches,
                                                                   r->block_len, pos_bits, s_qs, s_qe, s_ts, s_te, s_m, s_b, s_grp,
                                                                   blk_cnt));
      }
      SWG_KERNEL_CHECK(ctx);
      SWG_TRY(swg_inclusive_sum_scan_u64(ctx, blk_cnt, blk_cnt, n_blk));
      uint64_t h2[2] = {0, 0};
      SWG_TRY(swg_read_scalars(ctx, blk_cnt + (n_blk - 1), h2, (packedA && dropA) ? 2 : 1));
      tot = h2[0];
      if (!(packedA && dropA) || h2[1] == 0) break;
      // runs of equal truncated keys longer than the gather can order: once more, sorted on the whole key
      ++ctx->sort_drop_level;
      static const bool dbg = getenv("SWG_DEBUG") != nullptr;
      if (dbg) fprintf(stderr, "[swg] sort A: runs of equal q_start >> %d longer than %d: sorting again on the whole key\n", dropA, SWG_RUN_HALO);
      if (M != n) SWG_TRY(swg_flags_compact(ctx, alive_scan, B.idxA));  // (the failed gather wrote over the list of alive records)
      SWG_TRY(sortA_packed(0, sortA_key_bits, packed_idx_bits));
    }
    B.n_pairs = tot >> 32;
    n_groups = tot & 0xffffffffull;
    // the group-head flags as a column are only read by the scan-based reductions of few, long groups (below and in the chain
    // table): 4 bytes per member not written otherwise
    if (n_groups && m / n_groups > 8192) {
      head_flag = swg_alloc<uint32_t>(ctx, m);
      SWG_CHECK_ARENA(ctx);
    }
    SWG_LAUNCH(ctx, "group_pair", group_pair_kernel<<<nblk(M), EW, 0, st>>>(M, B.keyA, pos_bits, blk_cnt, B.a_dpair, s_gidx, head_flag,
                                                               group_begin));
    SWG_KERNEL_CHECK(ctx);
  } else {
    uint32_t *a_m = nullptr, *a_b = nullptr;  // matches / block length by A position (only with the record slots)
    if (slots) {
      a_m = swg_alloc<uint32_t>(ctx, M);
      a_b = swg_alloc<uint32_t>(ctx, M);
      SWG_CHECK_ARENA(ctx);
      if (wordsA)
        SWG_LAUNCH(ctx, "gatherA_slots", gatherA_slots_words_kernel<<<nblk(M), EW, 0, st>>>(M, wordsA, wordsA_idx_bits, slots, member, pos_bits, B.keyA,
                                                                                B.idxA, B.a_qe, B.a_ts, B.a_te, a_m, a_b, a_keep, pair_flag));
      else
        SWG_LAUNCH(ctx, "gatherA_slots", gatherA_slots_kernel<<<nblk(M), EW, 0, st>>>(M, B.keyA, B.idxA, slots, member, pos_bits, B.a_qe, B.a_ts, B.a_te,
                                                                          a_m, a_b, a_keep, pair_flag));
    } else {
      SWG_LAUNCH(ctx, "gatherA", gatherA_kernel<<<nblk(M), EW, 0, st>>>(M, B.keyA, B.idxA, r->q_end, r->t_start, r->t_end, member,
                                                            pos_bits, B.a_qe, B.a_ts, B.a_te, a_keep, pair_flag));
    }
    SWG_KERNEL_CHECK(ctx);
    SWG_TRY(swg_exclusive_scan_u32(ctx, pair_flag, pair_excl, M, d_tot + 1));
    SWG_LAUNCH(ctx, "dense_from_scan", dense_from_scan_kernel<<<nblk(M), EW, 0, st>>>(M, pair_excl, pair_flag, B.a_dpair));
    SWG_KERNEL_CHECK(ctx);
    // ---- survivors in A order
    swg_flag_scan keep_scan;
    SWG_TRY(swg_flags_count(ctx, a_keep, M, &keep_scan, d_tot + 2));
    uint64_t h3[3];
    SWG_TRY(swg_read_scalars(ctx, d_tot, h3, 3));
    B.n_pairs = h3[1];
    m = h3[2];
    B.m = m;
    if (m == 0) return SWG_OK;
    if (m >= 0xffffffffull) return swg_set_error(ctx, SWG_ERR_RANGE, "too many chain members");
    B.s_a = swg_alloc<uint32_t>(ctx, m);
    B.s_idx = swg_alloc<uint32_t>(ctx, m);
    B.s_chain = swg_alloc<uint32_t>(ctx, m);
    s_qs = swg_alloc<uint32_t>(ctx, m);
    s_qe = swg_alloc<uint32_t>(ctx, m);
    s_ts = swg_alloc<uint32_t>(ctx, m);
    s_te = swg_alloc<uint32_t>(ctx, m);
    s_m = swg_alloc<uint32_t>(ctx, m);
    s_b = swg_alloc<uint32_t>(ctx, m);
    s_grp = swg_alloc<uint64_t>(ctx, m);
    head_flag = swg_alloc<uint32_t>(ctx, m);
    uint32_t* gidx_excl = swg_alloc<uint32_t>(ctx, m);
    s_gidx = swg_alloc<uint32_t>(ctx, m);
    group_begin = swg_alloc<uint32_t>(ctx, m);
    bps = swg_alloc<unsigned long long>(ctx, m);
    pred = swg_alloc<uint32_t>(ctx, m);
    SWG_CHECK_ARENA(ctx);
    if (m == M)
      B.s_a = nullptr;  // every record of sort A is a member: positions coincide (nullptr = identity)
    else
      SWG_TRY(swg_flags_compact(ctx, keep_scan, B.s_a));
    if (m == M) {
      s_qe = B.a_qe;
      s_ts = B.a_ts;
      s_te = B.a_te;
      B.s_idx = B.idxA;
      SWG_LAUNCH(ctx, "gatherS_all", gatherS_all_kernel<<<nblk(m), EW, 0, st>>>(m, B.keyA, B.idxA, r->matches, r->block_len, pos_bits, s_qs,
                                                                s_m, s_b, s_grp, head_flag));
    } else {
      SWG_LAUNCH(ctx, "gatherS", gatherS_kernel<<<nblk(m), EW, 0, st>>>(m, B.s_a, B.keyA, B.idxA, B.a_qe, B.a_ts, B.a_te, r->matches,
                                                            r->block_len, a_m, a_b, pos_bits, s_qs, s_qe, s_ts, s_te, s_m, s_b,
                                                            B.s_idx, s_grp, head_flag));
    }
    SWG_KERNEL_CHECK(ctx);
    SWG_TRY(swg_exclusive_scan_u32(ctx, head_flag, gidx_excl, m, d_tot + 3));
    SWG_LAUNCH(ctx, "group_bounds", group_bounds_kernel<<<nblk(m), EW, 0, st>>>(m, head_flag, gidx_excl, s_gidx, group_begin));
    SWG_KERNEL_CHECK(ctx);
    SWG_TRY(swg_read_scalars(ctx, d_tot + 3, &n_groups, 1));
  }
  // ---- best-buddy chaining
  static const bool old_walk_fill = getenv("SWG_CHAIN_OLD") != nullptr;
  if (old_walk_fill) {  // the round-2 selection kernels read the score array from the start; the walk initialises what it uses
    SWG_LAUNCH(ctx, "fill", fill_u64_kernel<<<nblk(m), EW, 0, st>>>(m, reinterpret_cast<uint64_t*>(bps), ~0ull));
    SWG_KERNEL_CHECK(ctx);
  }
  SWG_LAUNCH(ctx, "fill", fill_u32_kernel<<<nblk(m), EW, 0, st>>>(m, pred, NONE));
  SWG_KERNEL_CHECK(ctx);
  {
    // units = groups cut where no window can straddle.  The walk's path keeps the cut flags as BYTES and gets the list of unit
    // begins from their compaction (per-tile counts, no element-wise scan, no per-element unit index: the few kernels that
    // need the unit of an element -- only members of long units ask -- search the long units' ranges); the round-2 kernels
    // (SWG_CHAIN_OLD) keep their u32 flags and the exclusive scan.
    static const bool old_walk = getenv("SWG_CHAIN_OLD") != nullptr;  // A/B knob: the per-step kernels of round 2
    uint32_t* unit_flag = old_walk ? swg_alloc<uint32_t>(ctx, m) : nullptr;
    uint32_t* unit_excl = old_walk ? swg_alloc<uint32_t>(ctx, m) : nullptr;
    uint8_t* unit_flag8 = old_walk ? nullptr : swg_alloc<uint8_t>(ctx, m);
    uint64_t* d_nu = swg_alloc<uint64_t>(ctx, 1);
    SWG_CHECK_ARENA(ctx);
    const bool long_groups = m / n_groups > 8192;  // few, long groups: scan-based reductions
    if (long_groups) {
      swg_arena_mark mk = swg_arena_save(ctx);
      uint64_t* comp = swg_alloc<uint64_t>(ctx, m);
      SWG_CHECK_ARENA(ctx);
      SWG_LAUNCH(ctx, "seg_compose", seg_compose_kernel<<<nblk(m), EW, 0, st>>>(m, s_gidx, s_qe, 0, comp));
      SWG_KERNEL_CHECK(ctx);
      SWG_TRY(swg_inclusive_max_scan_u64(ctx, comp, comp, m));
      if (old_walk)
        SWG_LAUNCH(ctx, "cuts_from_scan", cuts_from_scan_kernel<uint32_t><<<nblk(m), EW, 0, st>>>(m, head_flag, comp, s_qs, max_gap, unit_flag));
      else
        SWG_LAUNCH(ctx, "cuts_from_scan", cuts_from_scan_kernel<uint8_t><<<nblk(m), EW, 0, st>>>(m, head_flag, comp, s_qs, max_gap, unit_flag8));
      SWG_KERNEL_CHECK(ctx);
      swg_arena_restore(ctx, mk);
    } else {
      uint64_t blocks = (n_groups + 3) / 4;
      const uint64_t mb = (uint64_t)ctx->num_cu * 16;
      if (blocks > mb) blocks = mb;
      if (old_walk)
        SWG_LAUNCH(ctx, "chain_cuts", chain_cuts_kernel<uint32_t><<<(unsigned)blocks, EW, 0, st>>>((uint32_t)n_groups, group_begin, (uint32_t)m, s_qs,
                                                                                       s_qe, max_gap, unit_flag));
      else
        SWG_LAUNCH(ctx, "chain_cuts", chain_cuts_kernel<uint8_t><<<(unsigned)blocks, EW, 0, st>>>((uint32_t)n_groups, group_begin, (uint32_t)m, s_qs,
                                                                                      s_qe, max_gap, unit_flag8));
      SWG_KERNEL_CHECK(ctx);
    }
    uint64_t n_units = 0;
    uint32_t* unit_begin = nullptr;
    if (old_walk) {
      SWG_TRY(swg_exclusive_scan_u32(ctx, unit_flag, unit_excl, m, d_nu));
      SWG_TRY(swg_read_scalars(ctx, d_nu, &n_units, 1));
      unit_begin = swg_alloc<uint32_t>(ctx, n_units);
      SWG_CHECK_ARENA(ctx);
      SWG_LAUNCH(ctx, "unit_begin", unit_begin_kernel<<<nblk(m), EW, 0, st>>>(m, unit_flag, unit_excl, unit_begin));
      SWG_KERNEL_CHECK(ctx);
    } else {
      swg_flag_scan unit_scan;
      SWG_TRY(swg_flags_count(ctx, unit_flag8, m, &unit_scan, d_nu));
      SWG_TRY(swg_read_scalars(ctx, d_nu, &n_units, 1));
      unit_begin = swg_alloc<uint32_t>(ctx, n_units);
      SWG_CHECK_ARENA(ctx);
      SWG_TRY(swg_flags_compact(ctx, unit_scan, unit_begin));
    }
    static const bool force_deep = getenv("SWG_CHAIN_DEEP") != nullptr;  // test knob: the deep-group (wavefront per i) kernel at any size
    if (getenv("SWG_DEBUG"))
      fprintf(stderr, "[swg] chaining: m=%llu groups=%llu units=%llu\n", (unsigned long long)m,
              (unsigned long long)n_groups, (unsigned long long)n_units);
    if (!old_walk) {
      const bool lists_all = long_groups || force_deep;  // candidate lists for every element (wavefront per element)
      static const bool want_wstats = getenv("SWG_WALK_STATS") != nullptr;  // diagnostic counters of the walk kernels
      unsigned long long* wstats = nullptr;
      if (want_wstats) {
        wstats = swg_alloc<unsigned long long>(ctx, 16);
        SWG_CHECK_ARENA(ctx);
        SWG_HIP(ctx, hipMemsetAsync(wstats, 0, 16 * sizeof(unsigned long long), st));
      }
      uint8_t* is_big = swg_alloc<uint8_t>(ctx, n_units);
      uint8_t* chunk_flag = swg_alloc<uint8_t>(ctx, n_units);
      uint64_t* d_nb = swg_alloc<uint64_t>(ctx, 2);
      SWG_CHECK_ARENA(ctx);
      SWG_LAUNCH(ctx, "unit_big_flag", unit_big_flag_kernel<<<nblk(n_units), EW, 0, st>>>((uint32_t)n_units, unit_begin, (uint32_t)m, is_big));
      SWG_KERNEL_CHECK(ctx);
      SWG_LAUNCH(ctx, "chunk_flag", chunk_flag_kernel<<<nblk(n_units), EW, 0, st>>>((uint32_t)n_units, unit_begin, is_big, chunk_flag));
      SWG_KERNEL_CHECK(ctx);
      swg_flag_scan big_scan, chunk_scan;
      SWG_TRY(swg_flags_count(ctx, is_big, n_units, &big_scan, d_nb));
      SWG_TRY(swg_flags_count(ctx, chunk_flag, n_units, &chunk_scan, d_nb + 1));
      uint64_t h2[2];
      SWG_TRY(swg_read_scalars(ctx, d_nb, h2, 2));
      const uint64_t n_big = h2[0], n_chunks = h2[1];
      unsigned long long* c_d = nullptr;
      uint32_t *c_j = nullptr, *c_n = nullptr, *c_ext = nullptr;
      uint8_t* big_member = nullptr;  // members of long units (also what the chain table's generic path is restricted to)
      uint8_t* span_big = nullptr;    // the same per 1024-element span: lets that path skip whole work-groups
      uint32_t* big_list = nullptr;  // the long units (indices into unit_begin) ...
      uint32_t* big_rng = nullptr;   // ... and their [begin, end) ranges
      if (n_big) {
        big_member = swg_alloc<uint8_t>(ctx, m);
        const uint64_t n_span = (m + (uint64_t(1) << BIG_SPAN_SHIFT) - 1) >> BIG_SPAN_SHIFT;
        span_big = swg_alloc<uint8_t>(ctx, n_span + 1);
        big_list = swg_alloc<uint32_t>(ctx, n_big);
        big_rng = swg_alloc<uint32_t>(ctx, 2 * n_big);
        SWG_CHECK_ARENA(ctx);
        SWG_TRY(swg_flags_compact(ctx, big_scan, big_list));
        SWG_LAUNCH(ctx, "big_ranges", big_ranges_kernel<<<nblk(n_big), EW, 0, st>>>((uint32_t)n_big, big_list, (uint32_t)n_units, unit_begin, (uint32_t)m,
                                                                       big_rng));
        SWG_KERNEL_CHECK(ctx);
        SWG_HIP(ctx, hipMemsetAsync(span_big + n_span, 0, 1, st));  // (readers index spans up to m >> shift inclusive)
        SWG_LAUNCH(ctx, "big_member_fill", big_member_fill_kernel<<<(unsigned)n_span, EW, 0, st>>>(m, (uint32_t)n_big, big_rng, big_member, span_big));
        SWG_KERNEL_CHECK(ctx);
      }
      if (lists_all) {
        c_d = swg_alloc<unsigned long long>(ctx, (size_t)KC * m);
        c_j = swg_alloc<uint32_t>(ctx, (size_t)KC * m);
        c_n = swg_alloc<uint32_t>(ctx, m);
        c_ext = swg_alloc<uint32_t>(ctx, m);
        (void)swg_alloc<uint8_t>(ctx, CAND_PAD);  // readable bytes behind the coordinate arrays (cand_scan_fast)
        SWG_CHECK_ARENA(ctx);
        launch_candidates_wave(ctx, st, nblk((m + CW_PER_WAVE - 1) / CW_PER_WAVE * 64), m, s_gidx, group_begin, (uint32_t)n_groups, s_grp, s_qs, s_qe,
                               s_ts, s_te, max_gap, c_d, c_j, c_n, c_ext);
        SWG_KERNEL_CHECK(ctx);
      } else if (n_big) {
        // sparse data with a few long units: their blocks build their candidate lists while they walk, like the chunks (no
        // 56 bytes of candidate arrays per record for the sake of a tenth of the records); the block plan only needs the
        // window extents
        c_ext = swg_alloc<uint32_t>(ctx, m);
        SWG_CHECK_ARENA(ctx);
        SWG_LAUNCH(ctx, "window_extent", window_extent_kernel<<<nblk(m), EW, 0, st>>>(m, big_member, big_rng, (uint32_t)n_big, s_qs, s_qe, max_gap,
                                                                         c_ext, span_big));
        SWG_KERNEL_CHECK(ctx);
      }
      {
        // ---- units shorter than BIG_UNIT, glued into chunks: one wavefront per chunk, 64 elements per step
        uint32_t* chunk_unit = swg_alloc<uint32_t>(ctx, n_chunks);
        SpecBlock* cdesc = swg_alloc<SpecBlock>(ctx, n_chunks);
        SWG_CHECK_ARENA(ctx);
        SWG_TRY(swg_flags_compact(ctx, chunk_scan, chunk_unit));
        SWG_LAUNCH(ctx, "chunk_desc", chunk_desc_kernel<<<nblk(n_chunks), EW, 0, st>>>((uint32_t)n_chunks, chunk_unit, (uint32_t)n_units, unit_begin,
                                                                          is_big, (uint32_t)m, cdesc));
        SWG_KERNEL_CHECK(ctx);
        work->chunks = cdesc;
        work->n_chunks = n_chunks;
        work->big_member = big_member;
        work->span_big = span_big;
        static const uint64_t per_cu = getenv("SWG_WALK_BLOCKS") ? (uint64_t)atoi(getenv("SWG_WALK_BLOCKS")) : 128;  // (blocks per CU: 64 -> 128 took 10 % off the walk -- chunks are of uneven length, a finer grid balances them)
        const uint64_t wb = n_chunks < (uint64_t)ctx->num_cu * per_cu ? n_chunks : (uint64_t)ctx->num_cu * per_cu;
        if (lists_all)
          SWG_LAUNCH(ctx, "chain_walk", chain_walk_kernel<1024, false, false><<<(unsigned)wb, 64, 0, st>>>(
                                            (uint32_t)n_chunks, cdesc, (uint32_t)m, s_grp, s_qs, s_qe, s_ts, s_te, s_gidx, group_begin,
                                            (uint32_t)n_groups, max_gap, c_d, c_j, c_n, bps, bps, pred, pred, wstats, nullptr, nullptr, walk_plain_knob()));
        else
          SWG_LAUNCH(ctx, "chain_walk", chain_walk_kernel<256, true, false><<<(unsigned)wb, 64, 0, st>>>(
                                            (uint32_t)n_chunks, cdesc, (uint32_t)m, s_grp, s_qs, s_qe, s_ts, s_te, s_gidx, group_begin,
                                            (uint32_t)n_groups, max_gap, nullptr, nullptr, nullptr, bps, bps, pred, pred, wstats, nullptr, nullptr,
                                            walk_plain_knob()));
        SWG_KERNEL_CHECK(ctx);
      }
      if (n_big) {
        // ---- block-speculative selection of the long units
        uint32_t* S_u = swg_alloc<uint32_t>(ctx, n_big);
        uint32_t* nblk_u = swg_alloc<uint32_t>(ctx, n_big);
        uint32_t* blk_off = swg_alloc<uint32_t>(ctx, n_big);
        uint64_t* d_nblk = swg_alloc<uint64_t>(ctx, 1);
        unsigned long long* ext = swg_alloc<unsigned long long>(ctx, m);
        unsigned long long* v_own = swg_alloc<unsigned long long>(ctx, m);
        unsigned long long* v_prev = swg_alloc<unsigned long long>(ctx, m);
        uint32_t* p_own = swg_alloc<uint32_t>(ctx, m);
        uint32_t* p_prev = swg_alloc<uint32_t>(ctx, m);
        uint32_t* spec_changed = swg_alloc<uint32_t>(ctx, 2);
        uint64_t* d_smax = swg_alloc<uint64_t>(ctx, 1);  // adjacent to d_nblk: read back together
        SWG_CHECK_ARENA(ctx);
        SWG_HIP(ctx, hipMemsetAsync(d_smax, 0, 8, st));
        if (m / n_big > 65536) {  // few, very long units
          uint32_t* wmax_k = swg_alloc<uint32_t>(ctx, n_big);
          SWG_CHECK_ARENA(ctx);
          SWG_HIP(ctx, hipMemsetAsync(wmax_k, 0, n_big * sizeof(uint32_t), st));
          const uint64_t wb = nblk(m) < (uint64_t)ctx->num_cu * 16 ? nblk(m) : (uint64_t)ctx->num_cu * 16;
          SWG_LAUNCH(ctx, "big_wmax", big_wmax_kernel<<<(unsigned)wb, EW, 0, st>>>(m, big_member, big_rng, (uint32_t)n_big, c_ext, wmax_k, span_big));
          SWG_KERNEL_CHECK(ctx);
          SWG_LAUNCH(ctx, "spec_plan_from_wmax", spec_plan_from_wmax_kernel<<<nblk(n_big), EW, 0, st>>>((uint32_t)n_big, big_list, (uint32_t)n_units, unit_begin,
                                                                                (uint32_t)m, wmax_k, S_u, nblk_u,
                                                                                reinterpret_cast<uint32_t*>(d_smax), 1));
          SWG_KERNEL_CHECK(ctx);
        } else {
          SWG_LAUNCH(ctx, "spec_plan", spec_plan_kernel<<<(unsigned)n_big, EW, 0, st>>>((uint32_t)n_big, big_list, (uint32_t)n_units, unit_begin,
                                                                          (uint32_t)m, c_ext, S_u, nblk_u,
                                                                          reinterpret_cast<uint32_t*>(d_smax)));
          SWG_KERNEL_CHECK(ctx);
        }
        SWG_TRY(swg_exclusive_scan_u32(ctx, nblk_u, blk_off, n_big, d_nblk));
        uint64_t n_spec = 0, s_max = 0;
        SWG_TRY(swg_read_scalars(ctx, d_nblk, &n_spec, 1));
        SWG_TRY(swg_read_scalars(ctx, d_smax, &s_max, 1));
        s_max &= 0xffffffffull;
        SpecBlock* desc = swg_alloc<SpecBlock>(ctx, n_spec);
        SWG_CHECK_ARENA(ctx);
        SWG_LAUNCH(ctx, "spec_desc", spec_desc_kernel<<<(unsigned)n_big, EW, 0, st>>>((uint32_t)n_big, big_list, (uint32_t)n_units, unit_begin,
                                                                        (uint32_t)m, S_u, nblk_u, blk_off, desc));
        SWG_KERNEL_CHECK(ctx);
        const uint64_t rblocks = n_spec < (uint64_t)ctx->num_cu * 8 ? n_spec : (uint64_t)ctx->num_cu * 8;
        // `ext` (what the previous block offers each element) is only ever read and written inside the blocks: it starts as
        // "no offer" there and nowhere else (round 3 filled all m entries)
        SWG_LAUNCH(ctx, "spec_ext_init", spec_ext_init_kernel<<<(unsigned)rblocks, EW, 0, st>>>((uint32_t)n_spec, desc, ext));
        SWG_KERNEL_CHECK(ctx);
        // The ring is a cache (positions outside it are read from global memory), so its size only trades LDS hits for
        // resident wavefronts -- and a block is one dependent chain, bound by latency, so residency wins: 256 slots when no
        // window exceeds 511 elements, else 1024 (S-big1, windows up to 19,968: 12.4 ms with 4096 slots = one block per CU,
        // 7.1 ms with 1024, 9.8 ms with 256).
        static const char* ring_knob = getenv("SWG_SPEC_RING");
        const int ring = ring_knob ? atoi(ring_knob) : (s_max <= 512 ? 256 : 1024);
        int rounds = 0;
        for (uint64_t round = 0; round <= n_spec + 1; ++round) {
          SWG_LAUNCH(ctx, "spec_init", spec_init_kernel<<<(unsigned)rblocks, EW, 0, st>>>((uint32_t)n_spec, desc, ext, v_own, v_prev, p_own, p_prev));
          SWG_KERNEL_CHECK(ctx);
          const unsigned wblocks = (unsigned)(n_spec < (uint64_t)ctx->num_cu * 32 ? n_spec : (uint64_t)ctx->num_cu * 32);
#define SWG_WALK_SPEC(W, F)                                                                                                       \
  SWG_LAUNCH(ctx, "chain_walk_spec", chain_walk_kernel<W, F, true><<<wblocks, 64, 0, st>>>(                                         \
                                         (uint32_t)n_spec, desc, (uint32_t)m, s_grp, s_qs, s_qe, s_ts, s_te, s_gidx, group_begin, \
                                         (uint32_t)n_groups, max_gap, c_d, c_j, c_n, v_own, v_prev, p_own, p_prev, wstats ? wstats + 8 : nullptr, \
                                         nullptr, nullptr, walk_plain_knob()))
          if (!lists_all) {
            if (ring <= 256)
              SWG_WALK_SPEC(256, true);
            else
              SWG_WALK_SPEC(1024, true);
          } else if (ring <= 256) {
            SWG_WALK_SPEC(256, false);
          } else if (ring <= 1024) {
            SWG_WALK_SPEC(1024, false);
          } else {
            SWG_WALK_SPEC(4096, false);
          }
#undef SWG_WALK_SPEC
          SWG_KERNEL_CHECK(ctx);
          SWG_HIP(ctx, hipMemsetAsync(spec_changed, 0, 8, st));
          SWG_LAUNCH(ctx, "spec_check", spec_check_kernel<<<(unsigned)rblocks, EW, 0, st>>>((uint32_t)n_spec, desc, v_prev, ext, spec_changed));
          SWG_KERNEL_CHECK(ctx);
          uint64_t ch = 0;
          SWG_TRY(swg_read_scalars(ctx, reinterpret_cast<uint64_t*>(spec_changed), &ch, 1));
          ++rounds;
          if ((uint32_t)ch == 0) break;
        }
        if (getenv("SWG_DEBUG"))
          fprintf(stderr, "[swg] long units: %llu, blocks %llu (longest %llu), rounds %d\n", (unsigned long long)n_big,
                  (unsigned long long)n_spec, (unsigned long long)s_max, rounds);
        SWG_LAUNCH(ctx, "spec_final", spec_final_kernel<<<(unsigned)rblocks, EW, 0, st>>>((uint32_t)n_spec, desc, p_own, p_prev, pred));
        SWG_KERNEL_CHECK(ctx);
      }
      if (wstats) {
        uint64_t hs[16];
        SWG_TRY(swg_read_scalars(ctx, reinterpret_cast<uint64_t*>(wstats), hs, 16));
        for (int k = 0; k < 2; ++k)
          fprintf(stderr, "[swg] walk%s: batches %llu, work-list lanes %llu (+%llu woken), whole-window passes %llu (%llu batches)\n",
                  k ? " (speculative blocks)" : " (chunks)", (unsigned long long)hs[8 * k], (unsigned long long)hs[8 * k + 1],
                  (unsigned long long)hs[8 * k + 4], (unsigned long long)hs[8 * k + 2], (unsigned long long)hs[8 * k + 3]);
      }
    } else {
    unsigned long long* c_d = swg_alloc<unsigned long long>(ctx, (size_t)KC * m);
    uint32_t* c_j = swg_alloc<uint32_t>(ctx, (size_t)KC * m);
    uint32_t* c_n = swg_alloc<uint32_t>(ctx, m);
    uint32_t* c_ext = swg_alloc<uint32_t>(ctx, m);
    (void)swg_alloc<uint8_t>(ctx, CAND_PAD);  // readable bytes behind the coordinate arrays (cand_scan_fast)
    SWG_CHECK_ARENA(ctx);
    if (long_groups || force_deep)
      launch_candidates_wave(ctx, st, nblk((m + CW_PER_WAVE - 1) / CW_PER_WAVE * 64), m, s_gidx, group_begin, (uint32_t)n_groups, s_grp, s_qs, s_qe,
                             s_ts, s_te, max_gap, c_d, c_j, c_n, c_ext);
    else
      SWG_LAUNCH(ctx, "chain_candidates", chain_candidates_kernel<<<nblk(m), EW, 0, st>>>(m, s_gidx, group_begin, (uint32_t)n_groups, s_grp, s_qs,
                                                                              s_qe, s_ts, s_te, max_gap, c_d, c_j, c_n, c_ext));
    SWG_KERNEL_CHECK(ctx);
    SWG_LAUNCH(ctx, "chain_select_lanes", chain_select_lanes_kernel<<<nblk(n_units), EW, 0, st>>>((uint32_t)n_units, unit_begin, (uint32_t)m, s_grp,
                                                                                    s_qs, s_qe, s_ts, s_te, max_gap, c_d, c_j, c_n, c_ext, bps,
                                                                                    pred));
    SWG_KERNEL_CHECK(ctx);
    {
      uint8_t* is_big = swg_alloc<uint8_t>(ctx, n_units);
      uint64_t* d_nb = swg_alloc<uint64_t>(ctx, 1);
      SWG_CHECK_ARENA(ctx);
      SWG_LAUNCH(ctx, "unit_big_flag", unit_big_flag_kernel<<<nblk(n_units), EW, 0, st>>>((uint32_t)n_units, unit_begin, (uint32_t)m, is_big));
      SWG_KERNEL_CHECK(ctx);
      swg_flag_scan big_scan;
      SWG_TRY(swg_flags_count(ctx, is_big, n_units, &big_scan, d_nb));
      uint64_t n_big = 0;
      SWG_TRY(swg_read_scalars(ctx, d_nb, &n_big, 1));
      if (n_big) {
        uint32_t* big_list = swg_alloc<uint32_t>(ctx, n_big);
        SWG_CHECK_ARENA(ctx);
        SWG_TRY(swg_flags_compact(ctx, big_scan, big_list));
        // ---- block-speculative selection of the long units
        uint32_t* S_u = swg_alloc<uint32_t>(ctx, n_big);
        uint32_t* nblk_u = swg_alloc<uint32_t>(ctx, n_big);
        uint32_t* blk_off = swg_alloc<uint32_t>(ctx, n_big);
        uint64_t* d_nblk = swg_alloc<uint64_t>(ctx, 1);
        unsigned long long* ext = swg_alloc<unsigned long long>(ctx, m);
        unsigned long long* v_own = swg_alloc<unsigned long long>(ctx, m);
        unsigned long long* v_prev = swg_alloc<unsigned long long>(ctx, m);
        uint32_t* p_own = swg_alloc<uint32_t>(ctx, m);
        uint32_t* p_prev = swg_alloc<uint32_t>(ctx, m);
        uint32_t* spec_changed = swg_alloc<uint32_t>(ctx, 2);
        uint64_t* d_smax = swg_alloc<uint64_t>(ctx, 1);  // adjacent to d_nblk: read back together
        SWG_CHECK_ARENA(ctx);
        SWG_HIP(ctx, hipMemsetAsync(d_smax, 0, 8, st));
        if (m / n_big > 65536) {  // few, very long units
          uint32_t* wmax_u = swg_alloc<uint32_t>(ctx, n_units);
          SWG_CHECK_ARENA(ctx);
          SWG_HIP(ctx, hipMemsetAsync(wmax_u, 0, n_units * sizeof(uint32_t), st));
          const uint64_t wb = nblk(m) < (uint64_t)ctx->num_cu * 16 ? nblk(m) : (uint64_t)ctx->num_cu * 16;
          SWG_LAUNCH(ctx, "unit_wmax", unit_wmax_kernel<<<(unsigned)wb, EW, 0, st>>>(m, unit_flag, unit_excl, is_big, c_ext, wmax_u));
          SWG_KERNEL_CHECK(ctx);
          SWG_LAUNCH(ctx, "spec_plan_from_wmax", spec_plan_from_wmax_kernel<<<nblk(n_big), EW, 0, st>>>((uint32_t)n_big, big_list, (uint32_t)n_units, unit_begin,
                                                                                (uint32_t)m, wmax_u, S_u, nblk_u,
                                                                                reinterpret_cast<uint32_t*>(d_smax)));
          SWG_KERNEL_CHECK(ctx);
        } else {
          SWG_LAUNCH(ctx, "spec_plan", spec_plan_kernel<<<(unsigned)n_big, EW, 0, st>>>((uint32_t)n_big, big_list, (uint32_t)n_units, unit_begin,
                                                                          (uint32_t)m, c_ext, S_u, nblk_u,
                                                                          reinterpret_cast<uint32_t*>(d_smax)));
          SWG_KERNEL_CHECK(ctx);
        }
        SWG_TRY(swg_exclusive_scan_u32(ctx, nblk_u, blk_off, n_big, d_nblk));
        uint64_t n_spec = 0, s_max = 0;
        SWG_TRY(swg_read_scalars(ctx, d_nblk, &n_spec, 1));
        SWG_TRY(swg_read_scalars(ctx, d_smax, &s_max, 1));
        s_max &= 0xffffffffull;
        SpecBlock* desc = swg_alloc<SpecBlock>(ctx, n_spec);
        SWG_CHECK_ARENA(ctx);
        SWG_LAUNCH(ctx, "spec_desc", spec_desc_kernel<<<(unsigned)n_big, EW, 0, st>>>((uint32_t)n_big, big_list, (uint32_t)n_units, unit_begin,
                                                                        (uint32_t)m, S_u, nblk_u, blk_off, desc));
        SWG_KERNEL_CHECK(ctx);
        SWG_LAUNCH(ctx, "fill", fill_u64_kernel<<<nblk(m), EW, 0, st>>>(m, reinterpret_cast<uint64_t*>(ext), ~0ull));
        SWG_KERNEL_CHECK(ctx);
        const uint64_t rblocks = n_spec < (uint64_t)ctx->num_cu * 8 ? n_spec : (uint64_t)ctx->num_cu * 8;
        const bool small_ring = s_max + 64 <= 1024;  // every window fits a 1024-slot ring: 20 KB of LDS instead of 80
        // The ring is a cache (positions outside it are read from global memory), so its size only trades LDS hits for
        // resident wavefronts: with windows of at most a few hundred elements a 256-slot ring (5 KB) lets 32 one-wave
        // work-groups share a CU instead of 8.
        static const char* ring_knob = getenv("SWG_SPEC_RING");
        const bool tiny_ring = ring_knob ? atoi(ring_knob) == 256 : s_max <= 512;
        int rounds = 0;
        for (uint64_t round = 0; round <= n_spec + 1; ++round) {
          SWG_LAUNCH(ctx, "spec_init", spec_init_kernel<<<(unsigned)rblocks, EW, 0, st>>>((uint32_t)n_spec, desc, ext, v_own, v_prev, p_own, p_prev));
          SWG_KERNEL_CHECK(ctx);
          if (tiny_ring)
            SWG_LAUNCH(ctx, "spec_round", spec_round_kernel<256><<<(unsigned)(n_spec < (uint64_t)ctx->num_cu * 32 ? n_spec : (uint64_t)ctx->num_cu * 32), 64, 0, st>>>(
                                              (uint32_t)n_spec, desc, (uint32_t)m, s_grp, s_qs, s_qe, s_ts, s_te, max_gap, c_d, c_j, c_n, v_own,
                                              v_prev, p_own, p_prev));
          else if (small_ring)
            SWG_LAUNCH(ctx, "spec_round", spec_round_kernel<1024><<<(unsigned)rblocks, 64, 0, st>>>(
                                              (uint32_t)n_spec, desc, (uint32_t)m, s_grp, s_qs, s_qe, s_ts, s_te, max_gap, c_d, c_j, c_n, v_own,
                                              v_prev, p_own, p_prev));
          else
            SWG_LAUNCH(ctx, "spec_round", spec_round_kernel<4096><<<(unsigned)rblocks, 64, 0, st>>>(
                                              (uint32_t)n_spec, desc, (uint32_t)m, s_grp, s_qs, s_qe, s_ts, s_te, max_gap, c_d, c_j, c_n, v_own,
                                              v_prev, p_own, p_prev));
          SWG_KERNEL_CHECK(ctx);
          SWG_HIP(ctx, hipMemsetAsync(spec_changed, 0, 8, st));
          SWG_LAUNCH(ctx, "spec_check", spec_check_kernel<<<(unsigned)rblocks, EW, 0, st>>>((uint32_t)n_spec, desc, v_prev, ext, spec_changed));
          SWG_KERNEL_CHECK(ctx);
          uint64_t ch = 0;
          SWG_TRY(swg_read_scalars(ctx, reinterpret_cast<uint64_t*>(spec_changed), &ch, 1));
          ++rounds;
          if ((uint32_t)ch == 0) break;
        }
        if (getenv("SWG_DEBUG"))
          fprintf(stderr, "[swg] long units: %llu, blocks %llu (longest %llu), rounds %d\n", (unsigned long long)n_big,
                  (unsigned long long)n_spec, (unsigned long long)s_max, rounds);
#ifdef SWG_SPEC_STATS
        {
          unsigned long long hs[8];
          (void)hipStreamSynchronize(st);
          (void)hipMemcpyFromSymbol(hs, HIP_SYMBOL(g_spec_stats), sizeof hs);
          fprintf(stderr, "[swg] spec stats (cumulative): steps %llu, fallbacks %llu, fallback batches %llu\n", hs[0], hs[1], hs[2]);
        }
#endif
        SWG_LAUNCH(ctx, "spec_final", spec_final_kernel<<<(unsigned)rblocks, EW, 0, st>>>((uint32_t)n_spec, desc, p_own, p_prev, pred));
        SWG_KERNEL_CHECK(ctx);
      }
    }
    {
      // what is left: units longer than a lane should walk and shorter than BIG_UNIT, one wavefront each
      uint8_t* is_mid = swg_alloc<uint8_t>(ctx, n_units);
      uint64_t* d_nm = swg_alloc<uint64_t>(ctx, 1);
      SWG_CHECK_ARENA(ctx);
      SWG_LAUNCH(ctx, "unit_mid_flag", unit_mid_flag_kernel<<<nblk(n_units), EW, 0, st>>>((uint32_t)n_units, unit_begin, (uint32_t)m, is_mid));
      SWG_KERNEL_CHECK(ctx);
      swg_flag_scan mid_scan;
      SWG_TRY(swg_flags_count(ctx, is_mid, n_units, &mid_scan, d_nm));
      uint64_t n_mid = 0;
      SWG_TRY(swg_read_scalars(ctx, d_nm, &n_mid, 1));
      if (n_mid) {
        uint32_t* mid_list = swg_alloc<uint32_t>(ctx, n_mid);
        SWG_CHECK_ARENA(ctx);
        SWG_TRY(swg_flags_compact(ctx, mid_scan, mid_list));
        uint64_t blocks = (n_mid + 3) / 4;
        const uint64_t max_blocks = (uint64_t)ctx->num_cu * 8;
        if (blocks > max_blocks) blocks = max_blocks;
        SWG_LAUNCH(ctx, "chain_select", chain_select_kernel<<<(unsigned)blocks, 256, 0, st>>>((uint32_t)n_mid, mid_list, (uint32_t)n_units, unit_begin,
                                                                                  (uint32_t)m, s_grp, s_qs, s_qe, s_ts, s_te, max_gap, c_d, c_j,
                                                                                  c_n, s_gidx, group_begin, (uint32_t)n_groups, bps, pred));
        SWG_KERNEL_CHECK(ctx);
      }
    }
    }
  }
  ChainWork& W = *work;
  W.n_groups = n_groups;
  W.s_qs = s_qs;
  W.s_qe = s_qe;
  W.s_ts = s_ts;
  W.s_te = s_te;
  W.s_m = s_m;
  W.s_b = s_b;
  W.s_grp = s_grp;
  W.head_flag = head_flag;
  W.s_gidx = s_gidx;
  W.group_begin = group_begin;
  W.pred = pred;
  W.d_tot = d_tot;
  return SWG_OK;
}

// The walk over a chunk list made on the device (swg_pair.hip): every chunk lies inside one (query, target, strand) group and
// carries its strand; `cap_chunks` bounds the list, *n_chunks_dev is its length.  pred must hold NONE for every member.
int pair_walk_launch(swg_ctx* ctx, uint32_t cap_chunks, const uint32_t* n_chunks_dev, const SpecBlock* desc, const uint32_t* s_qs,
                     const uint32_t* s_qe, const uint32_t* s_ts, const uint32_t* s_te, uint64_t max_gap, unsigned long long* bps,
                     uint32_t* pred) {
  if (cap_chunks == 0) return SWG_OK;
  static const uint64_t per_cu = getenv("SWG_WALK_BLOCKS") ? (uint64_t)atoi(getenv("SWG_WALK_BLOCKS")) : 128;  // (blocks per CU: 64 -> 128 took 10 % off the walk -- chunks are of uneven length, a finer grid balances them)
  const uint64_t wb = cap_chunks < (uint64_t)ctx->num_cu * per_cu ? cap_chunks : (uint64_t)ctx->num_cu * per_cu;
  SWG_LAUNCH(ctx, "chain_walk", chain_walk_kernel<256, true, false><<<(unsigned)wb, 64, 0, ctx->stream>>>(
                                    cap_chunks, desc, 0u, nullptr, s_qs, s_qe, s_ts, s_te, nullptr, nullptr, 0u, max_gap, nullptr, nullptr,
                                    nullptr, bps, bps, pred, pred, nullptr, n_chunks_dev, nullptr, walk_plain_knob()));
  SWG_KERNEL_CHECK(ctx);
  return SWG_OK;
}

// The long chunks of the pair-resident path (a unit of LABEL_CAP_ELEMS members or more would pin one wavefront for its whole
// length): cut into blocks and walked block-speculatively like the long units of the global path (above spec_plan_kernel), but
// with everything decided on the device -- the plan, the list's length, and whether another round is needed: ROUNDS rounds
// are enqueued, a round behind a round that changed nothing returns at once, and rounds that have not settled by then set
// `fallback_bit` in *flags.  own / pred double as the walk's own score and predecessor arrays (the long chunks' positions belong
// to no other chunk).
int pair_walk_long_launch(swg_ctx* ctx, uint32_t cap_long, const uint32_t* n_long_dev, const uint32_t* long_list, const SpecBlock* chunks,
                          uint32_t n_members_cap, const uint32_t* s_qs, const uint32_t* s_qe, const uint32_t* s_ts, const uint32_t* s_te,
                          uint64_t max_gap, unsigned long long* own, uint32_t* pred, uint32_t* flags, uint32_t fallback_bit) {
  if (cap_long == 0) return SWG_OK;
  hipStream_t st = ctx->stream;
  constexpr int ROUNDS = 4;
  const uint32_t cap_spec = n_members_cap / 512 + cap_long + 1;
  SpecBlock* desc = swg_alloc<SpecBlock>(ctx, cap_spec);
  unsigned long long* ext = swg_alloc<unsigned long long>(ctx, n_members_cap);
  unsigned long long* v_prev = swg_alloc<unsigned long long>(ctx, n_members_cap);
  uint32_t* p_prev = swg_alloc<uint32_t>(ctx, n_members_cap);
  uint32_t* counters = swg_alloc<uint32_t>(ctx, 2 + ROUNDS);  // n_spec, overflow, changed[ROUNDS]
  SWG_CHECK_ARENA(ctx);
  SWG_HIP(ctx, hipMemsetAsync(counters, 0, (2 + ROUNDS) * sizeof(uint32_t), st));
  const unsigned pg = cap_long < (uint32_t)ctx->num_cu * 4 ? cap_long : (unsigned)ctx->num_cu * 4;
  SWG_LAUNCH(ctx, "spec_plan", pair_long_plan_kernel<<<pg, EW, 0, st>>>(cap_long, n_long_dev, long_list, chunks, s_qs, s_qe, max_gap, desc, cap_spec,
                                                              counters, ext, flags, fallback_bit));
  SWG_KERNEL_CHECK(ctx);
  const unsigned rblocks = cap_spec < (uint32_t)ctx->num_cu * 8 ? cap_spec : (unsigned)ctx->num_cu * 8;
  const unsigned wblocks = cap_spec < (uint32_t)ctx->num_cu * 32 ? cap_spec : (unsigned)ctx->num_cu * 32;
  for (int r = 0; r < ROUNDS; ++r) {
    const uint32_t* gate = r ? counters + 2 + (r - 1) : nullptr;
    SWG_LAUNCH(ctx, "spec_init", spec_init_kernel<<<rblocks, EW, 0, st>>>(cap_spec, desc, ext, own, v_prev, pred, p_prev, counters, gate));
    SWG_KERNEL_CHECK(ctx);
    SWG_LAUNCH(ctx, "chain_walk_spec", chain_walk_kernel<256, true, true><<<wblocks, 64, 0, st>>>(
                                           cap_spec, desc, 0u, nullptr, s_qs, s_qe, s_ts, s_te, nullptr, nullptr, 0u, max_gap, nullptr, nullptr,
                                           nullptr, own, v_prev, pred, p_prev, nullptr, counters, gate, walk_plain_knob()));
    SWG_KERNEL_CHECK(ctx);
    SWG_LAUNCH(ctx, "spec_check", spec_check_kernel<<<rblocks, EW, 0, st>>>(cap_spec, desc, v_prev, ext, counters + 2 + r, counters, gate));
    SWG_KERNEL_CHECK(ctx);
  }
  SWG_LAUNCH(ctx, "spec_final", spec_final_kernel<<<rblocks, EW, 0, st>>>(cap_spec, desc, pred, p_prev, pred, counters));
  SWG_KERNEL_CHECK(ctx);
  SWG_LAUNCH(ctx, "spec_final", pair_long_verdict_kernel<<<1, 64, 0, st>>>(counters, counters + 2 + (ROUNDS - 1), flags, fallback_bit));
  SWG_KERNEL_CHECK(ctx);
  return SWG_OK;
}

}  // namespace swg_scaf
