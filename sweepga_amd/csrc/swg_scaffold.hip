// Scaffold stage of the filter (src/paf_filter.rs:436-747) on the device: the driver and its last part.
//
//   chains      swg_chain.hip (sort A, best-buddy predecessors) + swg_chain_table.hip (labels, aggregates, order, filter)
//   sweep       swg_scaffold_sweep.hip: plane_sweep_both per chromosome pair + chain_N numbering
//   anchors     members of kept chains; inversion capture (paf_filter.rs:535-597)
//   rescue      per chromosome pair, anchors sorted by query centre, window search (paf_filter.rs:599-747)
#include "swg_scaffold_internal.h"

using namespace swg_scaf;

namespace {

// ---- anchors, inversions, rescue -------------------------------------------------------------------------------------
// anchor_num[i] = chain number of the kept chain record i belongs to (0 = not an anchor);
// a_state[a] (by A position, only with a rescue): 1 = anchor, 2 = member of a chain that passed the span / identity filter
// but not the scaffold sweep (pre_sweep_scaffold_members minus anchors: never rescued, paf_filter.rs:601-604, 675-678)
// First the chain numbers go to the heads (one entry per kept chain), then every member reads its head's: one dependent
// look-up per member instead of three (head -> position ordinal -> all_chains index -> number).
__global__ __launch_bounds__(EW) void head_numbers_kernel(uint64_t nc, const uint32_t* __restrict__ head_of_chain,
                                                          const uint32_t* __restrict__ C_num, uint32_t* __restrict__ head_num) {
  uint64_t c = (uint64_t)blockIdx.x * EW + threadIdx.x;
  if (c < nc) head_num[head_of_chain[c]] = C_num[c];
}
__global__ __launch_bounds__(EW) void member_marks_kernel(uint64_t m, const uint32_t* __restrict__ s_idx,
                                                          const uint32_t* __restrict__ hd,
                                                          const uint8_t* __restrict__ ok_head,
                                                          const uint32_t* __restrict__ head_num,
                                                          uint32_t* __restrict__ anchor_num,
                                                          const uint32_t* __restrict__ s_a, uint8_t* __restrict__ a_state) {
  uint64_t p = (uint64_t)swg_xcd_block(blockIdx.x, gridDim.x) * EW + threadIdx.x;
  if (p >= m) return;
  const uint32_t h = hd[p];
  if (!ok_head[h]) return;  // its chain failed the span / identity filter: neither anchor nor pre-sweep member (arrays pre-zeroed)
  const uint32_t i = s_idx[p];
  const uint32_t num = head_num[h];
  anchor_num[i] = num;
  if (a_state) a_state[s_a ? s_a[p] : p] = num ? 1 : 2;  // (nullptr: no rescue will run; pre-zeroed otherwise)
}

// Kept '+' chains compacted in all_chains order.  A chromosome pair's '+' group is contiguous in that order and
// sorted by chain q_start (head position order), so per pair the list is a q_start-sorted range: no CSR, no atomics.
__global__ __launch_bounds__(EW) void fwd_flag_kernel(uint64_t nc, const uint32_t* __restrict__ C_num,
                                                      const uint8_t* __restrict__ C_strand, uint8_t* __restrict__ flag) {
  uint64_t c = (uint64_t)blockIdx.x * EW + threadIdx.x;
  if (c < nc) flag[c] = (C_num[c] && C_strand[c] == 0) ? 1 : 0;
}
__global__ __launch_bounds__(EW) void fwd_cols_kernel(uint64_t nf, const uint32_t* __restrict__ f_c,
                                                      const uint32_t* __restrict__ C_qs, const uint32_t* __restrict__ C_qe,
                                                      const uint32_t* __restrict__ C_ts, const uint32_t* __restrict__ C_num,
                                                      const uint32_t* __restrict__ C_dpair, uint32_t* __restrict__ f_qs,
                                                      uint32_t* __restrict__ f_qe, uint32_t* __restrict__ f_ts,
                                                      uint32_t* __restrict__ f_num, uint32_t* __restrict__ f_dp) {
  uint64_t k = (uint64_t)blockIdx.x * EW + threadIdx.x;
  if (k >= nf) return;
  const uint32_t c = f_c[k];
  f_qs[k] = C_qs[c];
  f_qe[k] = C_qe[c];
  f_ts[k] = C_ts[c];
  f_num[k] = C_num[c];
  f_dp[k] = C_dpair[c];
}
// range of each pair in the list; one thread per list entry looks at its neighbours
__global__ __launch_bounds__(EW) void fwd_ranges_kernel(uint64_t nf, const uint32_t* __restrict__ f_dp,
                                                        uint32_t* __restrict__ pair_lo, uint32_t* __restrict__ pair_hi) {
  uint64_t k = (uint64_t)blockIdx.x * EW + threadIdx.x;
  if (k >= nf) return;
  const uint32_t dp = f_dp[k];
  if (k == 0 || f_dp[k - 1] != dp) pair_lo[dp] = (uint32_t)k;
  if (k + 1 == nf || f_dp[k + 1] != dp) pair_hi[dp] = (uint32_t)k + 1;
}
// running maximum of chain q_end inside each pair range (one wavefront per pair)
__global__ __launch_bounds__(EW) void fwd_prefmax_kernel(uint32_t n_pairs, const uint32_t* __restrict__ pair_lo,
                                                         const uint32_t* __restrict__ pair_hi,
                                                         const uint32_t* __restrict__ f_qe, uint32_t* __restrict__ f_pm) {
  const int lane = threadIdx.x & 63;
  const uint32_t wave_global = blockIdx.x * (EW / 64) + (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));  // wave-uniform, and visibly so
  const uint32_t n_waves = (gridDim.x * EW) >> 6;
  for (uint32_t dp = wave_global; dp < n_pairs; dp += n_waves) {
    const uint32_t lo = pair_lo[dp], hi = pair_hi[dp];
    uint32_t carry = 0;
    for (uint32_t k0 = lo; k0 < hi; k0 += 64) {
      const uint32_t k = k0 + lane;
      uint32_t v = k < hi ? f_qe[k] : 0u;
#pragma unroll
      for (int d = 1; d < 64; d <<= 1) {
        const uint32_t t = __shfl_up(v, d, 64);
        if (lane >= d && t > v) v = t;
      }
      if (carry > v) v = carry;
      if (k < hi) f_pm[k] = v;
      carry = __shfl(v, 63, 64);
    }
  }
}
// the same as one segmented scan: key = (first slot of the chain's pair << 32) | q_end; slots of a pair are contiguous
// and first slots grow with the slot index, so a u64 running maximum never crosses a pair boundary
__global__ __launch_bounds__(EW) void fwd_compose_kernel(uint64_t nf, const uint32_t* __restrict__ f_dp,
                                                         const uint32_t* __restrict__ pair_lo,
                                                         const uint32_t* __restrict__ f_qe, uint64_t* __restrict__ comp) {
  uint64_t k = (uint64_t)blockIdx.x * EW + threadIdx.x;
  if (k < nf) comp[k] = ((uint64_t)pair_lo[f_dp[k]] << 32) | f_qe[k];
}
__global__ __launch_bounds__(EW) void fwd_extract_kernel(uint64_t nf, const uint64_t* __restrict__ comp,
                                                         uint32_t* __restrict__ f_pm) {
  uint64_t k = (uint64_t)blockIdx.x * EW + threadIdx.x;
  if (k < nf) f_pm[k] = (uint32_t)comp[k];
}
// (fp_thresholds_kernel: swg_scaffold_internal.h)

// paf_filter.rs:535-597: a '-' record joins the first (lowest-numbered) kept '+' chain of its pair whose
// diagonal it sits on.  Candidates: chains with q_start <= q_end(rec) + gap (binary search) and
// q_end + gap >= q_start(rec) (backward scan, stopped by the running maximum).
__global__ __launch_bounds__(EW) void inversion_kernel(uint64_t M, const uint64_t* __restrict__ keyA,
                                                       const uint32_t* __restrict__ idxA,
                                                       const uint32_t* __restrict__ a_qe,
                                                       const uint32_t* __restrict__ a_ts,
                                                       const uint32_t* __restrict__ a_te,
                                                       const uint32_t* __restrict__ a_dpair, int pos_bits,
                                                       const uint32_t* __restrict__ pair_lo,
                                                       const uint32_t* __restrict__ pair_hi,
                                                       const uint32_t* __restrict__ f_qs,
                                                       const uint32_t* __restrict__ f_qe,
                                                       const uint32_t* __restrict__ f_pm,
                                                       const uint32_t* __restrict__ f_ts,
                                                       const uint32_t* __restrict__ f_num, uint64_t gap,
                                                       const uint64_t* __restrict__ fp_thr,
                                                       uint32_t* anchor_num, uint8_t* __restrict__ a_state) {
  uint64_t a = (uint64_t)blockIdx.x * EW + threadIdx.x;
  if (a >= M) return;
  const uint64_t k = keyA[a];
  if (((k >> pos_bits) & 1ull) == 0) return;  // only '-' records
  const uint32_t i = idxA[a];
  if (anchor_num[i]) return;
  const uint32_t dp = a_dpair[a];
  const uint32_t lb = pair_lo[dp], le = pair_hi[dp];
  if (lb >= le) return;
  const uint64_t qs = k & ((uint64_t(1) << pos_bits) - 1), qe = a_qe[a], ts = a_ts[a], te = a_te[a];
  const uint64_t qc = (qs + qe) / 2, tc = (ts + te) / 2;
  // chain.query_start.saturating_sub(gap) <= qe  <=>  chain.query_start <= qe + gap
  const uint64_t lim = qe > ~0ull - gap ? ~0ull : qe + gap;
  uint32_t l = lb, r = le;  // upper bound: first chain with q_start > lim
  while (l < r) {
    const uint32_t mid = l + ((r - l) >> 1);
    if ((uint64_t)f_qs[mid] <= lim)
      l = mid + 1;
    else
      r = mid;
  }
  // Among the pair's kept '+' chains [lb, l) (q_start order = chain-number order inside a pair) the record joins the FIRST
  // one that passes the window and diagonal tests.  No chain before c0 = the first slot whose running maximum of chain ends
  // reaches the record (a binary search: the running maximum never falls) can pass the window test, so the scan starts
  // there, runs forward and stops at the first hit (round 2 scanned backward from l through every chain the running maximum
  // could not rule out, keeping the minimum: ~3 ms per 10^7 records on one deep pair).
  uint32_t c0 = lb;
  {
    uint32_t lo = lb, hi = l;  // first slot in [lb, l) with f_pm + gap >= qs
    while (lo < hi) {
      const uint32_t mid = lo + ((hi - lo) >> 1);
      const uint64_t pm = f_pm[mid];
      if ((pm > ~0ull - gap ? ~0ull : pm + gap) < qs)
        lo = mid + 1;
      else
        hi = mid;
    }
    c0 = lo;
  }
  uint32_t best = 0;
  const uint64_t max_dev = fp_thr[0];  // (u64)(deviation / SQRT_2) <= gap  <=>  deviation <= max_dev (fp_thresholds_kernel)
  for (uint32_t c = c0; c < l; ++c) {
    const uint64_t cqe = f_qe[c];
    if ((cqe > ~0ull - gap ? ~0ull : cqe + gap) < qs) continue;  // mapping.query_start > extended_query_end
    const int64_t diag = (int64_t)f_ts[c] - (int64_t)f_qs[c];
    const int64_t dev = (int64_t)tc - (int64_t)qc - diag;
    const uint64_t deviation = dev < 0 ? (uint64_t)0 - (uint64_t)dev : (uint64_t)dev;
    if (deviation <= max_dev) {
      best = f_num[c];
      break;
    }
  }
  if (best) {
    anchor_num[i] = best;
    if (a_state) a_state[a] = 1;  // an anchor from here on, whatever it was (a member of a swept-away chain included)
  }
}

// anchors in A order -> keys (dense pair, query centre) for sort B.  (The states by A position are written where the anchors
// are made, member_marks and the inversion capture -- round 3 derived the anchor flags with one more gather pass over all
// records and kept the never-rescued members in a second, record-indexed array that the rescue had to gather from.)
__global__ __launch_bounds__(EW) void anchor_flag_kernel(uint64_t M, const uint8_t* __restrict__ a_state, uint8_t* __restrict__ flag) {
  const uint64_t a0 = ((uint64_t)blockIdx.x * EW + threadIdx.x) * 4;
  if (a0 + 4 <= M) {
    const uint32_t w = *reinterpret_cast<const uint32_t*>(a_state + a0);
    *reinterpret_cast<uint32_t*>(flag + a0) = w & 0x01010101u;  // state 1 = anchor, 2 = never-rescued member
  } else {
    for (uint64_t a = a0; a < M; ++a) flag[a] = a_state[a] & 1u;
  }
}
__global__ __launch_bounds__(EW) void anchor_keys_kernel(uint64_t na, const uint32_t* __restrict__ anchor_a,
                                                         const uint64_t* __restrict__ keyA,
                                                         const uint32_t* __restrict__ a_qe,
                                                         const uint32_t* __restrict__ a_dpair, int pos_bits,
                                                         uint64_t* __restrict__ key) {
  uint64_t j = (uint64_t)blockIdx.x * EW + threadIdx.x;
  if (j >= na) return;
  const uint32_t a = anchor_a[j];
  const uint64_t qs = keyA[a] & ((uint64_t(1) << pos_bits) - 1);
  const uint64_t qc = (qs + (uint64_t)a_qe[a]) / 2;
  key[j] = ((uint64_t)a_dpair[a] << pos_bits) | qc;
}
__global__ __launch_bounds__(EW) void anchor_cols_kernel(uint64_t na, const uint32_t* __restrict__ sorted_a,
                                                         const uint32_t* __restrict__ idxA,
                                                         const uint32_t* __restrict__ a_ts,
                                                         const uint32_t* __restrict__ a_te,
                                                         const uint32_t* __restrict__ anchor_num,
                                                         uint32_t* __restrict__ b_tc, uint32_t* __restrict__ b_idx,
                                                         uint32_t* __restrict__ b_num) {
  uint64_t j = (uint64_t)blockIdx.x * EW + threadIdx.x;
  if (j >= na) return;
  const uint32_t a = sorted_a[j];
  const uint32_t i = idxA[a];
  b_tc[j] = (uint32_t)(((uint64_t)a_ts[a] + (uint64_t)a_te[a]) / 2);
  b_idx[j] = i;
  b_num[j] = anchor_num[i];
}

// The same after the packed sort (swg_radix_sort_packed): P[j] = ((key >> 8) << val_bits) | A position.  The key's low 8
// bits -- the low 8 bits of the query centre, which the packed word drops -- come back from the record (pos_bits >= 8 here).
__global__ __launch_bounds__(EW) void anchor_cols_packed_kernel(uint64_t na, const uint64_t* __restrict__ P, int val_bits,
                                                                const uint64_t* __restrict__ keyA,
                                                                const uint32_t* __restrict__ idxA,
                                                                const uint32_t* __restrict__ a_qe,
                                                                const uint32_t* __restrict__ a_ts,
                                                                const uint32_t* __restrict__ a_te,
                                                                const uint32_t* __restrict__ anchor_num, int pos_bits,
                                                                uint64_t* __restrict__ b_key, uint32_t* __restrict__ b_tc,
                                                                uint32_t* __restrict__ b_idx, uint32_t* __restrict__ b_num) {
  uint64_t j = (uint64_t)blockIdx.x * EW + threadIdx.x;
  if (j >= na) return;
  const uint64_t w = P[j];
  const uint32_t a = (uint32_t)(w & ((uint64_t(1) << val_bits) - 1));
  const uint64_t qs = keyA[a] & ((uint64_t(1) << pos_bits) - 1);
  const uint64_t qc = (qs + (uint64_t)a_qe[a]) / 2;
  const uint32_t i = idxA[a];
  b_key[j] = ((w >> val_bits) << 8) | (qc & 0xffu);
  b_tc[j] = (uint32_t)(((uint64_t)a_ts[a] + (uint64_t)a_te[a]) / 2);
  b_idx[j] = i;
  b_num[j] = anchor_num[i];
}

// [lo, hi) of every dense pair in the sorted anchor table (pairs without anchors keep the zero-initialised empty range)
__global__ __launch_bounds__(EW) void anchor_ranges_kernel(uint64_t na, const uint64_t* __restrict__ b_key, int pos_bits,
                                                           uint32_t* __restrict__ pair_lo, uint32_t* __restrict__ pair_hi) {
  uint64_t j = (uint64_t)blockIdx.x * EW + threadIdx.x;
  if (j >= na) return;
  const uint64_t dp = b_key[j] >> pos_bits;
  if (j == 0 || (b_key[j - 1] >> pos_bits) != dp) pair_lo[dp] = (uint32_t)j;
  if (j + 1 == na || (b_key[j + 1] >> pos_bits) != dp) pair_hi[dp] = (uint32_t)(j + 1);
}

// lower_bound of `key` in b_key[lo, hi), found by the whole wavefront: 64 probes per round trip instead of one (a thread's own
// binary search over a pair's ~3,000 anchors is 12 dependent loads).  All arguments and the result are wave-uniform.
__device__ __forceinline__ uint32_t wave_lower_bound(const uint64_t* __restrict__ b_key, uint32_t lo, uint32_t hi, uint64_t key) {
  const uint32_t lane = threadIdx.x & 63;
  for (;;) {
    const uint32_t n = hi - lo;
    if (n == 0) return lo;
    if (n <= 64) {
      const bool ge = lane < n && b_key[lo + lane] >= key;
      const uint64_t m = __ballot(ge);
      return m ? lo + (uint32_t)__builtin_ctzll(m) : hi;
    }
    const uint32_t step = (n + 63) / 64;  // probe i looks at the LAST element of the i-th stretch of `step` elements
    uint32_t pos = lo + (lane + 1) * step;
    if (pos > hi) pos = hi;
    const bool ge = b_key[pos - 1] >= key;
    const uint64_t m = __ballot(ge);
    if (!m) return hi;  // (the last probe is element hi - 1)
    const uint32_t f = (uint32_t)__builtin_ctzll(m);
    uint32_t nlo = lo + f * step, nhi = lo + (f + 1) * step;
    if (nhi > hi) nhi = hi;
    lo = nlo;       // stretch f: its last element is >= key, every earlier stretch's last element is < key
    hi = nhi - 1;   // so the bound lies in [nlo, nhi - 1]; nhi - 1 itself qualifies (returned when nothing before it does)
    // (n shrinks by a factor of 64 per round)
  }
}

// paf_filter.rs:656-732 per record.  Rescued records take the chain of the lowest-index anchor in range
// (the reference iterates a HashSet here; ascending input order is the instance the oracle fixes).
// The records of a wavefront are consecutive in A order -- (pair, strand, q_start) -- so nearly always one pair's, with window
// starts close together: the wavefront finds the bounds of its smallest and largest window start together
// (wave_lower_bound), and each lane then searches only between them.
__global__ __launch_bounds__(EW) void rescue_kernel(uint64_t M, const uint64_t* __restrict__ keyA,
                                                    const uint32_t* __restrict__ idxA,
                                                    const uint32_t* __restrict__ a_qe,
                                                    const uint32_t* __restrict__ a_ts,
                                                    const uint32_t* __restrict__ a_te,
                                                    const uint32_t* __restrict__ a_dpair, int pos_bits,
                                                    const uint8_t* __restrict__ a_state,
                                                    const uint32_t* __restrict__ pair_lo,
                                                    const uint32_t* __restrict__ pair_hi,
                                                    const uint64_t* __restrict__ b_key,
                                                    const uint32_t* __restrict__ b_tc,
                                                    const uint32_t* __restrict__ b_idx,
                                                    const uint32_t* __restrict__ b_num, uint64_t D,
                                                    const uint64_t* __restrict__ fp_thr,
                                                    uint8_t* __restrict__ status, uint32_t* __restrict__ chain) {
  const uint64_t a = (uint64_t)swg_xcd_block(blockIdx.x, gridDim.x) * EW + threadIdx.x;
  // anchors got their status from the coalesced pass in input order; members of span-filtered chains stay DROPPED
  const bool act = a < M && a_state[a] == 0;
  if (__ballot(act) == 0) return;  // (wave-uniform)
  const uint64_t posmask = (uint64_t(1) << pos_bits) - 1;
  uint64_t qc = 0, tc = 0, lo = 0, hi = 0;
  uint32_t dp = NONE;
  if (act) {
    const uint64_t qs = keyA[a] & posmask;
    qc = (qs + (uint64_t)a_qe[a]) / 2;
    tc = ((uint64_t)a_ts[a] + (uint64_t)a_te[a]) / 2;
    dp = a_dpair[a];
    const uint64_t hi_part = (uint64_t)dp << pos_bits;
    lo = hi_part | (qc > D ? qc - D : 0);
    const uint64_t hi_q = qc + D < qc ? ~0ull : qc + D;
    hi = hi_part | (hi_q > posmask ? posmask : hi_q);
  }
  // the pair most of the wavefront works on: that of its first active lane
  const uint32_t dp0 = (uint32_t)__shfl((int)dp, (int)__builtin_ctzll(__ballot(act)), 64);
  const bool same = act && dp == dp0;
  uint64_t kmin = same ? lo : ~0ull, kmax = same ? lo : 0ull;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const uint64_t x = __shfl_xor(kmin, o, 64), y = __shfl_xor(kmax, o, 64);
    kmin = x < kmin ? x : kmin;
    kmax = y > kmax ? y : kmax;
  }
  const uint32_t s_lo = pair_lo[dp0], s_hi = pair_hi[dp0];
  const uint32_t L = wave_lower_bound(b_key, s_lo, s_hi, kmin);
  const uint32_t U = wave_lower_bound(b_key, L, s_hi, kmax);  // every lane's own bound lies in [L, U]
  if (!act) return;
  uint64_t slice_end, l, r;
  if (same) {
    slice_end = s_hi;
    l = L;
    r = U;
  } else {  // another pair begins inside this wavefront (rare): the lane's own search over its pair's slice
    slice_end = pair_hi[dp];
    l = pair_lo[dp];
    r = slice_end;
  }
  while (l < r) {  // lower_bound(b_key, lo) in [l, r]
    const uint64_t mid = (l + r) >> 1;
    if (b_key[mid] < lo)
      l = mid + 1;
    else
      r = mid;
  }
  uint32_t best_idx = NONE, best_num = 0;
  const uint64_t max_s2 = fp_thr[1];  // (u64)sqrt((q^2 + t^2) as f64) <= D  <=>  q^2 + t^2 <= max_s2 (fp_thresholds_kernel)
  for (uint64_t j = l; j < slice_end; ++j) {
    const uint64_t bk = b_key[j];
    if (bk > hi) break;
    const uint64_t aq = bk & posmask;
    const uint64_t q_diff = qc > aq ? qc - aq : aq - qc;
    if (q_diff > D) continue;
    const uint64_t at = b_tc[j];
    const uint64_t t_diff = tc > at ? tc - at : at - tc;
    if (q_diff * q_diff + t_diff * t_diff <= max_s2 && b_idx[j] < best_idx) {  // (wrapping sum, as the reference's)
      best_idx = b_idx[j];
      best_num = b_num[j];
    }
  }
  if (best_idx != NONE) {
    const uint32_t i = idxA[a];
    status[i] = SWG_ST_RESCUED;
    chain[i] = best_num;
  }
}

// The anchors' chain numbers are collected in the caller's chain column itself (pre-zeroed; member_marks and the inversion
// capture write it); this pass derives the status column from it, 16 records per thread: anchors are SCAFFOLD, the rest
// DROPPED until the rescue says otherwise.
__global__ __launch_bounds__(EW) void anchor_status_kernel(uint64_t n, const uint32_t* __restrict__ chain,
                                                           uint8_t* __restrict__ status, int aligned) {
  const uint64_t i0 = ((uint64_t)blockIdx.x * EW + threadIdx.x) * 16;
  if (i0 >= n) return;
  if (aligned && i0 + 16 <= n) {
    uint32_t w[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const uint4 c = *reinterpret_cast<const uint4*>(chain + i0 + 4 * j);
      w[j] = (c.x ? (uint32_t)SWG_ST_SCAFFOLD : 0u) | (c.y ? (uint32_t)SWG_ST_SCAFFOLD << 8 : 0u) |
             (c.z ? (uint32_t)SWG_ST_SCAFFOLD << 16 : 0u) | (c.w ? (uint32_t)SWG_ST_SCAFFOLD << 24 : 0u);
    }
    *reinterpret_cast<uint4*>(status + i0) = make_uint4(w[0], w[1], w[2], w[3]);
    return;
  }
  for (uint64_t i = i0; i < i0 + 16 && i < n; ++i) status[i] = chain[i] ? SWG_ST_SCAFFOLD : SWG_ST_DROPPED;
}

__global__ __launch_bounds__(EW) void count_status_kernel(uint64_t n, const uint8_t* __restrict__ status,
                                                          unsigned long long* __restrict__ out) {
  uint32_t cnt = 0;
  for (uint64_t i = (uint64_t)blockIdx.x * EW + threadIdx.x; i < n; i += (uint64_t)gridDim.x * EW) cnt += status[i] ? 1 : 0;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) cnt += __shfl_down(cnt, o, 64);
  if ((threadIdx.x & 63) == 0 && cnt) atomicAdd(out, (unsigned long long)cnt);
}

}  // namespace

int swg_scaffold_stage(swg_ctx* ctx, const swg_records* r, const swg_config* cfg, const uint8_t* alive,
                       const uint8_t* keep1, int pos_bits, uint8_t* status_out,
                       uint32_t* chain_out, swg_stats* stats, const uint32_t* q_order, uint64_t n_alive,
                       const swg_key_ends* slots) {
  const uint64_t n = r->n;
  hipStream_t st = ctx->stream;
  // chain_out doubles as the anchors' chain-number column (zero = not an anchor); status_out is written in full by
  // anchor_status below and only needs clearing on the early exits
  SWG_HIP(ctx, hipMemsetAsync(chain_out, 0, n * sizeof(uint32_t), st));
  ChainBuild B;
  B.want_s_chain = false;  // member_marks derives a member's chain from the labelling arrays
  SWG_TRY(build_chains(ctx, r, alive, keep1, cfg->scaffold_gap, cfg->min_scaffold_length, cfg->min_scaffold_identity,
                       pos_bits, true, &B, q_order, n_alive, slots));
  if (stats) {
    stats->n_swept = B.m;
    stats->n_chains = B.n_chains_all;
  }
  if (B.M == 0 || B.m == 0 || B.T.nc == 0) {  // nothing can be an anchor: everything is dropped
    SWG_HIP(ctx, hipMemsetAsync(status_out, 0, n, st));
    return SWG_OK;
  }
  const uint64_t nc = B.T.nc, M = B.M, m = B.m;
  const bool rescue_on = !cfg->scaffolds_only && cfg->scaffold_max_deviation != 0;
  uint8_t* C_kept = swg_alloc<uint8_t>(ctx, nc);
  uint32_t* C_num = swg_alloc<uint32_t>(ctx, nc);
  uint32_t* anchor_num = chain_out;
  uint8_t* a_state = rescue_on ? swg_alloc<uint8_t>(ctx, M) : nullptr;  // by A position: anchors, never-rescued members (the rescue's inputs)
  uint8_t* aflag = rescue_on ? swg_alloc<uint8_t>(ctx, M) : nullptr;    // a_state == 1, for the compaction of the anchors
  unsigned long long* d_cnt = swg_alloc<unsigned long long>(ctx, 2);
  SWG_CHECK_ARENA(ctx);
  uint64_t n_kept = 0;
  SWG_TRY(scaffold_sweep_and_number(ctx, B.T, r->n_seq, r->seq_genome_two, r->n_genome_two, cfg->scaffold_filter_mode,
                                    cfg->scaffold_max_per_query, cfg->scaffold_max_per_target,
                                    cfg->scaffold_overlap_threshold, cfg->scoring_function, pos_bits, C_kept, C_num,
                                    &n_kept));
  if (stats) stats->n_chains_kept = n_kept;
  if (a_state) SWG_HIP(ctx, hipMemsetAsync(a_state, 0, M, st));
  uint32_t* head_num = swg_alloc<uint32_t>(ctx, m);  // valid at the heads of passing chains
  SWG_CHECK_ARENA(ctx);
  SWG_LAUNCH(ctx, "head_numbers", head_numbers_kernel<<<nblk(nc), EW, 0, st>>>(nc, B.m_head_of_chain, C_num, head_num));
  SWG_KERNEL_CHECK(ctx);
  SWG_LAUNCH(ctx, "member_marks", member_marks_kernel<<<nblk(m), EW, 0, st>>>(m, B.s_idx, B.m_hd, B.m_ok_head, head_num, anchor_num, B.s_a, a_state));
  SWG_KERNEL_CHECK(ctx);

  auto finish_counts = [&]() -> int {
    if (!stats) return SWG_OK;
    SWG_HIP(ctx, hipMemsetAsync(d_cnt, 0, 16, st));
    SWG_LAUNCH(ctx, "count_status", count_status_kernel<<<ctx->num_cu * 4, EW, 0, st>>>(n, status_out, d_cnt));
    SWG_KERNEL_CHECK(ctx);
    uint64_t c = 0;
    SWG_TRY(swg_read_scalars(ctx, reinterpret_cast<uint64_t*>(d_cnt), &c, 1));
    stats->n_out = c;
    return SWG_OK;
  };

  const int out_aligned = ((reinterpret_cast<uintptr_t>(status_out) | reinterpret_cast<uintptr_t>(chain_out)) & 15) == 0;
  auto anchor_status = [&]() -> int {
    SWG_LAUNCH(ctx, "anchor_status", anchor_status_kernel<<<nblk((n + 15) / 16), EW, 0, st>>>(n, chain_out, status_out, out_aligned));
    SWG_KERNEL_CHECK(ctx);
    return SWG_OK;
  };
  if (cfg->scaffolds_only) {  // paf_filter.rs:486-513
    SWG_TRY(anchor_status());
    return finish_counts();
  }
  if (n_kept == 0) {  // no anchors anywhere: every pair is skipped (paf_filter.rs:658-660)
    SWG_HIP(ctx, hipMemsetAsync(status_out, 0, n, st));
    return finish_counts();
  }

  // the stage's two floating-point tests as integer thresholds (one thread, ~130 evaluations)
  uint64_t* fp_thr = swg_alloc<uint64_t>(ctx, 2);
  SWG_CHECK_ARENA(ctx);
  SWG_LAUNCH(ctx, "fp_thresholds", fp_thresholds_kernel<<<1, 64, 0, st>>>(cfg->scaffold_gap, cfg->scaffold_max_deviation, fp_thr));
  SWG_KERNEL_CHECK(ctx);
  // ---- inversion capture
  {
    const uint64_t np = B.n_pairs;
    uint8_t* fflag = swg_alloc<uint8_t>(ctx, nc);
    uint32_t* pair_lo = swg_alloc<uint32_t>(ctx, np + 1);
    uint32_t* pair_hi = swg_alloc<uint32_t>(ctx, np + 1);
    uint64_t* d_tot = swg_alloc<uint64_t>(ctx, 1);
    SWG_CHECK_ARENA(ctx);
    SWG_LAUNCH(ctx, "fwd_flag", fwd_flag_kernel<<<nblk(nc), EW, 0, st>>>(nc, C_num, B.C_strand, fflag));
    SWG_KERNEL_CHECK(ctx);
    swg_flag_scan fwd_scan;
    SWG_TRY(swg_flags_count(ctx, fflag, nc, &fwd_scan, d_tot));
    uint64_t nf = 0;
    SWG_TRY(swg_read_scalars(ctx, d_tot, &nf, 1));
    if (nf) {
      uint32_t* f_c = swg_alloc<uint32_t>(ctx, nf);
      uint32_t* f_qs = swg_alloc<uint32_t>(ctx, nf);
      uint32_t* f_qe = swg_alloc<uint32_t>(ctx, nf);
      uint32_t* f_pm = swg_alloc<uint32_t>(ctx, nf);
      uint32_t* f_ts = swg_alloc<uint32_t>(ctx, nf);
      uint32_t* f_num = swg_alloc<uint32_t>(ctx, nf);
      uint32_t* f_dp = swg_alloc<uint32_t>(ctx, nf);
      SWG_CHECK_ARENA(ctx);
      SWG_HIP(ctx, hipMemsetAsync(pair_lo, 0, (np + 1) * 4, st));
      SWG_HIP(ctx, hipMemsetAsync(pair_hi, 0, (np + 1) * 4, st));
      SWG_TRY(swg_flags_compact(ctx, fwd_scan, f_c));
      SWG_LAUNCH(ctx, "fwd_cols", fwd_cols_kernel<<<nblk(nf), EW, 0, st>>>(nf, f_c, B.T.qs, B.T.qe, B.T.ts, C_num, B.C_dpair, f_qs, f_qe,
                                                                f_ts, f_num, f_dp));
      SWG_KERNEL_CHECK(ctx);
      SWG_LAUNCH(ctx, "fwd_ranges", fwd_ranges_kernel<<<nblk(nf), EW, 0, st>>>(nf, f_dp, pair_lo, pair_hi));
      SWG_KERNEL_CHECK(ctx);
      {
        if (nf >= 4096 * np) {
          // few pairs with very many chains (one deep chromosome pair): a wavefront per pair would walk millions of
          // chains alone, so the running maximum becomes one u64 max-scan with the pair's first slot in the high word
          swg_arena_mark mk = swg_arena_save(ctx);
          uint64_t* comp = swg_alloc<uint64_t>(ctx, nf);
          SWG_CHECK_ARENA(ctx);
          SWG_LAUNCH(ctx, "fwd_compose", fwd_compose_kernel<<<nblk(nf), EW, 0, st>>>(nf, f_dp, pair_lo, f_qe, comp));
          SWG_KERNEL_CHECK(ctx);
          SWG_TRY(swg_inclusive_max_scan_u64(ctx, comp, comp, nf));
          SWG_LAUNCH(ctx, "fwd_extract", fwd_extract_kernel<<<nblk(nf), EW, 0, st>>>(nf, comp, f_pm));
          SWG_KERNEL_CHECK(ctx);
          swg_arena_restore(ctx, mk);
        } else {
          uint64_t blocks = (np + 3) / 4;
          const uint64_t max_blocks = (uint64_t)ctx->num_cu * 16;
          if (blocks > max_blocks) blocks = max_blocks;
          SWG_LAUNCH(ctx, "fwd_prefmax", fwd_prefmax_kernel<<<(unsigned)blocks, EW, 0, st>>>((uint32_t)np, pair_lo, pair_hi, f_qe, f_pm));
          SWG_KERNEL_CHECK(ctx);
        }
      }
      SWG_LAUNCH(ctx, "inversion", inversion_kernel<<<nblk(M), EW, 0, st>>>(M, B.keyA, B.idxA, B.a_qe, B.a_ts, B.a_te, B.a_dpair, pos_bits,
                                                                pair_lo, pair_hi, f_qs, f_qe, f_pm, f_ts, f_num,
                                                                cfg->scaffold_gap, fp_thr, anchor_num, a_state));
      SWG_KERNEL_CHECK(ctx);
    }
  }
  // ---- rescue
  // anchors: status + chain in input order (coalesced); everything else starts as dropped
  SWG_TRY(anchor_status());
  if (cfg->scaffold_max_deviation != 0) {  // with a rescue distance of 0 only anchors are kept (paf_filter.rs:680, 740): no anchor sort
    uint64_t* d_tot = swg_alloc<uint64_t>(ctx, 1);
    SWG_CHECK_ARENA(ctx);
    SWG_LAUNCH(ctx, "anchor_flag", anchor_flag_kernel<<<nblk((M + 3) / 4), EW, 0, st>>>(M, a_state, aflag));
    SWG_KERNEL_CHECK(ctx);
    swg_flag_scan anchor_scan;
    SWG_TRY(swg_flags_count(ctx, aflag, M, &anchor_scan, d_tot));
    uint64_t na = 0;
    SWG_TRY(swg_read_scalars(ctx, d_tot, &na, 1));
    if (na) {
      uint32_t* anchor_a = swg_alloc<uint32_t>(ctx, na + 1);
      uint32_t* anchor_tmp = swg_alloc<uint32_t>(ctx, na + 1);
      uint64_t* b_key = swg_alloc<uint64_t>(ctx, na + 1);
      uint64_t* b_key_tmp = swg_alloc<uint64_t>(ctx, na + 1);
      uint32_t* b_tc = swg_alloc<uint32_t>(ctx, na + 1);
      uint32_t* b_idx = swg_alloc<uint32_t>(ctx, na + 1);
      uint32_t* b_num = swg_alloc<uint32_t>(ctx, na + 1);
      uint32_t* a_pair_lo = swg_alloc<uint32_t>(ctx, B.n_pairs + 1);
      uint32_t* a_pair_hi = swg_alloc<uint32_t>(ctx, B.n_pairs + 1);
      SWG_CHECK_ARENA(ctx);
      SWG_TRY(swg_flags_compact(ctx, anchor_scan, anchor_a));
      SWG_LAUNCH(ctx, "anchor_keys", anchor_keys_kernel<<<nblk(na), EW, 0, st>>>(na, anchor_a, B.keyA, B.a_qe, B.a_dpair, pos_bits, b_key));
      SWG_KERNEL_CHECK(ctx);
      const int dp_bits = swg_bits_for(B.n_pairs) ? swg_bits_for(B.n_pairs) : 1;
      if (dp_bits + pos_bits > 64) return swg_set_error(ctx, SWG_ERR_RANGE, "anchor sort key exceeds 64 bits");
      // sort B: 8-byte packed passes (one pass fewer, two thirds of the bytes per pass) when the word has the room
      const int val_bits = swg_bits_for(M - 1) ? swg_bits_for(M - 1) : 1;
      uint64_t* packedB = nullptr;
      int prc = SWG_ERR_UNSUPPORTED;
      if (pos_bits >= 8) prc = swg_radix_sort_packed(ctx, b_key, anchor_a, b_key_tmp, na, dp_bits + pos_bits, val_bits, nullptr, &packedB);
      if (prc == SWG_OK) {
        uint64_t* b_key_out = packedB == b_key ? b_key_tmp : b_key;  // the buffer the words are not in
        SWG_LAUNCH(ctx, "anchor_cols_packed", anchor_cols_packed_kernel<<<nblk(na), EW, 0, st>>>(na, packedB, val_bits, B.keyA, B.idxA, B.a_qe, B.a_ts,
                                                                                   B.a_te, anchor_num, pos_bits, b_key_out, b_tc, b_idx, b_num));
        SWG_KERNEL_CHECK(ctx);
        b_key = b_key_out;
      } else if (prc != SWG_ERR_UNSUPPORTED) {
        return prc;
      } else {
        SWG_TRY(swg_radix_sort_pairs(ctx, &b_key, &anchor_a, &b_key_tmp, &anchor_tmp, na, 0, dp_bits + pos_bits));
        SWG_LAUNCH(ctx, "anchor_cols", anchor_cols_kernel<<<nblk(na), EW, 0, st>>>(na, anchor_a, B.idxA, B.a_ts, B.a_te, anchor_num, b_tc, b_idx,
                                                                       b_num));
        SWG_KERNEL_CHECK(ctx);
      }
      SWG_HIP(ctx, hipMemsetAsync(a_pair_lo, 0, (B.n_pairs + 1) * sizeof(uint32_t), st));
      SWG_HIP(ctx, hipMemsetAsync(a_pair_hi, 0, (B.n_pairs + 1) * sizeof(uint32_t), st));
      SWG_LAUNCH(ctx, "anchor_ranges", anchor_ranges_kernel<<<nblk(na), EW, 0, st>>>(na, b_key, pos_bits, a_pair_lo, a_pair_hi));
      SWG_KERNEL_CHECK(ctx);
      SWG_LAUNCH(ctx, "rescue", rescue_kernel<<<nblk(M), EW, 0, st>>>(M, B.keyA, B.idxA, B.a_qe, B.a_ts, B.a_te, B.a_dpair, pos_bits, a_state,
                                                          a_pair_lo, a_pair_hi, b_key, b_tc, b_idx, b_num,
                                                          cfg->scaffold_max_deviation, fp_thr, status_out, chain_out));
      SWG_KERNEL_CHECK(ctx);
    }
  }
  return finish_counts();
}

extern "C" int swg_merge_chains(swg_ctx* ctx, const swg_records* rec, uint64_t max_gap, uint32_t* chain_of,
                                uint32_t* c_q_start, uint32_t* c_q_end, uint32_t* c_t_start, uint32_t* c_t_end,
                                double* c_weighted_identity, uint64_t* n_chains) {
  if (!ctx) return SWG_ERR_INVALID;
  if (!rec || !n_chains) return swg_set_error(ctx, SWG_ERR_INVALID, "NULL argument");
  *n_chains = 0;
  const uint64_t n = rec->n;
  if (n == 0) return SWG_OK;
  if (!chain_of || !c_q_start || !c_q_end || !c_t_start || !c_t_end || !c_weighted_identity)
    return swg_set_error(ctx, SWG_ERR_INVALID, "NULL output");
  if (n >= (uint64_t(1) << 31)) return swg_set_error(ctx, SWG_ERR_RANGE, "too many records");
  SWG_HIP(ctx, hipSetDevice(ctx->device));
  hipStream_t st = ctx->stream;
  if (ctx->arena_cap == 0) SWG_TRY(swg_arena_reserve(ctx, (size_t)n * 400 + (size_t(16) << 20)));
  uint32_t mx = 0;
  for (uint64_t i = 0; i < n; ++i) {
    const uint32_t v[4] = {rec->q_start[i], rec->q_end[i], rec->t_start[i], rec->t_end[i]};
    for (uint32_t x : v)
      if (x > mx) mx = x;
  }
  const int pos_bits = swg_bits_for(mx) ? swg_bits_for(mx) : 1;
  return swg_run_with_arena(ctx, [&]() -> int {
    swg_records d = *rec;
    auto up32 = [&](const uint32_t* src, uint64_t cnt) -> uint32_t* {
      uint32_t* p = swg_alloc<uint32_t>(ctx, cnt);
      if (p) (void)hipMemcpyAsync(p, src, cnt * 4, hipMemcpyHostToDevice, st);
      return p;
    };
    d.q_id = up32(rec->q_id, n);
    d.t_id = up32(rec->t_id, n);
    d.q_start = up32(rec->q_start, n);
    d.q_end = up32(rec->q_end, n);
    d.t_start = up32(rec->t_start, n);
    d.t_end = up32(rec->t_end, n);
    d.matches = up32(rec->matches, n);
    d.block_len = up32(rec->block_len, n);
    d.seq_genome_last = up32(rec->seq_genome_last, rec->n_seq);
    d.seq_genome_two = up32(rec->seq_genome_two, rec->n_seq);
    uint8_t* d_strand = swg_alloc<uint8_t>(ctx, n);
    uint8_t* ones = swg_alloc<uint8_t>(ctx, n);
    SWG_CHECK_ARENA(ctx);
    SWG_HIP(ctx, hipMemcpyAsync(d_strand, rec->strand, n, hipMemcpyHostToDevice, st));
    d.strand = d_strand;
    d.identity = nullptr;
    SWG_HIP(ctx, hipMemsetAsync(ones, 1, n, st));
    ChainBuild B;
    SWG_TRY(build_chains(ctx, &d, ones, ones, max_gap, 0, 0.0, pos_bits, false, &B));
    const uint64_t nc = B.T.nc;
    // chain_of[original index] via the survivor list (every record is a survivor here)
    std::vector<uint32_t> s_idx(B.m), s_chain(B.m);
    SWG_HIP(ctx, hipMemcpyAsync(s_idx.data(), B.s_idx, B.m * 4, hipMemcpyDeviceToHost, st));
    SWG_HIP(ctx, hipMemcpyAsync(s_chain.data(), B.s_chain, B.m * 4, hipMemcpyDeviceToHost, st));
    SWG_HIP(ctx, hipMemcpyAsync(c_q_start, B.T.qs, nc * 4, hipMemcpyDeviceToHost, st));
    SWG_HIP(ctx, hipMemcpyAsync(c_q_end, B.T.qe, nc * 4, hipMemcpyDeviceToHost, st));
    SWG_HIP(ctx, hipMemcpyAsync(c_t_start, B.T.ts, nc * 4, hipMemcpyDeviceToHost, st));
    SWG_HIP(ctx, hipMemcpyAsync(c_t_end, B.T.te, nc * 4, hipMemcpyDeviceToHost, st));
    SWG_HIP(ctx, hipMemcpyAsync(c_weighted_identity, B.T.wid, nc * 8, hipMemcpyDeviceToHost, st));
    SWG_HIP(ctx, hipStreamSynchronize(st));
    for (uint64_t p = 0; p < B.m; ++p) chain_of[s_idx[p]] = s_chain[p];
    *n_chains = nc;
    return SWG_OK;
  });
}
