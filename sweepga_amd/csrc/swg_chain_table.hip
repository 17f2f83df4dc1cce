// Scaffold stage, part 2 (src/union_find.rs, src/paf_filter.rs:854-933, 449-455): chains from the predecessor forest.
//
//   labelling   chains = paths of the predecessor forest; heads by pointer jumping -- the role of
//               union_find.rs (every union joins a path's tail, so the union-find root is the path head)
//   aggregates  bounding box, sum of matches / block lengths per chain (atomics keyed by the head)
//   ordering    chains are put in the reference's `all_chains` order: groups by first appearance in the
//               plane-swept metadata order (genome-pair-major, paf_filter.rs:1037-1046, 761-770), then
//               by head position
//   filter      span / identity (paf_filter.rs:449-455), weighted identity with glibc-exact ln
#include "swg_scaffold_internal.h"

namespace swg_scaf {
namespace {

__global__ __launch_bounds__(EW) void seg_compose_kernel(uint64_t m, const uint32_t* __restrict__ s_gidx,
                                                         const uint32_t* __restrict__ v, int complement,
                                                         uint64_t* __restrict__ out) {
  uint64_t p = (uint64_t)blockIdx.x * EW + threadIdx.x;
  if (p < m) out[p] = ((uint64_t)s_gidx[p] << 32) | (complement ? 0xffffffffu - v[p] : v[p]);
}
__global__ __launch_bounds__(EW) void group_first_from_scan_kernel(uint64_t m, const uint32_t* __restrict__ head_flag,
                                                                   const uint64_t* __restrict__ run_max,
                                                                   uint32_t* __restrict__ group_first) {
  uint64_t p = (uint64_t)blockIdx.x * EW + threadIdx.x;
  if (p >= m) return;
  if (p + 1 == m || head_flag[p + 1]) {  // last member of its group
    const uint64_t v = run_max[p];
    group_first[(uint32_t)(v >> 32)] = 0xffffffffu - (uint32_t)v;
  }
}

// ---- chain labelling ---------------------------------------------------------------------------------------
// hd[p] = pred[p] (or p), then followed through LDS as far as the block's own 1024-element range goes: predecessors
// precede their successors and are usually close, so most elements reach their head here and the global pointer
// jumping below only has to connect chains across ranges.
constexpr int HEAD_SPAN = 1024;
__global__ __launch_bounds__(EW) void head_init_kernel(uint64_t m, const uint32_t* __restrict__ pred,
                                                       uint32_t* __restrict__ hd, const uint8_t* __restrict__ only,
                                                       const uint8_t* __restrict__ span) {
  __shared__ uint32_t l[HEAD_SPAN];
  static_assert(HEAD_SPAN == (1 << BIG_SPAN_SHIFT), "span flags");
  const uint64_t n_span = (m + HEAD_SPAN - 1) / HEAD_SPAN;
  for (uint64_t sp = blockIdx.x; sp < n_span; sp += gridDim.x) {  // (block-uniform)
    if (span && !span[sp]) continue;  // no member of a long unit in this span: nothing to write
    const uint64_t base = sp * HEAD_SPAN;
    __syncthreads();
    for (int k = threadIdx.x; k < HEAD_SPAN; k += EW) {
      const uint64_t p = base + k;
      if (p < m) l[k] = pred[p] == NONE ? (uint32_t)p : pred[p];
    }
    __syncthreads();
    for (int k = threadIdx.x; k < HEAD_SPAN; k += EW) {
      const uint64_t p = base + k;
      if (p >= m) break;
      uint32_t h = l[k];
      while (h >= base) {  // h <= p < base + HEAD_SPAN
        const uint32_t hh = l[h - base];
        if (hh == h) break;
        h = hh;
      }
      if (!only || only[p]) hd[p] = h;
    }
  }
}
// hd[p] <- hd[hd[p]]; in-place races are benign (every value read is an ancestor of p)
__global__ __launch_bounds__(EW) void head_jump_kernel(uint64_t m, uint32_t* hd, uint32_t* __restrict__ changed,
                                                       const uint8_t* __restrict__ only, const uint8_t* __restrict__ span) {
  const uint64_t n_span = (m + HEAD_SPAN - 1) / HEAD_SPAN;
  bool any = false;
  for (uint64_t sp = blockIdx.x; sp < n_span; sp += gridDim.x) {
    if (span && !span[sp]) continue;
    for (int k = threadIdx.x; k < HEAD_SPAN; k += EW) {
      const uint64_t p = sp * HEAD_SPAN + k;
      if (p >= m) break;
      if (only && !only[p]) continue;
      const uint32_t h = hd[p];
      const uint32_t hh = hd[h];
      if (hh != h) {
        hd[p] = hh;
        any = true;
      }
    }
  }
  if (any) *changed = 1;
}

// Chain aggregates live at the head's slot.  Pass 1 seeds every slot with the element's own values (plain
// stores, this is also the initialisation); pass 2 folds the non-head members into their head with atomics.
// Most chains are singletons, so most elements never issue an atomic.
__global__ __launch_bounds__(EW) void chain_aggregate_init_kernel(uint64_t m, const uint32_t* __restrict__ hd,
                                                                  const uint32_t* __restrict__ s_qe,
                                                                  const uint32_t* __restrict__ s_ts,
                                                                  const uint32_t* __restrict__ s_te,
                                                                  const uint32_t* __restrict__ s_m,
                                                                  const uint32_t* __restrict__ s_b,
                                                                  uint32_t* __restrict__ h_qe, uint32_t* __restrict__ h_ts,
                                                                  uint32_t* __restrict__ h_te,
                                                                  unsigned long long* __restrict__ h_sm,
                                                                  unsigned long long* __restrict__ h_sb,
                                                                  uint32_t* __restrict__ is_head,
                                                                  const uint8_t* __restrict__ only,
                                                                  const uint8_t* __restrict__ span,
                                                                  unsigned long long* __restrict__ n_heads) {
  const uint64_t n_span = (m + HEAD_SPAN - 1) / HEAD_SPAN;
  uint32_t heads = 0;
  for (uint64_t sp = blockIdx.x; sp < n_span; sp += gridDim.x) {
    if (span && !span[sp]) continue;  // (is_head is only read where `only` is set)
    for (int k = threadIdx.x; k < HEAD_SPAN; k += EW) {
      const uint64_t p = sp * HEAD_SPAN + k;
      if (p >= m) break;
      if (only && !only[p]) {
        is_head[p] = 0;  // short units: labelled, aggregated and filtered by chain_label_kernel
        continue;
      }
      const bool head = hd[p] == p;
      heads += head ? 1u : 0u;
      is_head[p] = head ? 1u : 0u;
      h_qe[p] = s_qe[p];
      h_ts[p] = s_ts[p];
      h_te[p] = s_te[p];
      h_sm[p] = s_m[p];
      h_sb[p] = s_b[p];
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) heads += __shfl_down(heads, o, 64);
  if ((threadIdx.x & 63) == 0 && heads) atomicAdd(n_heads, (unsigned long long)heads);
}
// A member is usually a few positions after its head, so a block first folds the members whose head lies inside its
// own 1024-element range into LDS (LDS atomics), then merges each touched partial aggregate into the head's seeded
// global slot with one set of atomics per chain instead of one per member; members whose head lies before the range
// go to global memory directly.
constexpr int AGG_SPAN = 1024;
__global__ __launch_bounds__(EW) void chain_aggregate_kernel(uint64_t m, const uint32_t* __restrict__ hd,
                                                             const uint32_t* __restrict__ s_qe,
                                                             const uint32_t* __restrict__ s_ts,
                                                             const uint32_t* __restrict__ s_te,
                                                             const uint32_t* __restrict__ s_m,
                                                             const uint32_t* __restrict__ s_b,
                                                             uint32_t* __restrict__ h_qe, uint32_t* __restrict__ h_ts,
                                                             uint32_t* __restrict__ h_te,
                                                             unsigned long long* __restrict__ h_sm,
                                                             unsigned long long* __restrict__ h_sb,
                                                             const uint8_t* __restrict__ only, const uint8_t* __restrict__ span) {
  __shared__ uint32_t l_qe[AGG_SPAN], l_ts[AGG_SPAN], l_te[AGG_SPAN], l_cnt[AGG_SPAN];
  __shared__ unsigned long long l_sm[AGG_SPAN], l_sb[AGG_SPAN];
  static_assert(AGG_SPAN == (1 << BIG_SPAN_SHIFT), "span flags");
  const uint64_t n_span = (m + AGG_SPAN - 1) / AGG_SPAN;
  for (uint64_t sp = blockIdx.x; sp < n_span; sp += gridDim.x) {  // (block-uniform)
    if (span && !span[sp]) continue;
    const uint64_t base = sp * AGG_SPAN;
    __syncthreads();
    for (int k = threadIdx.x; k < AGG_SPAN; k += EW) {
      l_qe[k] = 0;
      l_ts[k] = 0xffffffffu;
      l_te[k] = 0;
      l_cnt[k] = 0;
      l_sm[k] = 0;
      l_sb[k] = 0;
    }
    __syncthreads();
    for (int k = threadIdx.x; k < AGG_SPAN; k += EW) {
      const uint64_t p = base + k;
      if (p >= m) break;
      if (only && !only[p]) continue;
      const uint32_t h = hd[p];
      if (h == p) continue;
      if (h >= base) {  // heads precede their members, so h < p < base + AGG_SPAN
        const uint32_t l = (uint32_t)(h - base);
        atomicMax(&l_qe[l], s_qe[p]);
        atomicMin(&l_ts[l], s_ts[p]);
        atomicMax(&l_te[l], s_te[p]);
        atomicAdd(&l_sm[l], (unsigned long long)s_m[p]);
        atomicAdd(&l_sb[l], (unsigned long long)s_b[p]);
        l_cnt[l] = 1;
      } else {
        atomicMax(&h_qe[h], s_qe[p]);
        atomicMin(&h_ts[h], s_ts[p]);
        atomicMax(&h_te[h], s_te[p]);
        atomicAdd(&h_sm[h], (unsigned long long)s_m[p]);
        atomicAdd(&h_sb[h], (unsigned long long)s_b[p]);
      }
    }
    __syncthreads();
    for (int k = threadIdx.x; k < AGG_SPAN; k += EW) {
      if (!l_cnt[k]) continue;
      const uint64_t h = base + k;  // a head of this range with members in it; members of later ranges use atomics too
      atomicMax(&h_qe[h], l_qe[k]);
      atomicMin(&h_ts[h], l_ts[k]);
      atomicMax(&h_te[h], l_te[k]);
      atomicAdd(&h_sm[h], l_sm[k]);
      atomicAdd(&h_sb[h], l_sb[k]);
    }
  }
}

// min original index per (q,t,strand) group, and per genome pair (prefix-last) over ALL alive records
__global__ __launch_bounds__(EW) void group_first_kernel(uint32_t n_groups, const uint32_t* __restrict__ group_begin,
                                                         uint32_t m, const uint32_t* __restrict__ s_idx,
                                                         uint32_t* __restrict__ group_first) {
  // one wavefront per (q,t,strand) group: members are contiguous in survivor order
  const int lane = threadIdx.x & 63;
  const uint32_t wave_global = blockIdx.x * (EW / 64) + (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));  // wave-uniform, and visibly so
  const uint32_t n_waves = (gridDim.x * EW) >> 6;
  for (uint32_t g = wave_global; g < n_groups; g += n_waves) {
    const uint32_t b = group_begin[g];
    const uint32_t e = (g + 1 < n_groups) ? group_begin[g + 1] : m;
    uint32_t v = 0xffffffffu;
    for (uint32_t p = b + lane; p < e; p += 64) {
      const uint32_t x = s_idx[p];
      if (x < v) v = x;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const uint32_t t = __shfl_xor(v, o, 64);
      if (t < v) v = t;
    }
    if (lane == 0) group_first[g] = v;
  }
}

// First (lowest) original index of every genome pair over the alive records, in ORIGINAL order (coalesced
// reads).  Two filters keep the atomics rare: (1) a wavefront whose 256 records all belong to one pair (inputs
// grouped by pair) reduces to one atomic; (2) otherwise (interleaved pairs) a plain cached read of the table --
// it only ever decreases, so a stale value is merely conservative -- drops every record that cannot lower it.
__global__ __launch_bounds__(EW) void genome_pair_first_kernel(uint64_t n, const uint8_t* __restrict__ alive,
                                                               const uint32_t* __restrict__ q_id,
                                                               const uint32_t* __restrict__ t_id,
                                                               const uint32_t* __restrict__ seq_genome,
                                                               PairTable table) {
  constexpr int U = 4;
  const int lane = threadIdx.x & 63;
  const uint64_t stride = (uint64_t)gridDim.x * EW * U;
  // whole waves stay in the loop together (the bound is wave-uniform), so the cross-lane ops are safe
  for (uint64_t w0 = ((uint64_t)blockIdx.x * EW + (threadIdx.x & ~63)) * U; w0 < n; w0 += stride) {
    const uint64_t i0 = w0 + (uint64_t)lane * U;
    unsigned long long L[U];  // (gq << 32) | gt
    bool live[U];
    uint32_t first_i = 0xffffffffu;
    unsigned long long first_L = 0;
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const uint64_t i = i0 + u;
      live[u] = i < n && alive[i] != 0;
      L[u] = live[u] ? ((unsigned long long)seq_genome[q_id[i]] << 32) | seq_genome[t_id[i]] : 0ull;
      if (live[u] && first_i == 0xffffffffu) {
        first_i = (uint32_t)i;
        first_L = L[u];
      }
    }
    // wave minimum of first_i, and the pair of the lane holding it
    uint32_t vmin = first_i;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const uint32_t x = __shfl_xor(vmin, o, 64);
      if (x < vmin) vmin = x;
    }
    if (vmin == 0xffffffffu) continue;  // no live record in this wave's span
    const uint64_t holder = __ballot(first_i == vmin);
    const unsigned long long L0 = __shfl(first_L, __builtin_ctzll(holder), 64);
    bool same = true;
#pragma unroll
    for (int u = 0; u < U; ++u) same = same && (!live[u] || L[u] == L0);
    if (__all(same)) {
      if (lane == 0) {
        uint32_t* slot = pair_slot(table, (uint32_t)(L0 >> 32), (uint32_t)L0);
        if (*slot > vmin) atomicMin(slot, vmin);
      }
    } else {
#pragma unroll
      for (int u = 0; u < U; ++u)
        if (live[u]) {
          uint32_t* slot = pair_slot(table, (uint32_t)(L[u] >> 32), (uint32_t)L[u]);
          if (*slot > (uint32_t)(i0 + u)) atomicMin(slot, (uint32_t)(i0 + u));
        }
    }
  }
}

// The same from the groups' own first records (valid when every alive record is a member of some group's chains, i.e. the
// mapping-level sweep removed nothing): one atomic per (query, target, strand) group instead of a pass over the records.
__global__ __launch_bounds__(EW) void genome_pair_first_groups_kernel(uint32_t n_groups, const uint32_t* __restrict__ group_begin,
                                                                      const uint32_t* __restrict__ s_idx,
                                                                      const uint32_t* __restrict__ group_first,
                                                                      const uint32_t* __restrict__ q_id,
                                                                      const uint32_t* __restrict__ t_id,
                                                                      const uint32_t* __restrict__ seq_genome, PairTable table) {
  uint32_t g = blockIdx.x * EW + threadIdx.x;
  if (g >= n_groups) return;
  const uint32_t i = s_idx[group_begin[g]];  // any record of the group names its sequences
  atomicMin(pair_slot(table, seq_genome[q_id[i]], seq_genome[t_id[i]]), group_first[g]);
}

// all_chains order = (q,t,strand) groups by first appearance, chains of a group by head position.  Chains in
// head-position order are already contiguous per group, so only the GROUPS are sorted; a chain's place is its
// group's base plus its rank inside the group.
// number of passing heads before member position p = lower bound of p in the ascending list of passing heads
__device__ __forceinline__ uint32_t heads_before(const uint32_t* __restrict__ ch_head, uint32_t nc, uint32_t p) {
  uint32_t l = 0, r = nc;
  while (l < r) {
    const uint32_t mid = l + ((r - l) >> 1);
    if (ch_head[mid] < p)
      l = mid + 1;
    else
      r = mid;
  }
  return l;
}
__global__ __launch_bounds__(EW) void group_keys_kernel(uint32_t n_groups, const uint32_t* __restrict__ group_begin,
                                                        uint32_t m, const uint32_t* __restrict__ ch_head, uint32_t nc,
                                                        const uint32_t* __restrict__ s_idx,
                                                        const uint32_t* __restrict__ group_first,
                                                        const uint32_t* __restrict__ q_id,
                                                        const uint32_t* __restrict__ t_id,
                                                        const uint32_t* __restrict__ seq_genome, bool pair_major,
                                                        PairTable gp_first, int idx_bits,
                                                        uint64_t* __restrict__ g_key, uint32_t* __restrict__ g_val,
                                                        uint32_t* __restrict__ g_first_chain,
                                                        uint32_t* __restrict__ g_nchains) {
  uint32_t g = blockIdx.x * EW + threadIdx.x;
  if (g >= n_groups) return;
  const uint32_t b = group_begin[g];
  const uint32_t first = heads_before(ch_head, nc, b);  // passing heads before the group = its first chain
  const uint32_t next = (g + 1 < n_groups) ? heads_before(ch_head, nc, group_begin[g + 1]) : nc;
  (void)m;
  g_first_chain[g] = first;
  g_nchains[g] = next - first;
  const uint32_t i = s_idx[b];
  // !pair_major: groups in plain first-appearance order of the records as given
  // (merge_mappings_into_chains called on its own); otherwise genome-pair-major, which is the order
  // apply_plane_sweep_to_mappings leaves the metadata in (paf_filter.rs:1037-1046, 1117-1120).
  uint64_t hi = 0;
  if (pair_major) hi = pair_get(gp_first, seq_genome[q_id[i]], seq_genome[t_id[i]]);
  g_key[g] = (hi << idx_bits) | group_first[g];
  g_val[g] = g;
}
__global__ __launch_bounds__(EW) void group_sizes_sorted_kernel(uint32_t n_groups, const uint32_t* __restrict__ g_sorted,
                                                                const uint32_t* __restrict__ g_nchains,
                                                                uint32_t* __restrict__ sizes) {
  uint32_t r = blockIdx.x * EW + threadIdx.x;
  if (r < n_groups) sizes[r] = g_nchains[g_sorted[r]];
}
__global__ __launch_bounds__(EW) void group_base_kernel(uint32_t n_groups, const uint32_t* __restrict__ g_sorted,
                                                        const uint32_t* __restrict__ base_sorted,
                                                        uint32_t* __restrict__ g_base) {
  uint32_t r = blockIdx.x * EW + threadIdx.x;
  if (r < n_groups) g_base[g_sorted[r]] = base_sorted[r];
}
// per passing chain: position-order ordinal c -> all_chains index (one thread per chain: the list of passing heads comes
// out of the compaction of the byte flags; round 3 ran over all members with a u32 flag column and its element-wise scan)
__global__ __launch_bounds__(EW) void chain_place_kernel(uint64_t nc, const uint32_t* __restrict__ ch_head,
                                                         const uint32_t* __restrict__ s_gidx,
                                                         const uint32_t* __restrict__ g_first_chain,
                                                         const uint32_t* __restrict__ g_base, uint32_t* __restrict__ order) {
  uint64_t c = (uint64_t)blockIdx.x * EW + threadIdx.x;
  if (c >= nc) return;
  const uint32_t g = s_gidx[ch_head[c]];
  order[g_base[g] + ((uint32_t)c - g_first_chain[g])] = (uint32_t)c;
}

// What the chain table needs of a chain that passes the filter, written ONCE by whoever decides the filter at the chain's head
// (chain_label_kernel for the chunks, chain_ok_kernel for the long units) into the head's slot of a sparse array: one 32-byte
// sector per passing chain.  chain_columns_kernel then reads one sector per chain (round 3: six scattered 4 / 8-byte columns
// at the head position, 250 bytes of HBM traffic per chain).
// (HeadRec and chain_weighted_identity: swg_scaffold_internal.h)
// Span / identity filter (paf_filter.rs:449-455) decided at the head, in position order: ok_head[p] = 1 iff p heads a chain
// that passes.  Most chains are short singletons that fail the span test; only the passing ones are placed, materialised and
// swept.  The number of all chains (a statistic) is counted on the way.
__global__ __launch_bounds__(EW) void chain_ok_kernel(uint64_t m, const uint32_t* __restrict__ is_head,
                                                      const uint32_t* __restrict__ s_qs, const uint32_t* __restrict__ h_qe,
                                                      const uint32_t* __restrict__ h_ts, const uint32_t* __restrict__ h_te,
                                                      const unsigned long long* __restrict__ h_sm,
                                                      const unsigned long long* __restrict__ h_sb,
                                                      const uint64_t* __restrict__ s_grp, uint64_t min_len,
                                                      double min_ident, uint8_t* __restrict__ ok_head,
                                                      HeadRec* __restrict__ rec, const uint8_t* __restrict__ only,
                                                      const uint8_t* __restrict__ span) {
  const uint64_t n_span = (m + HEAD_SPAN - 1) / HEAD_SPAN;
  for (uint64_t sp = blockIdx.x; sp < n_span; sp += gridDim.x) {
    if (span && !span[sp]) continue;
    for (int k = threadIdx.x; k < HEAD_SPAN; k += EW) {
      const uint64_t p = sp * HEAD_SPAN + k;
      if (p >= m) break;
      if (only && !only[p]) continue;  // short units: chain_label_kernel wrote their flags
      bool ok = false;
      if (is_head[p] != 0) {
        const uint64_t total_length = (uint64_t)h_qe[p] - (uint64_t)s_qs[p];  // q_max - q_min
        ok = total_length >= min_len;
        if (ok) {
          const double wid = chain_weighted_identity(total_length, h_sm[p], h_sb[p]);
          ok = wid >= min_ident;
          if (ok) {
            HeadRec hr;
            hr.qs = s_qs[p];
            hr.qe = h_qe[p];
            hr.ts = h_ts[p];
            hr.te = h_te[p];
            hr.wid = wid;
            hr.grp = s_grp[p];
            rec[p] = hr;
          }
        }
      }
      ok_head[p] = ok ? 1 : 0;
    }
  }
}

// Chains of the short units, chunk by chunk (one work-group per chunk of whole units, swg_chain.hip): a chain never leaves
// its unit, so everything about it is inside the chunk.  Chains are simple paths -- every element proposes to at most one
// successor and has at most one predecessor -- so with the successor links in LDS every HEAD walks its own chain and folds
// its members' values in registers: no seeding pass over all elements, no atomics, no pointer-jumping rounds, and the span /
// identity filter (paf_filter.rs:449-455) is decided right there; only passing heads write aggregates.
constexpr int LABEL_CAP = (int)(WALK_CHUNK + BIG_UNIT);  // a chunk holds fewer elements than this
// Round 4: chunks of up to LABEL_FAST elements (nearly all: a chunk is ~WALK_CHUNK elements plus the tail of its last unit)
// first bring every element's five values into LDS with coalesced loads, and the heads then walk their chains in LDS.  With
// the values in memory a head's step was five scattered loads issued for the few lanes still walking -- 33 vector-memory
// instructions per 64 elements, most of them nearly empty, and the texture path was what the kernel waited for (SQ counters).
// The head of every element goes through LDS as well and is written in element order.  Longer chunks take the old path.
constexpr int LABEL_FAST = 1536;
constexpr int LABEL_NT = 512;
static_assert(LABEL_FAST * (2 + 2 + 3 * 4) >= LABEL_CAP * 2, "the two layouts share one LDS buffer");
// NT threads per chunk; MB: the matches / block-length columns exist (a weighted identity is asked for).  Round 6: 512 threads
// with three elements each instead of 256 with six (114 registers, 36 KB of LDS: four work-groups of four wavefronts per CU,
// and the kernel waited for its loads 76 % of the time) and no LDS for the two columns nobody reads under the CLI defaults.
template <int NT, bool MB>
__global__ __launch_bounds__(NT) void chain_label_kernel(uint32_t n_chunks, const SpecBlock* __restrict__ chunks,
                                                         const uint32_t* __restrict__ pred,
                                                         const uint32_t* __restrict__ s_qs, const uint32_t* __restrict__ s_qe,
                                                         const uint32_t* __restrict__ s_ts, const uint32_t* __restrict__ s_te,
                                                         const uint32_t* __restrict__ s_m, const uint32_t* __restrict__ s_b,
                                                         const uint64_t* __restrict__ s_grp,
                                                         uint64_t min_len, double min_ident, uint32_t* __restrict__ hd,
                                                         uint8_t* __restrict__ ok_head, HeadRec* __restrict__ rec,
                                                         unsigned long long* __restrict__ n_heads,
                                                         const uint32_t* __restrict__ n_chunks_dev = nullptr) {
  // (the pair-resident path, swg_pair.hip: the chunk list's length lives on the device and s_grp is nullptr -- a HeadRec's
  // group is not read there)
  if (n_chunks_dev) n_chunks = min(n_chunks, *n_chunks_dev);
  __shared__ uint32_t lds[LABEL_FAST * (MB ? 6 : 4)];  // 36,864 / 24,576 bytes
  uint16_t* succ = reinterpret_cast<uint16_t*>(lds);            // [LABEL_CAP] (long chunks) / [LABEL_FAST]
  uint16_t* l_hd = succ + LABEL_FAST;                           // head of every element, relative to the chunk
  uint32_t* l_qe = lds + LABEL_FAST;                            // (behind the two 16-bit arrays)
  uint32_t* l_ts = l_qe + LABEL_FAST;
  uint32_t* l_te = l_ts + LABEL_FAST;
  uint32_t* l_m = MB ? l_te + LABEL_FAST : nullptr;
  uint32_t* l_b = MB ? l_m + LABEL_FAST : nullptr;
  if (!MB) s_m = s_b = nullptr;
  constexpr uint16_t NO = 0xffffu;
  uint32_t heads = 0;  // chains headed in this thread's elements (a statistic: all chains, passing the filter or not)
  for (uint32_t c = blockIdx.x; c < n_chunks; c += gridDim.x) {
    const uint32_t b = chunks[c].bb, e = chunks[c].be;
    uint32_t len = e > b ? e - b : 0;  // 0: a long unit's place holder
    if (len >= (uint32_t)LABEL_CAP) len = 0;  // (pair-resident path: a chunk too long for this LDS layout is pair_label_long's)
    __syncthreads();
    for (uint32_t k = threadIdx.x; k < len; k += NT) succ[k] = NO;
    __syncthreads();
    if (len <= (uint32_t)LABEL_FAST) {
      // every load of the chunk is requested before the first value is used (U x 7 per thread): a thread's elements one after
      // the other would be U memory round trips in a row, per phase
      constexpr int U = LABEL_FAST / NT;
      static_assert(LABEL_FAST % NT == 0, "whole rounds");
      uint32_t r_pr[U], r_qe[U], r_ts[U], r_te[U], r_m[U], r_b[U], r_qs[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const uint32_t k = (uint32_t)u * NT + threadIdx.x;
        const uint32_t p = b + (k < len ? k : 0u);  // (clamped: a read that is not used)
        r_pr[u] = pred[p];
        r_qe[u] = s_qe[p];
        r_ts[u] = s_ts[p];
        r_te[u] = s_te[p];
        r_m[u] = s_m ? s_m[p] : 0u;  // (nullptr: the weighted identity is not asked for -- the pair-resident path without an identity floor)
        r_b[u] = s_m ? s_b[p] : 0u;
        r_qs[u] = s_qs[p];
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const uint32_t k = (uint32_t)u * NT + threadIdx.x;
        if (k < len) {
          if (r_pr[u] != NONE) succ[r_pr[u] - b] = (uint16_t)k;  // one successor per element: no two writers
          l_hd[k] = r_pr[u] != NONE ? NO : (uint16_t)k;          // (a member's entry is written by its head below)
          l_qe[k] = r_qe[u];
          l_ts[k] = r_ts[u];
          l_te[k] = r_te[u];
          if constexpr (MB) {
            l_m[k] = r_m[u];
            l_b[k] = r_b[u];
          }
        }
      }
      __syncthreads();
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const uint32_t k = (uint32_t)u * NT + threadIdx.x;
        if (k >= len) continue;
        const uint32_t p = b + k;
        if (r_pr[u] != NONE) {  // a member: its head writes its label
          ok_head[p] = 0;
          continue;
        }
        ++heads;
        uint32_t qe = r_qe[u], ts = r_ts[u], te = r_te[u];
        uint64_t sm = r_m[u], sb = r_b[u];
        for (uint16_t nx = succ[k]; nx != NO; nx = succ[nx]) {
          const uint32_t a = l_qe[nx], t0 = l_ts[nx], t1 = l_te[nx];
          l_hd[nx] = (uint16_t)k;
          qe = a > qe ? a : qe;
          ts = t0 < ts ? t0 : ts;
          te = t1 > te ? t1 : te;
          if constexpr (MB) {
            sm += l_m[nx];
            sb += l_b[nx];
          }
        }
        const uint32_t qs0 = r_qs[u];
        const uint64_t total_length = (uint64_t)qe - (uint64_t)qs0;  // q_max - q_min (the head has the smallest q_start)
        bool ok = total_length >= min_len;
        if (ok) {
          const double wid = chain_weighted_identity(total_length, sm, sb);
          ok = wid >= min_ident;
          if (ok) {
            HeadRec hr;
            hr.qs = qs0;
            hr.qe = qe;
            hr.ts = ts;
            hr.te = te;
            hr.wid = wid;
            hr.grp = s_grp ? s_grp[p] : 0ull;
            rec[p] = hr;
          }
        }
        // (pair-resident path, s_grp == nullptr: bits 1 and 2 say whether the chain's query / target span is positive -- what
        // pair_finish's closed-form sweep asks of a passing chain, so that it ranks the chains without reading their records)
        ok_head[p] = ok ? (uint8_t)(s_grp ? 1u : 1u | (qs0 < qe ? 2u : 0u) | (ts < te ? 4u : 0u)) : (uint8_t)0;
      }
      __syncthreads();
      for (uint32_t k = threadIdx.x; k < len; k += NT) hd[b + k] = b + l_hd[k];
      continue;
    }
    // (longer chunks: the values stay in memory.  The loads of LB elements per thread are requested together -- an element
    // after the other is a memory round trip each, and a chunk of the pair-resident path is a whole unit of thousands of
    // elements in ONE work-group: 35 round trips per pass and thread on S-pan, 150 us per chunk, before this was batched)
    constexpr int LB = 2048 / NT;  // (as many loads in flight per work-group as before)
    for (uint32_t k0 = 0; k0 < len; k0 += NT * LB) {
      uint32_t pr[LB];
#pragma unroll
      for (int u = 0; u < LB; ++u) {
        const uint32_t k = k0 + (uint32_t)u * NT + threadIdx.x;
        pr[u] = pred[b + (k < len ? k : 0u)];
      }
#pragma unroll
      for (int u = 0; u < LB; ++u) {
        const uint32_t k = k0 + (uint32_t)u * NT + threadIdx.x;
        if (k < len && pr[u] != NONE) succ[pr[u] - b] = (uint16_t)k;  // one successor per element: no two writers
      }
    }
    __syncthreads();
    for (uint32_t k0 = 0; k0 < len; k0 += NT * LB) {
      uint32_t pr[LB], v_qs[LB], v_qe[LB], v_ts[LB], v_te[LB], v_m[LB], v_b[LB];
#pragma unroll
      for (int u = 0; u < LB; ++u) {
        const uint32_t k = k0 + (uint32_t)u * NT + threadIdx.x;
        const uint32_t p = b + (k < len ? k : 0u);
        pr[u] = pred[p];
        v_qs[u] = s_qs[p];
        v_qe[u] = s_qe[p];
        v_ts[u] = s_ts[p];
        v_te[u] = s_te[p];
        v_m[u] = s_m ? s_m[p] : 0u;
        v_b[u] = s_m ? s_b[p] : 0u;
      }
#pragma unroll
      for (int u = 0; u < LB; ++u) {
        const uint32_t k = k0 + (uint32_t)u * NT + threadIdx.x;
        if (k >= len) continue;
        const uint32_t p = b + k;
        if (pr[u] != NONE) {  // a member: its head writes its label
          ok_head[p] = 0;
          continue;
        }
        // a head: walk the chain
        ++heads;
        hd[p] = p;
        uint32_t qe = v_qe[u], ts = v_ts[u], te = v_te[u];
        uint64_t sm = v_m[u], sb = v_b[u];
        for (uint16_t nx = succ[k]; nx != NO;) {
          const uint32_t q = b + nx;
          const uint16_t nn = succ[nx];  // requested together with the member's values
          const uint32_t a = s_qe[q], t0 = s_ts[q], t1 = s_te[q], mm = s_m ? s_m[q] : 0u, bb = s_m ? s_b[q] : 0u;
          hd[q] = p;
          qe = a > qe ? a : qe;
          ts = t0 < ts ? t0 : ts;
          te = t1 > te ? t1 : te;
          sm += mm;
          sb += bb;
          nx = nn;
        }
        const uint32_t qs0 = v_qs[u];
        const uint64_t total_length = (uint64_t)qe - (uint64_t)qs0;  // q_max - q_min (the head has the smallest q_start)
        bool ok = total_length >= min_len;
        if (ok) {
          const double wid = chain_weighted_identity(total_length, sm, sb);
          ok = wid >= min_ident;
          if (ok) {
            HeadRec hr;
            hr.qs = qs0;
            hr.qe = qe;
            hr.ts = ts;
            hr.te = te;
            hr.wid = wid;
            hr.grp = s_grp ? s_grp[p] : 0ull;
            rec[p] = hr;
          }
        }
        // (pair-resident path, s_grp == nullptr: bits 1 and 2 say whether the chain's query / target span is positive -- what
        // pair_finish's closed-form sweep asks of a passing chain, so that it ranks the chains without reading their records)
        ok_head[p] = ok ? (uint8_t)(s_grp ? 1u : 1u | (qs0 < qe ? 2u : 0u) | (ts < te ? 4u : 0u)) : (uint8_t)0;
      }
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) heads += __shfl_down(heads, o, 64);
  // one atomic per work-group (a statistic on ONE address: per wavefront it was 65,536 serialised atomics per call, and what
  // made a finer grid -- better balance over chunks of uneven length -- slower instead of faster)
  __syncthreads();
  if ((threadIdx.x & 63) == 0) lds[threadIdx.x >> 6] = heads;
  __syncthreads();
  if (threadIdx.x == 0) {
    uint32_t all = 0;
    for (int w = 0; w < NT / 64; ++w) all += lds[w];
    if (all) atomicAdd(n_heads, (unsigned long long)all);
  }
}
// The same for ONE chunk of any length per work-group (pair-resident path, swg_pair.hip: the long units of dense chromosome
// pairs): heads by pointer jumping in memory, the aggregates accumulated by atomics in the head's own HeadRec slot (the box in
// its coordinate fields, the sums of matches / block lengths in the two 8-byte fields), the filter decided by the head.
__global__ __launch_bounds__(1024) void chain_label_long_kernel(uint32_t cap_long, const uint32_t* __restrict__ n_long_dev,
                                                                const uint32_t* __restrict__ long_list,
                                                                const SpecBlock* __restrict__ chunks, const uint32_t* __restrict__ pred,
                                                                const uint32_t* __restrict__ s_qs, const uint32_t* __restrict__ s_qe,
                                                                const uint32_t* __restrict__ s_ts, const uint32_t* __restrict__ s_te,
                                                                const uint32_t* __restrict__ s_m, const uint32_t* __restrict__ s_b,
                                                                uint64_t min_len, double min_ident, uint32_t* hd,
                                                                uint8_t* __restrict__ ok_head, HeadRec* rec,
                                                                unsigned long long* __restrict__ n_heads) {
  __shared__ uint32_t changed;
  const uint32_t n_long = *n_long_dev < cap_long ? *n_long_dev : cap_long;
  uint32_t heads = 0;
  for (uint32_t c = blockIdx.x; c < n_long; c += gridDim.x) {
    const SpecBlock D = chunks[long_list[c]];
    const uint32_t b = D.bb, e = D.be;
    for (uint32_t p = b + threadIdx.x; p < e; p += 1024) {
      const uint32_t pr = pred[p];
      hd[p] = pr == NONE ? p : pr;
      if (pr == NONE) {  // a head: its slot starts from its own values
        HeadRec hr;
        hr.qs = s_qs[p];
        hr.qe = s_qe[p];
        hr.ts = s_ts[p];
        hr.te = s_te[p];
        hr.wid = __longlong_as_double((long long)(unsigned long long)(s_m ? s_b[p] : 0u));  // sum of block lengths, as bits
        hr.grp = s_m ? s_m[p] : 0u;                                                         // sum of matches
        rec[p] = hr;
      }
    }
    for (;;) {
      __syncthreads();
      if (threadIdx.x == 0) changed = 0;
      __syncthreads();
      bool ch = false;
      for (uint32_t p = b + threadIdx.x; p < e; p += 1024) {
        const uint32_t h = __hip_atomic_load(&hd[p], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        const uint32_t hh = __hip_atomic_load(&hd[h], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (hh != h) {
          __hip_atomic_store(&hd[p], hh, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          ch = true;
        }
      }
      if (ch) changed = 1;
      __syncthreads();
      if (!changed) break;
    }
    for (uint32_t p = b + threadIdx.x; p < e; p += 1024) {
      const uint32_t h = hd[p];
      if (h == p) continue;
      ok_head[p] = 0;
      atomicMax(&rec[h].qe, s_qe[p]);
      atomicMin(&rec[h].ts, s_ts[p]);
      atomicMax(&rec[h].te, s_te[p]);
      if (s_m) {
        atomicAdd(reinterpret_cast<unsigned long long*>(&rec[h].wid), (unsigned long long)s_b[p]);
        atomicAdd(reinterpret_cast<unsigned long long*>(&rec[h].grp), (unsigned long long)s_m[p]);
      }
    }
    __syncthreads();
    for (uint32_t p = b + threadIdx.x; p < e; p += 1024) {
      if (hd[p] != p) continue;
      ++heads;
      const HeadRec hr = rec[p];
      const uint64_t sb = (uint64_t)__double_as_longlong(hr.wid), sm = hr.grp;
      const uint64_t total_length = (uint64_t)hr.qe - (uint64_t)hr.qs;  // q_max - q_min (the head has the smallest q_start)
      bool ok = total_length >= min_len;
      if (ok) {
        const double wid = chain_weighted_identity(total_length, sm, sb);
        ok = wid >= min_ident;
        if (ok) {
          rec[p].wid = wid;
          rec[p].grp = 0;
        }
      }
      ok_head[p] = ok ? (uint8_t)(1u | (hr.qs < hr.qe ? 2u : 0u) | (hr.ts < hr.te ? 4u : 0u)) : (uint8_t)0;  // (span bits: see chain_label_kernel)
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) heads += __shfl_down(heads, o, 64);
  if ((threadIdx.x & 63) == 0 && heads) atomicAdd(n_heads, (unsigned long long)heads);
}
// chain columns in all_chains order.  weighted identity: paf_filter.rs:896-913
__global__ __launch_bounds__(EW) void chain_columns_kernel(
    uint64_t nc, const uint32_t* __restrict__ order, const uint32_t* __restrict__ ch_head, const HeadRec* __restrict__ rec,
    const uint32_t* __restrict__ s_a,
    const uint32_t* __restrict__ a_dpair, uint32_t n_seq,
    uint32_t* __restrict__ C_qid, uint32_t* __restrict__ C_tid, uint32_t* __restrict__ C_qs,
    uint32_t* __restrict__ C_qe, uint32_t* __restrict__ C_ts, uint32_t* __restrict__ C_te,
    double* __restrict__ C_wid, uint8_t* __restrict__ C_strand, uint32_t* __restrict__ C_dpair,
    uint32_t* __restrict__ rank_of_poschain, uint32_t* __restrict__ head_of_chain) {
  uint64_t c2 = (uint64_t)blockIdx.x * EW + threadIdx.x;
  if (c2 >= nc) return;
  const uint32_t c = order[c2];
  rank_of_poschain[c] = (uint32_t)c2;
  const uint32_t p = ch_head[c];
  head_of_chain[c2] = p;
  const HeadRec hr = rec[p];
  const uint64_t g = hr.grp;
  const uint64_t pair = g >> 1;
  const uint32_t qs = hr.qs, qe = hr.qe, ts = hr.ts, te = hr.te;
  C_qid[c2] = (uint32_t)(pair / n_seq);
  C_tid[c2] = (uint32_t)(pair % n_seq);
  C_strand[c2] = (uint8_t)(g & 1);
  C_qs[c2] = qs;
  C_qe[c2] = qe;
  C_ts[c2] = ts;
  C_te[c2] = te;
  C_dpair[c2] = a_dpair[s_a ? s_a[p] : p];  // s_a == nullptr: every record of sort A is a member
  C_wid[c2] = hr.wid;
}

// index into T of every member's chain (NONE: its chain fails the span / identity filter); only the merge_chains seam asks
__global__ __launch_bounds__(EW) void survivor_chain_kernel(uint64_t m, const uint32_t* __restrict__ hd,
                                                            const uint8_t* __restrict__ ok_head,
                                                            const uint32_t* __restrict__ ch_head, uint32_t nc,
                                                            const uint32_t* __restrict__ rank_of_poschain,
                                                            uint32_t* __restrict__ s_chain) {
  uint64_t p = (uint64_t)blockIdx.x * EW + threadIdx.x;
  if (p >= m) return;
  const uint32_t h = hd[p];
  s_chain[p] = ok_head[h] ? rank_of_poschain[heads_before(ch_head, nc, h)] : NONE;
}

}  // namespace

int pair_table_make(swg_ctx* ctx, uint32_t n_genome, uint64_t bound, PairTable* t) {
  hipStream_t st = ctx->stream;
  t->n_genome = n_genome;
  t->dense = nullptr;
  t->keys = nullptr;
  t->vals = nullptr;
  t->mask = 0;
  const uint64_t g2 = (uint64_t)n_genome * n_genome;
  if (g2 <= DENSE_PAIR_LIMIT) {
    t->dense = swg_alloc<uint32_t>(ctx, g2);
    SWG_CHECK_ARENA(ctx);
    SWG_HIP(ctx, hipMemsetAsync(t->dense, 0xff, g2 * sizeof(uint32_t), st));  // NONE in every entry
    return SWG_OK;
  }
  uint64_t cap = 1024;
  while (cap < 2 * bound) cap <<= 1;
  if (cap > (uint64_t(1) << 32)) return swg_set_error(ctx, SWG_ERR_RANGE, "genome-pair table beyond 2^32 slots");
  t->keys = swg_alloc<unsigned long long>(ctx, cap);
  t->vals = swg_alloc<uint32_t>(ctx, cap);
  SWG_CHECK_ARENA(ctx);
  t->mask = (uint32_t)(cap - 1);
  SWG_LAUNCH(ctx, "fill", fill_u64_kernel<<<nblk(cap), EW, 0, st>>>(cap, reinterpret_cast<uint64_t*>(t->keys), ~0ull));
  SWG_KERNEL_CHECK(ctx);
  SWG_LAUNCH(ctx, "fill", fill_u32_kernel<<<nblk(cap), EW, 0, st>>>(cap, t->vals, NONE));
  SWG_KERNEL_CHECK(ctx);
  return SWG_OK;
}

int chain_table_build(swg_ctx* ctx, const swg_records* r, const uint8_t* alive, uint64_t min_len, double min_ident,
                      bool genome_pair_major, ChainBuild* out, const ChainWork& W) {
  const uint64_t n = r->n;
  hipStream_t st = ctx->stream;
  ChainBuild& B = *out;
  const uint64_t m = B.m, n_groups = W.n_groups;
  uint32_t *s_qs = W.s_qs, *s_qe = W.s_qe, *s_ts = W.s_ts, *s_te = W.s_te, *s_m = W.s_m, *s_b = W.s_b;
  uint64_t* s_grp = W.s_grp;
  uint32_t *head_flag = W.head_flag, *s_gidx = W.s_gidx, *group_begin = W.group_begin, *pred = W.pred;
  uint64_t* d_tot = W.d_tot;
  uint32_t* hd = swg_alloc<uint32_t>(ctx, m);
  uint32_t* h_qe = swg_alloc<uint32_t>(ctx, m);
  uint32_t* h_ts = swg_alloc<uint32_t>(ctx, m);
  uint32_t* h_te = swg_alloc<uint32_t>(ctx, m);
  unsigned long long* h_sm = swg_alloc<unsigned long long>(ctx, m);
  unsigned long long* h_sb = swg_alloc<unsigned long long>(ctx, m);
  uint32_t* is_head = swg_alloc<uint32_t>(ctx, m);
  uint32_t* group_first = swg_alloc<uint32_t>(ctx, n_groups);
  PairTable gp_first;  // made where it is filled (below); its pairs are among the B.n_pairs (query, target, strand) groups
  uint32_t* changed = swg_alloc<uint32_t>(ctx, 2);
  SWG_CHECK_ARENA(ctx);
  HeadRec* head_rec = swg_alloc<HeadRec>(ctx, m);  // sparse: touched at the heads of passing chains only
  uint8_t* ok_head = swg_alloc<uint8_t>(ctx, m);
  SWG_CHECK_ARENA(ctx);
  SWG_HIP(ctx, hipMemsetAsync(d_tot + 1, 0, 8, st));
  unsigned long long* n_heads = reinterpret_cast<unsigned long long*>(d_tot + 1);  // all chains: counted where heads are found
  // ---- short units: labels, aggregates and the span / identity filter chunk by chunk
  const uint8_t* only = nullptr;  // what the generic path below is restricted to (nullptr: everything)
  const uint8_t* span = nullptr;  // ... and the 1024-element spans that hold any of it
  bool generic = true;
  if (W.n_chunks) {
    static const uint64_t lper = getenv("SWG_LABEL_BLOCKS") ? (uint64_t)atoi(getenv("SWG_LABEL_BLOCKS")) : 128;  // (work-groups per CU in the grid)
    const uint64_t lb = W.n_chunks < (uint64_t)ctx->num_cu * lper ? W.n_chunks : (uint64_t)ctx->num_cu * lper;
    if (s_m)
      SWG_LAUNCH(ctx, "chain_label", chain_label_kernel<LABEL_NT, true><<<(unsigned)lb, LABEL_NT, 0, st>>>((uint32_t)W.n_chunks, W.chunks, pred, s_qs, s_qe, s_ts, s_te,
                                                                        s_m, s_b, s_grp, min_len, min_ident, hd, ok_head, head_rec, n_heads));
    else
      SWG_LAUNCH(ctx, "chain_label", chain_label_kernel<LABEL_NT, false><<<(unsigned)lb, LABEL_NT, 0, st>>>((uint32_t)W.n_chunks, W.chunks, pred, s_qs, s_qe, s_ts, s_te,
                                                                        s_m, s_b, s_grp, min_len, min_ident, hd, ok_head, head_rec, n_heads));
    SWG_KERNEL_CHECK(ctx);
    only = W.big_member;
    span = W.span_big;
    generic = only != nullptr;  // long units exist
  }
  if (generic) {
    // ---- long units (or everything, without a chunk list): labelling by pointer jumping, aggregates by atomics at the head.
    // Work-groups stride over the 1024-element spans and skip those without a member of a long unit.
    const uint64_t n_span_all = (m + HEAD_SPAN - 1) / HEAD_SPAN;
    const unsigned span_grid = (unsigned)(n_span_all < (uint64_t)ctx->num_cu * 16 ? n_span_all : (uint64_t)ctx->num_cu * 16);
    SWG_LAUNCH(ctx, "head_init", head_init_kernel<<<span_grid, EW, 0, st>>>(m, pred, hd, only, span));
    SWG_KERNEL_CHECK(ctx);
    for (int round = 0; round < 64; ++round) {
      SWG_HIP(ctx, hipMemsetAsync(changed, 0, 8, st));
      SWG_LAUNCH(ctx, "head_jump", head_jump_kernel<<<span_grid, EW, 0, st>>>(m, hd, changed, only, span));
      SWG_KERNEL_CHECK(ctx);
      SWG_LAUNCH(ctx, "head_jump", head_jump_kernel<<<span_grid, EW, 0, st>>>(m, hd, changed, only, span));
      SWG_KERNEL_CHECK(ctx);
      uint64_t ch = 0;
      SWG_TRY(swg_read_scalars(ctx, reinterpret_cast<uint64_t*>(changed), &ch, 1));
      if ((uint32_t)ch == 0) break;
    }
    SWG_LAUNCH(ctx, "chain_aggregate_init", chain_aggregate_init_kernel<<<span_grid, EW, 0, st>>>(m, hd, s_qe, s_ts, s_te, s_m, s_b, h_qe, h_ts,
                                                                                    h_te, h_sm, h_sb, is_head, only, span, n_heads));
    SWG_KERNEL_CHECK(ctx);
    SWG_LAUNCH(ctx, "chain_aggregate", chain_aggregate_kernel<<<span_grid, EW, 0, st>>>(
                                           m, hd, s_qe, s_ts, s_te, s_m, s_b, h_qe, h_ts, h_te, h_sm, h_sb, only, span));
    SWG_KERNEL_CHECK(ctx);
    // span / identity filter at the heads; from here on "chain" means a chain that passes it
    SWG_LAUNCH(ctx, "chain_ok", chain_ok_kernel<<<span_grid, EW, 0, st>>>(
                                    m, is_head, s_qs, h_qe, h_ts, h_te, h_sm, h_sb, s_grp, min_len, min_ident, ok_head, head_rec, only, span));
    SWG_KERNEL_CHECK(ctx);
  }
  // the passing heads, in position order: byte flags -> per-tile counts (-> the list, below)
  swg_flag_scan ok_scan;
  SWG_TRY(swg_flags_count(ctx, ok_head, m, &ok_scan, d_tot));
  uint64_t h2[2];
  SWG_TRY(swg_read_scalars(ctx, d_tot, h2, 2));
  const uint64_t nc = h2[0];
  B.n_chains_all = h2[1];
  B.T.nc = nc;
  if (nc == 0) return SWG_OK;
  // ---- all_chains order
  SWG_LAUNCH(ctx, "fill", fill_u32_kernel<<<nblk(n_groups), EW, 0, st>>>(n_groups, group_first, NONE));
  SWG_KERNEL_CHECK(ctx);
  SWG_TRY(pair_table_make(ctx, r->n_genome_last, B.n_pairs, &gp_first));
  if (m / n_groups > 8192) {
    swg_arena_mark mk = swg_arena_save(ctx);
    uint64_t* comp = swg_alloc<uint64_t>(ctx, m);
    SWG_CHECK_ARENA(ctx);
    SWG_LAUNCH(ctx, "seg_compose", seg_compose_kernel<<<nblk(m), EW, 0, st>>>(m, s_gidx, B.s_idx, 1, comp));
    SWG_KERNEL_CHECK(ctx);
    SWG_TRY(swg_inclusive_max_scan_u64(ctx, comp, comp, m));
    SWG_LAUNCH(ctx, "group_first_from_scan", group_first_from_scan_kernel<<<nblk(m), EW, 0, st>>>(m, head_flag, comp, group_first));
    swg_arena_restore(ctx, mk);
  } else {
    uint64_t blocks = (n_groups + 3) / 4;
    const uint64_t max_blocks = (uint64_t)ctx->num_cu * 16;
    if (blocks > max_blocks) blocks = max_blocks;
    SWG_LAUNCH(ctx, "group_first", group_first_kernel<<<(unsigned)blocks, EW, 0, st>>>((uint32_t)n_groups, group_begin, (uint32_t)m,
                                                                               B.s_idx, group_first));
  }
  SWG_KERNEL_CHECK(ctx);
  if (B.m == B.M) {
    // members == alive records: a genome pair's first record is the smallest of its (query, target, strand) groups' firsts
    SWG_LAUNCH(ctx, "genome_pair_first_groups", genome_pair_first_groups_kernel<<<nblk(n_groups), EW, 0, st>>>(
                                             (uint32_t)n_groups, group_begin, B.s_idx, group_first, r->q_id, r->t_id,
                                             r->seq_genome_last, gp_first));
  } else {
    SWG_LAUNCH(ctx, "genome_pair_first", genome_pair_first_kernel<<<ctx->num_cu * 8, EW, 0, st>>>(n, alive, r->q_id, r->t_id,
                                                                                      r->seq_genome_last, gp_first));
  }
  SWG_KERNEL_CHECK(ctx);
  uint32_t* ch_head = swg_alloc<uint32_t>(ctx, nc);
  SWG_CHECK_ARENA(ctx);
  SWG_TRY(swg_flags_compact(ctx, ok_scan, ch_head));
  uint32_t* order = swg_alloc<uint32_t>(ctx, nc);
  uint32_t* rank_of = swg_alloc<uint32_t>(ctx, nc);
  uint32_t* head_of_chain = swg_alloc<uint32_t>(ctx, nc);
  uint64_t* g_key = swg_alloc<uint64_t>(ctx, n_groups);
  uint64_t* g_key_tmp = swg_alloc<uint64_t>(ctx, n_groups);
  uint32_t* g_sorted = swg_alloc<uint32_t>(ctx, n_groups);
  uint32_t* g_sorted_tmp = swg_alloc<uint32_t>(ctx, n_groups);
  uint32_t* g_first_chain = swg_alloc<uint32_t>(ctx, n_groups);
  uint32_t* g_nchains = swg_alloc<uint32_t>(ctx, n_groups);
  uint32_t* g_sizes = swg_alloc<uint32_t>(ctx, n_groups);
  uint32_t* g_base = swg_alloc<uint32_t>(ctx, n_groups);
  ChainTable& T = B.T;
  T.nc = nc;
  T.qid = swg_alloc<uint32_t>(ctx, nc);
  T.tid = swg_alloc<uint32_t>(ctx, nc);
  T.qs = swg_alloc<uint32_t>(ctx, nc);
  T.qe = swg_alloc<uint32_t>(ctx, nc);
  T.ts = swg_alloc<uint32_t>(ctx, nc);
  T.te = swg_alloc<uint32_t>(ctx, nc);
  T.wid = swg_alloc<double>(ctx, nc);
  B.C_strand = swg_alloc<uint8_t>(ctx, nc);
  B.C_dpair = swg_alloc<uint32_t>(ctx, nc);
  SWG_CHECK_ARENA(ctx);
  const int idx_bits = swg_bits_for(n) ? swg_bits_for(n) : 1;
  const unsigned gblk = nblk(n_groups);
  SWG_LAUNCH(ctx, "group_keys", group_keys_kernel<<<gblk, EW, 0, st>>>((uint32_t)n_groups, group_begin, (uint32_t)m, ch_head, (uint32_t)nc,
                                                           B.s_idx, group_first, r->q_id, r->t_id, r->seq_genome_last,
                                                           genome_pair_major, gp_first, idx_bits, g_key, g_sorted, g_first_chain,
                                                           g_nchains));
  SWG_KERNEL_CHECK(ctx);
  SWG_TRY(swg_radix_sort_pairs(ctx, &g_key, &g_sorted, &g_key_tmp, &g_sorted_tmp, n_groups, 0, 2 * idx_bits));
  SWG_LAUNCH(ctx, "group_sizes_sorted", group_sizes_sorted_kernel<<<gblk, EW, 0, st>>>((uint32_t)n_groups, g_sorted, g_nchains, g_sizes));
  SWG_KERNEL_CHECK(ctx);
  SWG_TRY(swg_exclusive_scan_u32(ctx, g_sizes, g_sizes, n_groups, nullptr));
  SWG_LAUNCH(ctx, "group_base", group_base_kernel<<<gblk, EW, 0, st>>>((uint32_t)n_groups, g_sorted, g_sizes, g_base));
  SWG_KERNEL_CHECK(ctx);
  SWG_LAUNCH(ctx, "chain_place", chain_place_kernel<<<nblk(nc), EW, 0, st>>>(nc, ch_head, s_gidx, g_first_chain, g_base, order));
  SWG_KERNEL_CHECK(ctx);
  SWG_LAUNCH(ctx, "chain_columns", chain_columns_kernel<<<nblk(nc), EW, 0, st>>>(
                                       nc, order, ch_head, head_rec, B.s_a, B.a_dpair,
                                       r->n_seq, T.qid, T.tid, T.qs, T.qe, T.ts, T.te, T.wid, B.C_strand, B.C_dpair, rank_of, head_of_chain));
  SWG_KERNEL_CHECK(ctx);
  B.m_head_of_chain = head_of_chain;
  B.m_hd = hd;
  B.m_ok_head = ok_head;
  B.m_rank_of = rank_of;
  if (B.want_s_chain) {
    SWG_LAUNCH(ctx, "survivor_chain", survivor_chain_kernel<<<nblk(m), EW, 0, st>>>(m, hd, ok_head, ch_head, (uint32_t)nc, rank_of, B.s_chain));
    SWG_KERNEL_CHECK(ctx);
  }
  return SWG_OK;
}

// chain_label_kernel over a chunk list made on the device (swg_pair.hip).
int pair_label_launch(swg_ctx* ctx, uint32_t cap_chunks, const uint32_t* n_chunks_dev, const SpecBlock* chunks, const uint32_t* pred,
                      const uint32_t* s_qs, const uint32_t* s_qe, const uint32_t* s_ts, const uint32_t* s_te, const uint32_t* s_m,
                      const uint32_t* s_b, uint64_t min_len, double min_ident, uint32_t* hd, uint8_t* ok_head, HeadRec* rec,
                      unsigned long long* n_heads, uint32_t cap_long, const uint32_t* n_long_dev, const uint32_t* long_list) {
  if (cap_chunks == 0) return SWG_OK;
  static const uint64_t lper = getenv("SWG_LABEL_BLOCKS") ? (uint64_t)atoi(getenv("SWG_LABEL_BLOCKS")) : 128;  // (work-groups per CU in the grid)
  const uint64_t lb = cap_chunks < (uint64_t)ctx->num_cu * lper ? cap_chunks : (uint64_t)ctx->num_cu * lper;
  if (s_m)
    SWG_LAUNCH(ctx, "chain_label", chain_label_kernel<LABEL_NT, true><<<(unsigned)lb, LABEL_NT, 0, ctx->stream>>>(cap_chunks, chunks, pred, s_qs, s_qe, s_ts, s_te, s_m, s_b,
                                                                             nullptr, min_len, min_ident, hd, ok_head, rec, n_heads,
                                                                             n_chunks_dev));
  else
    SWG_LAUNCH(ctx, "chain_label", chain_label_kernel<LABEL_NT, false><<<(unsigned)lb, LABEL_NT, 0, ctx->stream>>>(cap_chunks, chunks, pred, s_qs, s_qe, s_ts, s_te, s_m, s_b,
                                                                             nullptr, min_len, min_ident, hd, ok_head, rec, n_heads,
                                                                             n_chunks_dev));
  SWG_KERNEL_CHECK(ctx);
  if (cap_long) {
    const unsigned lg = cap_long < (uint32_t)ctx->num_cu * 2 ? cap_long : (unsigned)ctx->num_cu * 2;  // (32 registers: two work-groups of 1,024 per CU)
    SWG_LAUNCH(ctx, "chain_label_long", chain_label_long_kernel<<<lg, 1024, 0, ctx->stream>>>(cap_long, n_long_dev, long_list, chunks, pred, s_qs, s_qe, s_ts,
                                                                                  s_te, s_m, s_b, min_len, min_ident, hd, ok_head, rec, n_heads));
    SWG_KERNEL_CHECK(ctx);
  }
  return SWG_OK;
}

}  // namespace swg_scaf
