// The sorted begins of a sweep axis without a global sort (round 5) -- for inputs grouped by chromosome pair.
//
// swg_sweep_axis orders the begins of all live intervals by (segment, start, record index) -- the mapping-level sweep's
// segments are (sequence, genome of the other side), src/paf_filter.rs:1037-1100 -- and hands the tile kernels four arrays in
// that order: composite start S, record index I, end E, score key KEY (plus the tile-start keys and the `single` flags).  The
// general way there is an LSD radix sort of 8-byte words (four passes per axis at 10^8 records) and a gather of every
// begin's end and score through its record index (one random 32-byte sector per begin): 3.9 ms per axis on S-pan.
//
// When the input is grouped by (query, target) pair -- pair_plan (swg_pair.hip) has found its runs -- a segment is a handful
// of runs, all of whose records carry the segment's id, so the order falls apart into one small sort per segment:
//   run_alive   live records per run (once for both axes)
//   run_key     (segment id, run offset) per run, sorted (a few thousand to a few million keys)
//   seg_bounds  the segments: their stretch of the sorted runs, their offset among the live records (a prefix sum)
//   seg_perm    the live record indices, segment after segment, ascending inside a segment (stable compaction per run)
//   seg_sort    one work-group per segment: the pair_sort scheme (swg_pair.hip) -- keys dropped into buckets in LDS, ranked
//               inside their bucket by (key, position in the segment's list = record index order), and every column brought to
//               its sorted place through LDS by the thread that loaded it (coalesced loads, coalesced stores, no gather)
// Segments of more than 32,768 live records (rare: a work-group owns at most 32 records per thread) are sorted in key-range
// batches by a bitonic network and gather their columns; a coarse bin that holds more records than a batch (heavy ties)
// raises a flag and the caller sorts the axis the general way.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>

#include "swg_internal.h"
#include "swg_lds_sort.h"

namespace swg_seg {
namespace {

constexpr int EW = 256;
constexpr uint32_t NONE = 0xffffffffu;
constexpr uint32_t SEG_S_MAX = 1024, SEG_M_MAX = 4096, SEG_L_MAX = 32768, SEG_XL_MAX = 131072;
constexpr uint32_t TBF = 128;  // granularity of the tile-start keys (swg_sweep.hip)
inline unsigned nblk(uint64_t n) { return (unsigned)((n + EW - 1) / EW); }

struct Run {  // (layout of swg_scaf::PairRun)
  uint32_t a, n;
};

using swg_lds::block_excl_sum;
using swg_lds::lds_barrier;

// ---- the runs' live counts, keys, the segments -------------------------------------------------------------------------
__global__ __launch_bounds__(256) void run_alive_kernel(uint32_t n_runs, const Run* __restrict__ runs, const uint8_t* __restrict__ alive,
                                                        uint32_t* __restrict__ run_alive) {
  const uint32_t lane = threadIdx.x & 63;
  for (uint32_t k = blockIdx.x * 4u + (threadIdx.x >> 6); k < n_runs; k += gridDim.x * 4u) {
    const Run r = runs[k];
    uint32_t c = 0;
    if (alive) {
      // (four flags per lane and step where the run allows aligned 4-byte loads)
      const uint32_t head = (4u - (r.a & 3u)) & 3u, h = head < r.n ? head : r.n;
      if (lane < h) c += alive[r.a + lane] ? 1u : 0u;
      const uint32_t body = (r.n - h) / 4u;
      const uint32_t* a4 = reinterpret_cast<const uint32_t*>(alive + r.a + h);
      for (uint32_t j = lane; j < body; j += 64) {
        const uint32_t w = a4[j];
        c += ((w & 0xffu) ? 1u : 0u) + ((w & 0xff00u) ? 1u : 0u) + ((w & 0xff0000u) ? 1u : 0u) + ((w & 0xff000000u) ? 1u : 0u);
      }
      const uint32_t t0 = h + body * 4u;
      if (t0 + lane < r.n) c += alive[r.a + t0 + lane] ? 1u : 0u;
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o, 64);
    } else {
      c = r.n;
    }
    if (lane == 0) run_alive[k] = c;
  }
}
__global__ __launch_bounds__(EW) void run_key_kernel(uint32_t n_runs, const Run* __restrict__ runs, const uint32_t* __restrict__ seg_a,
                                                     const uint32_t* __restrict__ seg_b, const uint32_t* __restrict__ seg_table,
                                                     uint32_t seg_mul, const uint64_t* __restrict__ seg, int a_bits,
                                                     uint64_t* __restrict__ key, uint32_t* __restrict__ val) {
  const uint32_t k = blockIdx.x * EW + threadIdx.x;
  if (k >= n_runs) return;
  const uint32_t i = runs[k].a;
  uint64_t sg;
  if (seg) {
    sg = seg[i];
  } else {
    const uint32_t b = seg_b[i];
    sg = (uint64_t)seg_a[i] * seg_mul + (seg_table ? seg_table[b] : b);
  }
  key[k] = (sg << a_bits) | i;  // runs of one segment in ascending input order: the segment's list is then in record index order
  val[k] = k;
}
__global__ __launch_bounds__(EW) void seg_flags_kernel(uint32_t n_runs, const uint64_t* __restrict__ key, const uint32_t* __restrict__ val,
                                                       int a_bits, const uint32_t* __restrict__ run_alive, uint32_t* __restrict__ c,
                                                       uint32_t* __restrict__ f) {
  const uint32_t r = blockIdx.x * EW + threadIdx.x;
  if (r >= n_runs) return;
  c[r] = run_alive[val[r]];
  f[r] = (r == 0 || (key[r] >> a_bits) != (key[r - 1] >> a_bits)) ? 1u : 0u;
}
// off = exclusive sums of c, slot = exclusive sums of f: the segment of rank r is slot[r] + f[r] - 1
__global__ __launch_bounds__(EW) void seg_bounds_kernel(uint32_t n_runs, const uint64_t* __restrict__ key, int a_bits,
                                                        const uint32_t* __restrict__ c, const uint32_t* __restrict__ f,
                                                        const uint32_t* __restrict__ off, const uint32_t* __restrict__ slot,
                                                        uint32_t* __restrict__ seg_a, uint32_t* __restrict__ seg_e,
                                                        uint64_t* __restrict__ seg_id, const Run* __restrict__ runs,
                                                        const uint32_t* __restrict__ val, uint32_t* __restrict__ seg_base,
                                                        uint32_t* __restrict__ seg_len) {
  const uint32_t r = blockIdx.x * EW + threadIdx.x;
  if (r >= n_runs) return;
  const uint32_t s = slot[r] + f[r] - 1u;
  const uint64_t sg = key[r] >> a_bits;
  const bool last = r + 1 == n_runs || (key[r + 1] >> a_bits) != sg;
  if (f[r]) {
    seg_a[s] = off[r];
    seg_id[s] = sg;
    // a segment of ONE run: its records are a stretch of the input -- read in place, the dead ones masked (no list)
    const Run run = runs[val[r]];
    seg_base[s] = last ? run.a : NONE;
    seg_len[s] = last ? run.n : 0u;
  }
  if (last) seg_e[s] = off[r] + c[r];
}
// counters: [0..3] segments per size class, [4] flags (1: the sort must be done the general way)
__global__ __launch_bounds__(EW) void seg_class_kernel(const uint64_t* __restrict__ n_seg_dev, uint32_t cap, const uint32_t* __restrict__ seg_a,
                                                       const uint32_t* __restrict__ seg_e, const uint32_t* __restrict__ seg_base,
                                                       const uint32_t* __restrict__ seg_len, uint32_t* __restrict__ class_list,
                                                       uint32_t* __restrict__ counters, uint32_t* __restrict__ xl_len) {
  const uint32_t n_seg = (uint32_t)*n_seg_dev;
  const uint32_t s = blockIdx.x * EW + threadIdx.x;
  int cls = -1;
  if (s < n_seg && seg_e[s] != seg_a[s]) {  // (a segment without a live record: no class)
    const uint32_t m = seg_base[s] != NONE ? seg_len[s] : seg_e[s] - seg_a[s];  // places a work-group's threads own
    cls = m <= SEG_S_MAX ? 0 : (m <= SEG_M_MAX ? 1 : (m <= SEG_L_MAX ? 2 : 3));
    // the longest class passes over the whole segment once per batch of XCAP records: beyond SEG_XL_MAX places the general sort
    // is the faster way by far (one work-group would spend O(m^2 / XCAP) loads on it)
    if (m > SEG_XL_MAX) atomicOr(&counters[4], 1u);
  }
  if (xl_len && s < n_seg) xl_len[s] = cls == 3 ? seg_e[s] - seg_a[s] : 0u;
  // one atomic per wavefront and class (thousands of single increments of one counter take 10 ns each)
  const int lane = threadIdx.x & 63;
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const unsigned long long mk = __ballot(cls == c);
    if (mk == 0ull) continue;
    uint32_t base = 0;
    if (lane == __builtin_ctzll(mk)) base = atomicAdd(&counters[c], (uint32_t)__popcll(mk));
    base = (uint32_t)__shfl((int)base, __builtin_ctzll(mk), 64);
    if (cls == c) class_list[(size_t)c * cap + base + (uint32_t)__popcll(mk & ((1ull << lane) - 1ull))] = s;
  }
}
// the live records of every run, in input order, at the run's place in its segment's list: one wavefront per sorted run (a
// ballot per 64 records; millions of tiny runs must not each occupy a work-group and its barriers)
__global__ __launch_bounds__(256) void seg_perm_kernel(uint32_t n_runs, const Run* __restrict__ runs, const uint32_t* __restrict__ val,
                                                       const uint32_t* __restrict__ off, const uint32_t* __restrict__ f,
                                                       const uint8_t* __restrict__ alive, uint32_t* __restrict__ perm) {
  const uint32_t lane = threadIdx.x & 63;
  for (uint32_t r = blockIdx.x * 4u + (threadIdx.x >> 6); r < n_runs; r += gridDim.x * 4u) {
    if (f[r] && (r + 1 == n_runs || f[r + 1])) continue;  // a segment of one run is read in place
    const Run run = runs[val[r]];
    uint32_t dest = off[r];
    for (uint32_t j0 = 0; j0 < run.n; j0 += 64) {
      const uint32_t j = j0 + lane;
      const bool live = j < run.n && (!alive || alive[run.a + j] != 0);
      const unsigned long long mk = __ballot(live);
      if (live) perm[dest + (uint32_t)__popcll(mk & ((1ull << lane) - 1ull))] = run.a + j;
      dest += (uint32_t)__popcll(mk);
    }
  }
}

// ---- seg_sort -------------------------------------------------------------------------------------------------------------
struct SegSortArgs {
  const uint32_t* perm;
  const uint32_t *seg_a, *seg_e, *seg_base, *seg_len;
  const uint32_t* out_a;  // the segment's offset in the sorted arrays (seg_a, or the offsets among the longest segments alone)
  const uint64_t* seg_id;
  const uint8_t* alive;
  const uint32_t* list;
  const uint32_t *start, *end;
  const uint64_t* score;
  int pos_bits;
  uint32_t n_dead;
  uint64_t* S;
  uint32_t* I;
  uint32_t* E;
  uint64_t* KEY;
  uint64_t* tile_xf;
  uint8_t* single;
  uint32_t* counters;
  // optional (finite k): the intervals that overlap no other interval of their segment are settled here -- they are in the top k
  // wherever they are active and never overlapped (plane_sweep_exact.rs:197-352) -- and the ones that are never active
  // (start >= end) as well; tile_flag[sorted position] = 1 for every OTHER begin, the ones the caller compacts for the tile
  // kernels.  The kept lone ones are marked in `single` (the caller's combine keeps those whatever the tile kernels say).
  uint8_t* tile_flag;
};

constexpr size_t lds_align_up(size_t off, size_t align) { return (off + align - 1) / align * align; }
template <int NT, int ES, int ER, int NBK, int NBIN>
constexpr size_t seg_sort_lds_bytes() {
  return (size_t)NT * ES * 8 + (size_t)NBK * 4 + (size_t)NBIN * 4 + 17 * 4 + (size_t)(NT / 64 + 1) * 4 + 16 * 4 + (size_t)NT * 4 + 64;
}

// One work-group per segment of at most NT * ER live records, sorted in batches of at most NT * ES (one batch when they fit).
// The scheme and its idioms are pair_sort_body's (swg_pair.hip): a thread OWNS the records tid, tid + NT, ... of the segment's
// list, drops their keys into the batch's buckets, learns from the ranking where they ended up and puts their columns there.
template <int NT, int ES, int ER, int NBK, int NBIN, bool LONE>
__device__ __forceinline__ void seg_sort_body(const SegSortArgs& A, const uint32_t sg, char* lds_raw) {
  constexpr int CAP = NT * ES, MAXB = 16, H = 8;
  static_assert(ER <= 32 && ER % H == 0 && ES % 4 == 0 && NT * ER <= 65536 && CAP < 0xffff, "record masks are 32 bits wide, indices and ranks 16");
  static_assert(NBIN >= 1 && NBIN <= 4096 && NBK % NT == 0, "bins, bucket counters per thread");
  constexpr size_t O_K = 0, O_I = O_K + (size_t)CAP * 4, O_RR = O_I + (size_t)CAP * 2, O_CNT = lds_align_up(O_RR + (size_t)CAP * 2, 4),
                   O_BINS = O_CNT + (size_t)NBK * 4, O_BLO = O_BINS + (size_t)NBIN * 4, O_WS = O_BLO + (size_t)(MAXB + 1) * 4,
                   O_SH = O_WS + (size_t)(NT / 64 + 1) * 4, O_LONE = O_SH + 8 * 4;
  static_assert(O_LONE + (size_t)NT * 4 <= seg_sort_lds_bytes<NT, ES, ER, NBK, NBIN>() && ES <= 16, "LDS block of the work-group");
  uint16_t* const LBIT = reinterpret_cast<uint16_t*>(lds_raw + O_LONE);  // [NT] bit e: the thread's e-th consecutive position is a lone interval
  uint16_t* const LKEEP = LBIT + NT;                                       // ... and kept (it is active somewhere)
  uint32_t* const K = reinterpret_cast<uint32_t*>(lds_raw + O_K);
  uint16_t* const I = reinterpret_cast<uint16_t*>(lds_raw + O_I);
  uint16_t* const RR = reinterpret_cast<uint16_t*>(lds_raw + O_RR);
  uint32_t* const B2 = reinterpret_cast<uint32_t*>(lds_raw + O_I);  // (I and RR together, once both are done with: a second column buffer)
  uint32_t* const cnt = reinterpret_cast<uint32_t*>(lds_raw + O_CNT);
  uint32_t* const bins = reinterpret_cast<uint32_t*>(lds_raw + O_BINS);
  uint32_t* const b_lo = reinterpret_cast<uint32_t*>(lds_raw + O_BLO);
  uint32_t* const ws = reinterpret_cast<uint32_t*>(lds_raw + O_WS);
  uint32_t* const sh = reinterpret_cast<uint32_t*>(lds_raw + O_SH);  // [0] kmin, [1] kmax, [2] batches, [3] bad
  const int tid = threadIdx.x;
  const uint32_t a = A.seg_a[sg], n_live = A.seg_e[sg] - a;
  const uint32_t rbase = (uint32_t)__builtin_amdgcn_readfirstlane((int)A.seg_base[sg]);
  const bool in_place = rbase != NONE;  // one run: the list is the run itself, dead records and all
  const uint32_t m = in_place ? A.seg_len[sg] : n_live;  // places of the list
  const uint64_t seg_part = (A.seg_id[sg] + 1) << A.pos_bits;
  const uint32_t* __restrict__ c_perm = A.perm + a;
  if (tid == 0) {
    sh[0] = 0xffffffffu;
    sh[1] = 0u;
    sh[3] = 0u;
  }
  for (int b = tid; b < NBIN; b += NT) bins[b] = 0;
  lds_barrier();
  auto fresh_tid = [&]() -> uint32_t {
    uint32_t t = (uint32_t)tid;
    asm volatile("" : "+v"(t));
    return t;
  };
  uint32_t tid_v = (uint32_t)tid, m_v = m;
  auto rec_index = [&](int e) -> uint32_t {  // the record behind the thread's e-th place in the list
    const uint32_t li = tid_v + (uint32_t)e * NT;
    const uint32_t lc = li < m_v ? li : 0u;
    if (in_place) return rbase + lc;
    return c_perm[lc];
  };
  // ---- the key range
  uint32_t in_mask = 0;
  {
    uint32_t kmin = 0xffffffffu, kmax = 0;
#pragma unroll
    for (int g = 0; g < ER; g += H) {
      if ((uint32_t)g * NT >= m) continue;  // (block-uniform; `continue`, not `break`: the loop must unroll)
      uint32_t ixv[H], qv[H];
      uint8_t av[H];
#pragma unroll
      for (int e = 0; e < H; ++e) ixv[e] = rec_index(g + e);
#pragma unroll
      for (int e = 0; e < H; ++e) {
        qv[e] = A.start[ixv[e]];
        av[e] = (in_place && A.alive) ? A.alive[ixv[e]] : (uint8_t)1;
      }
#pragma unroll
      for (int e = 0; e < H; ++e) {
        const uint32_t li = (uint32_t)tid + (uint32_t)(g + e) * NT;
        if (li >= m || !av[e]) continue;
        if (n_live == 1u) {  // returned whole by the reference (plane_sweep_exact.rs:274-276)
          A.single[ixv[e]] = 1;
        }
        in_mask |= 1u << (g + e);
        kmin = qv[e] < kmin ? qv[e] : kmin;
        kmax = qv[e] > kmax ? qv[e] : kmax;
      }
      asm volatile("" ::: "memory");
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const uint32_t x = __shfl_xor(kmin, o, 64), y = __shfl_xor(kmax, o, 64);
      kmin = x < kmin ? x : kmin;
      kmax = y > kmax ? y : kmax;
    }
    if ((tid & 63) == 0) {
      atomicMin(&sh[0], kmin);
      atomicMax(&sh[1], kmax);
    }
  }
  lds_barrier();
  const uint32_t k_lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)sh[0]);
  const float scale = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int((float)NBIN / ((float)(sh[1] - sh[0]) + 1.0f))));
  // the fine bucket of a key inside the batch [bin_lo, bin_hi): the coarse map refined by a power of two (monotone in the key
  // whatever the rounding; f < NBIN <= 2^12, so (uint32)(f * 4096) >> 12 == (uint32)f)
  auto fine_of = [&](uint32_t k, int shift, uint32_t first, uint32_t* coarse) -> uint32_t {
    const float f = (float)(k - k_lo) * scale;
    uint32_t g = (uint32_t)(f * 4096.0f);
    if ((g >> 12) > (uint32_t)NBIN - 1u) g = (((uint32_t)NBIN - 1u) << 12) | 0xfffu;
    *coarse = g >> 12;
    const uint32_t b = (g >> shift) - first;
    return b < (uint32_t)NBK ? b : (uint32_t)NBK - 1u;
  };
  uint32_t n_batches = 1;
  if (n_live > (uint32_t)CAP) {
#pragma unroll
    for (int g = 0; g < ER; g += H) {
      if ((uint32_t)g * NT >= m) continue;
      uint32_t ixv[H], qv[H];
#pragma unroll
      for (int e = 0; e < H; ++e) ixv[e] = rec_index(g + e);
#pragma unroll
      for (int e = 0; e < H; ++e) qv[e] = A.start[ixv[e]];
#pragma unroll
      for (int e = 0; e < H; ++e)
        if ((in_mask >> (g + e)) & 1u) {
          uint32_t cb;
          (void)fine_of(qv[e], 12, 0u, &cb);
          atomicAdd(&bins[cb], 1u);
        }
      asm volatile("" ::: "memory");
    }
    lds_barrier();
    swg_lds::bins_to_offsets<NT, NBIN>(bins, ws);
    lds_barrier();
    if (tid == 0) {  // greedy: a batch takes as many bins as fit
      const uint32_t nb = swg_lds::plan_batches<NBIN>(bins, n_live, (uint32_t)CAP, (uint32_t)MAXB, b_lo);
      sh[2] = nb;
      if (nb == 0) sh[3] = 1;  // one bin denser than a batch: the general sort's case
    }
    lds_barrier();
    if (sh[3]) {
      if (tid == 0) atomicOr(&A.counters[4], 1u);
      return;
    }
    n_batches = (uint32_t)__builtin_amdgcn_readfirstlane((int)sh[2]);
  }
  uint32_t base = 0;
  uint32_t carry_end = 0;  // (lone intervals) the largest end of the batches so far
  for (uint32_t bt = 0; bt < n_batches; ++bt) {
    {  // (see pair_sort_body: keeps the loads of every batch inside the loop)
      uint32_t m_l = m_v;
      asm volatile("" : "+v"(tid_v), "+v"(m_l), "+v"(in_mask));
      m_v = (uint32_t)__builtin_amdgcn_readfirstlane((int)m_l);
    }
    const uint32_t bin_lo = n_batches > 1 ? (uint32_t)__builtin_amdgcn_readfirstlane((int)b_lo[bt]) : 0u;
    const uint32_t bin_hi = n_batches > 1 ? (uint32_t)__builtin_amdgcn_readfirstlane((int)b_lo[bt + 1]) : (uint32_t)NBIN;
    int shift = 0;
    while ((((bin_hi - bin_lo) << 12) >> shift) > (uint32_t)NBK) ++shift;
    const uint32_t first = (bin_lo << 12) >> shift;
    for (int b = tid; b < NBK; b += NT) cnt[b] = 0;
    if (tid == 0) sh[4] = 0xffffffffu;
    lds_barrier();
    // ---- count (and, for the lone intervals, the first start of the batches behind this one)
    uint32_t batch_mask = 0;
    {
      uint32_t s_after = 0xffffffffu;
#pragma unroll
      for (int g = 0; g < ER; g += H) {
        if ((uint32_t)g * NT >= m) continue;
        uint32_t ixv[H], qv[H];
#pragma unroll
        for (int e = 0; e < H; ++e) ixv[e] = rec_index(g + e);
#pragma unroll
        for (int e = 0; e < H; ++e) qv[e] = A.start[ixv[e]];
#pragma unroll
        for (int e = 0; e < H; ++e)
          if ((in_mask >> (g + e)) & 1u) {
            uint32_t cb;
            const uint32_t fb = fine_of(qv[e], shift, first, &cb);
            if (cb >= bin_lo && cb < bin_hi) {
              batch_mask |= 1u << (g + e);
              atomicAdd(&cnt[fb], 1u);
            } else if (LONE && cb >= bin_hi) {
              s_after = qv[e] < s_after ? qv[e] : s_after;
            }
          }
        asm volatile("" ::: "memory");
      }
      if (LONE && bt + 1 < n_batches) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
          const uint32_t x = __shfl_xor(s_after, o, 64);
          s_after = x < s_after ? x : s_after;
        }
        if ((tid & 63) == 0 && s_after != 0xffffffffu) atomicMin(&sh[4], s_after);
      }
    }
    lds_barrier();
    const uint32_t mb = swg_lds::bucket_offsets<NT, NBK>(cnt, ws);
    lds_barrier();
    // ---- scatter (unordered inside a bucket; cnt[b] ends as the bucket's end); the thread remembers where each key went
    uint32_t slotw[ER / 2];
#pragma unroll
    for (int j = 0; j < ER / 2; ++j) slotw[j] = 0;
    const uint32_t t_sc = fresh_tid();
#pragma unroll
    for (int g = 0; g < ER; g += H) {
      if ((uint32_t)g * NT >= m) continue;
      uint32_t ixv[H], qv[H];
#pragma unroll
      for (int e = 0; e < H; ++e) ixv[e] = rec_index(g + e);
#pragma unroll
      for (int e = 0; e < H; ++e) qv[e] = A.start[ixv[e]];
#pragma unroll
      for (int e = 0; e < H; ++e)
        if ((batch_mask >> (g + e)) & 1u) {
          uint32_t cb;
          const uint32_t fb = fine_of(qv[e], shift, first, &cb);
          const uint32_t pos = atomicAdd(&cnt[fb], 1u);
          K[pos] = qv[e];
          I[pos] = (uint16_t)(t_sc + (uint32_t)(g + e) * NT);
          slotw[(g + e) / 2] |= pos << (16 * ((g + e) & 1));
        }
      asm volatile("" ::: "memory");
    }
    lds_barrier();
    // order inside the buckets: final position = bucket begin + the bucket's elements that order before by (key, place in the list)
    swg_lds::rank_buckets<NT, ES, uint16_t, false>(K, I, RR, cnt, mb, fresh_tid(), [&](uint32_t, uint32_t key) {
      uint32_t cb;
      return fine_of(key, shift, first, &cb);
    });
    // ---- where the thread's own records went
    const uint32_t gbase = A.n_dead + A.out_a[sg] + base;
    swg_lds::slots_to_ranks<ER>(slotw, batch_mask, RR);
    lds_barrier();
    // ---- the other columns, transposed through LDS
    auto put_group = [&](int g, const uint32_t (&v)[H], uint32_t* buf) {
#pragma unroll
      for (int e = 0; e < H; ++e)
        if ((batch_mask >> (g + e)) & 1u) buf[(slotw[(g + e) / 2] >> (16 * ((g + e) & 1))) & 0xffffu] = v[e];
    };
    if constexpr (LONE) {
    // the ends, into the room of I and RR (both done with) while K still holds the sorted starts
    {
      tid_v = fresh_tid();
#pragma unroll
      for (int g = 0; g < ER; g += H) {
        if ((uint32_t)g * NT >= m) continue;
        uint32_t ixv[H], v[H];
#pragma unroll
        for (int e = 0; e < H; ++e) ixv[e] = rec_index(g + e);
#pragma unroll
        for (int e = 0; e < H; ++e) v[e] = A.end[ixv[e]];
        put_group(g, v, B2);
        asm volatile("" ::: "memory");
      }
      lds_barrier();
    }
    {
      // the lone intervals of the batch: no earlier interval of the segment reaches the start (the running maximum of the ends in
      // front of it, the batches so far included), no later one begins before the end (the next position's start; behind the
      // batch's last position the first start of the batches to come).  A thread takes ES consecutive positions.
      const uint32_t q0 = fresh_tid() * ES;
      const uint32_t s_after = bt + 1 < n_batches ? sh[4] : 0xffffffffu;
      uint32_t ks[ES + 1], es[ES], tmax = 0;
#pragma unroll
      for (int e = 0; e < ES; ++e) {
        ks[e] = q0 + e < mb ? K[q0 + e] : 0xffffffffu;
        es[e] = q0 + e < mb ? B2[q0 + e] : 0u;
      }
      ks[ES] = q0 + ES < mb ? K[q0 + ES] : s_after;
#pragma unroll
      for (int e = 0; e < ES; ++e)
        if (q0 + e < mb && es[e] > ks[e] && es[e] > tmax) tmax = es[e];
      uint32_t before = swg_lds::block_excl_max_u32<NT>(tmax, ws);
      before = before > carry_end ? before : carry_end;
      uint32_t lone = 0, lkeep = 0;
#pragma unroll
      for (int e = 0; e < ES; ++e) {
        if (q0 + e >= mb) continue;
        const bool active = es[e] > ks[e];
        const uint32_t next = q0 + e + 1 < mb ? ks[e + 1] : s_after;
        const bool alone = before <= ks[e] && next >= es[e];
        if (!active || alone) lone |= 1u << e;
        if (active && alone) lkeep |= 1u << e;
        if (active && es[e] > before) before = es[e];
      }
      LBIT[tid] = (uint16_t)lone;
      LKEEP[tid] = (uint16_t)lkeep;
      // (the batches so far: the largest end of all -- every thread ends with the batch's through the last thread's `before`)
      lds_barrier();
      if (tid == NT - 1) ws[NT / 64] = before;
      lds_barrier();
      {
        const uint32_t all = ws[NT / 64];
        // (the last thread holds positions beyond mb when the batch is short: its `before` is still the batch's maximum, the
        // running maximum in front of them)
        carry_end = all > carry_end ? all : carry_end;
      }
    }
    // ---- the composite starts out (and which of them go on to the tile kernels)
    const uint32_t t_out = fresh_tid();
#pragma unroll
    for (int e = 0; e < ES; ++e) {
      const uint32_t p = t_out + (uint32_t)e * NT;
      if (p < mb) {
        const uint64_t s = seg_part | K[p];
        A.S[gbase + p] = s;
        if (((gbase + p) % TBF) == 0u) A.tile_xf[(gbase + p) / TBF] = s;
        A.tile_flag[gbase + p] = ((LBIT[p / ES] >> (p % ES)) & 1u) ? 0 : 1;
      }
    }
    lds_barrier();
    // the record indices (K is free now)
    {
      tid_v = fresh_tid();
#pragma unroll
      for (int g = 0; g < ER; g += H) {
        if ((uint32_t)g * NT >= m) continue;
        uint32_t ixv[H];
#pragma unroll
        for (int e = 0; e < H; ++e) ixv[e] = rec_index(g + e);
        put_group(g, ixv, K);
        asm volatile("" ::: "memory");
      }
      lds_barrier();
      const uint32_t t_st = fresh_tid();
#pragma unroll
      for (int e = 0; e < ES; ++e) {
        const uint32_t p = t_st + (uint32_t)e * NT;
        if (p < mb) {
          const uint32_t ix = K[p];
          A.I[gbase + p] = ix;
          A.E[gbase + p] = B2[p];
          if ((LKEEP[p / ES] >> (p % ES)) & 1u) A.single[ix] = 1;  // (kept whatever the tile kernels say: combine)
        }
      }
      lds_barrier();
    }
    } else {
    // ---- the composite starts out
    {
      const uint32_t t_out = fresh_tid();
#pragma unroll
      for (int e = 0; e < ES; ++e) {
        const uint32_t p = t_out + (uint32_t)e * NT;
        if (p < mb) {
          const uint64_t s = seg_part | K[p];
          A.S[gbase + p] = s;
          if (((gbase + p) % TBF) == 0u) A.tile_xf[(gbase + p) / TBF] = s;
        }
      }
      lds_barrier();
    }
    // the record indices and the ends (two buffers: K, and the room of I and RR, both done with)
    {
      tid_v = fresh_tid();
#pragma unroll
      for (int g = 0; g < ER; g += H) {
        if ((uint32_t)g * NT >= m) continue;
        uint32_t ixv[H], v[H];
#pragma unroll
        for (int e = 0; e < H; ++e) ixv[e] = rec_index(g + e);
#pragma unroll
        for (int e = 0; e < H; ++e) v[e] = A.end[ixv[e]];
        put_group(g, ixv, K);
        put_group(g, v, B2);
        asm volatile("" ::: "memory");
      }
      lds_barrier();
      const uint32_t t_st = fresh_tid();
#pragma unroll
      for (int e = 0; e < ES; ++e) {
        const uint32_t p = t_st + (uint32_t)e * NT;
        if (p < mb) {
          A.I[gbase + p] = K[p];
          A.E[gbase + p] = B2[p];
        }
      }
      lds_barrier();
    }
    }
    // the score keys (8 bytes: the two halves through two buffers)
    {
      tid_v = fresh_tid();
#pragma unroll
      for (int g = 0; g < ER; g += H) {
        if ((uint32_t)g * NT >= m) continue;
        uint32_t ixv[H], lo[H], hi[H];
#pragma unroll
        for (int e = 0; e < H; ++e) ixv[e] = rec_index(g + e);
#pragma unroll
        for (int e = 0; e < H; ++e) {
          const uint64_t x = A.score[ixv[e]];
          lo[e] = (uint32_t)x;
          hi[e] = (uint32_t)(x >> 32);
        }
        put_group(g, lo, K);
        put_group(g, hi, B2);
        asm volatile("" ::: "memory");
      }
      lds_barrier();
      const uint32_t t_st = fresh_tid();
#pragma unroll
      for (int e = 0; e < ES; ++e) {
        const uint32_t p = t_st + (uint32_t)e * NT;
        if (p < mb) A.KEY[gbase + p] = ((uint64_t)B2[p] << 32) | K[p];
      }
    }
    base += mb;
    lds_barrier();
  }
}
template <int NT, int ES, int ER, int NBK, int NBIN, bool LONE>
__global__ __launch_bounds__(NT) void seg_sort_kernel(SegSortArgs A) {
  __shared__ __attribute__((aligned(16))) char raw[seg_sort_lds_bytes<NT, ES, ER, NBK, NBIN>()];
  seg_sort_body<NT, ES, ER, NBK, NBIN, LONE>(A, A.list[blockIdx.x], raw);
}

// Segments beyond SEG_L_MAX places: key-range batches of at most XCAP (coarse bins glued greedily), every batch picked out of
// the segment's list by a pass over its keys, sorted as 64-bit words (key << 32 | place in the list) by a bitonic network in
// LDS, its columns gathered through the record index.  Rare (a fraction of a per cent of S-pan's segments), so simple; they
// share the launch of the large class and go first, so that the bulk of the work runs beside them.
constexpr int XNT = 1024, XCAP = 8192, XBIN = 4096, XMAXB = 4096;
constexpr size_t SEG_XL_LDS = (size_t)XCAP * 8 + (size_t)(XBIN + 1) * 4 + (size_t)(XNT / 64 + 1) * 4 + 8 * 4 + 64;
__device__ __forceinline__ void seg_sort_xl_body(const SegSortArgs& A, const uint32_t sg, char* lds_raw) {
  constexpr size_t O_W = 0, O_BINS = O_W + (size_t)XCAP * 8, O_WS = O_BINS + (size_t)(XBIN + 1) * 4, O_SH = O_WS + (size_t)(XNT / 64 + 1) * 4;
  static_assert(O_SH + 8 * 4 <= SEG_XL_LDS, "LDS block of the work-group");
  unsigned long long* const W = reinterpret_cast<unsigned long long*>(lds_raw + O_W);
  uint32_t* const bins = reinterpret_cast<uint32_t*>(lds_raw + O_BINS);
  uint32_t* const ws = reinterpret_cast<uint32_t*>(lds_raw + O_WS);
  uint32_t* const sh = reinterpret_cast<uint32_t*>(lds_raw + O_SH);  // [0] kmin, [1] kmax, [2] cursor
  const int tid = threadIdx.x;
  const uint32_t a = A.seg_a[sg], n_live = A.seg_e[sg] - a;
  const uint32_t rbase = A.seg_base[sg];
  const bool in_place = rbase != NONE;
  const uint32_t m = in_place ? A.seg_len[sg] : n_live;
  const uint64_t seg_part = (A.seg_id[sg] + 1) << A.pos_bits;
  const uint32_t* __restrict__ c_perm = A.perm + a;
  auto rec_of = [&](uint32_t j) -> uint32_t { return in_place ? rbase + j : c_perm[j]; };
  auto live_at = [&](uint32_t i) -> bool { return !(in_place && A.alive) || A.alive[i] != 0; };
  if (tid == 0) {
    sh[0] = 0xffffffffu;
    sh[1] = 0u;
  }
  for (int b = tid; b <= XBIN; b += XNT) bins[b] = 0;
  __syncthreads();
  {
    uint32_t kmin = 0xffffffffu, kmax = 0;
    for (uint32_t j = tid; j < m; j += XNT) {
      const uint32_t i = rec_of(j);
      if (!live_at(i)) continue;
      const uint32_t k = A.start[i];
      kmin = k < kmin ? k : kmin;
      kmax = k > kmax ? k : kmax;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const uint32_t x = __shfl_xor(kmin, o, 64), y = __shfl_xor(kmax, o, 64);
      kmin = x < kmin ? x : kmin;
      kmax = y > kmax ? y : kmax;
    }
    if ((tid & 63) == 0) {
      atomicMin(&sh[0], kmin);
      atomicMax(&sh[1], kmax);
    }
  }
  __syncthreads();
  const uint32_t k_lo = sh[0];
  const float scale = (float)XBIN / ((float)(sh[1] - sh[0]) + 1.0f);
  auto bin_of = [&](uint32_t k) -> uint32_t {  // monotone in the key
    const uint32_t b = (uint32_t)((float)(k - k_lo) * scale);
    return b < (uint32_t)XBIN - 1u ? b : (uint32_t)XBIN - 1u;
  };
  for (uint32_t j = tid; j < m; j += XNT) {
    const uint32_t i = rec_of(j);
    if (live_at(i)) atomicAdd(&bins[bin_of(A.start[i])], 1u);
  }
  __syncthreads();
  {  // exclusive prefix sums over the bins (4 per thread), bins[XBIN] = the live records
    uint32_t c[XBIN / XNT], sum = 0, tot;
#pragma unroll
    for (int j = 0; j < XBIN / XNT; ++j) {
      c[j] = bins[tid * (XBIN / XNT) + j];
      sum += c[j];
    }
    uint32_t off = block_excl_sum<XNT>(sum, ws, &tot);
    __syncthreads();
#pragma unroll
    for (int j = 0; j < XBIN / XNT; ++j) {
      bins[tid * (XBIN / XNT) + j] = off;
      off += c[j];
    }
    if (tid == XNT - 1) bins[XBIN] = off;
  }
  __syncthreads();
  uint32_t lo = 0, base = 0;
  for (uint32_t guard = 0; lo < (uint32_t)XBIN && guard < (uint32_t)XMAXB; ++guard) {  // (uniform)
    // the batch [lo, hi): as many bins as fit (every thread runs the same search)
    const uint32_t start = bins[lo];
    uint32_t l = lo + 1, r = XBIN;
    while (l < r) {
      const uint32_t mid = l + ((r - l + 1) >> 1);
      if (bins[mid] - start <= (uint32_t)XCAP) l = mid; else r = mid - 1;
    }
    const uint32_t hi = l, mb = bins[hi] - start;
    if (mb > (uint32_t)XCAP) {  // one bin denser than a batch: the general sort's case
      if (tid == 0) atomicOr(&A.counters[4], 1u);
      return;
    }
    if (mb == 0u) {
      lo = hi;
      continue;
    }
    if (tid == 0) sh[2] = 0u;
    uint32_t np2 = 1;
    while (np2 < mb) np2 <<= 1;
    for (uint32_t x = tid; x < np2; x += XNT) W[x] = ~0ull;  // (padding sorts last)
    __syncthreads();
    for (uint32_t j = tid; j < m; j += XNT) {
      const uint32_t i = rec_of(j);
      if (!live_at(i)) continue;
      const uint32_t k = A.start[i];
      const uint32_t b = bin_of(k);
      if (b >= lo && b < hi) W[atomicAdd(&sh[2], 1u)] = ((unsigned long long)k << 32) | j;
    }
    __syncthreads();
    for (uint32_t kk = 2; kk <= np2; kk <<= 1)
      for (uint32_t jj = kk >> 1; jj > 0; jj >>= 1) {
        for (uint32_t x = tid; x < np2; x += XNT) {
          const uint32_t y = x ^ jj;
          if (y > x) {
            const unsigned long long u = W[x], v = W[y];
            const bool up = (x & kk) == 0;
            if ((u > v) == up) {
              W[x] = v;
              W[y] = u;
            }
          }
        }
        __syncthreads();
      }
    const uint32_t gbase = A.n_dead + A.out_a[sg] + base;
    for (uint32_t p = tid; p < mb; p += XNT) {
      const unsigned long long w = W[p];
      const uint32_t id = rec_of((uint32_t)w);
      const uint64_t s = seg_part | (w >> 32);
      A.S[gbase + p] = s;
      A.I[gbase + p] = id;
      A.E[gbase + p] = A.end[id];
      A.KEY[gbase + p] = A.score[id];
      if (A.tile_flag) A.tile_flag[gbase + p] = 1;  // (the longest segments: not looked at for lone intervals)
      if (((gbase + p) % TBF) == 0u) A.tile_xf[(gbase + p) / TBF] = s;
      if (n_live == 1u) {  // (a long run with one live record) returned whole by the reference
        A.single[id] = 1;
        if (A.tile_flag) A.tile_flag[gbase + p] = 0;
      }
    }
    __syncthreads();
    base += mb;
    lo = hi;
  }
}
constexpr size_t SEG_BIG_LDS = seg_sort_lds_bytes<1024, 16, 32, 4096, 1024>() > SEG_XL_LDS ? seg_sort_lds_bytes<1024, 16, 32, 4096, 1024>() : SEG_XL_LDS;
static_assert(SEG_BIG_LDS <= 160 * 1024, "LDS of a CU");
template <bool LONE>
__global__ __launch_bounds__(1024) void seg_sort_big_kernel(SegSortArgs A, const uint32_t* __restrict__ list_xl, uint32_t n_xl) {
  __shared__ __attribute__((aligned(16))) char raw[SEG_BIG_LDS];
  if (blockIdx.x < n_xl)
    seg_sort_xl_body(A, list_xl[blockIdx.x], raw);
  else
    seg_sort_body<1024, 16, 32, 4096, 1024, LONE>(A, A.list[blockIdx.x - n_xl], raw);
}


// ---- the k = 1 sweep of a segment, in the LDS residency of its sort (round 6) ------------------------------------------------
// src/plane_sweep_exact.rs:197-352 restated per segment (tools/model_segment_sweep.py::segment_sweep_k1_resident is
// the executable model of exactly this code, held to the oracle by tests/test_segment_sweep_model_cpu.py): with the segment's
// intervals in (start, record index) order in LDS -- start K, end and the running maximum of the ends EP, score key KEY -- the
// set of intervals active at a position P is a window of slots: from the first slot whose running maximum of ends exceeds P to
// the last slot that starts by P.  One thread per slot evaluates
//   * the interval's START event: the best active interval t at its start is marked `top`; if the slot's own interval is that
//     best, the top changed here and every other active interval is tested against it (overlap fraction > threshold -> the
//     sticky `overlapped`), otherwise only the slot's own interval is -- everything else met this top at an earlier event;
//   * its END event: the best active interval behind it is marked; the full pass runs only if the interval that ends
//     outranked it (the top changed here).
// keep = top & ~overlapped goes straight to the record's flag: no sorted columns in memory, no carry-in routing, no tiles.
// A segment larger than one LDS batch comes through in key-range batches (the sort's own); the intervals that may still be
// active at the next batch's first start are carried over in front of it (order-preserving compaction, at most CMAX), and an
// interval's END event is evaluated in the batch that holds every begin up to its end.  Deep data (more carried intervals than
// CMAX, or a thread's window scans beyond SWEEP_BUDGET steps per batch) raises counters[4] bit 1: the axis goes to the tile
// kernels (swg_sweep.hip), whose pruning is made for that.  start >= end is never active.
struct SegSweepArgs {
  const uint32_t* perm;
  const uint32_t *seg_a, *seg_e, *seg_base, *seg_len;
  const uint8_t* alive;
  const uint32_t* list;
  const uint32_t *start, *end;
  const uint64_t* score;
  const uint8_t* and_with;
  uint8_t* keep;
  double thr;
  uint32_t* counters;
};
constexpr uint32_t SWEEP_BUDGET = 4096;  // steps of 64 window slots a wavefront spends on the long windows of one round before the data counts as deep
#ifdef SWG_SEG_TIMING  // (a build knob: -DSWG_SEG_TIMING, SWG_DEFINES of sweepga_amd/build.py) the phases of seg_sweep_body in 100 MHz ticks
__device__ unsigned long long g_seg_t[16];
#define ST_STAMP(k) do { __syncthreads(); if (threadIdx.x == 0) { const unsigned long long t_ = wall_clock64(); atomicAdd(&g_seg_t[k], t_ - st_last); st_last = t_; } } while (0)
#else
#define ST_STAMP(k) do { } while (0)
#endif
constexpr uint32_t F_TOP = 1u, F_OVL = 2u;

__device__ __forceinline__ bool seg_overlap_exceeds(uint32_t as, uint32_t ae, uint32_t bs, uint32_t be, double thr) {
  // query_overlap / target_overlap, plane_sweep_exact.rs:113-144
  const uint32_t os = as > bs ? as : bs;
  const uint32_t oe = ae < be ? ae : be;
  const double ol = oe > os ? (double)(oe - os) : 0.0;
  const uint32_t la = ae - as, lb = be - bs;
  const double ml = (double)(la < lb ? la : lb);
  if (!(ml > 0.0)) return false;
  // ol / ml > thr, the quotient rounded to nearest as the reference computes it: decided without the division unless the exact
  // quotient lies within 2^-50 of thr (ol and ml are integers below 2^32, exact in f64; thr * ml and the factor each round once)
  const double tm = thr * ml;
  if (ol > tm * (1.0 + 0x1p-50)) return true;
  if (ol < tm * (1.0 - 0x1p-50)) return false;
  return __ddiv_rn(ol, ml) > thr;
}


// ---- one batch of the k = 1 sweep, shared by the fused body (seg_sweep_body) and the streaming one (seg_stream_body) -----------
// The batch's intervals sit in LDS in (start, record index) order: the nc carried ones in front, then mb new ones -- start K, end
// EP.x (EP.y is filled here: the running maximum of the ends), score key KEY, ID (what rec_of turns into the record's index),
// fresh flags TOPF / OVLF for the new ones.  The batch answers for the positions [s_first, s_next) (last: no bound above);
// intervals that reach beyond are carried over in front of the next batch (nc on return), the others' flags go to `keep`.
// false: the data is deep (counters[4] |= 2) -- the caller gives the call up.  Leaves through a barrier.
// (the LDS arrays as parameters of their own: a pointer that travels through a struct in memory loses its address space, and
// every access behind it becomes a flat instruction)
__device__ __forceinline__ uint32_t swg_fresh_tid() {
  uint32_t t = threadIdx.x;
  asm volatile("" : "+v"(t));
  return t;
}
template <int NT, int ES, int CMAX, int EVL, int N_LONG, typename IDT, class RecOf>
__device__ __forceinline__ bool sweep_batch(uint32_t* const K, uint2* const EP, uint64_t* const KEY, IDT* const ID, uint8_t* const TOPF,
                                            uint8_t* const OVLF, uint16_t* const ev_list /*[EVL] a round's listed slots*/,
                                            uint16_t* const ev_long /*[N_LONG] those with long windows*/, uint32_t* const ws,
                                            uint32_t* const sh /*[5] deep, [6] a round's long windows, [7] a round's listed intervals*/,
                                            uint32_t& nc, const uint32_t mb, const bool last, uint32_t& s_first, const uint32_t s_next,
                                            const double thr, const uint8_t* __restrict__ and_with, uint8_t* __restrict__ keep,
                                            uint32_t* __restrict__ counters, RecOf&& rec_of) {
  constexpr int CAP = NT * ES;
  const int tid = threadIdx.x;
  const bool ovl_on = thr < 1.0;
  {
    const uint32_t nb = nc + mb;
    // ---- the running maximum of the ends (an interval that is never active does not raise it)
    {
      const uint32_t q0 = swg_fresh_tid() * ES;
      uint32_t run[ES], tmax = 0;
#pragma unroll
      for (int e = 0; e < ES; ++e) {
        const uint32_t q = q0 + e;
        const uint32_t s = q < nb ? K[q] : 0u, en = q < nb ? EP[q].x : 0u;
        if (q < nb && en <= s && en) EP[q].x = 0u;  // never active: end 0 from here on (no position lies below it)
        tmax = (en > s && en > tmax) ? en : tmax;
        run[e] = tmax;
      }
      const uint32_t before = swg_lds::block_excl_max_u32<NT>(tmax, ws);
#pragma unroll
      for (int e = 0; e < ES; ++e)
        if (q0 + e < nb) EP[q0 + e].y = run[e] > before ? run[e] : before;
    }
    lds_barrier();
    // ---- the sweep of the batch's range of positions [s_first, s_next): one evaluation per interval t, over the part [a, b) of
    // its span that lies in the range.  Among the intervals that intersect [a, b) -- a window of slots: from the first slot whose
    // running maximum of ends exceeds a to the last slot that starts before b -- the BETTER ones (score key, then slot) are
    // walked in start order with the position `reach` up to which they cover [a, b) without a gap; every gap [g, h) is a stretch
    // where t is the top of the active set: t is marked, and every other interval active somewhere in [g, h) is tested against
    // it (overlap fraction above the threshold -> the sticky `overlapped`).  (g is an event position: t's start, the range's first
    // position or a better interval's end; the active set only changes at event positions.)
    // Everything here is bound by instruction issue (one work-group of 16 wavefronts on the CU, and a wavefront pays for every
    // path one of its lanes takes), so the work is sorted before it is done:
    //   classify   one thread per slot: an interval with nobody else in [a, b) -- no earlier interval reaches a (the running
    //              maximum in front of the slot), no later one begins before b (the next slot's start) -- is the top there,
    //              settled on the spot; the others go onto a list (the bucket counters' room)
    //   evaluate   one thread per listed interval: the lanes are full again
    {
      constexpr int ROUND = ES < EVL / NT ? ES : EVL / NT;  // slots per thread and round
      constexpr int EV_STEPS = 12;
      static_assert(ROUND >= 1 && CAP < 65536, "a round's slots (16 bits each) fit the bucket counters' room");
      const uint32_t lane = (uint32_t)tid & 63u;
      // (the work-group waits for its slowest thread at the next barrier, and window lengths are heavy-tailed -- a long
      // interval's window holds every begin inside its span, an interval in the shadow of a long one every slot back to it: a
      // thread gives up beyond EV_STEPS window slots and leaves the interval to a whole wavefront, evaluate_long)
      auto evaluate = [&](const uint32_t t) {
        const uint32_t s = K[t], en = EP[t].x;
        const uint64_t kx = KEY[t];
        const uint32_t a_ = s > s_first ? s : s_first, b_ = (last || en < s_next) ? en : s_next;
        uint32_t lo = t, hi = t;
        while (lo > 0 && EP[lo - 1].y > a_ && t - lo <= (uint32_t)EV_STEPS) --lo;
        while (hi + 1 < nb && K[hi + 1] < b_ && hi - lo <= (uint32_t)EV_STEPS) ++hi;
        if (hi - lo > (uint32_t)EV_STEPS) {
          const uint32_t at = atomicAdd(&sh[6], 1u);
          if (at < (uint32_t)N_LONG) ev_long[at] = (uint16_t)t; else sh[5] = 1u;  // (more long intervals than the list holds: deep data)
          return;
        }
        const uint32_t ts = s, te = en;
        auto stretch = [&](const uint32_t g, const uint32_t h) {
          TOPF[t] = 1;
          if (!ovl_on) return;
          for (uint32_t x = lo; x <= hi; ++x) {
            const uint32_t ex = EP[x].x, sx = K[x];
            if (x != t && ex > g && sx < h && seg_overlap_exceeds(sx, ex, ts, te, thr)) OVLF[x] = 1;
          }
        };
        uint32_t reach = a_;
        for (uint32_t j = lo; j <= hi && reach < b_; ++j) {
          const uint32_t ej = EP[j].x;
          const uint64_t kj = KEY[j];
          if (j != t && ej > reach && (kj < kx || (kj == kx && j < t))) {  // (an interval that is never active has end 0 here)
            const uint32_t sj = K[j];
            if (sj > reach) stretch(reach, sj < b_ ? sj : b_);
            reach = ej;
          }
        }
        if (reach < b_) stretch(reach, b_);
      };
      // the same by a whole wavefront, 64 window slots per step (every lane is passed the same t)
      auto evaluate_long = [&](const uint32_t t, uint32_t* rounds) {
        const uint32_t s = K[t], en = EP[t].x;
        const uint64_t kx = KEY[t];
        const uint32_t a_ = s > s_first ? s : s_first, b_ = (last || en < s_next) ? en : s_next;
        uint32_t lo = t, hi = t;
        for (;;) {  // backwards while the running maximum in front of the slot exceeds a
          const int j = (int)lo - 1 - (int)lane;
          const unsigned long long stop = ~__ballot(j >= 0 && EP[j >= 0 ? j : 0].y > a_);
          const uint32_t c = stop ? (uint32_t)__builtin_ctzll(stop) : 64u;
          lo -= c;
          ++*rounds;
          if (c < 64u) break;
        }
        for (;;) {  // forwards while the next slot starts before b
          const uint32_t j = hi + 1u + lane;
          const unsigned long long stop = ~__ballot(j < nb && K[j < nb ? j : 0u] < b_);
          const uint32_t c = stop ? (uint32_t)__builtin_ctzll(stop) : 64u;
          hi += c;
          ++*rounds;
          if (c < 64u) break;
        }
        const uint32_t ts = s, te = en;
        auto stretch = [&](const uint32_t g, const uint32_t h) {
          if (lane == 0) TOPF[t] = 1;
          if (!ovl_on) return;
          for (uint32_t x0 = lo; x0 <= hi; x0 += 64u) {
            const uint32_t x = x0 + lane;
            if (x <= hi && x != t) {
              const uint32_t ex = EP[x].x, sx = K[x];
              if (ex > g && sx < h && seg_overlap_exceeds(sx, ex, ts, te, thr)) OVLF[x] = 1;
            }
            ++*rounds;
          }
        };
        uint32_t reach = a_;
        for (uint32_t j0 = lo; j0 <= hi && reach < b_; j0 += 64u) {
          const uint32_t j = j0 + lane;
          const bool in = j <= hi && j != t;
          const uint32_t jc = in ? j : t;
          const uint32_t ej = EP[jc].x, sj = K[jc];
          const uint64_t kj = KEY[jc];
          const bool isb = in && ej > a_ && (kj < kx || (kj == kx && j < t));
          // the position the better intervals in front of this lane's slot reach (the sequential walk's `reach` there)
          uint32_t inc = isb ? ej : 0u;
#pragma unroll
          for (int d = 1; d < 64; d <<= 1) {
            const uint32_t y = __shfl_up(inc, d, 64);
            if (lane >= (uint32_t)d) inc = y > inc ? y : inc;
          }
          uint32_t before = __shfl_up(inc, 1, 64);
          if (lane == 0) before = 0u;
          before = before > reach ? before : reach;
          unsigned long long gaps = __ballot(isb && ej > before && sj > before && before < b_);
          while (gaps) {
            const int l = __builtin_ctzll(gaps);
            gaps &= gaps - 1ull;
            const uint32_t g = (uint32_t)__shfl((int)before, l, 64), h0 = (uint32_t)__shfl((int)sj, l, 64);
            stretch(g, h0 < b_ ? h0 : b_);
          }
          const uint32_t all = (uint32_t)__shfl((int)inc, 63, 64);
          reach = all > reach ? all : reach;
          ++*rounds;
        }
        if (reach < b_) stretch(reach, b_);
      };
#pragma unroll 1
      for (int e0 = 0; e0 < ES; e0 += ROUND) {
        if ((uint32_t)e0 * NT >= nb) break;  // (uniform)
        if (tid == 0) {
          sh[6] = 0u;
          sh[7] = 0u;
        }
        lds_barrier();
        // classify
        {
          const uint32_t t_ev = swg_fresh_tid();
          uint32_t ent[ROUND];
          uint32_t n_ev = 0;
#pragma unroll
          for (int e = 0; e < ROUND; ++e) {
            const uint32_t p = t_ev + (uint32_t)(e0 + e) * NT;
            const bool in = e0 + e < ES && p < nb;
            const uint32_t pc = in ? p : 0u;
            const uint32_t s = K[pc], en = EP[pc].x;
            const uint32_t k_next = pc + 1 < nb ? K[pc + 1] : 0xffffffffu;
            const uint32_t pm_prev = pc ? EP[pc - 1].y : 0u;
            const uint32_t a_ = s > s_first ? s : s_first, b_ = (last || en < s_next) ? en : s_next;
            const bool has = in && en > s && b_ > a_;
            const bool alone = pm_prev <= a_ && (pc + 1 >= nb || k_next >= b_);
            if (has && alone) TOPF[p] = 1;
            ent[e] = (has && !alone) ? p : NONE;
            n_ev += ent[e] != NONE ? 1u : 0u;
          }
          // the wavefront's intervals onto the list: one atomic per wavefront
          uint32_t inc = n_ev;
#pragma unroll
          for (int d = 1; d < 64; d <<= 1) {
            const uint32_t x = __shfl_up(inc, d, 64);
            if (lane >= (uint32_t)d) inc += x;
          }
          uint32_t base = 0;
          if (lane == 63u && inc) base = atomicAdd(&sh[7], inc);
          base = (uint32_t)__shfl((int)base, 63, 64) + inc - n_ev;
#pragma unroll
          for (int e = 0; e < ROUND; ++e)
            if (ent[e] != NONE) ev_list[base++] = (uint16_t)ent[e];
        }
        lds_barrier();
        // evaluate
        {
          const uint32_t n_list = sh[7];
          for (uint32_t i = swg_fresh_tid(); i < n_list; i += NT) evaluate(ev_list[i]);
        }
        lds_barrier();
        {
          const uint32_t n_long = sh[6] < (uint32_t)N_LONG ? sh[6] : (uint32_t)N_LONG;
          uint32_t rounds = 0;
          for (uint32_t i = (uint32_t)__builtin_amdgcn_readfirstlane(tid >> 6); i < n_long; i += NT / 64) {
            evaluate_long((uint32_t)__builtin_amdgcn_readfirstlane((int)ev_long[i]), &rounds);
            if (rounds > SWEEP_BUDGET) {
              sh[5] = 1u;
              break;
            }
          }
        }
      }
    }
    lds_barrier();
    if (sh[5]) {
      if (tid == 0) atomicOr(&counters[4], 2u);
      return false;
    }
    // ---- retire (the flag goes to the record) or carry over
    {
      const uint32_t q0 = swg_fresh_tid() * ES;
      uint32_t ck[ES], ce[ES], cw[ES], cf[ES];  // (cw: the interval's ID, cf: its flags so far)
      uint64_t cs[ES];
      uint32_t cmask = 0;
#pragma unroll
      for (int e = 0; e < ES; ++e) {
        const uint32_t q = q0 + e;
        ck[e] = ce[e] = cw[e] = cf[e] = 0;
        cs[e] = 0;
        if (q >= nb) continue;
        const uint32_t s = K[q], en = EP[q].x, f = (TOPF[q] ? F_TOP : 0u) | (OVLF[q] ? F_OVL : 0u), place = ID[q];
        if (!last && en > s && en > s_next) {  // (reaches into the next batch's range)
          cmask |= 1u << e;
          ck[e] = s;
          ce[e] = en;
          cw[e] = place;
          cf[e] = f;
          cs[e] = KEY[q];
        } else {
          const uint32_t ix = rec_of(place);
          keep[ix] = ((f & F_TOP) && !(f & F_OVL) && (!and_with || and_with[ix])) ? 1 : 0;
        }
      }
      if (!last) {
        uint32_t tot;
        lds_barrier();
        const uint32_t at = block_excl_sum<NT>((uint32_t)__popc(cmask), ws, &tot);
        lds_barrier();
        tot = (uint32_t)__builtin_amdgcn_readfirstlane((int)tot);
        if (tot > (uint32_t)CMAX) {
          if (tid == 0) {
            atomicOr(&counters[4], 2u);
            atomicAdd(&counters[7], 1u);
          }
          return false;
        }
#pragma unroll
        for (int e = 0; e < ES; ++e)
          if ((cmask >> e) & 1u) {
            const uint32_t d = at + (uint32_t)__popc(cmask & ((1u << e) - 1u));
            K[d] = ck[e];
            EP[d].x = ce[e];
            KEY[d] = cs[e];
            ID[d] = (IDT)cw[e];
            TOPF[d] = (uint8_t)(cf[e] & F_TOP);
            OVLF[d] = (uint8_t)(cf[e] & F_OVL);
          }
        nc = tot;
      }
    }
    s_first = s_next;
    lds_barrier();
  }
  return true;
}
template <int NT, int ES, int NBK>
constexpr size_t seg_sweep_lds_bytes() {
  // K, EP, KEY, ID, F per slot; bucket counters; b_lo, ws, scalars; slack for the alignment of each piece
  return (size_t)NT * ES * (4 + 8 + 8 + 2 + 2) + (size_t)NBK * 4 + 17 * 4 + (size_t)(NT / 64 + 1) * 4 + 16 * 4 + (size_t)NT * 2 + 64;
}

template <int NT, int ES, int ER, int NBK, int NBIN, int CMAX>
__device__ __forceinline__ void seg_sweep_body(const SegSweepArgs& A, const uint32_t sg, char* lds_raw) {
  constexpr int CAP = NT * ES, CAPN = CAP - CMAX, MAXB = 16, H = 8;
  static_assert(ER <= 32 && ER % H == 0 && ES % 2 == 0 && NT * ER <= 65536 && CAP < 0xffff, "record masks are 32 bits wide, places and ranks 16");
  static_assert(NBIN >= 1 && NBIN <= 4096 && NBK % NT == 0 && CMAX < CAP, "bins, bucket counters per thread");
  static_assert((size_t)NBIN * 4 <= (size_t)CAP * 8 && CAP % 4 == 0, "the bins borrow EP; the flags are read as words");
  constexpr size_t O_K = 0, O_EP = O_K + (size_t)CAP * 4, O_KEY = O_EP + (size_t)CAP * 8, O_ID = O_KEY + (size_t)CAP * 8,
                   O_F = O_ID + (size_t)CAP * 2, O_CNT = lds_align_up(O_F + (size_t)CAP * 2, 4), O_BLO = O_CNT + (size_t)NBK * 4,
                   O_WS = O_BLO + (size_t)(MAXB + 1) * 4, O_SH = O_WS + (size_t)(NT / 64 + 1) * 4, O_LONG = O_SH + 8 * 4;
  constexpr int N_LONG = NT;  // intervals of one round that get a wavefront of their own
  static_assert(O_LONG + (size_t)N_LONG * 2 <= seg_sweep_lds_bytes<NT, ES, NBK>(), "LDS block of the work-group");
  uint32_t* const K = reinterpret_cast<uint32_t*>(lds_raw + O_K);
  uint2* const EP = reinterpret_cast<uint2*>(lds_raw + O_EP);  // x: end, y: running maximum of the ends up to and including this slot
  uint64_t* const KEY = reinterpret_cast<uint64_t*>(lds_raw + O_KEY);
  uint16_t* const ID = reinterpret_cast<uint16_t*>(lds_raw + O_ID);  // the interval's place in the segment's list
  uint8_t* const TOPF = reinterpret_cast<uint8_t*>(lds_raw + O_F);  // (two byte arrays: the events set them by plain stores)
  uint8_t* const OVLF = TOPF + CAP;
  uint32_t* const cnt = reinterpret_cast<uint32_t*>(lds_raw + O_CNT);
  uint32_t* const bins = reinterpret_cast<uint32_t*>(lds_raw + O_EP);  // (only before the first batch)
  uint32_t* const b_lo = reinterpret_cast<uint32_t*>(lds_raw + O_BLO);
  uint32_t* const ws = reinterpret_cast<uint32_t*>(lds_raw + O_WS);
  uint32_t* const sh = reinterpret_cast<uint32_t*>(lds_raw + O_SH);  // [0] kmin, [1] kmax, [2] batches, [3] bad, [4] next batch's first start, [5] deep, [6] a round's intervals with long windows, [7] a round's listed intervals
  const int tid = threadIdx.x;
#ifdef SWG_SEG_TIMING
  unsigned long long st_last = wall_clock64();
#endif
  const uint32_t a = A.seg_a[sg], n_live = A.seg_e[sg] - a;
  const uint32_t rbase = (uint32_t)__builtin_amdgcn_readfirstlane((int)A.seg_base[sg]);
  const bool in_place = rbase != NONE;
  const uint32_t m = in_place ? A.seg_len[sg] : n_live;
  const uint32_t* __restrict__ c_perm = A.perm + a;
  if (tid == 0) {
    sh[0] = 0xffffffffu;
    sh[1] = 0u;
    sh[3] = 0u;
    sh[5] = 0u;
  }
  if (n_live > (uint32_t)CAP)
    for (int b = tid; b < NBIN; b += NT) bins[b] = 0;
  lds_barrier();
  auto fresh_tid = [&]() -> uint32_t {
    uint32_t t = (uint32_t)tid;
    asm volatile("" : "+v"(t));
    return t;
  };
  uint32_t tid_v = (uint32_t)tid, m_v = m;
  auto rec_index = [&](int e) -> uint32_t {
    const uint32_t li = tid_v + (uint32_t)e * NT;
    const uint32_t lc = li < m_v ? li : 0u;
    if (in_place) return rbase + lc;
    return c_perm[lc];
  };
  auto rec_of_place = [&](uint32_t place) -> uint32_t { return in_place ? rbase + place : c_perm[place]; };
  // ---- the key range
  uint32_t in_mask = 0;
  {
    uint32_t kmin = 0xffffffffu, kmax = 0;
#pragma unroll
    for (int g = 0; g < ER; g += H) {
      if ((uint32_t)g * NT >= m) continue;
      uint32_t ixv[H], qv[H];
      uint8_t av[H];
#pragma unroll
      for (int e = 0; e < H; ++e) ixv[e] = rec_index(g + e);
#pragma unroll
      for (int e = 0; e < H; ++e) {
        qv[e] = A.start[ixv[e]];
        av[e] = (in_place && A.alive) ? A.alive[ixv[e]] : (uint8_t)1;
      }
#pragma unroll
      for (int e = 0; e < H; ++e) {
        const uint32_t li = (uint32_t)tid + (uint32_t)(g + e) * NT;
        if (li >= m || !av[e]) continue;
        if (n_live == 1u) A.keep[ixv[e]] = (!A.and_with || A.and_with[ixv[e]]) ? 1 : 0;  // returned whole (plane_sweep_exact.rs:274-276)
        in_mask |= 1u << (g + e);
        kmin = qv[e] < kmin ? qv[e] : kmin;
        kmax = qv[e] > kmax ? qv[e] : kmax;
      }
      asm volatile("" ::: "memory");
    }
    if (n_live == 1u) return;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const uint32_t x = __shfl_xor(kmin, o, 64), y = __shfl_xor(kmax, o, 64);
      kmin = x < kmin ? x : kmin;
      kmax = y > kmax ? y : kmax;
    }
    if ((tid & 63) == 0) {
      atomicMin(&sh[0], kmin);
      atomicMax(&sh[1], kmax);
    }
  }
  lds_barrier();
  ST_STAMP(1);
  const uint32_t k_lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)sh[0]);
  const float scale = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int((float)NBIN / ((float)(sh[1] - sh[0]) + 1.0f))));
  auto fine_of = [&](uint32_t k, int shift, uint32_t first, uint32_t* coarse) -> uint32_t {  // (seg_sort_body's)
    const float f = (float)(k - k_lo) * scale;
    uint32_t g = (uint32_t)(f * 4096.0f);
    if ((g >> 12) > (uint32_t)NBIN - 1u) g = (((uint32_t)NBIN - 1u) << 12) | 0xfffu;
    *coarse = g >> 12;
    const uint32_t b = (g >> shift) - first;
    return b < (uint32_t)NBK ? b : (uint32_t)NBK - 1u;
  };
  uint32_t n_batches = 1;
  if (n_live > (uint32_t)CAP) {
#pragma unroll
    for (int g = 0; g < ER; g += H) {
      if ((uint32_t)g * NT >= m) continue;
      uint32_t ixv[H], qv[H];
#pragma unroll
      for (int e = 0; e < H; ++e) ixv[e] = rec_index(g + e);
#pragma unroll
      for (int e = 0; e < H; ++e) qv[e] = A.start[ixv[e]];
#pragma unroll
      for (int e = 0; e < H; ++e)
        if ((in_mask >> (g + e)) & 1u) {
          uint32_t cb;
          (void)fine_of(qv[e], 12, 0u, &cb);
          atomicAdd(&bins[cb], 1u);
        }
      asm volatile("" ::: "memory");
    }
    lds_barrier();
    swg_lds::bins_to_offsets<NT, NBIN>(bins, ws);
    lds_barrier();
    if (tid == 0) {  // (a batch leaves room for the carried intervals)
      const uint32_t nb = swg_lds::plan_batches<NBIN>(bins, n_live, (uint32_t)CAPN, (uint32_t)MAXB, b_lo);
      sh[2] = nb;
      if (nb == 0) sh[3] = 1;
    }
    lds_barrier();
    if (sh[3]) {
      if (tid == 0) atomicOr(&A.counters[4], 1u);
      return;
    }
    n_batches = (uint32_t)__builtin_amdgcn_readfirstlane((int)sh[2]);
  }
  ST_STAMP(2);
  const bool ovl_on = A.thr < 1.0;
  uint32_t nc = 0;  // carried intervals, in slots [0, nc)
  uint32_t s_first = 0;  // the batch answers for the positions [s_first, s_next): its first start on (the first batch: from 0)
  for (uint32_t bt = 0; bt < n_batches; ++bt) {
    {  // (see pair_sort_body: keeps the loads of every batch inside the loop)
      uint32_t m_l = m_v;
      asm volatile("" : "+v"(tid_v), "+v"(m_l), "+v"(in_mask));
      m_v = (uint32_t)__builtin_amdgcn_readfirstlane((int)m_l);
    }
    const bool last = bt + 1 == n_batches;
    const uint32_t bin_lo = n_batches > 1 ? (uint32_t)__builtin_amdgcn_readfirstlane((int)b_lo[bt]) : 0u;
    const uint32_t bin_hi = n_batches > 1 ? (uint32_t)__builtin_amdgcn_readfirstlane((int)b_lo[bt + 1]) : (uint32_t)NBIN;
    int shift = 0;
    while ((((bin_hi - bin_lo) << 12) >> shift) > (uint32_t)NBK) ++shift;
    const uint32_t first = (bin_lo << 12) >> shift;
    for (int b = tid; b < NBK; b += NT) cnt[b] = 0;
    if (tid == 0) sh[4] = 0xffffffffu;
    lds_barrier();
    // the sort's scratch (scatter place -> list place, scatter place -> rank) borrows the score keys' room behind the carried ones
    uint32_t* const Kb = K + nc;
    uint16_t* const I = reinterpret_cast<uint16_t*>(KEY + nc);
    uint16_t* const RR = I + (last && n_batches == 1 ? CAP : CAPN);
    // ---- count (and the first start of the batches behind this one)
    uint32_t batch_mask = 0;
    {
      uint32_t s_next = 0xffffffffu;
#pragma unroll
      for (int g = 0; g < ER; g += H) {
        if ((uint32_t)g * NT >= m) continue;
        uint32_t ixv[H], qv[H];
#pragma unroll
        for (int e = 0; e < H; ++e) ixv[e] = rec_index(g + e);
#pragma unroll
        for (int e = 0; e < H; ++e) qv[e] = A.start[ixv[e]];
#pragma unroll
        for (int e = 0; e < H; ++e)
          if ((in_mask >> (g + e)) & 1u) {
            uint32_t cb;
            const uint32_t fb = fine_of(qv[e], shift, first, &cb);
            if (cb >= bin_lo && cb < bin_hi) {
              batch_mask |= 1u << (g + e);
              atomicAdd(&cnt[fb], 1u);
            } else if (cb >= bin_hi) {
              s_next = qv[e] < s_next ? qv[e] : s_next;
            }
          }
        asm volatile("" ::: "memory");
      }
      if (!last) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
          const uint32_t x = __shfl_xor(s_next, o, 64);
          s_next = x < s_next ? x : s_next;
        }
        if ((tid & 63) == 0 && s_next != 0xffffffffu) atomicMin(&sh[4], s_next);
      }
    }
    lds_barrier();
    ST_STAMP(3);
    const uint32_t mb = swg_lds::bucket_offsets<NT, NBK>(cnt, ws);
    lds_barrier();
    // ---- scatter
    uint32_t slotw[ER / 2];
#pragma unroll
    for (int j = 0; j < ER / 2; ++j) slotw[j] = 0;
    const uint32_t t_sc = fresh_tid();
#pragma unroll
    for (int g = 0; g < ER; g += H) {
      if ((uint32_t)g * NT >= m) continue;
      uint32_t ixv[H], qv[H];
#pragma unroll
      for (int e = 0; e < H; ++e) ixv[e] = rec_index(g + e);
#pragma unroll
      for (int e = 0; e < H; ++e) qv[e] = A.start[ixv[e]];
#pragma unroll
      for (int e = 0; e < H; ++e)
        if ((batch_mask >> (g + e)) & 1u) {
          uint32_t cb;
          const uint32_t fb = fine_of(qv[e], shift, first, &cb);
          const uint32_t pos = atomicAdd(&cnt[fb], 1u);
          Kb[pos] = qv[e];
          I[pos] = (uint16_t)(t_sc + (uint32_t)(g + e) * NT);
          slotw[(g + e) / 2] |= pos << (16 * ((g + e) & 1));
        }
      asm volatile("" ::: "memory");
    }
    lds_barrier();
    ST_STAMP(4);
    swg_lds::rank_buckets<NT, ES, uint16_t, false>(Kb, I, RR, cnt, mb, fresh_tid(), [&](uint32_t, uint32_t key) {
      uint32_t cb;
      return fine_of(key, shift, first, &cb);
    });
    swg_lds::slots_to_ranks<ER>(slotw, batch_mask, RR);
    lds_barrier();
    ST_STAMP(5);
    // ---- the ends, the score keys, the list places and fresh flags at the sorted slots (by the thread that owns the record)
    {
      tid_v = fresh_tid();
      const uint32_t t_pl = fresh_tid();
#pragma unroll
      for (int g = 0; g < ER; g += H) {
        if ((uint32_t)g * NT >= m) continue;
        uint32_t ixv[H], ev[H];
        uint64_t sv[H];
#pragma unroll
        for (int e = 0; e < H; ++e) ixv[e] = rec_index(g + e);
#pragma unroll
        for (int e = 0; e < H; ++e) {
          ev[e] = A.end[ixv[e]];
          sv[e] = A.score[ixv[e]];
        }
#pragma unroll
        for (int e = 0; e < H; ++e)
          if ((batch_mask >> (g + e)) & 1u) {
            const uint32_t r = nc + ((slotw[(g + e) / 2] >> (16 * ((g + e) & 1))) & 0xffffu);
            EP[r].x = ev[e];
            KEY[r] = sv[e];
            ID[r] = (uint16_t)(t_pl + (uint32_t)(g + e) * NT);
            TOPF[r] = 0;
            OVLF[r] = 0;
          }
        asm volatile("" ::: "memory");
      }
    }
    lds_barrier();
    ST_STAMP(6);
    {
      const uint32_t s_next = (uint32_t)__builtin_amdgcn_readfirstlane((int)sh[4]);  // (the last batch: no bound)
      if (!sweep_batch<NT, ES, CMAX, 2 * NBK, N_LONG>(K, EP, KEY, ID, TOPF, OVLF, reinterpret_cast<uint16_t*>(cnt), reinterpret_cast<uint16_t*>(lds_raw + O_LONG), ws,
                                                      sh, nc, mb, last, s_first, s_next, A.thr, A.and_with, A.keep, A.counters, rec_of_place))
        return;
    }
    ST_STAMP(9);
  }
}
template <int NT, int ES, int ER, int NBK, int NBIN, int CMAX>
__global__ __launch_bounds__(NT) void seg_sweep_kernel(SegSweepArgs A) {
  __shared__ __attribute__((aligned(16))) char raw[seg_sweep_lds_bytes<NT, ES, NBK>()];
  seg_sweep_body<NT, ES, ER, NBK, NBIN, CMAX>(A, A.list[blockIdx.x], raw);
}
// the 1,024-thread class and, in front of it in the same launch (so that the bulk of the work runs beside them), the sorts of
// the longest segments for the tile kernels
constexpr size_t SEG_SWEEP_BIG_LDS = seg_sweep_lds_bytes<1024, 6, 2048>() > SEG_XL_LDS ? seg_sweep_lds_bytes<1024, 6, 2048>() : SEG_XL_LDS;
static_assert(SEG_SWEEP_BIG_LDS <= 160 * 1024, "LDS of a CU");
__global__ __launch_bounds__(1024) void seg_sweep_big_kernel(SegSweepArgs W, SegSortArgs A, const uint32_t* __restrict__ list_xl, uint32_t n_xl) {
  __shared__ __attribute__((aligned(16))) char raw[SEG_SWEEP_BIG_LDS];
  if (blockIdx.x < n_xl)
    seg_sort_xl_body(A, list_xl[blockIdx.x], raw);
  else
    seg_sweep_body<1024, 6, 32, 2048, 1024, 512>(W, W.list[blockIdx.x - n_xl], raw);
}


// ---- the k = 1 sweep of a segment over its SORTED begins, streamed through LDS (round 6; the default behind seg_sort) -------------
// seg_sort leaves every segment's begins in (start, record index) order in memory -- S, I, E, KEY -- for the tile kernels
// (swg_sweep.hip), which cut the whole axis into tiles of 128 begins and need every interval that reaches into a tile routed
// to it first.  A segment's sweep needs none of that: a work-group takes the segment's stretch through LDS chunk by chunk
// (coalesced loads), the intervals that reach beyond a chunk's positions carried over in front of the next one, and evaluates
// each chunk with sweep_batch -- the same code the fused kernel runs behind its own sort (seg_sweep_body), but with nothing in
// LDS except the sweep's own state: several work-groups share a CU, and a segment of any length streams through.  keep[] is
// written directly (the other axis' result folded in); deep data raises counters[4] bit 1 and the axis goes on to the tile
// kernels with the arrays it already has.
struct SegStreamArgs {
  const uint32_t *seg_a, *seg_e;
  const uint32_t* class_list;  // [4][n_runs]
  uint32_t n_runs;
  uint32_t n_cls[4];
  const uint64_t* S;  // composite: (segment + 1) << pos_bits | start
  const uint32_t* I;
  const uint32_t* E;
  const uint64_t* KEY;
  uint32_t n_dead;
  uint32_t pos_mask;  // the start's bits of S
  const uint8_t* and_with;
  uint8_t* keep;
  double thr;
  uint32_t* counters;
};
template <int NT, int ES>
constexpr size_t seg_stream_lds_bytes() {
  // K, EP, KEY, ID, two flag bytes and a list entry per slot; the long windows' list; ws, scalars; slack
  return (size_t)NT * ES * (4 + 8 + 8 + 4 + 2 + 2 + 1) + (size_t)(NT / 64 + 1) * 4 + 16 * 4 + 64;
}
template <int NT, int ES, int CMAX>
__device__ __forceinline__ void seg_stream_body(const SegStreamArgs& A, const uint32_t sg, char* lds_raw) {
  constexpr int CAP = NT * ES, CAPN = CAP - CMAX, N_LONG = CAP / 2;
  static_assert(CMAX < CAP && CAP % 4 == 0, "room for the new intervals of a chunk");
  constexpr size_t O_K = 0, O_EP = O_K + (size_t)CAP * 4, O_KEY = O_EP + (size_t)CAP * 8, O_ID = O_KEY + (size_t)CAP * 8, O_F = O_ID + (size_t)CAP * 4,
                   O_LIST = O_F + (size_t)CAP * 2, O_LONG = O_LIST + (size_t)CAP * 2, O_WS = lds_align_up(O_LONG + (size_t)N_LONG * 2, 4),
                   O_SH = O_WS + (size_t)(NT / 64 + 1) * 4;
  static_assert(O_SH + 8 * 4 <= seg_stream_lds_bytes<NT, ES>(), "LDS block of the work-group");
  uint32_t* const K = reinterpret_cast<uint32_t*>(lds_raw + O_K);
  uint2* const EP = reinterpret_cast<uint2*>(lds_raw + O_EP);
  uint64_t* const KEY = reinterpret_cast<uint64_t*>(lds_raw + O_KEY);
  uint32_t* const ID = reinterpret_cast<uint32_t*>(lds_raw + O_ID);  // the record's own index
  uint8_t* const TOPF = reinterpret_cast<uint8_t*>(lds_raw + O_F);
  uint8_t* const OVLF = TOPF + CAP;
  uint16_t* const ev_list = reinterpret_cast<uint16_t*>(lds_raw + O_LIST);
  uint16_t* const ev_long = reinterpret_cast<uint16_t*>(lds_raw + O_LONG);
  uint32_t* const ws = reinterpret_cast<uint32_t*>(lds_raw + O_WS);
  uint32_t* const sh = reinterpret_cast<uint32_t*>(lds_raw + O_SH);  // [4] the chunk's intervals that start with the next chunk, [5] deep, [6], [7]: sweep_batch's
  const int tid = threadIdx.x;
  const uint32_t a = A.seg_a[sg], n_live = A.seg_e[sg] - a;
  const size_t base = (size_t)A.n_dead + a;
  if (n_live <= 1u) {  // returned whole by the reference (plane_sweep_exact.rs:274-276)
    if (tid == 0 && n_live) {
      const uint32_t ix = A.I[base];
      A.keep[ix] = (!A.and_with || A.and_with[ix]) ? 1 : 0;
    }
    return;
  }
  if (tid == 0) sh[5] = 0u;
  uint32_t nc = 0, s_first = 0, pos = 0;
  while (pos < n_live) {
    const uint32_t take = n_live - pos < (uint32_t)CAPN ? n_live - pos : (uint32_t)CAPN;
    bool last = pos + take == n_live;
    // the chunk must not end inside a run of equal starts: its intervals that start where the NEXT chunk starts are left to it
    const uint32_t s_after = last ? 0u : (uint32_t)A.S[base + pos + take] & A.pos_mask;
    if (tid == 0) sh[4] = 0u;
    lds_barrier();
    uint32_t tail = 0;
    const uint32_t t_ld = swg_fresh_tid();
#pragma unroll
    for (int e = 0; e < ES; ++e) {
      const uint32_t j = t_ld + (uint32_t)e * NT;
      if (j < take) {
        const size_t g = base + pos + j;
        const uint32_t st = (uint32_t)A.S[g] & A.pos_mask;
        K[nc + j] = st;
        EP[nc + j].x = A.E[g];
        KEY[nc + j] = A.KEY[g];
        ID[nc + j] = A.I[g];
        TOPF[nc + j] = 0;
        OVLF[nc + j] = 0;
        tail += (!last && st == s_after) ? 1u : 0u;
      }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) tail += __shfl_xor(tail, o, 64);
    if ((tid & 63) == 0 && tail) atomicAdd(&sh[4], tail);
    lds_barrier();
    const uint32_t n_tail = (uint32_t)__builtin_amdgcn_readfirstlane((int)sh[4]);
    if (n_tail >= take) {  // a run of equal starts longer than a chunk: the tile kernels' case
      if (tid == 0) atomicOr(&A.counters[4], 2u);
      return;
    }
    const uint32_t mb = take - n_tail;
    const uint32_t s_next = s_after;
    if (!sweep_batch<NT, ES, CMAX, CAP, N_LONG>(K, EP, KEY, ID, TOPF, OVLF, ev_list, ev_long, ws, sh, nc, mb, last, s_first, s_next, A.thr, A.and_with,
                                                A.keep, A.counters, [](uint32_t ix) { return ix; }))
      return;
    pos += mb;
  }
}
// the segments of more than SEG_S_MAX places, the longest ones first (blockIdx -> class 3, 2, 1)
template <int NT, int ES, int CMAX>
__global__ __launch_bounds__(NT) void seg_stream_kernel(SegStreamArgs A) {
  __shared__ __attribute__((aligned(16))) char raw[seg_stream_lds_bytes<NT, ES>()];
  uint32_t b = blockIdx.x;
  int c = 3;
  while (c > 1 && b >= A.n_cls[c]) {
    b -= A.n_cls[c];
    --c;
  }
  seg_stream_body<NT, ES, CMAX>(A, A.class_list[(size_t)c * A.n_runs + b], raw);
}
template <int NT, int ES>
__global__ __launch_bounds__(NT) void seg_stream_small_kernel(SegStreamArgs A) {
  __shared__ __attribute__((aligned(16))) char raw[seg_stream_lds_bytes<NT, ES>()];
  seg_stream_body<NT, ES, 0>(A, A.class_list[blockIdx.x], raw);
}

}  // namespace

}  // namespace swg_seg

// live records per run of a pair-grouped input (alive == nullptr: every record), for both axes of a mapping sweep
int swg_seg_run_alive(swg_ctx* ctx, const void* runs, uint32_t n_runs, const uint8_t* alive, uint32_t* run_alive) {
  using namespace swg_seg;
  if (n_runs == 0) return SWG_OK;
  const unsigned g = (n_runs + 3) / 4 < (uint32_t)ctx->num_cu * 32u ? (n_runs + 3) / 4 : (unsigned)ctx->num_cu * 32u;
  SWG_LAUNCH(ctx, "seg_run_alive", run_alive_kernel<<<g, 256, 0, ctx->stream>>>(n_runs, static_cast<const Run*>(runs), alive, run_alive));
  SWG_KERNEL_CHECK(ctx);
  return SWG_OK;
}

namespace swg_seg {
namespace {
// The segments of a sweep axis over the runs of a pair-grouped input, on the device; the host knows their number per size class.
struct SegPlan {
  uint32_t *seg_a, *seg_e, *seg_base, *seg_len, *class_list, *perm, *counters, *xl_len;
  uint64_t* seg_id;
  uint32_t ncls[4];
  uint64_t n_seg, n_alive;
};
// *ok = 0: not this path's input (the caller sorts the general way).  With xl_len the live records of every segment of the
// longest class are left there (0 for the others), for the caller's offsets among those segments alone.
int seg_plan_make(swg_ctx* ctx, const swg_axis_input& in, bool want_xl_len, SegPlan* P, int* ok) {
  *ok = 0;
  const uint64_t n = in.n;
  const uint32_t n_runs = in.n_seg_runs;
  const int a_bits = swg_bits_for(n - 1) ? swg_bits_for(n - 1) : 1;
  const bool count_known = in.n_alive != ~0ull;  // (~0: the live records' number is the runs' total, read back with the plan)
  if (in.seg_bits + a_bits > 64 || n >= (uint64_t(1) << 31) || (count_known && in.n_alive > n)) return SWG_OK;
  hipStream_t st = ctx->stream;
  const Run* runs = static_cast<const Run*>(in.seg_runs);
  uint64_t* key = swg_alloc<uint64_t>(ctx, n_runs);
  uint64_t* key2 = swg_alloc<uint64_t>(ctx, n_runs);
  uint32_t* val = swg_alloc<uint32_t>(ctx, n_runs);
  uint32_t* val2 = swg_alloc<uint32_t>(ctx, n_runs);
  uint32_t* c = swg_alloc<uint32_t>(ctx, n_runs + 1);
  uint32_t* f = swg_alloc<uint32_t>(ctx, n_runs + 1);
  uint32_t* off = swg_alloc<uint32_t>(ctx, n_runs + 1);
  uint32_t* slot = swg_alloc<uint32_t>(ctx, n_runs + 1);
  P->seg_a = swg_alloc<uint32_t>(ctx, n_runs);
  P->seg_e = swg_alloc<uint32_t>(ctx, n_runs);
  P->seg_id = swg_alloc<uint64_t>(ctx, n_runs);
  P->seg_base = swg_alloc<uint32_t>(ctx, n_runs);
  P->seg_len = swg_alloc<uint32_t>(ctx, n_runs);
  P->class_list = swg_alloc<uint32_t>(ctx, (size_t)4 * n_runs);
  P->xl_len = want_xl_len ? swg_alloc<uint32_t>(ctx, (size_t)n_runs + 1) : nullptr;
  P->perm = swg_alloc<uint32_t>(ctx, (count_known ? in.n_alive : n) + 1);
  uint64_t* d_tot = swg_alloc<uint64_t>(ctx, 2);
  P->counters = swg_alloc<uint32_t>(ctx, 8);
  SWG_CHECK_ARENA(ctx);
  SWG_HIP(ctx, hipMemsetAsync(P->counters, 0, 8 * sizeof(uint32_t), st));
  SWG_HIP(ctx, hipMemsetAsync(d_tot, 0, 2 * sizeof(uint64_t), st));
  SWG_LAUNCH(ctx, "seg_run_key", run_key_kernel<<<nblk(n_runs), EW, 0, st>>>(n_runs, runs, in.seg_a, in.seg_b, in.seg_table, in.seg_mul, in.seg, a_bits, key, val));
  SWG_KERNEL_CHECK(ctx);
  SWG_TRY(swg_radix_sort_pairs(ctx, &key, &val, &key2, &val2, n_runs, 0, in.seg_bits + a_bits));
  SWG_LAUNCH(ctx, "seg_flags", seg_flags_kernel<<<nblk(n_runs), EW, 0, st>>>(n_runs, key, val, a_bits, in.seg_run_alive, c, f));
  SWG_KERNEL_CHECK(ctx);
  SWG_TRY(swg_exclusive_scan_u32(ctx, c, off, n_runs, d_tot));
  SWG_TRY(swg_exclusive_scan_u32(ctx, f, slot, n_runs, d_tot + 1));
  SWG_LAUNCH(ctx, "seg_bounds", seg_bounds_kernel<<<nblk(n_runs), EW, 0, st>>>(n_runs, key, a_bits, c, f, off, slot, P->seg_a, P->seg_e, P->seg_id, runs, val, P->seg_base, P->seg_len));
  SWG_KERNEL_CHECK(ctx);
  SWG_LAUNCH(ctx, "seg_class", seg_class_kernel<<<nblk(n_runs), EW, 0, st>>>(d_tot + 1, n_runs, P->seg_a, P->seg_e, P->seg_base, P->seg_len, P->class_list, P->counters, P->xl_len));
  SWG_KERNEL_CHECK(ctx);
  {
    const unsigned pb = (n_runs + 3) / 4 < (uint32_t)ctx->num_cu * 32u ? (n_runs + 3) / 4 : (unsigned)ctx->num_cu * 32u;
    SWG_LAUNCH(ctx, "seg_perm", seg_perm_kernel<<<pb, 256, 0, st>>>(n_runs, runs, val, off, f, in.alive, P->perm));
    SWG_KERNEL_CHECK(ctx);
  }
  uint64_t h[6];
  {
    // one read-back: the live total (a check), the segments, the segments per size class, the too-long flag
    uint64_t* d_all = swg_alloc<uint64_t>(ctx, 6);
    SWG_CHECK_ARENA(ctx);
    SWG_HIP(ctx, hipMemsetAsync(d_all, 0, 6 * sizeof(uint64_t), st));
    SWG_HIP(ctx, hipMemcpyAsync(d_all, d_tot, 2 * sizeof(uint64_t), hipMemcpyDeviceToDevice, st));
    SWG_HIP(ctx, hipMemcpyAsync(d_all + 2, P->counters, 5 * sizeof(uint32_t), hipMemcpyDeviceToDevice, st));
    SWG_TRY(swg_read_scalars(ctx, d_all, h, 5));
  }
  P->ncls[0] = (uint32_t)h[2];
  P->ncls[1] = (uint32_t)(h[2] >> 32);
  P->ncls[2] = (uint32_t)h[3];
  P->ncls[3] = (uint32_t)(h[3] >> 32);
  P->n_seg = h[1];
  static const bool dbg = getenv("SWG_DEBUG") != nullptr;
  P->n_alive = count_known ? in.n_alive : h[0];
  if (h[0] != P->n_alive || P->n_alive > n) {  // (the caller's count of live records and the runs' disagree: not this path's input)
    if (dbg) fprintf(stderr, "[swg] segment sort: %llu live records in the runs, %llu expected: the general sort takes the axis\n",
                     (unsigned long long)h[0], (unsigned long long)in.n_alive);
    return SWG_OK;
  }
  if ((uint32_t)h[4] & 1u) {  // (a segment beyond SEG_XL_MAX places: its batch passes are quadratic in its size)
    if (dbg) fprintf(stderr, "[swg] segment sort: a segment of more than %u places: the general sort takes the axis\n", SEG_XL_MAX);
    return SWG_OK;
  }
  if (dbg)
    fprintf(stderr, "[swg] segment sort: %llu segments over %u runs (%u / %u / %u / %u by size class)\n", (unsigned long long)h[1], n_runs, P->ncls[0],
            P->ncls[1], P->ncls[2], P->ncls[3]);
  *ok = 1;
  return SWG_OK;
}
// counters[4] after the segment launches: bit 0 a coarse bin denser than an LDS batch, bit 1 the resident sweep met deep data
int seg_flags_read(swg_ctx* ctx, const SegPlan& P, uint32_t* flags) {
  uint64_t fl = 0;
  uint64_t* d_f = swg_alloc<uint64_t>(ctx, 1);
  SWG_CHECK_ARENA(ctx);
  SWG_HIP(ctx, hipMemsetAsync(d_f, 0, sizeof(uint64_t), ctx->stream));
  SWG_HIP(ctx, hipMemcpyAsync(d_f, P.counters + 4, sizeof(uint32_t), hipMemcpyDeviceToDevice, ctx->stream));
  SWG_TRY(swg_read_scalars(ctx, d_f, &fl, 1));
  *flags = (uint32_t)fl;
  static const bool dbg = getenv("SWG_DEBUG") != nullptr;
  if (dbg && fl) {  // why: batches whose list of long events overflowed / wavefronts beyond their budget / batches with too many carried intervals
    uint32_t why[4] = {0, 0, 0, 0};
    (void)hipMemcpy(why, P.counters + 4, sizeof why, hipMemcpyDeviceToHost);
    fprintf(stderr, "[swg] segment flags %u: long-event lists %u, budgets %u, carried %u\n", why[0], why[1], why[2], why[3]);
  }
  return SWG_OK;
}
}  // namespace
}  // namespace swg_seg

// The axis' sorted begins from the runs of a pair-grouped input.  *done = 0: not applicable here, or a segment too dense for
// the LDS batches -- the caller sorts the general way (nothing it cannot overwrite was written).
int swg_seg_sort_begins(swg_ctx* ctx, const swg_axis_input& in, uint64_t* S, uint32_t* I, uint32_t* E, uint64_t* KEY, uint64_t* tile_xf,
                        uint32_t ntilesf, uint8_t* single, int* done, swg_seg_plan_view* view, uint8_t* tile_flag) {
  using namespace swg_seg;
  *done = 0;
  if (view) *view = swg_seg_plan_view{};
  static const int knob = getenv("SWG_SEG_SORT") ? atoi(getenv("SWG_SEG_SORT")) : -1;
  if (knob == 0) return SWG_OK;
  if (!in.seg_runs || in.n_seg_runs == 0 || !in.score_key || in.packed || !in.seg_run_alive) return SWG_OK;
  const uint64_t n = in.n;
  const uint32_t n_runs = in.n_seg_runs;
  hipStream_t st = ctx->stream;
  const swg_arena_mark mark = swg_arena_save(ctx);
  SegPlan P{};
  int ok = 0;
  SWG_TRY(seg_plan_make(ctx, in, false, &P, &ok));
  if (!ok) {
    swg_arena_restore(ctx, mark);
    return SWG_OK;
  }
  const uint64_t n_dead = n - P.n_alive;
  // the dead records' places (in front) and every tile start among them: zero
  if (n_dead) {
    SWG_HIP(ctx, hipMemsetAsync(S, 0, n_dead * sizeof(uint64_t), st));
    SWG_HIP(ctx, hipMemsetAsync(I, 0, n_dead * sizeof(uint32_t), st));
    SWG_HIP(ctx, hipMemsetAsync(E, 0, n_dead * sizeof(uint32_t), st));
    SWG_HIP(ctx, hipMemsetAsync(KEY, 0, n_dead * sizeof(uint64_t), st));
    SWG_HIP(ctx, hipMemsetAsync(tile_xf, 0, ((n_dead + TBF - 1) / TBF) * sizeof(uint64_t), st));
  }
  (void)ntilesf;
  SegSortArgs A{};
  A.perm = P.perm; A.seg_a = P.seg_a; A.seg_e = P.seg_e; A.seg_base = P.seg_base; A.seg_len = P.seg_len; A.alive = in.alive; A.seg_id = P.seg_id; A.start = in.start; A.end = in.end; A.score = in.score_key;
  A.out_a = P.seg_a;
  A.pos_bits = in.pos_bits; A.n_dead = (uint32_t)n_dead; A.S = S; A.I = I; A.E = E; A.KEY = KEY; A.tile_xf = tile_xf; A.single = single;
  A.counters = P.counters;
  A.tile_flag = tile_flag;  // (zero before -- the dead records' places stay so)
  if (P.ncls[2] + P.ncls[3]) {  // (the longest segments first in the same launch)
    A.list = P.class_list + (size_t)2 * n_runs;
    if (tile_flag)
      SWG_LAUNCH(ctx, "seg_sort_big", seg_sort_big_kernel<true><<<P.ncls[2] + P.ncls[3], 1024, 0, st>>>(A, P.class_list + (size_t)3 * n_runs, P.ncls[3]));
    else
      SWG_LAUNCH(ctx, "seg_sort_big", seg_sort_big_kernel<false><<<P.ncls[2] + P.ncls[3], 1024, 0, st>>>(A, P.class_list + (size_t)3 * n_runs, P.ncls[3]));
    SWG_KERNEL_CHECK(ctx);
  }
  if (P.ncls[1]) {
    A.list = P.class_list + (size_t)1 * n_runs;
    if (tile_flag)
      SWG_LAUNCH(ctx, "seg_sort_m", seg_sort_kernel<256, 16, 16, 1024, 64, true><<<P.ncls[1], 256, 0, st>>>(A));
    else
      SWG_LAUNCH(ctx, "seg_sort_m", seg_sort_kernel<256, 16, 16, 1024, 64, false><<<P.ncls[1], 256, 0, st>>>(A));
    SWG_KERNEL_CHECK(ctx);
  }
  if (P.ncls[0]) {
    A.list = P.class_list;
    if (tile_flag)
      SWG_LAUNCH(ctx, "seg_sort_s", seg_sort_kernel<64, 16, 16, 256, 64, true><<<P.ncls[0], 64, 0, st>>>(A));
    else
      SWG_LAUNCH(ctx, "seg_sort_s", seg_sort_kernel<64, 16, 16, 256, 64, false><<<P.ncls[0], 64, 0, st>>>(A));
    SWG_KERNEL_CHECK(ctx);
  }
  // the dense-bin flag: read with the caller's next read-back would be cheaper, but the caller must know before it routes
  uint32_t fl = 0;
  SWG_TRY(seg_flags_read(ctx, P, &fl));
  if (fl || !view) swg_arena_restore(ctx, mark);
  if (fl) {
    static const bool dbg = getenv("SWG_DEBUG") != nullptr;
    if (dbg) fprintf(stderr, "[swg] segment sort: a coarse bin denser than an LDS batch: the general sort takes the axis\n");
    return SWG_OK;
  }
  if (view) {  // (the plan stays in the arena for the caller's streaming sweep: it restores its own mark)
    view->valid = 1;
    view->seg_a = P.seg_a;
    view->seg_e = P.seg_e;
    view->class_list = P.class_list;
    view->counters = P.counters;
    view->n_runs = n_runs;
    for (int c = 0; c < 4; ++c) view->ncls[c] = P.ncls[c];
    view->n_dead = n_dead;
  }
  *done = 1;
  return SWG_OK;
}

// The k = 1 sweep of every segment over the begins swg_seg_sort_begins left sorted (seg_stream_body).  `keep` must be zero.
// *done = 0: deep data (or a run of equal starts longer than a chunk) -- the caller goes on with the tile kernels over the same
// arrays; `keep` then holds garbage.
int swg_seg_stream_sweep_k1(swg_ctx* ctx, const swg_seg_plan_view& v, const uint64_t* S, const uint32_t* I, const uint32_t* E, const uint64_t* KEY,
                            int pos_bits, double thr, const uint8_t* and_with, uint8_t* keep, uint64_t n, int* done) {
  using namespace swg_seg;
  *done = 0;
  // Off unless SWG_SEG_STREAM=1 (read at every call).  Measured on 10^8 records, sweep flags (round 6, profiles/README.md): on
  // segments of thousands of records the streamed sweep equals the routing + tile kernels it replaces (S-pan 13.4 against 13.6 ms,
  // 100 genomes x 5 chromosomes 11.7 against 12.3), on segments of a few hundred it loses badly (100 x 20: 17.1 against 12.3 -- a
  // work-group per segment again).  Exact either way (tests/test_gpu_segsweep.py).
  const char* knob_s = getenv("SWG_SEG_STREAM");
  const int knob = knob_s ? atoi(knob_s) : 0;  // (2: also where the context remembers deep data -- tests)
  if (knob < 1 || !v.valid) return SWG_OK;
  // deep data last time, on a call of about this size: the tile kernels' case (remembered like the sort's dropped bits)
  if (knob != 2 && ctx->seg_sweep_deep_n && n >= ctx->seg_sweep_deep_n / 2 && n <= ctx->seg_sweep_deep_n * 2) return SWG_OK;
  hipStream_t st = ctx->stream;
  SegStreamArgs A{};
  A.seg_a = v.seg_a; A.seg_e = v.seg_e; A.class_list = v.class_list; A.n_runs = v.n_runs;
  for (int c = 0; c < 4; ++c) A.n_cls[c] = v.ncls[c];
  A.S = S; A.I = I; A.E = E; A.KEY = KEY; A.n_dead = (uint32_t)v.n_dead; A.pos_mask = pos_bits >= 32 ? 0xffffffffu : (1u << pos_bits) - 1u; A.and_with = and_with; A.keep = keep; A.thr = thr; A.counters = v.counters;
  const uint32_t n_big = v.ncls[1] + v.ncls[2] + v.ncls[3];
  if (n_big) {
    SWG_LAUNCH(ctx, "seg_stream", seg_stream_kernel<512, 4, 256><<<n_big, 512, 0, st>>>(A));
    SWG_KERNEL_CHECK(ctx);
  }
  if (v.ncls[0]) {
    SWG_LAUNCH(ctx, "seg_stream_s", seg_stream_small_kernel<64, 16><<<v.ncls[0], 64, 0, st>>>(A));
    SWG_KERNEL_CHECK(ctx);
  }
  uint32_t fl = 0;
  {
    SegPlan P{};
    P.counters = v.counters;
    SWG_TRY(seg_flags_read(ctx, P, &fl));
  }
  if (fl) {
    ctx->seg_sweep_deep_n = n;
    static const bool dbg = getenv("SWG_DEBUG") != nullptr;
    if (dbg) fprintf(stderr, "[swg] streaming segment sweep: deep data: the axis goes on to the tile kernels\n");
    return SWG_OK;
  }
  *done = 1;
  return SWG_OK;
}

// The k = 1 sweep of an axis over the runs of a pair-grouped input, segment-resident (seg_sweep_body).  `keep` must be zero.
// *outcome = 0: declined (not this path's input, a bin denser than an LDS batch, or deep data -- the caller runs the axis its
//               other ways; `keep` holds garbage);
//            1: keep[i] is the axis' answer for every record (and_with folded in);
//            2: ... for the records of every segment but the longest ones (more than SEG_L_MAX places), whose *nb_left begins
//               sit sorted in S / I / E / KEY / tile_xf (nothing in front of them) for the tile kernels.
int swg_seg_sweep_k1(swg_ctx* ctx, const swg_axis_input& in, double thr, uint8_t* keep, uint64_t* S, uint32_t* I, uint32_t* E, uint64_t* KEY,
                     uint64_t* tile_xf, uint8_t* single, uint64_t* nb_left, int* outcome) {
  using namespace swg_seg;
  *outcome = 0;
  *nb_left = 0;
  // Off unless SWG_SEG_SWEEP=1 (read at every call: the tests switch it inside one process).  Measured on S-pan (10^8 records,
  // round 6, profiles/README.md): the sort + sweep of both axes 12.1 ms this way against 11.3 ms through seg_sort + the tile
  // kernels -- the work-group is alone on its CU and bound by instruction issue, and the windows' heavy-tailed lengths make every
  // pass cost what its slowest lane costs.  Exact either way (tests/test_gpu_segsweep.py).
  const char* knob_s = getenv("SWG_SEG_SWEEP");
  const int knob = knob_s ? atoi(knob_s) : 0;
  static const int sort_knob = getenv("SWG_SEG_SORT") ? atoi(getenv("SWG_SEG_SORT")) : -1;
  if (knob != 1 || sort_knob == 0) return SWG_OK;
  if (!in.seg_runs || in.n_seg_runs == 0 || !in.score_key || in.packed || !in.seg_run_alive || in.sorted_idx_out) return SWG_OK;
  // deep data last time, on a call of about this size: the tile kernels' case (remembered like the sort's dropped bits)
  if (ctx->seg_sweep_deep_n && in.n >= ctx->seg_sweep_deep_n / 2 && in.n <= ctx->seg_sweep_deep_n * 2) return SWG_OK;
  const uint32_t n_runs = in.n_seg_runs;
  hipStream_t st = ctx->stream;
  const swg_arena_mark mark = swg_arena_save(ctx);
  SegPlan P{};
  int ok = 0;
  SWG_TRY(seg_plan_make(ctx, in, true, &P, &ok));
  if (!ok) {
    swg_arena_restore(ctx, mark);
    return SWG_OK;
  }
  static const bool dbg = getenv("SWG_DEBUG") != nullptr;
  SegSweepArgs W{};
  W.perm = P.perm; W.seg_a = P.seg_a; W.seg_e = P.seg_e; W.seg_base = P.seg_base; W.seg_len = P.seg_len; W.alive = in.alive;
  W.start = in.start; W.end = in.end; W.score = in.score_key; W.and_with = in.and_with; W.keep = keep; W.thr = thr; W.counters = P.counters;
  uint64_t left = 0;
  SegSortArgs A{};
  if (P.ncls[3]) {
    // the longest segments: sorted in key-range batches into S / I / E / KEY at their offsets among themselves
    uint32_t* xl_off = swg_alloc<uint32_t>(ctx, P.n_seg + 1);
    uint64_t* d_left = swg_alloc<uint64_t>(ctx, 1);
    SWG_CHECK_ARENA(ctx);
    SWG_HIP(ctx, hipMemsetAsync(d_left, 0, sizeof(uint64_t), st));
    SWG_TRY(swg_exclusive_scan_u32(ctx, P.xl_len, xl_off, P.n_seg, d_left));
    SWG_TRY(swg_read_scalars(ctx, d_left, &left, 1));
    A.perm = P.perm; A.seg_a = P.seg_a; A.seg_e = P.seg_e; A.seg_base = P.seg_base; A.seg_len = P.seg_len; A.alive = in.alive; A.seg_id = P.seg_id; A.start = in.start; A.end = in.end; A.score = in.score_key;
    A.out_a = xl_off;
    A.pos_bits = in.pos_bits; A.n_dead = 0; A.S = S; A.I = I; A.E = E; A.KEY = KEY; A.tile_xf = tile_xf; A.single = single;
    A.counters = P.counters;
  }
  if (P.ncls[2] + P.ncls[3]) {
    W.list = P.class_list + (size_t)2 * n_runs;
    SWG_LAUNCH(ctx, "seg_sweep_big", seg_sweep_big_kernel<<<P.ncls[2] + P.ncls[3], 1024, 0, st>>>(W, A, P.class_list + (size_t)3 * n_runs, P.ncls[3]));
    SWG_KERNEL_CHECK(ctx);
  }
  if (P.ncls[1]) {
    W.list = P.class_list + (size_t)1 * n_runs;
    SWG_LAUNCH(ctx, "seg_sweep_m", seg_sweep_kernel<256, 8, 16, 1024, 256, 256><<<P.ncls[1], 256, 0, st>>>(W));
    SWG_KERNEL_CHECK(ctx);
  }
  if (P.ncls[0]) {
    W.list = P.class_list;
    SWG_LAUNCH(ctx, "seg_sweep_s", seg_sweep_kernel<64, 16, 16, 256, 64, 0><<<P.ncls[0], 64, 0, st>>>(W));
    SWG_KERNEL_CHECK(ctx);
  }
  uint32_t fl = 0;
  SWG_TRY(seg_flags_read(ctx, P, &fl));
  swg_arena_restore(ctx, mark);
  if (fl) {
    if (fl & 2u) ctx->seg_sweep_deep_n = in.n;
    if (dbg)
      fprintf(stderr, "[swg] segment sweep: %s: the axis goes to the tile kernels\n",
              (fl & 2u) ? "deep data (carried intervals or window scans beyond the resident sweep's bounds)" : "a coarse bin denser than an LDS batch");
    return SWG_OK;
  }
#ifdef SWG_SEG_TIMING
  {
    unsigned long long ht[16], z[16] = {0};
    (void)hipMemcpyFromSymbol(ht, HIP_SYMBOL(g_seg_t), sizeof ht);
    fprintf(stderr, "[swg] seg_sweep phases (100 MHz ticks summed over work-groups):");
    for (int k = 0; k < 16; ++k) fprintf(stderr, " %d:%llu", k, ht[k]);
    fprintf(stderr, "\n");
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_seg_t), z, sizeof z);
  }
#endif
  if (dbg) fprintf(stderr, "[swg] segment sweep: %u / %u / %u segments resident, %u long ones (%llu begins) left to the tile kernels\n", P.ncls[0], P.ncls[1], P.ncls[2], P.ncls[3], (unsigned long long)left);
  *nb_left = left;
  *outcome = left ? 2 : 1;
  return SWG_OK;
}
