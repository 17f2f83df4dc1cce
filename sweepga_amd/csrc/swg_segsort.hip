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

namespace swg_seg {
namespace {

constexpr int EW = 256;
constexpr uint32_t NONE = 0xffffffffu;
constexpr uint32_t SEG_S_MAX = 1024, SEG_M_MAX = 4096, SEG_L_MAX = 32768;
constexpr uint32_t TBF = 128;  // granularity of the tile-start keys (swg_sweep.hip)
inline unsigned nblk(uint64_t n) { return (unsigned)((n + EW - 1) / EW); }

struct Run {  // (layout of swg_scaf::PairRun)
  uint32_t a, n;
};

__device__ __forceinline__ void lds_barrier() {  // orders LDS accesses only (swg_pair.hip)
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_s_waitcnt(0xc07f);
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
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
                                                       uint32_t* __restrict__ counters) {
  const uint32_t n_seg = (uint32_t)*n_seg_dev;
  const uint32_t s = blockIdx.x * EW + threadIdx.x;
  int cls = -1;
  if (s < n_seg && seg_e[s] != seg_a[s]) {  // (a segment without a live record: no class)
    const uint32_t m = seg_base[s] != NONE ? seg_len[s] : seg_e[s] - seg_a[s];  // places a work-group's threads own
    cls = m <= SEG_S_MAX ? 0 : (m <= SEG_M_MAX ? 1 : (m <= SEG_L_MAX ? 2 : 3));
  }
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
};

constexpr size_t lds_align_up(size_t off, size_t align) { return (off + align - 1) / align * align; }
template <int NT, int ES, int ER, int NBK, int NBIN>
constexpr size_t seg_sort_lds_bytes() {
  return (size_t)NT * ES * 8 + (size_t)NBK * 4 + (size_t)NBIN * 4 + 17 * 4 + (size_t)(NT / 64 + 1) * 4 + 16 * 4 + 64;
}

// One work-group per segment of at most NT * ER live records, sorted in batches of at most NT * ES (one batch when they fit).
// The scheme and its idioms are pair_sort_body's (swg_pair.hip): a thread OWNS the records tid, tid + NT, ... of the segment's
// list, drops their keys into the batch's buckets, learns from the ranking where they ended up and puts their columns there.
template <int NT, int ES, int ER, int NBK, int NBIN>
__device__ __forceinline__ void seg_sort_body(const SegSortArgs& A, const uint32_t sg, char* lds_raw) {
  constexpr int CAP = NT * ES, MAXB = 16, H = 8;
  static_assert(ER <= 32 && ER % H == 0 && ES % 4 == 0 && NT * ER <= 65536 && CAP < 0xffff, "record masks are 32 bits wide, indices and ranks 16");
  static_assert(NBIN >= 1 && NBIN <= 4096 && NBK % NT == 0, "bins, bucket counters per thread");
  constexpr size_t O_K = 0, O_I = O_K + (size_t)CAP * 4, O_RR = O_I + (size_t)CAP * 2, O_CNT = lds_align_up(O_RR + (size_t)CAP * 2, 4),
                   O_BINS = O_CNT + (size_t)NBK * 4, O_BLO = O_BINS + (size_t)NBIN * 4, O_WS = O_BLO + (size_t)(MAXB + 1) * 4,
                   O_SH = O_WS + (size_t)(NT / 64 + 1) * 4;
  static_assert(O_SH + 8 * 4 <= seg_sort_lds_bytes<NT, ES, ER, NBK, NBIN>(), "LDS block of the work-group");
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
        if (n_live == 1u) A.single[ixv[e]] = 1;  // returned whole by the reference (plane_sweep_exact.rs:274-276)
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
    if (tid == 0) {  // greedy: a batch takes as many bins as fit
      uint32_t nb = 0, lo = 0;
      b_lo[0] = 0;
      while (lo < (uint32_t)NBIN) {
        const uint32_t start = bins[lo];
        uint32_t l = lo + 1, r = NBIN;
        while (l < r) {
          const uint32_t mid = l + ((r - l + 1) >> 1);
          const uint32_t pm = mid < (uint32_t)NBIN ? bins[mid] : n_live;
          if (pm - start <= (uint32_t)CAP) l = mid; else r = mid - 1;
        }
        const uint32_t p1 = l < (uint32_t)NBIN ? bins[l] : n_live;
        if (p1 - start > (uint32_t)CAP || nb + 1 >= (uint32_t)MAXB) {  // one bin denser than a batch: the general sort's case
          sh[3] = 1;
          break;
        }
        b_lo[++nb] = l;
        lo = l;
      }
      sh[2] = nb;
    }
    lds_barrier();
    if (sh[3]) {
      if (tid == 0) atomicOr(&A.counters[4], 1u);
      return;
    }
    n_batches = (uint32_t)__builtin_amdgcn_readfirstlane((int)sh[2]);
  }
  uint32_t base = 0;
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
    lds_barrier();
    // ---- count
    uint32_t batch_mask = 0;
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
          }
        }
      asm volatile("" ::: "memory");
    }
    lds_barrier();
    uint32_t mb;
    {
      constexpr int PER = NBK / NT;
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
    {
      // order inside the buckets: final position = bucket begin + the bucket's elements that order before by (key, place in the
      // list).  (the slot's list place and its rank share a word, place << 16 | rank; rank 0xffff = an empty slot)
      uint32_t rk[ES], rp[ES];
      const uint32_t t_rk = fresh_tid();
#pragma unroll
      for (int e = 0; e < ES; ++e) {
        const uint32_t pos = t_rk + (uint32_t)e * NT;
        rk[e] = pos < mb ? K[pos] : 0u;
        const uint32_t ix = pos < mb ? (uint32_t)I[pos] : 0u;
        rp[e] = (ix << 16) | 0xffffu;
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
            const uint32_t b = fine_of(rk[OFF + e], shift, first, &cb);
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
              if (kx == rk[OFF + e]) before = (uint32_t)I[x] < (rp[OFF + e] >> 16) ? 1u : 0u;
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
          RR[t_rk + (uint32_t)e * NT] = (uint16_t)r;
        }
    }
    lds_barrier();
    // ---- the composite starts out; where the thread's own records went
    const uint32_t gbase = A.n_dead + a + base;
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
#pragma unroll
    for (int j = 0; j < ER / 2; ++j) {
      const uint32_t w = slotw[j];
      const uint32_t r0 = (batch_mask >> (2 * j)) & 1u ? RR[w & 0xffffu] : 0u, r1 = (batch_mask >> (2 * j + 1)) & 1u ? RR[w >> 16] : 0u;
      slotw[j] = r0 | (r1 << 16);
      asm volatile("" : "+v"(slotw[j]));  // (kept packed)
    }
    lds_barrier();
    // ---- the other columns, transposed through LDS
    auto put_group = [&](int g, const uint32_t (&v)[H], uint32_t* buf) {
#pragma unroll
      for (int e = 0; e < H; ++e)
        if ((batch_mask >> (g + e)) & 1u) buf[(slotw[(g + e) / 2] >> (16 * ((g + e) & 1))) & 0xffffu] = v[e];
    };
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
template <int NT, int ES, int ER, int NBK, int NBIN>
__global__ __launch_bounds__(NT) void seg_sort_kernel(SegSortArgs A) {
  __shared__ __attribute__((aligned(16))) char raw[seg_sort_lds_bytes<NT, ES, ER, NBK, NBIN>()];
  seg_sort_body<NT, ES, ER, NBK, NBIN>(A, A.list[blockIdx.x], raw);
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
    const uint32_t gbase = A.n_dead + a + base;
    for (uint32_t p = tid; p < mb; p += XNT) {
      const unsigned long long w = W[p];
      const uint32_t id = rec_of((uint32_t)w);
      const uint64_t s = seg_part | (w >> 32);
      A.S[gbase + p] = s;
      A.I[gbase + p] = id;
      A.E[gbase + p] = A.end[id];
      A.KEY[gbase + p] = A.score[id];
      if (((gbase + p) % TBF) == 0u) A.tile_xf[(gbase + p) / TBF] = s;
      if (n_live == 1u) A.single[id] = 1;  // (a long run with one live record) returned whole by the reference
    }
    __syncthreads();
    base += mb;
    lo = hi;
  }
}
constexpr size_t SEG_BIG_LDS = seg_sort_lds_bytes<1024, 16, 32, 4096, 1024>() > SEG_XL_LDS ? seg_sort_lds_bytes<1024, 16, 32, 4096, 1024>() : SEG_XL_LDS;
static_assert(SEG_BIG_LDS <= 160 * 1024, "LDS of a CU");
__global__ __launch_bounds__(1024) void seg_sort_big_kernel(SegSortArgs A, const uint32_t* __restrict__ list_xl, uint32_t n_xl) {
  __shared__ __attribute__((aligned(16))) char raw[SEG_BIG_LDS];
  if (blockIdx.x < n_xl)
    seg_sort_xl_body(A, list_xl[blockIdx.x], raw);
  else
    seg_sort_body<1024, 16, 32, 4096, 1024>(A, A.list[blockIdx.x - n_xl], raw);
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

// The axis' sorted begins from the runs of a pair-grouped input.  *done = 0: not applicable here, or a segment too dense for
// the LDS batches -- the caller sorts the general way (nothing it cannot overwrite was written).
int swg_seg_sort_begins(swg_ctx* ctx, const swg_axis_input& in, uint64_t* S, uint32_t* I, uint32_t* E, uint64_t* KEY, uint64_t* tile_xf,
                        uint32_t ntilesf, uint8_t* single, int* done) {
  using namespace swg_seg;
  *done = 0;
  static const int knob = getenv("SWG_SEG_SORT") ? atoi(getenv("SWG_SEG_SORT")) : -1;
  if (knob == 0) return SWG_OK;
  if (!in.seg_runs || in.n_seg_runs == 0 || !in.score_key || in.packed || !in.seg_run_alive) return SWG_OK;
  const uint64_t n = in.n;
  const uint32_t n_runs = in.n_seg_runs;
  const int a_bits = swg_bits_for(n - 1) ? swg_bits_for(n - 1) : 1;
  const bool count_known = in.n_alive != ~0ull;  // (~0: the live records' number is the runs' total, read back with the plan)
  if (in.seg_bits + a_bits > 64 || n >= (uint64_t(1) << 31) || (count_known && in.n_alive > n)) return SWG_OK;
  hipStream_t st = ctx->stream;
  const Run* runs = static_cast<const Run*>(in.seg_runs);
  const swg_arena_mark mark = swg_arena_save(ctx);
  uint64_t* key = swg_alloc<uint64_t>(ctx, n_runs);
  uint64_t* key2 = swg_alloc<uint64_t>(ctx, n_runs);
  uint32_t* val = swg_alloc<uint32_t>(ctx, n_runs);
  uint32_t* val2 = swg_alloc<uint32_t>(ctx, n_runs);
  uint32_t* c = swg_alloc<uint32_t>(ctx, n_runs + 1);
  uint32_t* f = swg_alloc<uint32_t>(ctx, n_runs + 1);
  uint32_t* off = swg_alloc<uint32_t>(ctx, n_runs + 1);
  uint32_t* slot = swg_alloc<uint32_t>(ctx, n_runs + 1);
  uint32_t* seg_a = swg_alloc<uint32_t>(ctx, n_runs);
  uint32_t* seg_e = swg_alloc<uint32_t>(ctx, n_runs);
  uint64_t* seg_id = swg_alloc<uint64_t>(ctx, n_runs);
  uint32_t* seg_base = swg_alloc<uint32_t>(ctx, n_runs);
  uint32_t* seg_len = swg_alloc<uint32_t>(ctx, n_runs);
  uint32_t* class_list = swg_alloc<uint32_t>(ctx, (size_t)4 * n_runs);
  uint32_t* perm = swg_alloc<uint32_t>(ctx, (count_known ? in.n_alive : n) + 1);
  uint64_t* d_tot = swg_alloc<uint64_t>(ctx, 2);
  uint32_t* counters = swg_alloc<uint32_t>(ctx, 8);
  SWG_CHECK_ARENA(ctx);
  SWG_HIP(ctx, hipMemsetAsync(counters, 0, 8 * sizeof(uint32_t), st));
  SWG_HIP(ctx, hipMemsetAsync(d_tot, 0, 2 * sizeof(uint64_t), st));
  SWG_LAUNCH(ctx, "seg_run_key", run_key_kernel<<<nblk(n_runs), EW, 0, st>>>(n_runs, runs, in.seg_a, in.seg_b, in.seg_table, in.seg_mul, in.seg, a_bits, key, val));
  SWG_KERNEL_CHECK(ctx);
  SWG_TRY(swg_radix_sort_pairs(ctx, &key, &val, &key2, &val2, n_runs, 0, in.seg_bits + a_bits));
  SWG_LAUNCH(ctx, "seg_flags", seg_flags_kernel<<<nblk(n_runs), EW, 0, st>>>(n_runs, key, val, a_bits, in.seg_run_alive, c, f));
  SWG_KERNEL_CHECK(ctx);
  SWG_TRY(swg_exclusive_scan_u32(ctx, c, off, n_runs, d_tot));
  SWG_TRY(swg_exclusive_scan_u32(ctx, f, slot, n_runs, d_tot + 1));
  SWG_LAUNCH(ctx, "seg_bounds", seg_bounds_kernel<<<nblk(n_runs), EW, 0, st>>>(n_runs, key, a_bits, c, f, off, slot, seg_a, seg_e, seg_id, runs, val, seg_base, seg_len));
  SWG_KERNEL_CHECK(ctx);
  SWG_LAUNCH(ctx, "seg_class", seg_class_kernel<<<nblk(n_runs), EW, 0, st>>>(d_tot + 1, n_runs, seg_a, seg_e, seg_base, seg_len, class_list, counters));
  SWG_KERNEL_CHECK(ctx);
  {
    const unsigned pb = (n_runs + 3) / 4 < (uint32_t)ctx->num_cu * 32u ? (n_runs + 3) / 4 : (unsigned)ctx->num_cu * 32u;
    SWG_LAUNCH(ctx, "seg_perm", seg_perm_kernel<<<pb, 256, 0, st>>>(n_runs, runs, val, off, f, in.alive, perm));
    SWG_KERNEL_CHECK(ctx);
  }
  uint64_t h[6];
  {
    // one read-back: the live total (a check), the segments per size class
    uint64_t* d_all = swg_alloc<uint64_t>(ctx, 6);
    SWG_CHECK_ARENA(ctx);
    SWG_HIP(ctx, hipMemcpyAsync(d_all, d_tot, 2 * sizeof(uint64_t), hipMemcpyDeviceToDevice, st));
    SWG_HIP(ctx, hipMemcpyAsync(d_all + 2, counters, 4 * sizeof(uint32_t), hipMemcpyDeviceToDevice, st));
    SWG_TRY(swg_read_scalars(ctx, d_all, h, 4));
  }
  const uint32_t ncls[4] = {(uint32_t)h[2], (uint32_t)(h[2] >> 32), (uint32_t)h[3], (uint32_t)(h[3] >> 32)};
  static const bool dbg = getenv("SWG_DEBUG") != nullptr;
  const uint64_t n_alive = count_known ? in.n_alive : h[0];
  if (h[0] != n_alive || n_alive > n) {  // (the caller's count of live records and the runs' disagree: not this path's input)
    if (dbg) fprintf(stderr, "[swg] segment sort: %llu live records in the runs, %llu expected: the general sort takes the axis\n",
                     (unsigned long long)h[0], (unsigned long long)in.n_alive);
    swg_arena_restore(ctx, mark);
    return SWG_OK;
  }
  if (dbg)
    fprintf(stderr, "[swg] segment sort: %llu segments over %u runs (%u / %u / %u / %u by size class)\n", (unsigned long long)h[1], n_runs, ncls[0],
            ncls[1], ncls[2], ncls[3]);
  const uint64_t n_dead = n - n_alive;
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
  A.perm = perm; A.seg_a = seg_a; A.seg_e = seg_e; A.seg_base = seg_base; A.seg_len = seg_len; A.alive = in.alive; A.seg_id = seg_id; A.start = in.start; A.end = in.end; A.score = in.score_key;
  A.pos_bits = in.pos_bits; A.n_dead = (uint32_t)n_dead; A.S = S; A.I = I; A.E = E; A.KEY = KEY; A.tile_xf = tile_xf; A.single = single;
  A.counters = counters;
  if (ncls[2] + ncls[3]) {  // (the longest segments first in the same launch)
    A.list = class_list + (size_t)2 * n_runs;
    SWG_LAUNCH(ctx, "seg_sort_big", seg_sort_big_kernel<<<ncls[2] + ncls[3], 1024, 0, st>>>(A, class_list + (size_t)3 * n_runs, ncls[3]));
    SWG_KERNEL_CHECK(ctx);
  }
  if (ncls[1]) {
    A.list = class_list + (size_t)1 * n_runs;
    SWG_LAUNCH(ctx, "seg_sort_m", seg_sort_kernel<256, 16, 16, 1024, 64><<<ncls[1], 256, 0, st>>>(A));
    SWG_KERNEL_CHECK(ctx);
  }
  if (ncls[0]) {
    A.list = class_list;
    SWG_LAUNCH(ctx, "seg_sort_s", seg_sort_kernel<64, 16, 16, 256, 64><<<ncls[0], 64, 0, st>>>(A));
    SWG_KERNEL_CHECK(ctx);
  }
  // the dense-bin flag: read with the caller's next read-back would be cheaper, but the caller must know before it routes
  uint64_t fl = 0;
  {
    uint64_t* d_f = swg_alloc<uint64_t>(ctx, 1);
    SWG_CHECK_ARENA(ctx);
    SWG_HIP(ctx, hipMemsetAsync(d_f, 0, sizeof(uint64_t), st));
    SWG_HIP(ctx, hipMemcpyAsync(d_f, counters + 4, sizeof(uint32_t), hipMemcpyDeviceToDevice, st));
    SWG_TRY(swg_read_scalars(ctx, d_f, &fl, 1));
  }
  swg_arena_restore(ctx, mark);
  if ((uint32_t)fl) {
    if (dbg) fprintf(stderr, "[swg] segment sort: a coarse bin denser than an LDS batch: the general sort takes the axis\n");
    return SWG_OK;
  }
  *done = 1;
  return SWG_OK;
}
