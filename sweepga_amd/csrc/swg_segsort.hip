// Segmented sort of the begins of one sweep axis: the per-(query, target)-group sort of the plane sweep
// (src/plane_sweep_exact.rs:300: events of ONE segment sorted by position; src/paf_filter.rs:1037-1100: one segment per
// (sequence, genome of the other side)).
//
// The global LSD sort of swg_sort.hip treats the n composite keys ((segment + 1) << pos_bits | start) as one array: 6
// read+write passes of 12-byte pairs for the 42-bit keys of a 100-genome pangenome.  But a PAF arrives grouped: an
// aligner writes all mappings of one (query, target) pair together, so the records of one segment form a few long RUNS
// of the input.  This file sorts by segment WITHOUT moving the records through radix passes:
//
//   1. runs       every block of 2048 records finds its runs of equal segment (a run never crosses a block); the runs
//                 -- thousands, not millions -- are stable-sorted by segment with the ordinary radix sort, and a scan
//                 of their lengths gives every run its place: all records of a segment become contiguous, in record
//                 order.  If the input is not grouped (more than n / 32 runs) the caller falls back to the LSD sort.
//   2. scatter    every record goes to its place as one 8-byte word (start << 32 | record index).
//   3. LDS sort   consecutive segments are packed into buckets that fit a CU's LDS; one 1024-thread work-group per
//                 bucket loads it, runs stable 8-bit radix passes over the start bits entirely in LDS (wave match +
//                 per-wave digit counters, the ranking of os_pass_kernel) and writes the sorted composite keys and
//                 record indices.  A bucket of many small segments sorts (local segment, start, position) packed in one
//                 word.  Segments too large for LDS ("giants") are gathered, sorted by the LSD sort and put back.
//
// The result is bit-identical to begin_build + swg_radix_sort_pairs: keys ascending, dead records (key 0) first, ties
// in record order.  HBM traffic per record: 3 reads of the 9-byte segment columns, 8 B written + 8 B read for the
// scatter, 12 B written = ~55 B instead of ~170 B.
#include <cstdlib>

#include "swg_internal.h"

namespace {

constexpr int RT = 256;             // threads of the run kernels
constexpr int RPT = 8;              // consecutive records per thread
constexpr int RBLK = RT * RPT;      // records per block; a run never crosses a block
constexpr int LT = 1024;            // threads of the LDS sort
constexpr int LW = LT / 64;         // its waves
constexpr int ITEMS_L = 17;         // single-segment bucket: up to 17408 records (136 KB of LDS)
constexpr int CAP_L = LT * ITEMS_L;
constexpr int ITEMS_S = 8;          // bucket of small segments: up to 8192 records (2 x 64 KB of LDS)
constexpr int CAP_S = LT * ITEMS_S;
constexpr uint32_t SMALL = CAP_S / 2;  // segments up to this size are packed several per bucket
constexpr uint32_t NONE32 = 0xffffffffu;

enum : uint8_t { CLS_DEAD = 0, CLS_SMALL = 1, CLS_SINGLE = 2, CLS_GIANT = 3 };

struct BinSrc {
  const uint64_t* seg;
  const uint32_t* seg_a;
  const uint32_t* seg_b;
  const uint32_t* seg_table;
  uint32_t seg_mul;
  const uint8_t* alive;
};

__device__ __forceinline__ uint64_t bin_of(const BinSrc& b, uint64_t i) {  // segment + 1, 0 for a dead record
  if (b.alive && !b.alive[i]) return 0;
  if (b.seg) return b.seg[i] + 1;
  const uint32_t sb = b.seg_b[i];
  return (uint64_t)b.seg_a[i] * b.seg_mul + (b.seg_table ? b.seg_table[sb] : sb) + 1;
}

// Bins of the thread's RPT consecutive records and the mask of run heads among them (bit j: record j starts a run).
// The block's first record always starts a run.  `s_last` is RT u64 of LDS.  Contains one barrier.
__device__ __forceinline__ uint32_t run_heads(const BinSrc& src, uint64_t n, uint64_t i0, uint64_t* s_last, uint64_t bins[RPT]) {
#pragma unroll
  for (int j = 0; j < RPT; ++j) bins[j] = i0 + j < n ? bin_of(src, i0 + j) : ~0ull;
  s_last[threadIdx.x] = bins[RPT - 1];
  __syncthreads();
  uint32_t heads = 0;
  uint64_t prev = threadIdx.x ? s_last[threadIdx.x - 1] : ~0ull;
#pragma unroll
  for (int j = 0; j < RPT; ++j) {
    if (i0 + j < n && ((threadIdx.x == 0 && j == 0) || bins[j] != prev)) heads |= 1u << j;
    prev = bins[j];
  }
  return heads;
}

// exclusive prefix of one u32 per thread over the block (RT threads); *total = block sum.  Two barriers.
__device__ __forceinline__ uint32_t block_excl_sum(uint32_t v, uint32_t* s_wave, uint32_t* total) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  uint32_t inc = v;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const uint32_t t = __shfl_up(inc, d, 64);
    if (lane >= d) inc += t;
  }
  if (lane == 63) s_wave[wave] = inc;
  __syncthreads();
  uint32_t base = 0, tot = 0;
#pragma unroll
  for (int w = 0; w < RT / 64; ++w) {
    const uint32_t s = s_wave[w];
    if (w < wave) base += s;
    tot += s;
  }
  __syncthreads();
  *total = tot;
  return base + inc - v;
}

__global__ __launch_bounds__(RT) void ss_runs_count_kernel(BinSrc src, uint64_t n, uint32_t* __restrict__ run_cnt) {
  __shared__ uint64_t s_last[RT];
  __shared__ uint32_t s_wave[RT / 64];
  const uint64_t i0 = (uint64_t)blockIdx.x * RBLK + (uint64_t)threadIdx.x * RPT;
  uint64_t bins[RPT];
  const uint32_t heads = run_heads(src, n, i0, s_last, bins);
  uint32_t total;
  (void)block_excl_sum(__popc(heads), s_wave, &total);
  if (threadIdx.x == 0) run_cnt[blockIdx.x] = total;
}

// run records: key = bin, length, and the slot id as the value the radix sort carries
__global__ __launch_bounds__(RT) void ss_runs_emit_kernel(BinSrc src, uint64_t n, const uint32_t* __restrict__ run_off,
                                                          uint64_t* __restrict__ rkey, uint32_t* __restrict__ rlen,
                                                          uint32_t* __restrict__ rslot) {
  __shared__ uint64_t s_last[RT];
  __shared__ uint32_t s_wave[RT / 64];
  __shared__ uint32_t s_hpos[RBLK + 1];  // local position of every run head of the block, then the block's end
  const uint64_t b0 = (uint64_t)blockIdx.x * RBLK;
  const uint64_t i0 = b0 + (uint64_t)threadIdx.x * RPT;
  uint64_t bins[RPT];
  const uint32_t heads = run_heads(src, n, i0, s_last, bins);
  uint32_t total;
  uint32_t r = block_excl_sum(__popc(heads), s_wave, &total);
  const uint32_t slot0 = run_off[blockIdx.x];
#pragma unroll
  for (int j = 0; j < RPT; ++j)
    if (heads & (1u << j)) {
      s_hpos[r] = threadIdx.x * RPT + j;
      rkey[slot0 + r] = bins[j];
      rslot[slot0 + r] = slot0 + r;
      ++r;
    }
  if (threadIdx.x == 0) s_hpos[total] = (uint32_t)((n - b0) < (uint64_t)RBLK ? (n - b0) : (uint64_t)RBLK);
  __syncthreads();
  for (uint32_t k = threadIdx.x; k < total; k += RT) rlen[slot0 + k] = s_hpos[k + 1] - s_hpos[k];
}

// sorted run j: its length (for the scan that places the runs) and whether it is the first run of its segment
__global__ __launch_bounds__(RT) void ss_runs_sorted_kernel(uint64_t R, const uint64_t* __restrict__ rk,
                                                            const uint32_t* __restrict__ rs, const uint32_t* __restrict__ rlen,
                                                            uint32_t* __restrict__ len_sorted, uint32_t* __restrict__ seg_flag) {
  const uint64_t j = (uint64_t)blockIdx.x * RT + threadIdx.x;
  if (j >= R) return;
  len_sorted[j] = rlen[rs[j]];
  seg_flag[j] = (j == 0 || rk[j] != rk[j - 1]) ? 1u : 0u;
}

__global__ __launch_bounds__(RT) void ss_runs_finish_kernel(uint64_t R, uint64_t n, const uint64_t* __restrict__ rk,
                                                            const uint32_t* __restrict__ rs, const uint32_t* __restrict__ base_sorted,
                                                            const uint32_t* __restrict__ seg_flag, const uint32_t* __restrict__ seg_idx,
                                                            uint32_t n_segs, uint32_t* __restrict__ rbase,
                                                            uint32_t* __restrict__ seg_off, uint64_t* __restrict__ seg_key) {
  const uint64_t j = (uint64_t)blockIdx.x * RT + threadIdx.x;
  if (j == 0) seg_off[n_segs] = (uint32_t)n;
  if (j >= R) return;
  const uint32_t b = base_sorted[j];
  rbase[rs[j]] = b;
  if (seg_flag[j]) {
    seg_off[seg_idx[j]] = b;
    seg_key[seg_idx[j]] = rk[j];
  }
}

// record -> its place: P[place] = start << 32 | record index
__global__ __launch_bounds__(RT) void ss_scatter_kernel(BinSrc src, uint64_t n, const uint32_t* __restrict__ start,
                                                        const uint32_t* __restrict__ run_off, const uint32_t* __restrict__ rbase,
                                                        uint64_t* __restrict__ P) {
  __shared__ uint64_t s_last[RT];
  __shared__ uint32_t s_wave[RT / 64];
  __shared__ uint32_t s_hpos[RBLK];
  const uint64_t b0 = (uint64_t)blockIdx.x * RBLK;
  const uint64_t i0 = b0 + (uint64_t)threadIdx.x * RPT;
  uint64_t bins[RPT];
  const uint32_t heads = run_heads(src, n, i0, s_last, bins);
  uint32_t total;
  const uint32_t r0 = block_excl_sum(__popc(heads), s_wave, &total);
  {
    uint32_t r = r0;
#pragma unroll
    for (int j = 0; j < RPT; ++j)
      if (heads & (1u << j)) s_hpos[r++] = threadIdx.x * RPT + j;
  }
  __syncthreads();
  const uint32_t slot0 = run_off[blockIdx.x];
  uint32_t r = r0;  // runs that start before this thread's first record; its first record belongs to run r - 1 unless it is a head
#pragma unroll
  for (int j = 0; j < RPT; ++j) {
    const uint64_t i = i0 + j;
    if (i >= n) break;
    if (heads & (1u << j)) ++r;
    const uint32_t run = r - 1;
    const uint32_t local = threadIdx.x * RPT + j;
    P[(uint64_t)rbase[slot0 + run] + (local - s_hpos[run])] = ((uint64_t)start[i] << 32) | (uint32_t)i;
  }
}

// per segment: class, and the two sizes the bucket / giant plans are scanned from
__global__ __launch_bounds__(RT) void ss_seg_class_kernel(uint32_t n_segs, const uint32_t* __restrict__ seg_off,
                                                          const uint64_t* __restrict__ seg_key, uint8_t* __restrict__ cls,
                                                          uint32_t* __restrict__ small_cnt, uint32_t* __restrict__ giant_cnt,
                                                          uint8_t* __restrict__ giant_flag) {
  const uint32_t s = blockIdx.x * RT + threadIdx.x;
  if (s >= n_segs) return;
  const uint32_t c = seg_off[s + 1] - seg_off[s];
  const uint8_t k = seg_key[s] == 0 ? CLS_DEAD : (c <= SMALL ? CLS_SMALL : (c <= (uint32_t)CAP_L ? CLS_SINGLE : CLS_GIANT));
  cls[s] = k;
  small_cnt[s] = k == CLS_SMALL ? c : 0u;
  giant_cnt[s] = k == CLS_GIANT ? c : 0u;
  giant_flag[s] = k == CLS_GIANT ? 1 : 0;
}

// A bucket = one work-group of the LDS sort: a dead / single / giant segment alone, or consecutive small segments whose
// start offsets (counted over small segments only) fall into the same window of SMALL records (< 2 * SMALL = CAP_S in all).
__global__ __launch_bounds__(RT) void ss_bucket_flag_kernel(uint32_t n_segs, const uint8_t* __restrict__ cls,
                                                            const uint32_t* __restrict__ small_pre, uint8_t* __restrict__ bucket_head) {
  const uint32_t s = blockIdx.x * RT + threadIdx.x;
  if (s >= n_segs) return;
  bool h = true;
  if (s > 0 && cls[s] == CLS_SMALL && cls[s - 1] == CLS_SMALL) h = small_pre[s] / SMALL != small_pre[s - 1] / SMALL;
  bucket_head[s] = h ? 1 : 0;
}

__global__ __launch_bounds__(RT) void ss_dead_out_kernel(uint64_t n_dead, const uint64_t* __restrict__ P, uint64_t* __restrict__ S,
                                                         uint32_t* __restrict__ I) {
  const uint64_t k = (uint64_t)blockIdx.x * RT + threadIdx.x;
  if (k >= n_dead) return;
  S[k] = 0;
  I[k] = (uint32_t)P[k];
}

// ---- the LDS sort ---------------------------------------------------------------------------------------------------
// Stable 8-bit radix pass over `arr` (LDS, NI * LT words in wave-blocked order: wave w owns [w * 64 * NI, (w+1) * 64 * NI),
// row r of it = 64 consecutive words).  Words past the bucket's end hold ~0 and stay at the end.  Digit = bits
// [shift, shift + 8) of the word.
template <int NI>
__device__ __forceinline__ void lds_radix_pass(uint64_t* arr, uint32_t (*cnt)[256], uint32_t* s_scan, int shift, uint32_t dmask) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const uint64_t lt_mask = lane == 0 ? 0ull : (~0ull >> (64 - lane));
  uint64_t key[NI];
  uint32_t rank[NI];
#pragma unroll
  for (int r = 0; r < NI; ++r) key[r] = arr[wave * 64 * NI + r * 64 + lane];
  for (int k = tid; k < LW * 256; k += LT) (&cnt[0][0])[k] = 0;
  __syncthreads();
#pragma unroll
  for (int r = 0; r < NI; ++r) {
    const uint32_t d = (uint32_t)(key[r] >> shift) & dmask;
    uint64_t peers = ~0ull;
#pragma unroll
    for (int b = 0; b < 8; ++b) {
      const bool bit = (d >> b) & 1u;
      const uint64_t m = __ballot(bit);
      peers &= bit ? m : ~m;
    }
    const int leader = __builtin_ctzll(peers);
    uint32_t prev = 0;
    if (lane == leader) {
      prev = cnt[wave][d];
      cnt[wave][d] = prev + (uint32_t)__popcll(peers);
    }
    prev = __shfl(prev, leader, 64);
    rank[r] = prev + (uint32_t)__popcll(peers & lt_mask);
  }
  __syncthreads();
  // per digit: exclusive over the waves, then exclusive over the digits
  uint32_t tot = 0;
  if (tid < 256) {
#pragma unroll
    for (int w = 0; w < LW; ++w) {
      const uint32_t c = cnt[w][tid];
      cnt[w][tid] = tot;
      tot += c;
    }
    uint32_t inc = tot;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const uint32_t t = __shfl_up(inc, d, 64);
      if (lane >= d) inc += t;
    }
    if (lane == 63) s_scan[wave] = inc;
    s_scan[8 + tid] = inc - tot;  // exclusive inside the wave
  }
  __syncthreads();
  if (tid < 256) {
    uint32_t base = s_scan[8 + tid];
    for (int w = 0; w < wave; ++w) base += s_scan[w];
#pragma unroll
    for (int w = 0; w < LW; ++w) cnt[w][tid] += base;
  }
  __syncthreads();
#pragma unroll
  for (int r = 0; r < NI; ++r) {
    const uint32_t d = (uint32_t)(key[r] >> shift) & dmask;
    arr[cnt[wave][d] + rank[r]] = key[r];
  }
  __syncthreads();
}

struct LdsSortArgs {
  const uint64_t* P;
  const uint32_t* bucket_seg;  // [n_buckets + 1] first segment of every bucket
  const uint32_t* seg_off;
  const uint64_t* seg_key;
  const uint8_t* cls;
  int pos_bits;
  uint64_t* S;
  uint32_t* I;
};

__global__ __launch_bounds__(LT) void ss_lds_sort_kernel(LdsSortArgs a) {
  __shared__ uint64_t arr[CAP_L];        // the words being sorted
  __shared__ uint32_t cnt[LW][256];      // per-wave digit counters
  __shared__ uint32_t s_scan[8 + 256];
  __shared__ uint32_t s_wtot[LW];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const uint32_t s0 = a.bucket_seg[blockIdx.x], s1 = a.bucket_seg[blockIdx.x + 1];
  const uint8_t k = a.cls[s0];
  if (k == CLS_DEAD || k == CLS_GIANT) return;  // block-uniform
  const uint32_t lo = a.seg_off[s0], hi = a.seg_off[s1];
  const uint32_t N = hi - lo;
  if (k == CLS_SINGLE) {
    // one segment: sort the words (start << 32 | index) by their start bits; the scatter left them in index order
    for (uint32_t p = tid; p < (uint32_t)CAP_L; p += LT) arr[p] = p < N ? a.P[lo + p] : ~0ull;
    __syncthreads();
    for (int shift = 0; shift < a.pos_bits; shift += 8) {
      const int bits = a.pos_bits - shift < 8 ? a.pos_bits - shift : 8;
      lds_radix_pass<ITEMS_L>(arr, cnt, s_scan, 32 + shift, (1u << bits) - 1u);
    }
    const uint64_t hi_key = a.seg_key[s0] << a.pos_bits;
    for (uint32_t p = tid; p < N; p += LT) {
      const uint64_t w = arr[p];
      a.S[lo + p] = hi_key | (w >> 32);
      a.I[lo + p] = (uint32_t)w;
    }
    return;
  }
  // several small segments (N < CAP_S): words = (local segment << (pos_bits + 13)) | (start << 13) | local position, sorted
  // over the start and segment bits; the original words stay in the upper half of `arr`
  uint64_t* orig = arr + CAP_S;
  for (uint32_t p = tid; p < (uint32_t)CAP_S; p += LT) orig[p] = p < N ? a.P[lo + p] : ~0ull;
  // segment heads -> local segment number of every position (flags live in cnt, which the first pass re-zeroes)
  uint32_t* flag = &cnt[0][0];  // CAP_S bits would do; LW * 256 = 4096 u32 >= CAP_S / 2 ... use bytes
  uint8_t* fb = reinterpret_cast<uint8_t*>(flag);
  for (uint32_t p = tid; p < (uint32_t)CAP_S; p += LT) fb[p] = 0;
  __syncthreads();
  for (uint32_t s = s0 + tid; s < s1; s += LT) fb[a.seg_off[s] - lo] = 1;
  __syncthreads();
  uint32_t segn[ITEMS_S];
  {
    uint32_t run = 0;  // heads seen in this wave's earlier rows
#pragma unroll
    for (int r = 0; r < ITEMS_S; ++r) {
      const uint32_t p = wave * 64 * ITEMS_S + r * 64 + lane;
      const bool f = fb[p] != 0;
      const uint64_t m = __ballot(f);
      const uint64_t le_mask = lane == 63 ? ~0ull : ((1ull << (lane + 1)) - 1ull);
      segn[r] = run + (uint32_t)__popcll(m & le_mask);  // heads at or before p inside this wave's span
      run += (uint32_t)__popcll(m);
    }
    if (lane == 0) s_wtot[wave] = run;
  }
  __syncthreads();
  {
    uint32_t before = 0;
    for (int w = 0; w < wave; ++w) before += s_wtot[w];
    const int shift_seg = a.pos_bits + 13;
#pragma unroll
    for (int r = 0; r < ITEMS_S; ++r) {
      const uint32_t p = wave * 64 * ITEMS_S + r * 64 + lane;
      const uint64_t w = orig[p];
      arr[p] = p < N ? ((uint64_t)(before + segn[r] - 1) << shift_seg) | ((w >> 32) << 13) | p : ~0ull;
    }
  }
  __syncthreads();
  const uint32_t nseg = s1 - s0;
  int seg_bits = 0;
  while ((1u << seg_bits) < nseg) ++seg_bits;
  const int top = a.pos_bits + seg_bits;  // bits to sort above the 13 position bits (already ascending: stability)
  for (int shift = 0; shift < top; shift += 8) {
    const int bits = top - shift < 8 ? top - shift : 8;
    lds_radix_pass<ITEMS_S>(arr, cnt, s_scan, 13 + shift, (1u << bits) - 1u);
  }
  const uint64_t pos_mask = (a.pos_bits >= 64 ? ~0ull : ((1ull << a.pos_bits) - 1ull));
  for (uint32_t p = tid; p < N; p += LT) {
    const uint64_t w = arr[p];
    const uint32_t ls = (uint32_t)(w >> (a.pos_bits + 13));
    a.S[lo + p] = (a.seg_key[s0 + ls] << a.pos_bits) | ((w >> 13) & pos_mask);
    a.I[lo + p] = (uint32_t)orig[w & 0x1fffu];
  }
}

// ---- giants: gathered, sorted by the global radix sort, put back ----------------------------------------------------
__device__ __forceinline__ uint32_t giant_of(const uint32_t* __restrict__ gpre, uint32_t ng, uint32_t j) {
  uint32_t l = 0, r = ng;  // last g with gpre[g] <= j
  while (r - l > 1) {
    const uint32_t m = (l + r) >> 1;
    if (gpre[m] <= j)
      l = m;
    else
      r = m;
  }
  return l;
}
__global__ __launch_bounds__(RT) void ss_giant_gather_kernel(uint32_t G, uint32_t ng, const uint32_t* __restrict__ giant_segs,
                                                             const uint32_t* __restrict__ gpre, const uint32_t* __restrict__ seg_off,
                                                             const uint64_t* __restrict__ seg_key, int pos_bits,
                                                             const uint64_t* __restrict__ P, uint64_t* __restrict__ gk,
                                                             uint32_t* __restrict__ gv) {
  const uint32_t j = blockIdx.x * RT + threadIdx.x;
  if (j >= G) return;
  const uint32_t g = giant_of(gpre, ng, j);
  const uint32_t s = giant_segs[g];
  const uint64_t w = P[seg_off[s] + (j - gpre[g])];
  gk[j] = (seg_key[s] << pos_bits) | (w >> 32);
  gv[j] = (uint32_t)w;
}
__global__ __launch_bounds__(RT) void ss_giant_gpre_kernel(uint32_t ng, const uint32_t* __restrict__ giant_segs,
                                                           const uint32_t* __restrict__ giant_pre, uint32_t* __restrict__ gpre) {
  const uint32_t g = blockIdx.x * RT + threadIdx.x;
  if (g < ng) gpre[g] = giant_pre[giant_segs[g]];
}
__global__ __launch_bounds__(RT) void ss_giant_scatter_kernel(uint32_t G, uint32_t ng, const uint32_t* __restrict__ giant_segs,
                                                              const uint32_t* __restrict__ gpre, const uint32_t* __restrict__ seg_off,
                                                              const uint64_t* __restrict__ gk, const uint32_t* __restrict__ gv,
                                                              uint64_t* __restrict__ S, uint32_t* __restrict__ I) {
  const uint32_t j = blockIdx.x * RT + threadIdx.x;
  if (j >= G) return;
  const uint32_t g = giant_of(gpre, ng, j);
  const uint32_t pos = seg_off[giant_segs[g]] + (j - gpre[g]);
  S[pos] = gk[j];
  I[pos] = gv[j];
}

inline unsigned nb(uint64_t n, int t) { return (unsigned)((n + t - 1) / t); }

}  // namespace

// Returns SWG_OK with *taken = 1 and the sorted begins in S / I, or *taken = 0 when the input is not grouped (or too small,
// or switched off) and the caller should use begin_build + swg_radix_sort_pairs.  `P` is scratch of n u64.
int swg_segsort_begins(swg_ctx* ctx, const swg_axis_input& in, uint64_t* S, uint32_t* I, uint64_t* P, int* taken) {
  *taken = 0;
  const uint64_t n = in.n;
  static const char* knob = getenv("SWG_SEGSORT");  // "0": never, "1": whenever the input is grouped (tests), unset: large inputs
  if (knob && knob[0] == '0') return SWG_OK;
  const bool force = knob && knob[0] == '1';
  if (!force && n < (uint64_t(1) << 20)) return SWG_OK;  // launch-bound below this: the plain sort is as fast
  if (n < 2 || n >= (uint64_t(1) << 31) || in.pos_bits > 32 || in.pos_bits + 13 + 13 > 64) return SWG_OK;
  hipStream_t st = ctx->stream;
  swg_arena_mark mark = swg_arena_save(ctx);
  BinSrc src{in.seg, in.seg_a, in.seg_b, in.seg_table, in.seg_mul, in.alive};
  const uint32_t nblocks = nb(n, RBLK);
  uint32_t* run_cnt = swg_alloc<uint32_t>(ctx, (size_t)nblocks + 1);
  uint64_t* d_tot = swg_alloc<uint64_t>(ctx, 4);
  SWG_CHECK_ARENA(ctx);
  SWG_LAUNCH(ctx, "ss_runs_count", ss_runs_count_kernel<<<nblocks, RT, 0, st>>>(src, n, run_cnt));
  SWG_KERNEL_CHECK(ctx);
  SWG_TRY(swg_exclusive_scan_u32(ctx, run_cnt, run_cnt, nblocks, d_tot));
  uint64_t R = 0;
  SWG_TRY(swg_read_scalars(ctx, d_tot, &R, 1));
  if (R > n / 32 + nblocks && !(force && R <= (uint64_t(1) << 24))) {  // not grouped: the radix passes are the better plan
    swg_arena_restore(ctx, mark);
    return SWG_OK;
  }
  uint64_t* rk = swg_alloc<uint64_t>(ctx, R);
  uint64_t* rk2 = swg_alloc<uint64_t>(ctx, R);
  uint32_t* rs = swg_alloc<uint32_t>(ctx, R);
  uint32_t* rs2 = swg_alloc<uint32_t>(ctx, R);
  uint32_t* rlen = swg_alloc<uint32_t>(ctx, R);
  uint32_t* len_sorted = swg_alloc<uint32_t>(ctx, R);
  uint32_t* seg_flag = swg_alloc<uint32_t>(ctx, R);
  uint32_t* rbase = swg_alloc<uint32_t>(ctx, R);
  SWG_CHECK_ARENA(ctx);
  SWG_LAUNCH(ctx, "ss_runs_emit", ss_runs_emit_kernel<<<nblocks, RT, 0, st>>>(src, n, run_cnt, rk, rlen, rs));
  SWG_KERNEL_CHECK(ctx);
  SWG_TRY(swg_radix_sort_pairs(ctx, &rk, &rs, &rk2, &rs2, R, 0, in.seg_bits + 1 > 64 ? 64 : in.seg_bits + 1));
  SWG_LAUNCH(ctx, "ss_runs_sorted", ss_runs_sorted_kernel<<<nb(R, RT), RT, 0, st>>>(R, rk, rs, rlen, len_sorted, seg_flag));
  SWG_KERNEL_CHECK(ctx);
  SWG_TRY(swg_exclusive_scan_u32(ctx, len_sorted, len_sorted, R, nullptr));
  SWG_TRY(swg_exclusive_scan_u32(ctx, seg_flag, rs2, R, d_tot + 1));  // rs2: the sort's scratch values, free again -> segment index
  uint64_t n_segs = 0;
  SWG_TRY(swg_read_scalars(ctx, d_tot + 1, &n_segs, 1));
  uint32_t* seg_off = swg_alloc<uint32_t>(ctx, n_segs + 1);
  uint64_t* seg_key = swg_alloc<uint64_t>(ctx, n_segs);
  uint8_t* cls = swg_alloc<uint8_t>(ctx, n_segs + 1);
  uint8_t* giant_flag = swg_alloc<uint8_t>(ctx, n_segs);
  uint8_t* bucket_head = swg_alloc<uint8_t>(ctx, n_segs);
  uint32_t* small_pre = swg_alloc<uint32_t>(ctx, n_segs);
  uint32_t* giant_pre = swg_alloc<uint32_t>(ctx, n_segs);
  SWG_CHECK_ARENA(ctx);
  SWG_LAUNCH(ctx, "ss_runs_finish", ss_runs_finish_kernel<<<nb(R, RT), RT, 0, st>>>(R, n, rk, rs, len_sorted, seg_flag, rs2, (uint32_t)n_segs,
                                                                         rbase, seg_off, seg_key));
  SWG_KERNEL_CHECK(ctx);
  SWG_LAUNCH(ctx, "ss_seg_class", ss_seg_class_kernel<<<nb(n_segs, RT), RT, 0, st>>>((uint32_t)n_segs, seg_off, seg_key, cls, small_pre, giant_pre,
                                                                           giant_flag));
  SWG_KERNEL_CHECK(ctx);
  SWG_TRY(swg_exclusive_scan_u32(ctx, small_pre, small_pre, n_segs, nullptr));
  SWG_TRY(swg_exclusive_scan_u32(ctx, giant_pre, giant_pre, n_segs, d_tot + 2));
  SWG_LAUNCH(ctx, "ss_bucket_flag", ss_bucket_flag_kernel<<<nb(n_segs, RT), RT, 0, st>>>((uint32_t)n_segs, cls, small_pre, bucket_head));
  SWG_KERNEL_CHECK(ctx);
  swg_flag_scan bscan, gscan;
  SWG_TRY(swg_flags_count(ctx, bucket_head, n_segs, &bscan, d_tot + 3));
  SWG_TRY(swg_flags_count(ctx, giant_flag, n_segs, &gscan, d_tot));
  uint64_t h[4];
  SWG_TRY(swg_read_scalars(ctx, d_tot, h, 4));
  const uint64_t n_giant_segs = h[0], G = h[2], n_buckets = h[3];
  if (G > n / 2 && !force) {  // mostly segments that do not fit LDS (one deep chromosome pair): the plain sort does it in one go
    swg_arena_restore(ctx, mark);
    return SWG_OK;
  }
  SWG_LAUNCH(ctx, "ss_scatter", ss_scatter_kernel<<<nblocks, RT, 0, st>>>(src, n, in.start, run_cnt, rbase, P));
  SWG_KERNEL_CHECK(ctx);
  uint32_t* bucket_seg = swg_alloc<uint32_t>(ctx, n_buckets + 1);
  SWG_CHECK_ARENA(ctx);
  SWG_TRY(swg_flags_compact(ctx, bscan, bucket_seg));
  {
    const uint32_t ns32 = (uint32_t)n_segs;
    SWG_HIP(ctx, hipMemcpyAsync(bucket_seg + n_buckets, &ns32, sizeof(uint32_t), hipMemcpyHostToDevice, st));
    SWG_HIP(ctx, hipStreamSynchronize(st));  // ns32 lives on this stack frame
  }
  // dead records: the first segment when its key is 0
  {
    uint64_t first[2];
    SWG_TRY(swg_read_scalars(ctx, seg_key, first, 1));
    if (first[0] == 0) {
      uint64_t off2 = 0;
      SWG_HIP(ctx, hipMemcpyAsync(ctx->h_scalars, seg_off + 1, sizeof(uint32_t), hipMemcpyDeviceToHost, st));
      SWG_HIP(ctx, hipStreamSynchronize(st));
      off2 = reinterpret_cast<uint32_t*>(ctx->h_scalars)[0];
      SWG_LAUNCH(ctx, "ss_dead_out", ss_dead_out_kernel<<<nb(off2, RT), RT, 0, st>>>(off2, P, S, I));
      SWG_KERNEL_CHECK(ctx);
    }
  }
  LdsSortArgs la{P, bucket_seg, seg_off, seg_key, cls, in.pos_bits, S, I};
  SWG_LAUNCH(ctx, "ss_lds_sort", ss_lds_sort_kernel<<<(unsigned)n_buckets, LT, 0, st>>>(la));
  SWG_KERNEL_CHECK(ctx);
  if (G) {
    uint32_t* giant_segs = swg_alloc<uint32_t>(ctx, n_giant_segs);
    uint32_t* gpre = swg_alloc<uint32_t>(ctx, n_giant_segs);
    uint64_t* gk = swg_alloc<uint64_t>(ctx, G);
    uint64_t* gk2 = swg_alloc<uint64_t>(ctx, G);
    uint32_t* gv = swg_alloc<uint32_t>(ctx, G);
    uint32_t* gv2 = swg_alloc<uint32_t>(ctx, G);
    SWG_CHECK_ARENA(ctx);
    SWG_TRY(swg_flags_compact(ctx, gscan, giant_segs));
    SWG_LAUNCH(ctx, "ss_giant_gpre", ss_giant_gpre_kernel<<<nb(n_giant_segs, RT), RT, 0, st>>>((uint32_t)n_giant_segs, giant_segs, giant_pre, gpre));
    SWG_KERNEL_CHECK(ctx);
    SWG_LAUNCH(ctx, "ss_giant_gather", ss_giant_gather_kernel<<<nb(G, RT), RT, 0, st>>>((uint32_t)G, (uint32_t)n_giant_segs, giant_segs, gpre, seg_off,
                                                                           seg_key, in.pos_bits, P, gk, gv));
    SWG_KERNEL_CHECK(ctx);
    SWG_TRY(swg_radix_sort_pairs(ctx, &gk, &gv, &gk2, &gv2, G, 0, in.seg_bits + in.pos_bits));
    SWG_LAUNCH(ctx, "ss_giant_scatter", ss_giant_scatter_kernel<<<nb(G, RT), RT, 0, st>>>((uint32_t)G, (uint32_t)n_giant_segs, giant_segs, gpre, seg_off, gk,
                                                                             gv, S, I));
    SWG_KERNEL_CHECK(ctx);
  }
  swg_arena_restore(ctx, mark);
  *taken = 1;
  return SWG_OK;
}
