// ANI pre-pass on the device: calculate_ani_stats (src/main.rs:334-498) and calculate_ani_n_percentile
// (src/main.rs:500-688) after the host has parsed the ANI view of every line (csrc/host/paf_io.cpp).
//
//   lines taking part  -> compaction of (eligible & select) in file order
//   N-percentile order -> stable radix sort by the descending-sortable f64 key (length / identity /
//                         identity * max(ln(length), 1)), ties keep file order like Rust's sort_by
//   prefix cut         -> inclusive u64 scan of the block lengths in that order, first position whose running
//                         total reaches total_genome_size * percentile / 100 (block lengths must be integral, so
//                         the f64 running sum of the reference is exact and equals the integer one)
//   per-pair sums      -> stable radix sort of the selected prefix by pair id, then ONE thread per pair adds its
//                         matches and block lengths in reference order: the f64 sums are bit-identical to the
//                         reference's sequential accumulation (dv:f: makes matches non-integral, so order matters)
//   median             -> per-pair ratios come back to the host (at most n_pairs values) and are sorted there
#include <algorithm>
#include <cmath>
#include <vector>

#include "swg_internal.h"
#include "swg_log.h"

namespace {

constexpr int EW = 256;
inline unsigned nblk(uint64_t n) { return (unsigned)((n + EW - 1) / EW); }

__device__ __forceinline__ uint64_t desc_key(double v) {  // larger value -> smaller key; -0 == +0
  if (v == 0.0) v = 0.0;
  const uint64_t b = (uint64_t)__double_as_longlong(v);
  const uint64_t asc = (b >> 63) ? ~b : (b | 0x8000000000000000ull);
  return ~asc;
}

// err[0] |= 1: NaN sort key (the reference panics), |= 2: block length not an integer in [0, 2^53)
__global__ __launch_bounds__(EW) void ani_keys_kernel(uint64_t m, const uint32_t* __restrict__ list,
                                                      const double* __restrict__ matches,
                                                      const double* __restrict__ block, int sort,
                                                      uint64_t* __restrict__ key, uint32_t* __restrict__ err) {
  uint64_t j = (uint64_t)blockIdx.x * EW + threadIdx.x;
  if (j >= m) return;
  const uint32_t i = list[j];
  const double b = block[i], mt = matches[i];
  uint32_t e = 0;
  if (!(b >= 0.0 && b < 9007199254740992.0) || b != (double)(uint64_t)b) e |= 2u;
  const double identity = mt / (b > 1.0 ? b : 1.0);  // block_len.max(1.0)
  double v;
  if (sort == SWG_NSORT_LENGTH) {
    v = b;
  } else if (sort == SWG_NSORT_IDENTITY) {
    v = identity;
  } else {
    const double l = b >= 3.0 ? swg_log_glibc(b) : 1.0;  // ln(b).max(1.0); ln(b) > 1 iff b >= 3 for integral b
    v = __dmul_rn(identity, l);
  }
  if (v != v) e |= 1u;
  if (e) atomicOr(err, e);
  key[j] = desc_key(v);
}

__global__ __launch_bounds__(EW) void ani_block_u64_kernel(uint64_t m, const uint32_t* __restrict__ list,
                                                           const double* __restrict__ block,
                                                           uint64_t* __restrict__ out) {
  uint64_t j = (uint64_t)blockIdx.x * EW + threadIdx.x;
  if (j < m) out[j] = (uint64_t)block[list[j]];
}

// first position whose inclusive running total reaches the threshold (cum is non-decreasing)
__global__ __launch_bounds__(EW) void ani_cut_kernel(uint64_t m, const uint64_t* __restrict__ cum, double thr,
                                                     unsigned long long* __restrict__ cut) {
  uint64_t j = (uint64_t)blockIdx.x * EW + threadIdx.x;
  if (j >= m) return;
  const bool here = (double)cum[j] >= thr;
  const bool before = j > 0 && (double)cum[j - 1] >= thr;
  if (here && !before) atomicMin(cut, (unsigned long long)j);
}

__global__ __launch_bounds__(EW) void ani_pair_keys_kernel(uint64_t c, const uint32_t* __restrict__ list,
                                                           const uint32_t* __restrict__ pair,
                                                           uint64_t* __restrict__ key, uint32_t* __restrict__ pos) {
  uint64_t j = (uint64_t)blockIdx.x * EW + threadIdx.x;
  if (j >= c) return;
  key[j] = pair[list[j]];
  pos[j] = (uint32_t)j;
}

__global__ __launch_bounds__(EW) void ani_run_flag_kernel(uint64_t c, const uint64_t* __restrict__ key,
                                                          uint8_t* __restrict__ flag) {
  uint64_t j = (uint64_t)blockIdx.x * EW + threadIdx.x;
  if (j < c) flag[j] = (j == 0 || key[j] != key[j - 1]) ? 1 : 0;
}

// one thread per genome pair: sequential f64 accumulation in reference order
__global__ __launch_bounds__(64) void ani_run_sums_kernel(uint64_t n_runs, uint64_t c, const uint32_t* __restrict__ run_start,
                                                          const uint32_t* __restrict__ pos,
                                                          const uint32_t* __restrict__ list,
                                                          const double* __restrict__ matches,
                                                          const double* __restrict__ block,
                                                          double* __restrict__ sum_m, double* __restrict__ sum_b) {
  uint64_t r = (uint64_t)blockIdx.x * 64 + threadIdx.x;
  if (r >= n_runs) return;
  const uint64_t b = run_start[r], e = r + 1 < n_runs ? run_start[r + 1] : c;
  double sm = 0.0, sb = 0.0;
  for (uint64_t j = b; j < e; ++j) {
    const uint32_t i = list[pos[j]];
    sm = __dadd_rn(sm, matches[i]);
    sb = __dadd_rn(sb, block[i]);
  }
  sum_m[r] = sm;
  sum_b[r] = sb;
}

}  // namespace

extern "C" int swg_ani_median(swg_ctx* ctx, const swg_ani_input* in, const uint8_t* select, int kind, double percentile, int sort,
                              double* ani50) {
  if (!ctx) return SWG_ERR_INVALID;
  if (!in || !ani50) return swg_set_error(ctx, SWG_ERR_INVALID, "swg_ani_median: NULL argument");
  if (kind < SWG_ANI_ALL || kind > SWG_ANI_NPERCENTILE) return swg_set_error(ctx, SWG_ERR_INVALID, "bad ANI method");
  if (kind == SWG_ANI_NPERCENTILE && (sort < SWG_NSORT_LENGTH || sort > SWG_NSORT_SCORE || !(percentile > 0.0 && percentile <= 100.0)))
    return swg_set_error(ctx, SWG_ERR_INVALID, "bad N-percentile parameters");
  *ani50 = 0.0;
  const uint64_t n = in->n;
  if (n == 0) return SWG_OK;
  if (n >= (uint64_t(1) << 31)) return swg_set_error(ctx, SWG_ERR_RANGE, "too many records");
  if (!in->eligible || !in->pair || !in->matches || !in->block_len) return swg_set_error(ctx, SWG_ERR_INVALID, "NULL column");
  SWG_HIP(ctx, hipSetDevice(ctx->device));
  std::vector<uint8_t> flag(n);
  for (uint64_t k = 0; k < n; ++k) flag[k] = in->eligible[k] && (!select || select[k]);
  const int pair_bits = swg_bits_for(in->n_pairs ? in->n_pairs - 1 : 0) ? swg_bits_for(in->n_pairs ? in->n_pairs - 1 : 0) : 1;
  hipStream_t st = ctx->stream;
  if (ctx->arena_cap == 0) SWG_TRY(swg_arena_reserve(ctx, (size_t)n * 80 + (size_t(8) << 20)));
  std::vector<double> h_m, h_b;
  int rc = swg_run_with_arena(ctx, [&]() -> int {
    h_m.clear();
    h_b.clear();
    uint8_t* d_flag = swg_alloc<uint8_t>(ctx, n);
    uint32_t* d_pair = swg_alloc<uint32_t>(ctx, n);
    double* d_m = swg_alloc<double>(ctx, n);
    double* d_b = swg_alloc<double>(ctx, n);
    uint64_t* d_scal = swg_alloc<uint64_t>(ctx, 4);  // [0] m  [1] cut  [2] err  [3] n_runs
    SWG_CHECK_ARENA(ctx);
    SWG_HIP(ctx, hipMemcpyAsync(d_flag, flag.data(), n, hipMemcpyHostToDevice, st));
    SWG_HIP(ctx, hipMemcpyAsync(d_pair, in->pair, n * 4, hipMemcpyHostToDevice, st));
    SWG_HIP(ctx, hipMemcpyAsync(d_m, in->matches, n * 8, hipMemcpyHostToDevice, st));
    SWG_HIP(ctx, hipMemcpyAsync(d_b, in->block_len, n * 8, hipMemcpyHostToDevice, st));
    SWG_HIP(ctx, hipMemsetAsync(d_scal, 0, 4 * 8, st));
    swg_flag_scan fs;
    SWG_TRY(swg_flags_count(ctx, d_flag, n, &fs, d_scal));
    uint64_t m = 0;
    SWG_TRY(swg_read_scalars(ctx, d_scal, &m, 1));
    if (m == 0) return SWG_OK;  // "No inter-genome alignments": 0.0
    uint32_t* list = swg_alloc<uint32_t>(ctx, m);
    uint32_t* list_alt = swg_alloc<uint32_t>(ctx, m);
    uint64_t* key = swg_alloc<uint64_t>(ctx, m);
    uint64_t* key_alt = swg_alloc<uint64_t>(ctx, m);
    SWG_CHECK_ARENA(ctx);
    SWG_TRY(swg_flags_compact(ctx, fs, list));
    uint64_t cut = m;
    if (kind == SWG_ANI_NPERCENTILE) {
      SWG_LAUNCH(ctx, "ani_keys", ani_keys_kernel<<<nblk(m), EW, 0, st>>>(m, list, d_m, d_b, sort, key, reinterpret_cast<uint32_t*>(d_scal + 2)));
      SWG_KERNEL_CHECK(ctx);
      uint64_t err = 0;
      SWG_TRY(swg_read_scalars(ctx, d_scal + 2, &err, 1));
      if (err & 1) return swg_set_error(ctx, SWG_ERR_INVALID, "NaN sort key in the ANI pass (the reference panics here)");
      if (err & 2)
        return swg_set_error(ctx, SWG_ERR_UNSUPPORTED, "non-integral or negative block length (PAF column 11) in the N-percentile ANI pass");
      SWG_TRY(swg_radix_sort_pairs(ctx, &key, &list, &key_alt, &list_alt, m, 0, 64));
      uint64_t* cum = key_alt;  // free again after the sort
      SWG_LAUNCH(ctx, "ani_block_u64", ani_block_u64_kernel<<<nblk(m), EW, 0, st>>>(m, list, d_b, cum));
      SWG_KERNEL_CHECK(ctx);
      SWG_TRY(swg_inclusive_sum_scan_u64(ctx, cum, cum, m));
      const unsigned long long none = ~0ull;
      SWG_HIP(ctx, hipMemcpyAsync(d_scal + 1, &none, 8, hipMemcpyHostToDevice, st));
      const double thr = in->total_genome_size * (percentile / 100.0);  // main.rs:630
      SWG_LAUNCH(ctx, "ani_cut", ani_cut_kernel<<<nblk(m), EW, 0, st>>>(m, cum, thr, reinterpret_cast<unsigned long long*>(d_scal + 1)));
      SWG_KERNEL_CHECK(ctx);
      uint64_t first = 0;
      SWG_TRY(swg_read_scalars(ctx, d_scal + 1, &first, 1));
      if (first != ~0ull) cut = first + 1;  // the line that crosses the threshold is still counted (main.rs:655-657)
    }
    // ---- per-pair sums over list[0, cut) in that order
    uint32_t* pos = swg_alloc<uint32_t>(ctx, cut);
    uint32_t* pos_alt = swg_alloc<uint32_t>(ctx, cut);
    uint8_t* run_flag = swg_alloc<uint8_t>(ctx, cut);
    SWG_CHECK_ARENA(ctx);
    SWG_LAUNCH(ctx, "ani_pair_keys", ani_pair_keys_kernel<<<nblk(cut), EW, 0, st>>>(cut, list, d_pair, key, pos));
    SWG_KERNEL_CHECK(ctx);
    SWG_TRY(swg_radix_sort_pairs(ctx, &key, &pos, &key_alt, &pos_alt, cut, 0, pair_bits));
    SWG_LAUNCH(ctx, "ani_run_flag", ani_run_flag_kernel<<<nblk(cut), EW, 0, st>>>(cut, key, run_flag));
    SWG_KERNEL_CHECK(ctx);
    swg_flag_scan rs;
    SWG_TRY(swg_flags_count(ctx, run_flag, cut, &rs, d_scal + 3));
    uint64_t n_runs = 0;
    SWG_TRY(swg_read_scalars(ctx, d_scal + 3, &n_runs, 1));
    uint32_t* run_start = swg_alloc<uint32_t>(ctx, n_runs);
    double* sum_m = swg_alloc<double>(ctx, n_runs);
    double* sum_b = swg_alloc<double>(ctx, n_runs);
    SWG_CHECK_ARENA(ctx);
    SWG_TRY(swg_flags_compact(ctx, rs, run_start));
    SWG_LAUNCH(ctx, "ani_run_sums", ani_run_sums_kernel<<<(unsigned)((n_runs + 63) / 64), 64, 0, st>>>(n_runs, cut, run_start, pos, list, d_m,
                                                                                        d_b, sum_m, sum_b));
    SWG_KERNEL_CHECK(ctx);
    h_m.resize(n_runs);
    h_b.resize(n_runs);
    SWG_HIP(ctx, hipMemcpyAsync(h_m.data(), sum_m, n_runs * 8, hipMemcpyDeviceToHost, st));
    SWG_HIP(ctx, hipMemcpyAsync(h_b.data(), sum_b, n_runs * 8, hipMemcpyDeviceToHost, st));
    SWG_HIP(ctx, hipStreamSynchronize(st));
    return SWG_OK;
  });
  if (rc != SWG_OK) return rc;
  if (h_m.empty()) return SWG_OK;
  std::vector<double> ani(h_m.size());
  for (size_t r = 0; r < h_m.size(); ++r) {  // main.rs:464-472
    ani[r] = h_b[r] > 0.0 ? h_m[r] / h_b[r] : 0.0;
    if (ani[r] != ani[r]) return swg_set_error(ctx, SWG_ERR_INVALID, "NaN per-pair ANI (the reference panics here)");
  }
  std::sort(ani.begin(), ani.end());
  const size_t mid = ani.size() / 2;  // main.rs:477-483
  *ani50 = (ani.size() % 2 == 0 && ani.size() > 1) ? (ani[mid - 1] + ani[mid]) / 2.0 : ani[mid];
  return SWG_OK;
}
