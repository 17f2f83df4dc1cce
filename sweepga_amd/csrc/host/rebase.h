// RecordMeta holds u64 coordinates (src/paf_filter.rs:58-62); the device layout is 32-bit.  Every quantity
// apply_filters derives from coordinates is a difference, an order or a midpoint of positions ON ONE SEQUENCE (spans,
// overlaps, chaining gaps, bounding boxes, diagonals t - q of one sequence pair, rescue distances, the saturating
// window starts, which only ever clamp below a sequence's smallest coordinate), so subtracting a per-sequence constant
// from all of a sequence's coordinates -- query and target appearances alike -- changes no result.  The constant is the
// smallest coordinate the sequence has anywhere in the record set; what must fit 32 bits is then only the stretch of
// each sequence that mappings touch, not its absolute position.  Host version (threads); the device version of the same
// two steps is in swg_filter.hip.  A sequence whose touched stretch is wider than that gets the finer partition of
// columns_by_axis (below) before anything is refused.
#ifndef SWG_HOST_REBASE_H
#define SWG_HOST_REBASE_H

#include <cstdint>
#include <thread>
#include <vector>

#include "threads.h"

namespace swg_rebase {

struct Result {
  bool ok = true;
  uint64_t bad_record = 0;
  int bad_field = -1;  // 0..3 = q_start, q_end, t_start, t_end (mapped stretch of the sequence >= 2^32); 4 matches; 5 block length;
                       // 6 = a sequence id >= n_seq (nothing was rebased)
};

inline const char* field_name(int f) {
  static const char* const N[6] = {"query_start", "query_end", "target_start", "target_end", "matches", "block_length"};
  return (f >= 0 && f < 6) ? N[f] : "?";
}

using swg_host::run;

// c64 = {q_start, q_end, t_start, t_end, matches, block_len}; matches / block_len may be NULL (then c32[4] / c32[5] are
// left alone).  lo[n_seq] receives the offsets (UINT64_MAX for a sequence no record names).
inline Result columns(uint64_t n, const uint32_t* q_id, const uint32_t* t_id, const uint64_t* const c64[6], uint32_t n_seq,
                      int threads, uint32_t* const c32[6], uint64_t* lo) {
  if (threads < 1) threads = 1;
  if ((uint64_t)threads > n / 65536 + 1) threads = (int)(n / 65536 + 1);
  std::vector<std::vector<uint64_t>> part(threads > 1 ? threads : 0);
  for (uint32_t s = 0; s < n_seq; ++s) lo[s] = UINT64_MAX;
  std::vector<uint64_t> bad_id(threads, UINT64_MAX);
  for (auto& pt : part) pt.assign(n_seq, UINT64_MAX);  // before the threads start: a worker body allocates nothing
  run(threads, [&](int t) {
    uint64_t* m = threads > 1 ? part[t].data() : lo;
    const uint64_t b = n * (uint64_t)t / threads, e = n * (uint64_t)(t + 1) / threads;
    for (uint64_t i = b; i < e; ++i) {
      if (q_id[i] >= n_seq || t_id[i] >= n_seq) {  // the ids index the table: checked in the same pass
        if (bad_id[t] == UINT64_MAX) bad_id[t] = i;
        continue;
      }
      uint64_t& mq = m[q_id[i]];
      const uint64_t q = c64[0][i] < c64[1][i] ? c64[0][i] : c64[1][i];
      if (q < mq) mq = q;
      uint64_t& mt = m[t_id[i]];
      const uint64_t tt = c64[2][i] < c64[3][i] ? c64[2][i] : c64[3][i];
      if (tt < mt) mt = tt;
    }
  });
  for (int t = 0; t < threads; ++t)
    if (bad_id[t] != UINT64_MAX) {
      Result r;
      r.ok = false;
      r.bad_record = bad_id[t];
      r.bad_field = 6;
      return r;
    }
  for (int t = 0; t < threads && threads > 1; ++t)
    for (uint32_t s = 0; s < n_seq; ++s)
      if (part[t][s] < lo[s]) lo[s] = part[t][s];
  std::vector<Result> res(threads);
  run(threads, [&](int t) {
    Result& r = res[t];
    const uint64_t b = n * (uint64_t)t / threads, e = n * (uint64_t)(t + 1) / threads;
    for (uint64_t i = b; i < e; ++i) {
      const uint64_t oq = lo[q_id[i]], ot = lo[t_id[i]];
      const uint64_t v[6] = {c64[0][i] - oq, c64[1][i] - oq, c64[2][i] - ot, c64[3][i] - ot, c64[4] ? c64[4][i] : 0,
                             c64[5] ? c64[5][i] : 0};
      for (int f = 0; f < 6; ++f) {
        if ((v[f] >> 32) && r.ok) {
          r.ok = false;
          r.bad_record = i;
          r.bad_field = f;
        }
        if (f < 4 || c64[f]) c32[f][i] = (uint32_t)v[f];
      }
    }
  });
  for (auto& r : res)
    if (!r.ok) return r;
  return Result{};
}

// The finer partition, for a sequence whose mapped stretch does not fit 32 bits as a whole (round 6).  Nothing in
// apply_filters ever compares the QUERY coordinates of two records unless they share the query sequence and the genome of the
// target -- the mapping-level sweep's query-axis segment (src/paf_filter.rs:1037-1100); chains, the scaffold sweep, the inversion
// capture and the rescue all nest inside one (query, target) pair, which nests inside that segment -- and likewise the TARGET
// coordinates unless they share the target sequence and the genome of the query.  So the constant may differ from segment to
// segment: q coordinates are rebased to the smallest one of their (q_id, genome(t_id)) segment, t coordinates to the smallest
// one of their (t_id, genome(q_id)) segment, and what must fit 32 bits is the stretch of a sequence touched by the mappings
// against ONE genome.  seq_genome = the records' seq_genome_last table.  Two tables of n_seq * n_genome offsets (the caller
// checks that this is affordable: axis_tables_fit).  Same result convention as columns(); ids are taken as checked.
inline bool axis_tables_fit(uint32_t n_seq, uint32_t n_genome) { return (uint64_t)n_seq * n_genome <= (uint64_t(1) << 24); }
inline Result columns_by_axis(uint64_t n, const uint32_t* q_id, const uint32_t* t_id, const uint64_t* const c64[6], uint32_t n_seq,
                              const uint32_t* seq_genome, uint32_t n_genome, int threads, uint32_t* const c32[6],
                              uint64_t* rec_off_q = nullptr, uint64_t* rec_off_t = nullptr) {
  if (threads < 1) threads = 1;
  if ((uint64_t)threads > n / 65536 + 1) threads = (int)(n / 65536 + 1);
  const size_t cells = (size_t)n_seq * n_genome;
  std::vector<uint64_t> lo_q(cells, UINT64_MAX), lo_t(cells, UINT64_MAX);
  auto amin = [](uint64_t* cell, uint64_t v) {
    uint64_t cur = __atomic_load_n(cell, __ATOMIC_RELAXED);
    while (v < cur && !__atomic_compare_exchange_n(cell, &cur, v, true, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {
    }
  };
  run(threads, [&](int t) {
    const uint64_t b = n * (uint64_t)t / threads, e = n * (uint64_t)(t + 1) / threads;
    for (uint64_t i = b; i < e; ++i) {
      const size_t cq = (size_t)q_id[i] * n_genome + seq_genome[t_id[i]], ct = (size_t)t_id[i] * n_genome + seq_genome[q_id[i]];
      amin(&lo_q[cq], c64[0][i] < c64[1][i] ? c64[0][i] : c64[1][i]);
      amin(&lo_t[ct], c64[2][i] < c64[3][i] ? c64[2][i] : c64[3][i]);
    }
  });
  std::vector<Result> res(threads);
  run(threads, [&](int t) {
    Result& r = res[t];
    const uint64_t b = n * (uint64_t)t / threads, e = n * (uint64_t)(t + 1) / threads;
    for (uint64_t i = b; i < e; ++i) {
      const uint64_t oq = lo_q[(size_t)q_id[i] * n_genome + seq_genome[t_id[i]]], ot = lo_t[(size_t)t_id[i] * n_genome + seq_genome[q_id[i]]];
      if (rec_off_q) rec_off_q[i] = oq;  // (what was taken off this record's coordinates, for a front end that publishes it)
      if (rec_off_t) rec_off_t[i] = ot;
      const uint64_t v[6] = {c64[0][i] - oq, c64[1][i] - oq, c64[2][i] - ot, c64[3][i] - ot, c64[4] ? c64[4][i] : 0,
                             c64[5] ? c64[5][i] : 0};
      for (int f = 0; f < 6; ++f) {
        if ((v[f] >> 32) && r.ok) {
          r.ok = false;
          r.bad_record = i;
          r.bad_field = f;
        }
        if (f < 4 || c64[f]) c32[f][i] = (uint32_t)v[f];
      }
    }
  });
  for (auto& r : res)
    if (!r.ok) return r;
  return Result{};
}

}  // namespace swg_rebase

#endif
