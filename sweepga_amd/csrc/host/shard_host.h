// Host side of swg_filter_multi: one record set split over the devices of a node by genome pair.
// Genome pairs are independent units of the filter -- every sweep segment (src/paf_filter.rs:1037-1100), chain group
// (:761-770), scaffold chromosome pair (src/plane_sweep_scaffold.rs:116-130) and rescue pair (src/paf_filter.rs:625-629)
// nests inside one pair -- so the data path has no collective; what the host does around the per-device filter calls is
//
//   plan      genome pair of every record (dense ids in first-appearance order), pair sizes, first retained record of
//             every pair, longest-processing-time packing of the pairs onto the shards
//   scatter   every record copied once into its shard's columns (positions by a counting sort: ascending record index
//             inside a shard)
//   merge     results back to record order; chain numbers are global in the reference (kept chains are numbered genome
//             pair by genome pair in the order the pairs first appear, src/paf_filter.rs:517-521), so every shard-local
//             number is shifted by the number of kept chains of all pairs that appear earlier
//
// all of it on host threads over slices of the record set (round 2 did these passes on one thread: seconds per 10^8
// records around 40 ms of device time).  No HIP in this file: tests/native/shard_host_bench.cpp times it on any machine.
#ifndef SWG_HOST_SHARD_H
#define SWG_HOST_SHARD_H

#include <algorithm>
#include <cstdint>
#include <cstdlib>
#include <new>
#include <cstring>
#include <memory>
#include <thread>
#include <unordered_map>
#include <vector>

#include "../../../include/sweepga_gpu.h"
#include "threads.h"

namespace swg_shard {

using swg_host::run;

inline int default_threads(uint64_t n) {
  int t = (int)std::thread::hardware_concurrency();
  if (t < 1) t = 1;
  if (t > 16) t = 16;  // measured: more slices in flight make the first-touch pass slower, see scatter()
  if ((uint64_t)t > n / 65536 + 1) t = (int)(n / 65536 + 1);
  return t;
}

// Uninitialised column storage (std::vector would zero-fill on one thread first).  Plain malloc: 2 MB-aligned blocks
// marked MADV_HUGEPAGE were tried and made the first touch slower, not faster, where transparent huge pages defragment on
// fault (this build box: scatter 81 -> 253 ms per 3*10^6 records).
template <class T>
struct Column {
  T* p = nullptr;
  Column() = default;
  Column(const Column&) = delete;
  Column& operator=(const Column&) = delete;
  Column(Column&& o) noexcept : p(o.p) { o.p = nullptr; }
  Column& operator=(Column&& o) noexcept {
    if (this != &o) {
      std::free(p);
      p = o.p;
      o.p = nullptr;
    }
    return *this;
  }
  ~Column() { std::free(p); }
  void alloc(size_t n) {
    std::free(p);
    p = nullptr;
    if (!n) return;
    p = static_cast<T*>(std::malloc(n * sizeof(T)));
    if (!p) throw std::bad_alloc();
  }
  T* data() const { return p; }
};

struct Shard {
  uint64_t m = 0;
  Column<uint32_t> idx, q_id, t_id, qs, qe, ts, te, matches, block, chain;
  Column<double> identity;
  Column<uint8_t> strand, status;
  swg_stats stats{};
  int rc = SWG_OK;
  swg_records view(const swg_records& whole) const {
    swg_records sub = whole;
    sub.n = m;
    sub.q_id = q_id.data();
    sub.t_id = t_id.data();
    sub.q_start = qs.data();
    sub.q_end = qe.data();
    sub.t_start = ts.data();
    sub.t_end = te.data();
    sub.identity = whole.identity ? identity.data() : nullptr;  // NULL = derived from matches / block length on the device
    sub.matches = matches.data();
    sub.block_len = block.data();
    sub.strand = strand.data();
    return sub;
  }
};

struct Plan {
  int threads = 1, n_shards = 1;
  uint64_t n = 0;
  uint32_t n_pairs = 0;
  Column<uint32_t> pair;            // [n] dense genome-pair id, first-appearance order
  std::vector<uint64_t> count;      // [n_pairs]
  std::vector<uint64_t> first;      // [n_pairs] first record passing the step-1 predicate (n = none)
  std::vector<int> shard_of_pair;   // [n_pairs]
  std::vector<uint64_t> load;       // [n_shards]
  std::vector<uint64_t> slice_off;  // [threads + 1][n_shards] write positions of every slice in every shard
  uint64_t bad_record = UINT64_MAX; // a sequence id >= n_seq
};

// Step 1: pairs, sizes, first retained records, LPT packing.  Returns false when a sequence id is out of range.
inline bool make_plan(const swg_records& r, const swg_config& cfg, int n_shards, int threads, Plan* P) {
  const uint64_t n = r.n;
  P->n = n;
  P->n_shards = n_shards;
  if (threads < 1) threads = 1;
  if ((uint64_t)threads > n / 65536 + 1) threads = (int)(n / 65536 + 1);
  P->threads = threads;
  P->pair.alloc(n);
  uint32_t* pair = P->pair.data();
  const uint32_t G = r.n_genome_two;
  // pass A (threads): the distinct pair keys of every slice in order of first appearance; slice-local ids for now
  struct Slice {
    std::vector<uint64_t> keys;  // distinct, first-appearance order inside the slice
    uint64_t bad = UINT64_MAX;
  };
  std::vector<Slice> sl(threads);
  run(threads, [&](int t) {
    const uint64_t b = n * (uint64_t)t / threads, e = n * (uint64_t)(t + 1) / threads;
    Slice& S = sl[t];
    std::unordered_map<uint64_t, uint32_t> ids;
    uint64_t last_key = ~0ull;
    uint32_t last_id = 0;
    for (uint64_t i = b; i < e; ++i) {
      const uint32_t q = r.q_id[i], tt = r.t_id[i];
      if (q >= r.n_seq || tt >= r.n_seq) {
        S.bad = i;
        return;
      }
      const uint64_t key = (uint64_t)r.seq_genome_two[q] * G + r.seq_genome_two[tt];
      if (key != last_key) {  // PAFs are written pair by pair: the map is consulted once per run
        auto it = ids.find(key);
        if (it == ids.end()) {
          it = ids.emplace(key, (uint32_t)S.keys.size()).first;
          S.keys.push_back(key);
        }
        last_key = key;
        last_id = it->second;
      }
      pair[i] = last_id;
    }
  });
  for (int t = 0; t < threads; ++t)
    if (sl[t].bad != UINT64_MAX) {
      P->bad_record = sl[t].bad;
      return false;
    }
  // merge in slice order = first appearance over the whole record set
  std::unordered_map<uint64_t, uint32_t> gid;
  std::vector<std::vector<uint32_t>> remap(threads);
  for (int t = 0; t < threads; ++t) {
    remap[t].resize(sl[t].keys.size());
    for (size_t k = 0; k < sl[t].keys.size(); ++k) {
      auto it = gid.find(sl[t].keys[k]);
      if (it == gid.end()) it = gid.emplace(sl[t].keys[k], (uint32_t)gid.size()).first;
      remap[t][k] = it->second;
    }
  }
  const uint32_t np = (uint32_t)gid.size();
  P->n_pairs = np;
  // pass B (threads): global ids, sizes, first retained record (step-1 predicate, src/paf_filter.rs:384-388)
  std::vector<std::vector<uint64_t>> cnt(threads), fst(threads);
  for (int t = 0; t < threads; ++t) {
    cnt[t].assign(np, 0);
    fst[t].assign(np, n);
  }
  run(threads, [&](int t) {
    const uint64_t b = n * (uint64_t)t / threads, e = n * (uint64_t)(t + 1) / threads;
    const uint32_t* mp = remap[t].data();
    uint64_t* c = cnt[t].data();
    uint64_t* f = fst[t].data();
    for (uint64_t i = b; i < e; ++i) {
      const uint32_t p = mp[pair[i]];
      pair[i] = p;
      ++c[p];
      if (f[p] == n && (uint64_t)r.block_len[i] >= cfg.min_block_length && (cfg.keep_self || r.q_id[i] != r.t_id[i]) &&
          (r.identity ? r.identity[i] : (double)r.matches[i] / (double)(r.block_len[i] > 1 ? r.block_len[i] : 1)) >= cfg.min_identity)
        f[p] = i;
    }
  });
  P->count.assign(np, 0);
  P->first.assign(np, n);
  for (int t = 0; t < threads; ++t)
    for (uint32_t p = 0; p < np; ++p) {
      P->count[p] += cnt[t][p];
      if (fst[t][p] < P->first[p]) P->first[p] = fst[t][p];
    }
  // longest-processing-time bin packing by mapping count (deterministic: ties by pair id)
  std::vector<uint32_t> order(np);
  for (uint32_t p = 0; p < np; ++p) order[p] = p;
  std::sort(order.begin(), order.end(),
            [&](uint32_t a, uint32_t b) { return P->count[a] != P->count[b] ? P->count[a] > P->count[b] : a < b; });
  P->shard_of_pair.assign(np, 0);
  P->load.assign(n_shards, 0);
  for (uint32_t p : order) {
    const int s = (int)(std::min_element(P->load.begin(), P->load.end()) - P->load.begin());
    P->shard_of_pair[p] = s;
    P->load[s] += P->count[p];
  }
  // counting sort positions: slice t writes shard s from slice_off[t][s] on (ascending record index inside a shard)
  P->slice_off.assign((size_t)(threads + 1) * n_shards, 0);
  for (int t = 0; t < threads; ++t)
    for (uint32_t p = 0; p < np; ++p) P->slice_off[(size_t)(t + 1) * n_shards + P->shard_of_pair[p]] += cnt[t][p];
  for (int s = 0; s < n_shards; ++s) {
    uint64_t acc = 0;
    for (int t = 0; t <= threads; ++t) {
      const uint64_t c = P->slice_off[(size_t)t * n_shards + s];
      acc += c;
      P->slice_off[(size_t)t * n_shards + s] = acc;  // inclusive over slices 0..t-1 (row 0 is zero)
    }
  }
  return true;
}

// Step 2: every record copied once into its shard (threads over slices; positions from the plan).
inline void scatter(const swg_records& r, const Plan& P, std::vector<Shard>* shards) {
  std::vector<Shard>& sh = *shards;
  const int ns = P.n_shards, threads = P.threads;
  sh.resize(ns);
  run(std::min(ns, threads), [&](int s0) {
    for (int s = s0; s < ns; s += std::min(ns, threads)) {
      Shard& S = sh[s];
      S.m = P.load[s];
      S.idx.alloc(S.m); S.q_id.alloc(S.m); S.t_id.alloc(S.m); S.qs.alloc(S.m); S.qe.alloc(S.m); S.ts.alloc(S.m);
      S.te.alloc(S.m); S.matches.alloc(S.m); S.block.alloc(S.m); S.chain.alloc(S.m); if (r.identity) S.identity.alloc(S.m);
      S.strand.alloc(S.m); S.status.alloc(S.m);
    }
  });
  const uint32_t* pair = P.pair.data();
  const uint64_t n = P.n;
  // (the shard columns are fresh memory: the first touch of ~50 bytes per record is what this pass costs, and page faults of
  // many threads in one address space get in each other's way -- 10^8 records on a 256-thread host: 273 ms with 16 slices in
  // flight, 372 ms with 64, 410 ms with 128; the plan's slices are kept, every slice is still written by one thread)
  run(threads, [&](int t) {
    const uint64_t b = n * (uint64_t)t / threads, e = n * (uint64_t)(t + 1) / threads;
    std::vector<uint64_t> pos(P.slice_off.begin() + (size_t)t * ns, P.slice_off.begin() + (size_t)(t + 1) * ns);
    for (uint64_t i = b; i < e; ++i) {
      const int s = P.shard_of_pair[pair[i]];
      Shard& S = sh[s];
      const uint64_t k = pos[s]++;
      S.idx.data()[k] = (uint32_t)i;
      S.q_id.data()[k] = r.q_id[i];
      S.t_id.data()[k] = r.t_id[i];
      S.qs.data()[k] = r.q_start[i];
      S.qe.data()[k] = r.q_end[i];
      S.ts.data()[k] = r.t_start[i];
      S.te.data()[k] = r.t_end[i];
      S.matches.data()[k] = r.matches[i];
      S.block.data()[k] = r.block_len[i];
      if (r.identity) S.identity.data()[k] = r.identity[i];
      S.strand.data()[k] = r.strand[i];
    }
  });
}

// Step 2, without the copies (round 6): only every shard's list of record indices (ascending) and room for its results; the
// records themselves go from the caller's columns to the device through a pinned ring (swg_filter_gathered).  4 bytes per
// record instead of ~50 of fresh memory.
inline void scatter_indices(const Plan& P, std::vector<Shard>* shards) {
  std::vector<Shard>& sh = *shards;
  const int ns = P.n_shards, threads = P.threads;
  sh.resize(ns);
  run(std::min(ns, threads), [&](int s0) {
    for (int s = s0; s < ns; s += std::min(ns, threads)) {
      Shard& S = sh[s];
      S.m = P.load[s];
      S.idx.alloc(S.m);
      S.chain.alloc(S.m);
      S.status.alloc(S.m);
    }
  });
  const uint32_t* pair = P.pair.data();
  const uint64_t n = P.n;
  run(threads, [&](int t) {
    const uint64_t b = n * (uint64_t)t / threads, e = n * (uint64_t)(t + 1) / threads;
    std::vector<uint64_t> pos(P.slice_off.begin() + (size_t)t * ns, P.slice_off.begin() + (size_t)(t + 1) * ns);
    for (uint64_t i = b; i < e; ++i) {
      const int s = P.shard_of_pair[pair[i]];
      sh[s].idx.data()[pos[s]++] = (uint32_t)i;
    }
  });
}

// Step 3: results back to record order, chain numbers made global.
inline void merge(const Plan& P, const std::vector<Shard>& sh, uint8_t* status_out, uint32_t* chain_out) {
  const int ns = P.n_shards, threads = P.threads;
  const uint32_t np = P.n_pairs;
  const uint32_t* pair = P.pair.data();
  // (shard, slice-of-shard) tasks, dealt round-robin over the threads
  struct Task { int s; uint64_t b, e; };
  std::vector<Task> tasks;
  for (int s = 0; s < ns; ++s) {
    const uint64_t m = sh[s].m;
    const int parts = (int)std::min<uint64_t>((uint64_t)threads, m / 65536 + 1);
    for (int k = 0; k < parts; ++k) tasks.push_back({s, m * (uint64_t)k / parts, m * (uint64_t)(k + 1) / parts});
  }
  // per pair: range of shard-local chain numbers
  std::vector<std::vector<uint32_t>> lo_t(threads), hi_t(threads);
  run(threads, [&](int t) {
    lo_t[t].assign(np, 0xffffffffu);
    hi_t[t].assign(np, 0);
    uint32_t* lo = lo_t[t].data();
    uint32_t* hi = hi_t[t].data();
    for (size_t k = (size_t)t; k < tasks.size(); k += (size_t)threads) {
      const Shard& S = sh[tasks[k].s];
      const uint32_t* idx = S.idx.data();
      const uint32_t* ch = S.chain.data();
      for (uint64_t j = tasks[k].b; j < tasks[k].e; ++j) {
        const uint32_t c = ch[j];
        if (!c) continue;
        const uint32_t p = pair[idx[j]];
        if (c < lo[p]) lo[p] = c;
        if (c > hi[p]) hi[p] = c;
      }
    }
  });
  std::vector<uint32_t> lo(np, 0xffffffffu), hi(np, 0);
  for (int t = 0; t < threads; ++t)
    for (uint32_t p = 0; p < np; ++p) {
      if (lo_t[t][p] < lo[p]) lo[p] = lo_t[t][p];
      if (hi_t[t][p] > hi[p]) hi[p] = hi_t[t][p];
    }
  std::vector<uint32_t> with;
  for (uint32_t p = 0; p < np; ++p)
    if (hi[p]) with.push_back(p);
  std::stable_sort(with.begin(), with.end(), [&](uint32_t a, uint32_t b) { return P.first[a] < P.first[b]; });
  std::vector<int64_t> shift(np, 0);
  int64_t offset = 0;
  for (uint32_t p : with) {
    shift[p] = offset - ((int64_t)lo[p] - 1);
    offset += (int64_t)hi[p] - (int64_t)lo[p] + 1;
  }
  run(threads, [&](int t) {
    for (size_t k = (size_t)t; k < tasks.size(); k += (size_t)threads) {
      const Shard& S = sh[tasks[k].s];
      const uint32_t* idx = S.idx.data();
      const uint32_t* ch = S.chain.data();
      const uint8_t* st = S.status.data();
      for (uint64_t j = tasks[k].b; j < tasks[k].e; ++j) {
        const uint32_t i = idx[j];
        status_out[i] = st[j];
        const uint32_t c = ch[j];
        chain_out[i] = c ? (uint32_t)((int64_t)c + shift[pair[i]]) : 0u;
      }
    }
  });
}

}  // namespace swg_shard
#endif
