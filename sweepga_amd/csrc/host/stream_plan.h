// Chunk plan of the streamed host path (swg_filter / swg_filter_multi, csrc/swg_stream.hip).  No HIP in this file.
//
// Genome pairs are independent units of the filter (every sweep segment, chain group, scaffold chromosome
// pair and rescue pair nests inside one pair), and every pair (A, B) has ONE query genome, A.  An aligner writes its PAF
// query by query, so the records of a query genome normally form one contiguous run of the input.  When they do, any cut
// of the input at run boundaries splits it into record ranges that are closed under genome pairs: each range can be
// filtered on its own -- on the same device while the next range is still crossing PCIe, or on another device -- with
// NO host-side scatter or merge: a range is a slice [lo, hi) of the caller's own columns and its results land in the same
// slice of the caller's output arrays.  Chain numbers: the reference numbers kept chains genome pair by genome pair in the
// order the pairs first appear (src/paf_filter.rs:517-521); every pair of an earlier range appears before every pair of a
// later one, so a range's local numbers are shifted by the kept chains of all earlier ranges.
//
// plan(): runs of equal query genome (the coarser, first-two-'#'-parts prefix of src/plane_sweep_scaffold.rs:13-22, which is
// safe for both prefix rules when they induce the same partition -- the caller checks that), found on host threads; the
// plan is refused (false) when a genome has two runs (records not grouped by query genome: the caller takes the monolithic
// path) or a sequence id is out of range.
#ifndef SWG_HOST_STREAM_PLAN_H
#define SWG_HOST_STREAM_PLAN_H

#include <algorithm>
#include <cstdint>
#include <vector>

#include "../../../include/sweepga_gpu.h"
#include "threads.h"

namespace swg_streamed {

struct Chunk {
  uint64_t lo, hi;  // records [lo, hi)
};

// Whether the reference's two genome-prefix rules (up to the last '#', src/paf_filter.rs:1022-1030; first two '#' parts,
// src/plane_sweep_scaffold.rs:13-22) split the sequences into the same genomes.  Only then is "genome pair" one notion and
// a cut along it exact for every stage of the filter.
inline bool same_partition(const swg_records* r) {
  std::vector<int64_t> l2t(r->n_genome_last, -1), t2l(r->n_genome_two, -1);
  for (uint32_t s = 0; s < r->n_seq; ++s) {
    const uint32_t a = r->seq_genome_last[s], b = r->seq_genome_two[s];
    if (a >= r->n_genome_last || b >= r->n_genome_two) return false;
    if (l2t[a] < 0) l2t[a] = b;
    if (t2l[b] < 0) t2l[b] = a;
    if (l2t[a] != (int64_t)b || t2l[b] != (int64_t)a) return false;
  }
  return true;
}

// Runs of equal query genome -> chunks of whole runs, each at least `target` records (the last one may be smaller; a run
// longer than `target` is a chunk of its own).  Returns false when the input is not grouped by query genome.
inline bool plan(uint64_t n, const uint32_t* q_id, const uint32_t* seq_genome, uint32_t n_seq, uint32_t n_genome, uint64_t target,
                 int threads, std::vector<Chunk>* out) {
  out->clear();
  if (n == 0) return true;
  if (threads < 1) threads = 1;
  if ((uint64_t)threads > n / 65536 + 1) threads = (int)(n / 65536 + 1);
  struct Run {
    uint64_t start;
    uint32_t genome;
  };
  struct Slice {
    std::vector<Run> runs;
    bool bad = false;
  };
  std::vector<Slice> sl((size_t)threads);
  constexpr size_t MAX_RUNS_PER_SLICE = size_t(1) << 16;  // more boundaries than this: interleaved input, not worth chunking
  auto lo = [&](int t) { return n / (uint64_t)threads * (uint64_t)t + ((uint64_t)t < n % (uint64_t)threads ? (uint64_t)t : n % (uint64_t)threads); };
  swg_host::run(threads, [&](int t) {
    Slice& s = sl[(size_t)t];
    uint32_t cur_seq = 0xffffffffu, cur_g = 0xffffffffu;
    for (uint64_t i = lo(t), e = lo(t + 1); i < e; ++i) {
      const uint32_t q = q_id[i];
      if (q == cur_seq) continue;  // the common case: same query sequence as the record before
      if (q >= n_seq) {
        s.bad = true;
        return;
      }
      cur_seq = q;
      const uint32_t g = seq_genome[q];
      if (g == cur_g) continue;
      if (g >= n_genome || s.runs.size() >= MAX_RUNS_PER_SLICE) {
        s.bad = true;
        return;
      }
      cur_g = g;
      s.runs.push_back({i, g});
    }
  });
  std::vector<uint8_t> seen((size_t)n_genome, 0);
  std::vector<uint64_t> starts;  // run starts, merged over the slices
  uint32_t last_g = 0xffffffffu;
  for (const Slice& s : sl) {
    if (s.bad) return false;
    for (const Run& r : s.runs) {
      if (r.genome == last_g) continue;  // a run that continues across a slice boundary
      if (seen[r.genome]) return false;  // second run of a genome: not grouped
      seen[r.genome] = 1;
      last_g = r.genome;
      starts.push_back(r.start);
    }
  }
  starts.push_back(n);
  if (target < 1) target = 1;
  uint64_t begin = 0;
  for (size_t k = 1; k < starts.size(); ++k) {
    if (starts[k] - begin >= target || k + 1 == starts.size()) {
      out->push_back({begin, starts[k]});
      begin = starts[k];
    }
  }
  // a small tail is glued to the chunk before it (a chunk's fixed cost is ~100 launches)
  if (out->size() >= 2 && out->back().hi - out->back().lo < target / 2) {
    const uint64_t hi = out->back().hi;
    out->pop_back();
    out->back().hi = hi;
  }
  return true;
}

// Longest-processing-time assignment of the chunks to `n_dev` devices (chunk sizes are what the devices' times follow).
// Returns for every device its chunk indices in ascending (= input) order.
inline std::vector<std::vector<int>> assign(const std::vector<Chunk>& chunks, int n_dev) {
  std::vector<std::vector<int>> per((size_t)n_dev);
  std::vector<uint64_t> load((size_t)n_dev, 0);
  std::vector<int> order(chunks.size());
  for (size_t k = 0; k < chunks.size(); ++k) order[k] = (int)k;
  std::stable_sort(order.begin(), order.end(), [&](int a, int b) {
    return chunks[(size_t)a].hi - chunks[(size_t)a].lo > chunks[(size_t)b].hi - chunks[(size_t)b].lo;
  });
  for (int k : order) {
    int best = 0;
    for (int d = 1; d < n_dev; ++d)
      if (load[(size_t)d] < load[(size_t)best]) best = d;
    per[(size_t)best].push_back(k);
    load[(size_t)best] += chunks[(size_t)k].hi - chunks[(size_t)k].lo;
  }
  for (auto& v : per) std::sort(v.begin(), v.end());
  return per;
}

}  // namespace swg_streamed
#endif
