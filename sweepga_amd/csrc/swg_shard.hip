// One record set over several devices of a node: host-side partition, one swg_filter per
// context on its own host thread, host-side merge.  No collective: genome pairs are independent units of the
// filter -- every sweep segment (src/paf_filter.rs:1037-1100), chain group (:761-770), scaffold chromosome pair
// (src/plane_sweep_scaffold.rs:116-130) and rescue pair (src/paf_filter.rs:625-629) nests inside one pair.
//
// Chain numbers are global in the reference: kept chains are numbered genome pair by genome pair, in the order
// the pairs first appear (src/paf_filter.rs:517-521 over plane_sweep_scaffolds' output order).  After the
// gather every shard-local number is shifted by the number of kept chains of all pairs that appear earlier.
// That is exact when the reference's two genome-prefix rules (last '#' vs first two '#' parts) induce the same
// partition of the sequences; otherwise the call falls back to one device.
#include <algorithm>
#include <atomic>
#include <thread>
#include <unordered_map>
#include <vector>

#include "swg_internal.h"

namespace {

bool same_partition(const swg_records* r) {
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

struct Shard {
  std::vector<uint32_t> idx;  // record indices, ascending
  std::vector<uint32_t> q_id, t_id, qs, qe, ts, te, matches, block;
  std::vector<double> identity;
  std::vector<uint8_t> strand, status;
  std::vector<uint32_t> chain;
  swg_stats stats{};
  int rc = SWG_OK;
};

}  // namespace

extern "C" int swg_filter_multi(swg_ctx* const* ctxs, int n_ctx, const swg_records* r, const swg_config* cfg, uint8_t* status_out,
                                uint32_t* chain_out, swg_stats* stats) {
  if (!ctxs || n_ctx < 1 || !ctxs[0]) return SWG_ERR_INVALID;
  swg_ctx* ctx0 = ctxs[0];
  if (!r || !cfg) return swg_set_error(ctx0, SWG_ERR_INVALID, "records/config is NULL");
  for (int d = 0; d < n_ctx; ++d)
    if (!ctxs[d]) return swg_set_error(ctx0, SWG_ERR_INVALID, "context %d is NULL", d);
  const uint64_t n = r->n;
  if (n_ctx == 1 || n == 0 || !r->seq_genome_last || !r->seq_genome_two || !same_partition(r))
    return swg_filter(ctx0, r, cfg, status_out, chain_out, stats);
  if (n >= (uint64_t(1) << 31)) return swg_set_error(ctx0, SWG_ERR_RANGE, "more than 2^31-1 records");
  if (!r->q_id || !r->t_id || !r->q_start || !r->q_end || !r->t_start || !r->t_end || !r->identity || !r->matches || !r->block_len ||
      !r->strand || !status_out || !chain_out)
    return swg_set_error(ctx0, SWG_ERR_INVALID, "a record column or an output buffer is NULL");
  for (uint64_t i = 0; i < n; ++i)
    if (r->q_id[i] >= r->n_seq || r->t_id[i] >= r->n_seq) return swg_set_error(ctx0, SWG_ERR_INVALID, "sequence id out of range");

  // ---- genome pairs (dense ids in first-appearance order) and their sizes
  std::vector<uint32_t> pair(n);
  std::vector<uint64_t> count;
  {
    std::unordered_map<uint64_t, uint32_t> ids;
    uint64_t last_key = ~0ull;
    uint32_t last_id = 0;
    for (uint64_t i = 0; i < n; ++i) {
      const uint64_t key = (uint64_t)r->seq_genome_two[r->q_id[i]] * r->n_genome_two + r->seq_genome_two[r->t_id[i]];
      if (key != last_key) {
        auto it = ids.find(key);
        if (it == ids.end()) {
          it = ids.emplace(key, (uint32_t)count.size()).first;
          count.push_back(0);
        }
        last_key = key;
        last_id = it->second;
      }
      pair[i] = last_id;
      ++count[last_id];
    }
  }
  const uint32_t n_pairs = (uint32_t)count.size();
  // ---- longest-processing-time bin packing by mapping count (deterministic: ties by pair id)
  std::vector<uint32_t> order(n_pairs);
  for (uint32_t p = 0; p < n_pairs; ++p) order[p] = p;
  std::sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return count[a] != count[b] ? count[a] > count[b] : a < b; });
  std::vector<int> shard_of_pair(n_pairs, 0);
  std::vector<uint64_t> load(n_ctx, 0);
  for (uint32_t p : order) {
    const int s = (int)(std::min_element(load.begin(), load.end()) - load.begin());
    shard_of_pair[p] = s;
    load[s] += count[p];
  }
  std::vector<Shard> sh(n_ctx);
  for (int s = 0; s < n_ctx; ++s) sh[s].idx.reserve(load[s]);
  for (uint64_t i = 0; i < n; ++i) sh[shard_of_pair[pair[i]]].idx.push_back((uint32_t)i);

  // ---- one host thread per device: gather the shard's columns, filter, keep the results
  std::vector<std::thread> pool;
  auto work = [&](int s) {
    Shard& S = sh[s];
    const size_t m = S.idx.size();
    if (m == 0) return;
    auto take32 = [&](const uint32_t* src, std::vector<uint32_t>* dst) {
      dst->resize(m);
      for (size_t k = 0; k < m; ++k) (*dst)[k] = src[S.idx[k]];
    };
    take32(r->q_id, &S.q_id);
    take32(r->t_id, &S.t_id);
    take32(r->q_start, &S.qs);
    take32(r->q_end, &S.qe);
    take32(r->t_start, &S.ts);
    take32(r->t_end, &S.te);
    take32(r->matches, &S.matches);
    take32(r->block_len, &S.block);
    S.identity.resize(m);
    S.strand.resize(m);
    for (size_t k = 0; k < m; ++k) {
      S.identity[k] = r->identity[S.idx[k]];
      S.strand[k] = r->strand[S.idx[k]];
    }
    S.status.assign(m, 0);
    S.chain.assign(m, 0);
    swg_records sub = *r;
    sub.n = m;
    sub.q_id = S.q_id.data();
    sub.t_id = S.t_id.data();
    sub.q_start = S.qs.data();
    sub.q_end = S.qe.data();
    sub.t_start = S.ts.data();
    sub.t_end = S.te.data();
    sub.identity = S.identity.data();
    sub.matches = S.matches.data();
    sub.block_len = S.block.data();
    sub.strand = S.strand.data();
    S.rc = swg_filter(ctxs[s], &sub, cfg, S.status.data(), S.chain.data(), &S.stats);
  };
  for (int s = 1; s < n_ctx; ++s) pool.emplace_back(work, s);
  work(0);
  for (auto& t : pool) t.join();
  for (int s = 0; s < n_ctx; ++s)
    if (sh[s].rc != SWG_OK) {
      const std::string msg = swg_last_error(ctxs[s]);
      return swg_set_error(ctx0, sh[s].rc, "shard %d of %d: %s", s, n_ctx, msg.c_str());
    }

  // ---- merge
  bool any_chain = false;
  for (int s = 0; s < n_ctx; ++s) {
    const Shard& S = sh[s];
    for (size_t k = 0; k < S.idx.size(); ++k) {
      status_out[S.idx[k]] = S.status[k];
      chain_out[S.idx[k]] = S.chain[k];
      any_chain |= S.chain[k] != 0;
    }
  }
  if (any_chain) {
    // per pair: range of shard-local chain numbers, and first retained record (step-1 predicate, src/paf_filter.rs:384-388)
    std::vector<uint32_t> lo(n_pairs, 0xffffffffu), hi(n_pairs, 0);
    std::vector<uint64_t> first(n_pairs, n);
    for (uint64_t i = 0; i < n; ++i) {
      const uint32_t p = pair[i];
      const uint32_t c = chain_out[i];
      if (c) {
        if (c < lo[p]) lo[p] = c;
        if (c > hi[p]) hi[p] = c;
      }
      if (first[p] == n && (uint64_t)r->block_len[i] >= cfg->min_block_length && (cfg->keep_self || r->q_id[i] != r->t_id[i]) &&
          r->identity[i] >= cfg->min_identity)
        first[p] = i;
    }
    std::vector<uint32_t> with;
    for (uint32_t p = 0; p < n_pairs; ++p)
      if (hi[p]) with.push_back(p);
    std::stable_sort(with.begin(), with.end(), [&](uint32_t a, uint32_t b) { return first[a] < first[b]; });
    std::vector<int64_t> shift(n_pairs, 0);
    int64_t offset = 0;
    for (uint32_t p : with) {
      shift[p] = offset - ((int64_t)lo[p] - 1);
      offset += (int64_t)hi[p] - (int64_t)lo[p] + 1;
    }
    for (uint64_t i = 0; i < n; ++i)
      if (chain_out[i]) chain_out[i] = (uint32_t)((int64_t)chain_out[i] + shift[pair[i]]);
  }
  if (stats) {
    *stats = swg_stats{};
    stats->n_in = n;
    for (int s = 0; s < n_ctx; ++s) {
      const swg_stats& a = sh[s].stats;
      stats->n_retained += a.n_retained;
      stats->n_swept += a.n_swept;
      stats->n_chains += a.n_chains;
      stats->n_chains_kept += a.n_chains_kept;
      stats->n_out += a.n_out;
      stats->device_ms = std::max(stats->device_ms, a.device_ms);  // devices run concurrently
      stats->h2d_ms = std::max(stats->h2d_ms, a.h2d_ms);
      stats->d2h_ms = std::max(stats->d2h_ms, a.d2h_ms);
    }
  }
  return SWG_OK;
}

// RecordMeta's own widths: rebased on the host first (ctxs[0] keeps the 32-bit columns), then the same sharding.
extern "C" int swg_filter_multi64(swg_ctx* const* ctxs, int n_ctx, const swg_records64* r, const swg_config* cfg, uint8_t* status_out,
                                  uint32_t* chain_out, swg_stats* stats) {
  if (!ctxs || n_ctx < 1 || !ctxs[0]) return SWG_ERR_INVALID;
  swg_records v;
  SWG_TRY(swg_rebase_host(ctxs[0], r, cfg, &v));
  return swg_filter_multi(ctxs, n_ctx, &v, cfg, status_out, chain_out, stats);
}
