// Host side of swg_filter_multi (sweepga_amd/csrc/host/shard_host.h) without a GPU: plan + scatter + merge on an S-pan-shaped
// record set (BASELINE.json configs[3]: G single-chromosome genomes, every ordered non-self pair, pair-major order), the
// per-device filter call replaced by a stand-in that numbers "chains" locally per shard, and the result checked against a
// serial restatement of the same protocol.  Prints one JSON line with the phase times.
//
//   g++ -O2 -std=c++17 -pthread -o shard_host_bench tests/native/shard_host_bench.cpp && ./shard_host_bench 100000000 100 8 64
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <vector>

#include "../../sweepga_amd/csrc/host/shard_host.h"

static double now() {
  return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

int main(int argc, char** argv) {
  const uint64_t n = argc > 1 ? strtoull(argv[1], nullptr, 10) : 1000000;
  const uint32_t G = argc > 2 ? (uint32_t)atoi(argv[2]) : 100;
  const int shards = argc > 3 ? atoi(argv[3]) : 8;
  const int threads = argc > 4 ? atoi(argv[4]) : swg_shard::default_threads(n);
  const bool check = argc > 5 ? atoi(argv[5]) != 0 : n <= 20000000;
  // records: pair-major, sizes varying 1 : 3 over the pairs; a few records of an earlier pair come back later (interleaving)
  std::vector<uint32_t> q_id(n), t_id(n), c32(n), table(G);
  std::vector<double> ident(n);
  std::vector<uint8_t> strand(n);
  for (uint32_t g = 0; g < G; ++g) table[g] = g;
  const uint64_t P = (uint64_t)G * (G - 1);
  {
    uint64_t i = 0, x = 88172645463325252ull;
    for (uint64_t p = 0; p < P && i < n; ++p) {
      const uint64_t size = p + 1 == P ? n - i : std::min<uint64_t>(n - i, (n / P) / 2 + (uint64_t)((n / P) * (double)(p % 7) / 4.0));
      const uint32_t q = (uint32_t)(p / (G - 1));
      uint32_t t = (uint32_t)(p % (G - 1));
      t += t >= q;
      for (uint64_t k = 0; k < size; ++k, ++i) {
        x ^= x << 13; x ^= x >> 7; x ^= x << 17;
        const bool stray = (x & 1023) == 0 && p > 0;  // a record of the first pair, out of place
        q_id[i] = stray ? 0 : q;
        t_id[i] = stray ? 1 : t;
        c32[i] = (uint32_t)(x >> 40);
        ident[i] = 0.5 + (double)(x & 0xffff) / 131072.0;
        strand[i] = (uint8_t)((x >> 20) & 1);
      }
    }
    for (; i < n; ++i) { q_id[i] = 0; t_id[i] = 1; c32[i] = 1; ident[i] = 0.9; strand[i] = 0; }
  }
  swg_records r{};
  r.n = n;
  r.q_id = q_id.data(); r.t_id = t_id.data();
  r.q_start = c32.data(); r.q_end = c32.data(); r.t_start = c32.data(); r.t_end = c32.data();
  r.matches = c32.data(); r.block_len = c32.data();
  r.identity = ident.data(); r.strand = strand.data();
  r.n_seq = G; r.seq_genome_last = table.data(); r.n_genome_last = G; r.seq_genome_two = table.data(); r.n_genome_two = G;
  swg_config cfg{};
  cfg.min_identity = 0.6;  // some records fail step 1: `first` is not simply the pair's first record
  cfg.keep_self = 0;

  swg_shard::Plan plan;
  std::vector<swg_shard::Shard> sh;
  const double t0 = now();
  if (!swg_shard::make_plan(r, cfg, shards, threads, &plan)) return 2;
  const double t1 = now();
  swg_shard::scatter(r, plan, &sh);
  const double t2 = now();
  // stand-in for swg_filter on every shard: status = low bit of the coordinate; "kept chains" numbered 1.. per shard in
  // order of the shard's pairs' first records, 3 chains per pair with retained records, a record's chain = 1 + (x % 3)
  swg_shard::run(shards, [&](int s) {
    swg_shard::Shard& S = sh[s];
    std::map<uint32_t, uint32_t> base;  // pair -> first local chain number, in order of first appearance inside the shard
    uint32_t next = 1;
    for (uint64_t k = 0; k < S.m; ++k) {
      const uint32_t i = S.idx.data()[k];
      const uint32_t p = plan.pair.data()[i];
      const bool retained = ident[i] >= cfg.min_identity && q_id[i] != t_id[i];
      S.status.data()[k] = (uint8_t)(retained ? 1 + (c32[i] & 1) : 0);
      uint32_t c = 0;
      if (retained) {
        auto it = base.find(p);
        if (it == base.end()) { it = base.emplace(p, next).first; next += 3; }
        c = it->second + c32[i] % 3;
      }
      S.chain.data()[k] = c;
    }
  });
  const double t3 = now();
  std::vector<uint8_t> status(n);
  std::vector<uint32_t> chain(n);
  swg_shard::merge(plan, sh, status.data(), chain.data());
  const double t4 = now();
  // serial check of the protocol's result: pairs with chains in order of their first retained record get consecutive
  // number ranges [base, base + (hi - lo)]
  bool ok = true;
  uint64_t maxload = 0, sum = 0;
  for (int s = 0; s < shards; ++s) { maxload = std::max(maxload, plan.load[s]); sum += plan.load[s]; }
  ok = ok && sum == n;
  if (check) {
    std::map<uint64_t, uint32_t> order;  // first retained record -> pair
    std::vector<uint32_t> pr(n);
    std::map<uint64_t, uint32_t> ids;
    for (uint64_t i = 0; i < n; ++i) {
      const uint64_t key = (uint64_t)q_id[i] * G + t_id[i];
      auto it = ids.find(key);
      if (it == ids.end()) it = ids.emplace(key, (uint32_t)ids.size()).first;
      pr[i] = it->second;
      ok = ok && pr[i] == plan.pair.data()[i];
      const bool retained = ident[i] >= cfg.min_identity && q_id[i] != t_id[i];
      ok = ok && status[i] == (uint8_t)(retained ? 1 + (c32[i] & 1) : 0);
    }
    std::vector<uint64_t> first(ids.size(), n);
    std::vector<uint32_t> mn(ids.size(), 0xffffffffu), mx(ids.size(), 0);
    for (uint64_t i = 0; i < n; ++i) {
      const bool retained = ident[i] >= cfg.min_identity && q_id[i] != t_id[i];
      if (retained && first[pr[i]] == n) first[pr[i]] = i;
      if (retained) { mn[pr[i]] = std::min(mn[pr[i]], c32[i] % 3); mx[pr[i]] = std::max(mx[pr[i]], c32[i] % 3); }
    }
    for (uint32_t p = 0; p < ids.size(); ++p)
      if (first[p] != n) order[first[p]] = p;
    std::vector<uint32_t> base(ids.size(), 0);
    uint32_t next = 1;
    for (auto& kv : order) { base[kv.second] = next; next += mx[kv.second] - mn[kv.second] + 1; }
    for (uint64_t i = 0; i < n && ok; ++i) {
      const bool retained = ident[i] >= cfg.min_identity && q_id[i] != t_id[i];
      const uint32_t want = retained ? base[pr[i]] + (c32[i] % 3 - mn[pr[i]]) : 0;
      if (chain[i] != want) { ok = false; fprintf(stderr, "record %llu: chain %u, expected %u\n", (unsigned long long)i, chain[i], want); }
    }
  }
  printf("{\"records\": %llu, \"genomes\": %u, \"pairs\": %u, \"shards\": %d, \"threads\": %d, \"plan_ms\": %.1f, \"scatter_ms\": %.1f, "
         "\"merge_ms\": %.1f, \"host_total_ms\": %.1f, \"stand_in_filter_ms\": %.1f, \"load_max_over_mean\": %.5f, \"checked\": %s, "
         "\"ok\": %s}\n",
         (unsigned long long)n, G, plan.n_pairs, shards, plan.threads, (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t4 - t3) * 1e3,
         ((t2 - t0) + (t4 - t3)) * 1e3, (t3 - t2) * 1e3, (double)maxload / ((double)n / shards), check ? "true" : "false",
         ok ? "true" : "false");
  return ok ? 0 : 1;
}
