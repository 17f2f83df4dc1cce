// TEST INFRASTRUCTURE ONLY -- C entry points over the CPU oracle so that pytest and
// bench.py's cpu_baseline leg can drive it through ctypes.  Not part of the product.
#include <chrono>
#include <cmath>
#include <cstring>
#include <exception>
#include <string>
#include <vector>

#include "sweepga_oracle.h"

using namespace orc;

extern "C" {

struct orc_config {
  uint64_t min_block_length;
  int32_t mapping_filter_mode;
  uint64_t mapping_max_per_query;   // 0 == None
  uint64_t mapping_max_per_target;  // 0 == None
  int32_t scaffold_filter_mode;
  uint64_t scaffold_max_per_query;
  uint64_t scaffold_max_per_target;
  double overlap_threshold;
  uint64_t scaffold_gap;
  uint64_t min_scaffold_length;
  double scaffold_overlap_threshold;
  uint64_t scaffold_max_deviation;
  int32_t scoring_function;
  double min_identity;
  double min_scaffold_identity;
  int32_t keep_self;
  int32_t scaffolds_only;
};

static FilterConfig to_cfg(const orc_config* c) {
  FilterConfig f;
  f.min_block_length = c->min_block_length;
  f.mapping_filter_mode = c->mapping_filter_mode;
  f.mapping_max_per_query = c->mapping_max_per_query;
  f.mapping_max_per_target = c->mapping_max_per_target;
  f.scaffold_filter_mode = c->scaffold_filter_mode;
  f.scaffold_max_per_query = c->scaffold_max_per_query;
  f.scaffold_max_per_target = c->scaffold_max_per_target;
  f.overlap_threshold = c->overlap_threshold;
  f.scaffold_gap = c->scaffold_gap;
  f.min_scaffold_length = c->min_scaffold_length;
  f.scaffold_overlap_threshold = c->scaffold_overlap_threshold;
  f.scaffold_max_deviation = c->scaffold_max_deviation;
  f.scoring_function = c->scoring_function;
  f.min_identity = c->min_identity;
  f.min_scaffold_identity = c->min_scaffold_identity;
  f.keep_self = c->keep_self != 0;
  f.scaffolds_only = c->scaffolds_only != 0;
  return f;
}

// axis: 0 = plane_sweep_query, 1 = plane_sweep_target, 2 = plane_sweep_both.
// keep_out[i] = 1 iff index i is returned.  k == UINT64_MAX means usize::MAX.
int orc_plane_sweep(int axis, uint64_t n, const uint64_t* qs, const uint64_t* qe,
                    const uint64_t* ts, const uint64_t* te, const double* identity, uint64_t k_q,
                    uint64_t k_t, double thr, int scoring, uint8_t* keep_out) {
  std::vector<PlaneSweepMapping> m(n);
  for (uint64_t i = 0; i < n; ++i) m[i] = {(size_t)i, qs[i], qe[i], ts[i], te[i], identity[i], 0};
  std::vector<size_t> kept;
  if (axis == 0)
    kept = plane_sweep_query(m, k_q, thr, scoring);
  else if (axis == 1)
    kept = plane_sweep_target(m, k_t, thr, scoring);
  else
    kept = plane_sweep_both(m, k_q, k_t, thr, scoring);
  std::memset(keep_out, 0, n);
  for (size_t i : kept) keep_out[i] = 1;
  return (int)kept.size();
}

double orc_score(uint64_t qs, uint64_t qe, double identity, int scoring) {
  PlaneSweepMapping m{0, qs, qe, 0, 0, identity, 0};
  return score_with_function(m, scoring);
}

// order_out receives the kept indices in the reference's output order; returns their count.
int64_t orc_plane_sweep_scaffolds(uint64_t n, const char* const* qnames, const char* const* tnames,
                                  const uint64_t* qs, const uint64_t* qe, const uint64_t* ts,
                                  const uint64_t* te, const double* identity, int mode,
                                  uint64_t max_q, uint64_t max_t, double thr, int scoring,
                                  uint64_t* order_out) {
  std::vector<ChainView> chains(n);
  for (uint64_t i = 0; i < n; ++i)
    chains[i] = {qnames[i], tnames[i], qs[i], qe[i], ts[i], te[i], identity[i]};
  std::vector<size_t> kept = plane_sweep_scaffolds(chains, mode, max_q, max_t, thr, scoring);
  for (size_t i = 0; i < kept.size(); ++i) order_out[i] = kept[i];
  return (int64_t)kept.size();
}

// Unions (xs[e], ys[e]) in order; set_of[i] = position of i's set in get_sets() order;
// returns the number of sets.
int64_t orc_union_find_sets(uint64_t n, uint64_t m, const uint64_t* xs, const uint64_t* ys,
                            uint64_t* set_of) {
  UnionFind uf(n);
  for (uint64_t e = 0; e < m; ++e) uf.unite(xs[e], ys[e]);
  auto sets = uf.get_sets();
  for (size_t s = 0; s < sets.size(); ++s)
    for (size_t i : sets[s]) set_of[i] = s;
  return (int64_t)sets.size();
}

int64_t orc_plane_sweep_core(uint64_t n, const uint32_t* begin, const uint32_t* end,
                             const double* score, uint64_t max_to_keep, double thr,
                             uint64_t* order_out) {
  std::vector<Interval> iv(n);
  for (uint64_t i = 0; i < n; ++i) iv[i] = {(size_t)i, begin[i], end[i], score[i], 0};
  std::vector<size_t> kept = plane_sweep_core(iv, max_to_keep, thr);
  for (size_t i = 0; i < kept.size(); ++i) order_out[i] = kept[i];
  return (int64_t)kept.size();
}

static std::vector<RecordMeta> build_records(uint64_t n, const uint64_t* rank,
                                             const char* const* qnames, const char* const* tnames,
                                             const uint64_t* qs, const uint64_t* qe,
                                             const uint64_t* ts, const uint64_t* te,
                                             const uint64_t* block_length, const double* identity,
                                             const uint64_t* matches, const char* strand) {
  std::vector<RecordMeta> md(n);
  for (uint64_t i = 0; i < n; ++i) {
    RecordMeta& m = md[i];
    m.rank = rank ? (size_t)rank[i] : (size_t)i;
    m.query_name = qnames[i];
    m.target_name = tnames[i];
    m.query_start = qs[i];
    m.query_end = qe[i];
    m.target_start = ts[i];
    m.target_end = te[i];
    m.block_length = block_length[i];
    m.identity = identity[i];
    m.matches = matches[i];
    m.alignment_length = block_length[i];
    m.strand = strand[i] == '+' ? '+' : '-';
  }
  return md;
}

// PafFilter::apply_filters.  status_out[i]: 0 dropped, 1 scaffold, 2 rescued, 3 unassigned.
// chain_out[i]: N of "chain_N", 0 = no ch:Z: tag.  Returns number kept, or -1 on error.
// seconds_out (optional) receives the wall time of apply_filters alone (records already built).
// 1: apply_filters evaluates step 4b (paf_filter.rs:535-597) through a bucket index instead of the literal
// chains x reverse-mappings loop (same result, tests/test_oracle_fast_cpu.py); for full-size checks of deep chromosome pairs.
void orc_set_fast_inversion(int on) { g_fast_inversion = on != 0; }

int64_t orc_apply_filters(const orc_config* cfg, uint64_t n, const uint64_t* rank,
                          const char* const* qnames, const char* const* tnames,
                          const uint64_t* qs, const uint64_t* qe, const uint64_t* ts,
                          const uint64_t* te, const uint64_t* block_length, const double* identity,
                          const uint64_t* matches, const char* strand, uint8_t* status_out,
                          uint32_t* chain_out, double* seconds_out) {
  try {
    std::vector<RecordMeta> md =
        build_records(n, rank, qnames, tnames, qs, qe, ts, te, block_length, identity, matches, strand);
    std::vector<size_t> ranks(n);
    for (uint64_t i = 0; i < n; ++i) ranks[i] = md[i].rank;
    PafFilter f(to_cfg(cfg));
    auto t0 = std::chrono::steady_clock::now();
    auto passing = f.apply_filters(std::move(md));
    auto t1 = std::chrono::steady_clock::now();
    if (seconds_out) *seconds_out = std::chrono::duration<double>(t1 - t0).count();
    for (uint64_t i = 0; i < n; ++i) {
      auto it = passing.find(ranks[i]);
      if (it == passing.end()) {
        status_out[i] = 0;
        chain_out[i] = 0;
      } else {
        status_out[i] = (uint8_t)it->second.chain_status;
        chain_out[i] = it->second.has_chain_id
                           ? (uint32_t)std::stoul(it->second.chain_id.substr(6))
                           : 0u;
      }
    }
    return (int64_t)passing.size();
  } catch (const std::exception&) {
    return -1;
  }
}

// The same over SoA columns with integer sequence ids (u32 coordinates, strand 0 = '+'), the layout the device
// library takes: lets the checker run on millions of records without a Python string per record.  The records
// are [lo, hi) of the columns; rank = index - lo.
int64_t orc_apply_filters_ids(const orc_config* cfg, uint64_t lo, uint64_t hi, const uint32_t* q_id, const uint32_t* t_id,
                              const char* const* names, const uint32_t* qs, const uint32_t* qe, const uint32_t* ts,
                              const uint32_t* te, const uint32_t* block_length, const double* identity,
                              const uint32_t* matches, const uint8_t* strand, uint8_t* status_out, uint32_t* chain_out,
                              double* seconds_out) {
  try {
    const uint64_t n = hi - lo;
    std::vector<RecordMeta> md(n);
    for (uint64_t k = 0; k < n; ++k) {
      const uint64_t i = lo + k;
      RecordMeta& m = md[k];
      m.rank = k;
      m.query_name = names[q_id[i]];
      m.target_name = names[t_id[i]];
      m.query_start = qs[i];
      m.query_end = qe[i];
      m.target_start = ts[i];
      m.target_end = te[i];
      m.block_length = block_length[i];
      m.identity = identity[i];
      m.matches = matches[i];
      m.alignment_length = block_length[i];
      m.strand = strand[i] ? '-' : '+';
    }
    PafFilter f(to_cfg(cfg));
    auto t0 = std::chrono::steady_clock::now();
    auto passing = f.apply_filters(std::move(md));
    auto t1 = std::chrono::steady_clock::now();
    if (seconds_out) *seconds_out = std::chrono::duration<double>(t1 - t0).count();
    for (uint64_t k = 0; k < n; ++k) {
      auto it = passing.find(k);
      if (it == passing.end()) {
        status_out[k] = 0;
        chain_out[k] = 0;
      } else {
        status_out[k] = (uint8_t)it->second.chain_status;
        chain_out[k] = it->second.has_chain_id ? (uint32_t)std::stoul(it->second.chain_id.substr(6)) : 0u;
      }
    }
    return (int64_t)passing.size();
  } catch (const std::exception&) {
    return -1;
  }
}

// merge_mappings_into_chains on the given records (no retain, no sweep).  chain_of[i] = index
// of record i's chain in all_chains order.  Per-chain outputs are sized n.  Returns #chains.
int64_t orc_merge_chains(uint64_t n, const char* const* qnames, const char* const* tnames,
                         const uint64_t* qs, const uint64_t* qe, const uint64_t* ts,
                         const uint64_t* te, const uint64_t* block_length, const uint64_t* matches,
                         const char* strand, uint64_t max_gap, uint32_t* chain_of, uint64_t* c_qs,
                         uint64_t* c_qe, uint64_t* c_ts, uint64_t* c_te, uint64_t* c_total_length,
                         double* c_weighted_identity) {
  std::vector<double> ident(n, 1.0);
  std::vector<RecordMeta> md =
      build_records(n, nullptr, qnames, tnames, qs, qe, ts, te, block_length, ident.data(), matches, strand);
  FilterConfig cfg;
  PafFilter f(cfg);
  std::vector<MergedChain> chains = f.merge_mappings_into_chains(md, max_gap);
  for (size_t c = 0; c < chains.size(); ++c) {
    for (size_t r : chains[c].member_indices) chain_of[r] = (uint32_t)c;
    c_qs[c] = chains[c].query_start;
    c_qe[c] = chains[c].query_end;
    c_ts[c] = chains[c].target_start;
    c_te[c] = chains[c].target_end;
    c_total_length[c] = chains[c].total_length;
    c_weighted_identity[c] = chains[c].weighted_identity;
  }
  return (int64_t)chains.size();
}

int orc_filter_paf(const orc_config* cfg, const char* in_path, const char* out_path) {
  try {
    PafFilter f(to_cfg(cfg));
    f.filter_paf(in_path, out_path);
    return 0;
  } catch (const std::exception&) {
    return -1;
  }
}

// Parses PAF text into column arrays (extract_metadata); names are returned as indices into a
// caller-visible table via callbacks being overkill, so this entry only reports counts and the
// numeric columns; tests that need names parse them in Python.  Returns #records or -1.
int64_t orc_extract_metadata(const char* path, uint64_t cap, uint64_t* rank, uint64_t* qs,
                             uint64_t* qe, uint64_t* ts, uint64_t* te, uint64_t* block_length,
                             double* identity, uint64_t* matches, char* strand) {
  try {
    FilterConfig cfg;
    PafFilter f(cfg);
    std::vector<RecordMeta> md = f.extract_metadata(path);
    if (md.size() > cap) return -2;
    for (size_t i = 0; i < md.size(); ++i) {
      rank[i] = md[i].rank;
      qs[i] = md[i].query_start;
      qe[i] = md[i].query_end;
      ts[i] = md[i].target_start;
      te[i] = md[i].target_end;
      block_length[i] = md[i].block_length;
      identity[i] = md[i].identity;
      matches[i] = md[i].matches;
      strand[i] = md[i].strand;
    }
    return (int64_t)md.size();
  } catch (const std::exception&) {
    return -1;
  }
}

int orc_parse_filter_mode(const char* s, int32_t* mode, uint64_t* pq, uint64_t* pt) {
  int m;
  if (!parse_filter_mode(s, &m, pq, pt)) return 0;
  *mode = m;
  return 1;
}
int orc_parse_metric_number(const char* s, uint64_t* out) { return parse_metric_number(s, out) ? 1 : 0; }
int orc_parse_identity_value(const char* s, double* out) { return parse_identity_value(s, out) ? 1 : 0; }
// ani_percentile < 0 = None
int orc_parse_identity_value_ani(const char* s, double ani_percentile, double* out) {
  return parse_identity_value(s, out, ani_percentile) ? 1 : 0;
}
// main.rs:296-330 ; returns 0 for None
int orc_parse_ani_method(const char* s, int* kind, double* percentile, int* sort) {
  AniMethod m;
  if (!parse_ani_method(s, &m)) return 0;
  *kind = m.kind;
  *percentile = m.percentile;
  *sort = m.sort;
  return 1;
}
// main.rs:334-688 ; returns 0 on success, -1 on error (I/O, NaN where the reference panics)
int orc_calculate_ani_stats(const char* path, int kind, double percentile, int sort, double* out) {
  try {
    AniMethod m;
    m.kind = kind;
    m.percentile = percentile;
    m.sort = sort;
    *out = calculate_ani_stats(path, m);
    return 0;
  } catch (const std::exception&) {
    return -1;
  }
}
int orc_parse_scoring(const char* s) { return parse_scoring(s); }
uint64_t orc_round_nice(uint64_t v) { return round_nice(v); }
void orc_clamp_scaffold_params(uint64_t j, uint64_t m, int have_avg, uint64_t avg, int adaptive,
                               uint64_t* jo, uint64_t* mo) {
  clamp_scaffold_params(j, m, have_avg != 0, avg, adaptive != 0, jo, mo);
}
int orc_parse_cigar_counts(const char* s, uint64_t* m, uint64_t* x, uint64_t* i, uint64_t* d) {
  return parse_cigar_counts(s, m, x, i, d) ? 1 : 0;
}

// Host libm log (what Rust's f64::ln calls on Linux/glibc) for checking the device log.
double orc_log(double x) { return std::log(x); }
void orc_log_array(uint64_t n, const double* x, double* y) {
  for (uint64_t i = 0; i < n; ++i) y[i] = std::log(x[i]);
}
// ln(L) for L = first .. first+n-1 (integer lengths), for exhaustive device-log checks.
void orc_log_range(uint64_t first, uint64_t n, double* y) {
  for (uint64_t i = 0; i < n; ++i) y[i] = std::log((double)(first + i));
}

// extract_1aln_metadata's record derivation (src/unified_filter.rs:83-142) over decoded alignments.  Names come back
// through name_out (n query names then n target names, name_cap bytes each, NUL-terminated).
int64_t orc_records_from_1aln(uint64_t n, const char* const* qname, const char* const* tname, const uint64_t* qs,
                              const uint64_t* qe, const uint64_t* ts, const uint64_t* te, const uint64_t* matches,
                              const char* strand, uint64_t* block_length, double* identity, char* name_out,
                              uint64_t name_cap) {
  try {
    std::vector<AlnRecord> alns(n);
    for (uint64_t i = 0; i < n; ++i) {
      alns[i].query_name = qname[i];
      alns[i].target_name = tname[i];
      alns[i].query_start = qs[i];
      alns[i].query_end = qe[i];
      alns[i].target_start = ts[i];
      alns[i].target_end = te[i];
      alns[i].matches = matches[i];
      alns[i].strand = strand[i];
    }
    const std::vector<RecordMeta> md = records_from_1aln(alns);
    for (uint64_t i = 0; i < n; ++i) {
      block_length[i] = md[i].block_length;
      identity[i] = md[i].identity;
      if (md[i].query_name.size() + 1 > name_cap || md[i].target_name.size() + 1 > name_cap) return -2;
      std::memcpy(name_out + i * name_cap, md[i].query_name.c_str(), md[i].query_name.size() + 1);
      std::memcpy(name_out + (n + i) * name_cap, md[i].target_name.c_str(), md[i].target_name.size() + 1);
    }
    return (int64_t)n;
  } catch (const std::exception&) {
    return -1;
  }
}

// tree_filter::apply_tree_filter_to_paf (src/tree_filter.rs:205-285) over PAF text in memory (lines split like BufRead::lines:
// "\n" or "\r\n").  Returns the length of the filtered text written to out (cap bytes), or -2 if it does not fit.
int64_t orc_tree_filter_text(const char* text, uint64_t len, uint64_t k_nearest, uint64_t k_farthest, double random_fraction,
                             char* out, uint64_t cap) {
  try {
    std::vector<std::string> lines;
    uint64_t pos = 0;
    while (pos < len) {
      const void* nl = std::memchr(text + pos, '\n', len - pos);
      uint64_t end = nl ? (uint64_t)((const char*)nl - text) : len;
      uint64_t ll = end - pos;
      if (ll && text[pos + ll - 1] == '\r' && nl) --ll;
      lines.emplace_back(text + pos, ll);
      pos = end + 1;
    }
    const std::vector<std::string> kept = tree_filter_paf_lines(lines, (size_t)k_nearest, (size_t)k_farthest, random_fraction);
    uint64_t o = 0;
    for (const std::string& l : kept) {
      if (o + l.size() + 1 > cap) return -2;
      std::memcpy(out + o, l.data(), l.size());
      o += l.size();
      out[o++] = '\n';
    }
    return (int64_t)o;
  } catch (const std::exception&) {
    return -1;
  }
}
uint64_t orc_default_hash_str_pair(const char* a, const char* b) { return default_hash_str_pair(a, b); }

}  // extern "C"
