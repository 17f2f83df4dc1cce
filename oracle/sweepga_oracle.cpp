// TEST INFRASTRUCTURE ONLY -- see sweepga_oracle.h.
//
// Literal CPU restatement of the reference filter path.  Every function cites the
// reference file:line it follows (paths relative to pangenome/sweepga).  Containers
// mirror the reference's: IndexMap -> insertion-ordered map, BTreeSet -> std::set with
// the same comparator, stable sort_by_key -> std::stable_sort, f64::ln -> glibc log().
//
// Parity: pinned by the reference's known-answer tests only (no Rust toolchain here, so
// no oracle/_ref build).  One reference behaviour is not deterministic -- the ch:Z: tag
// of *rescued* mappings depends on HashSet iteration order (paf_filter.rs:637-644,
// 690-716).  This restatement iterates anchors in ascending input order, which is one
// admissible instance of the reference's behaviour.
#include "sweepga_oracle.h"

#include <chrono>
#include <cstdio>
#include <cstdlib>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <limits>
#include <map>
#include <set>
#include <stdexcept>
#include <tuple>
#include <unordered_set>
#include <unistd.h>

namespace orc {

// ---------------------------------------------------------------------------------------
// Insertion-ordered map (indexmap::IndexMap): iteration order == first-insertion order.
// ---------------------------------------------------------------------------------------
template <class K, class V>
struct IndexMap {
  std::vector<std::pair<K, V>> items;
  std::map<K, size_t> index;
  V& entry(const K& k) {
    auto it = index.find(k);
    if (it == index.end()) {
      index.emplace(k, items.size());
      items.emplace_back(k, V());
      return items.back().second;
    }
    return items[it->second].second;
  }
  const V* get(const K& k) const {
    auto it = index.find(k);
    return it == index.end() ? nullptr : &items[it->second].second;
  }
};

// ---------------------------------------------------------------------------------------
// plane_sweep_exact.rs
// ---------------------------------------------------------------------------------------
static const uint8_t FLAG_DISCARD = 0x01, FLAG_OVERLAPPED = 0x02;

// plane_sweep_exact.rs:29-86.  Length is always the QUERY span.
double score_with_function(const PlaneSweepMapping& m, int scoring) {
  const double NEG_INF = -std::numeric_limits<double>::infinity();
  const double length = (double)(m.query_end - m.query_start);  // u64 wrapping sub, as f64
  switch (scoring) {
    case SCORE_IDENTITY:  // :39-46
      return m.identity <= 0.0 ? NEG_INF : m.identity;
    case SCORE_LENGTH:  // :48-56
      return length <= 0.0 ? NEG_INF : length;
    case SCORE_LENGTH_IDENTITY:  // :58-66
    case SCORE_MATCHES:          // :78-86
      return (length <= 0.0 || m.identity <= 0.0) ? NEG_INF : length * m.identity;
    case SCORE_LOG_LENGTH_IDENTITY:  // :68-76
    default:
      return (length <= 0.0 || m.identity <= 0.0) ? NEG_INF : m.identity * std::log(length);
  }
}

// plane_sweep_exact.rs:113-144
static double axis_overlap(const PlaneSweepMapping& a, const PlaneSweepMapping& b, bool query) {
  uint64_t as = query ? a.query_start : a.target_start, ae = query ? a.query_end : a.target_end;
  uint64_t bs = query ? b.query_start : b.target_start, be = query ? b.query_end : b.target_end;
  uint64_t ostart = std::max(as, bs), oend = std::min(ae, be);
  int64_t d = (int64_t)oend - (int64_t)ostart;
  double overlap_len = (double)std::max<int64_t>(d, 0);
  double self_len = (double)(ae - as), other_len = (double)(be - bs);
  double min_len = std::min(self_len, other_len);
  return min_len > 0.0 ? overlap_len / min_len : 0.0;
}

// plane_sweep_exact.rs:163-194 (MappingOrder::cmp)
struct MappingOrder {
  size_t idx;
  double score;
  uint64_t start_pos;
};
struct MappingOrderLess {
  bool operator()(const MappingOrder& a, const MappingOrder& b) const {
    // other.score.partial_cmp(&self.score).unwrap_or(Equal): descending score, NaN -> Equal
    if (b.score < a.score) return true;   // a has higher score -> a first
    if (b.score > a.score) return false;
    if (a.start_pos != b.start_pos) return a.start_pos < b.start_pos;
    return a.idx < b.idx;
  }
};
using Bst = std::set<MappingOrder, MappingOrderLess>;

// plane_sweep_exact.rs:197-259
static void mark_good(const Bst& bst, std::vector<PlaneSweepMapping>& m, uint64_t keep, double thr,
                      bool query_axis) {
  if (bst.empty()) return;
  std::vector<size_t> kept_indices;
  uint64_t kept = 0;
  for (const auto& mo : bst) {
    if (kept >= keep) break;
    m[mo.idx].flags &= (uint8_t)~FLAG_DISCARD;
    kept_indices.push_back(mo.idx);
    ++kept;
  }
  if (thr < 1.0) {
    std::unordered_set<size_t> kept_set(kept_indices.begin(), kept_indices.end());
    for (const auto& mo : bst) {
      size_t idx = mo.idx;
      if (kept_set.count(idx)) continue;
      for (size_t kidx : kept_indices) {
        double ov = axis_overlap(m[idx], m[kidx], query_axis);
        if (ov > thr) {
          m[idx].flags |= FLAG_OVERLAPPED;
          m[idx].flags |= FLAG_DISCARD;
          break;
        }
      }
    }
  }
}

struct Event {
  uint64_t position;
  int type;  // Begin = 0, End = 1
  size_t mapping_idx;
};

// plane_sweep_exact.rs:268-352 (query) and :355-433 (target) share one body here.
static std::vector<size_t> plane_sweep_axis(std::vector<PlaneSweepMapping>& m, uint64_t keep,
                                            double thr, int scoring, bool query_axis) {
  std::vector<size_t> out;
  if (m.size() <= 1) {  // :274-276
    for (size_t i = 0; i < m.size(); ++i) out.push_back(i);
    return out;
  }
  for (auto& x : m) {  // :279-282
    x.flags |= FLAG_DISCARD;
    x.flags &= (uint8_t)~FLAG_OVERLAPPED;
  }
  std::vector<Event> events;
  events.reserve(m.size() * 2);
  for (size_t i = 0; i < m.size(); ++i) {
    events.push_back({query_axis ? m[i].query_start : m[i].target_start, 0, i});
    events.push_back({query_axis ? m[i].query_end : m[i].target_end, 1, i});
  }
  // :300 sort_by_key(|e| (e.position, e.event_type)) -- stable
  std::stable_sort(events.begin(), events.end(), [](const Event& a, const Event& b) {
    if (a.position != b.position) return a.position < b.position;
    return a.type < b.type;
  });
  Bst bst;
  size_t i = 0;
  while (i < events.size()) {
    uint64_t pos = events[i].position;
    size_t j = i;
    while (j < events.size() && events[j].position == pos) ++j;
    for (size_t e = i; e < j; ++e) {
      const Event& ev = events[e];
      MappingOrder mo{ev.mapping_idx, score_with_function(m[ev.mapping_idx], scoring),
                      query_axis ? m[ev.mapping_idx].query_start : m[ev.mapping_idx].target_start};
      if (ev.type == 0)
        bst.insert(mo);
      else
        bst.erase(mo);
    }
    mark_good(bst, m, keep, thr, query_axis);
    i = j;
  }
  for (size_t k = 0; k < m.size(); ++k)
    if (!(m[k].flags & FLAG_DISCARD) && !(m[k].flags & FLAG_OVERLAPPED)) out.push_back(k);
  return out;
}

std::vector<size_t> plane_sweep_query(std::vector<PlaneSweepMapping>& m, uint64_t keep, double thr,
                                      int scoring) {
  return plane_sweep_axis(m, keep, thr, scoring, true);
}
std::vector<size_t> plane_sweep_target(std::vector<PlaneSweepMapping>& m, uint64_t keep,
                                       double thr, int scoring) {
  return plane_sweep_axis(m, keep, thr, scoring, false);
}

// plane_sweep_exact.rs:436-461
std::vector<size_t> plane_sweep_both(std::vector<PlaneSweepMapping>& m, uint64_t qkeep,
                                     uint64_t tkeep, double thr, int scoring) {
  std::vector<size_t> query_kept = plane_sweep_query(m, qkeep, thr, scoring);
  std::vector<PlaneSweepMapping> filtered;
  filtered.reserve(query_kept.size());
  for (size_t idx : query_kept) filtered.push_back(m[idx]);
  std::vector<size_t> target_kept = plane_sweep_target(filtered, tkeep, thr, scoring);
  std::vector<size_t> out;
  for (size_t idx : target_kept) out.push_back(query_kept[idx]);
  return out;
}

// ---------------------------------------------------------------------------------------
// plane_sweep_scaffold.rs
// ---------------------------------------------------------------------------------------
// plane_sweep_scaffold.rs:13-22 : first two '#'-separated parts + '#', else whole name
static std::string extract_genome_prefix2(const std::string& name) {
  size_t p1 = name.find('#');
  if (p1 == std::string::npos) return name;  // parts.len() == 1
  size_t p2 = name.find('#', p1 + 1);
  std::string part0 = name.substr(0, p1);
  std::string part1 = p2 == std::string::npos ? name.substr(p1 + 1) : name.substr(p1 + 1, p2 - p1 - 1);
  return part0 + "#" + part1 + "#";
}

// plane_sweep_scaffold.rs:47-251.  apply_one_to_one_sweep and apply_many_sweep differ only in
// the limits handed to plane_sweep_both.
std::vector<size_t> plane_sweep_scaffolds(const std::vector<ChainView>& chains, int mode,
                                          uint64_t max_per_query, uint64_t max_per_target,
                                          double thr, int scoring) {
  std::vector<size_t> all_kept;
  if (chains.size() <= 1) {  // :55-57
    for (size_t i = 0; i < chains.size(); ++i) all_kept.push_back(i);
    return all_kept;
  }
  std::vector<PlaneSweepMapping> psm(chains.size());
  for (size_t i = 0; i < chains.size(); ++i)
    psm[i] = {i, chains[i].query_start, chains[i].query_end, chains[i].target_start,
              chains[i].target_end, chains[i].identity, 0};
  uint64_t qlim, tlim;
  if (mode == ONE_TO_ONE) {  // :81-84, :168-174
    qlim = 1;
    tlim = 1;
  } else {  // :85-91, :199-200
    qlim = max_per_query ? max_per_query : K_INF;
    tlim = max_per_target ? max_per_target : K_INF;
  }
  using Key = std::pair<std::string, std::string>;
  IndexMap<Key, IndexMap<Key, std::vector<size_t>>> genome_pairs;  // :113-130 / :204-218
  for (size_t i = 0; i < chains.size(); ++i) {
    Key gp(extract_genome_prefix2(chains[i].query_name), extract_genome_prefix2(chains[i].target_name));
    Key cp(chains[i].query_name, chains[i].target_name);
    genome_pairs.entry(gp).entry(cp).push_back(i);
  }
  for (auto& gp : genome_pairs.items) {
    for (auto& cp : gp.second.items) {
      const std::vector<size_t>& indices = cp.second;
      if (indices.empty()) continue;
      std::vector<PlaneSweepMapping> pair_mappings;
      for (size_t i : indices) pair_mappings.push_back(psm[i]);
      std::vector<size_t> kept = plane_sweep_both(pair_mappings, qlim, tlim, thr, scoring);
      for (size_t local : kept) all_kept.push_back(indices[local]);
    }
  }
  return all_kept;
}

// ---------------------------------------------------------------------------------------
// union_find.rs
// ---------------------------------------------------------------------------------------
UnionFind::UnionFind(size_t n) : parent(n), rank(n, 0) {
  for (size_t i = 0; i < n; ++i) parent[i] = i;
}
size_t UnionFind::find(size_t x) {  // :17-22 (recursive path compression; iterative here, same result)
  size_t root = x;
  while (parent[root] != root) root = parent[root];
  while (parent[x] != root) {
    size_t next = parent[x];
    parent[x] = root;
    x = next;
  }
  return root;
}
void UnionFind::unite(size_t x, size_t y) {  // :25-40
  size_t rx = find(x), ry = find(y);
  if (rx != ry) {
    if (rank[rx] < rank[ry]) {
      parent[rx] = ry;
    } else if (rank[rx] > rank[ry]) {
      parent[ry] = rx;
    } else {
      parent[ry] = rx;
      rank[rx] += 1;
    }
  }
}
std::vector<std::vector<size_t>> UnionFind::get_sets() {  // :52-63
  std::map<size_t, std::vector<size_t>> root_to_group;
  for (size_t i = 0; i < parent.size(); ++i) root_to_group[find(i)].push_back(i);
  std::vector<std::vector<size_t>> out;
  for (auto& kv : root_to_group) out.push_back(std::move(kv.second));
  return out;
}

// ---------------------------------------------------------------------------------------
// plane_sweep_core.rs (off the CLI path; only tests/test_plane_sweep_symmetry.rs calls it)
// ---------------------------------------------------------------------------------------
static bool core_overlaps(const Interval& a, const Interval& b, double thr) {  // :21-33
  uint32_t os = std::max(a.begin, b.begin), oe = std::min(a.end, b.end);
  if (os >= oe) return false;
  uint32_t ol = oe - os;
  uint32_t min_len = std::min(a.end - a.begin, b.end - b.begin);
  return (double)ol / (double)min_len > thr;
}

std::vector<size_t> plane_sweep_core(std::vector<Interval>& iv, uint64_t max_to_keep, double thr) {
  std::vector<size_t> kept;
  if (iv.empty()) return kept;
  if (iv.size() == 1) return {0};
  if (max_to_keep == K_INF) {  // :94-96
    for (size_t i = 0; i < iv.size(); ++i) kept.push_back(i);
    return kept;
  }
  struct Ev {
    uint32_t pos;
    int type;
    size_t idx;
  };
  std::vector<Ev> events;
  for (size_t i = 0; i < iv.size(); ++i) {
    events.push_back({iv[i].begin, 0, i});
    events.push_back({iv[i].end, 1, i});
  }
  // :114 sort_unstable by (position, Begin<End).  Order among equal keys is unspecified in
  // the reference; a stable sort is one admissible instance (parity only claimed on the
  // symmetry-test vectors, which have no same-position Begin ties that matter).
  std::stable_sort(events.begin(), events.end(), [](const Ev& a, const Ev& b) {
    if (a.pos != b.pos) return a.pos < b.pos;
    return a.type < b.type;
  });
  std::set<std::pair<int64_t, size_t>> active;  // (-score_bits, idx)  :117
  for (const Ev& ev : events) {
    uint64_t bits;
    std::memcpy(&bits, &iv[ev.idx].score, 8);
    int64_t sb = (int64_t)bits;
    if (ev.type == 0) {
      active.insert({-sb, ev.idx});
      uint64_t count = 0;  // mark_best :152-164
      for (const auto& a : active) {
        if (count >= max_to_keep) break;
        kept.push_back(a.second);
        ++count;
      }
    } else {
      active.erase({-sb, ev.idx});
    }
  }
  std::sort(kept.begin(), kept.end());
  kept.erase(std::unique(kept.begin(), kept.end()), kept.end());
  if (thr < 1.0 && kept.size() > 1) {  // filter_by_overlap :167-201
    std::stable_sort(kept.begin(), kept.end(), [&](size_t a, size_t b) {
      return iv[b].score < iv[a].score;  // descending score; partial_cmp ties -> Equal (stable)
    });
    std::vector<size_t> final_kept{kept[0]};
    for (size_t k = 1; k < kept.size(); ++k) {
      bool keep = true;
      for (size_t f : final_kept)
        if (core_overlaps(iv[kept[k]], iv[f], thr)) {
          keep = false;
          break;
        }
      if (keep) final_kept.push_back(kept[k]);
    }
    kept = final_kept;
  }
  return kept;
}

// ---------------------------------------------------------------------------------------
// Parsing helpers
// ---------------------------------------------------------------------------------------
// Rust str::parse::<u64>: optional '+', then >=1 ASCII digits, overflow is an error.
static bool rust_parse_u64(const std::string& s, uint64_t* out) {
  size_t i = 0;
  if (s.empty()) return false;
  if (s[0] == '+') i = 1;
  if (i >= s.size()) return false;
  uint64_t v = 0;
  for (; i < s.size(); ++i) {
    char c = s[i];
    if (c < '0' || c > '9') return false;
    uint64_t d = (uint64_t)(c - '0');
    if (v > (UINT64_MAX - d) / 10) return false;
    v = v * 10 + d;
  }
  *out = v;
  return true;
}
// Rust str::parse::<f64>: decimal/exponent forms, "inf"/"infinity"/"nan" (any case), no
// surrounding whitespace, no hex floats.
static bool rust_parse_f64(const std::string& s, double* out) {
  if (s.empty()) return false;
  for (char c : s)
    if (c == 'x' || c == 'X' || c == ' ' || c == '\t' || c == '\n' || c == '(') return false;
  const char* b = s.c_str();
  char* e = nullptr;
  double v = std::strtod(b, &e);
  if (e == b || *e != '\0') return false;
  *out = v;
  return true;
}

// paf.rs:32-64
bool parse_cigar_counts(const std::string& cigar, uint64_t* m, uint64_t* x, uint64_t* ins,
                        uint64_t* del) {
  *m = *x = *ins = *del = 0;
  std::string num;
  for (char ch : cigar) {
    if (ch >= '0' && ch <= '9') {
      num.push_back(ch);
    } else {
      uint64_t count;
      if (!rust_parse_u64(num, &count)) return false;  // includes the empty-number case
      num.clear();
      switch (ch) {
        case '=': *m += count; break;
        case 'X': *x += count; break;
        case 'I': *ins += count; break;
        case 'D': *del += count; break;
        default: break;  // 'M' and everything else contribute nothing
      }
    }
  }
  return true;
}

// cli.rs:26-61
bool parse_metric_number(const std::string& s, uint64_t* out) {
  if (s.empty()) return false;
  std::string num = s;
  char suffix = 0;
  char last = s.back();
  if ((last >= 'a' && last <= 'z') || (last >= 'A' && last <= 'Z')) {
    suffix = last;
    num = s.substr(0, s.size() - 1);
  }
  double base;
  if (!rust_parse_f64(num, &base)) return false;
  double mult = 1.0;
  switch (suffix) {
    case 0: break;
    case 'k': case 'K': mult = 1000.0; break;
    case 'm': case 'M': mult = 1000000.0; break;
    case 'g': case 'G': mult = 1000000000.0; break;
    default: return false;
  }
  double r = base * mult;
  if (r > (double)UINT64_MAX) return false;
  // Rust `as u64` saturates: NaN -> 0, negative -> 0
  if (!(r > 0.0)) {
    *out = 0;
  } else if (r >= 18446744073709551616.0) {
    *out = UINT64_MAX;
  } else {
    *out = (uint64_t)r;
  }
  return true;
}

// cli.rs:76-130
bool parse_identity_value(const std::string& s, double* out, double ani_percentile) {
  std::string lower = s;
  for (auto& c : lower) c = (char)std::tolower((unsigned char)c);
  if (lower.rfind("ani", 0) == 0) {
    if (ani_percentile < 0.0) return false;  // "Cannot use ANI-based threshold without input alignments"
    const std::string rem = lower.substr(3);
    if (rem.empty()) {
      *out = ani_percentile;
      return true;
    }
    // aniN, aniN+X, aniN-X: the percentile number is parsed away and ignored (only the median is honoured)
    char sign = 0;
    std::string off;
    const size_t plus = rem.find('+');
    if (plus != std::string::npos) {
      sign = '+';
      off = rem.substr(plus + 1);
    } else {
      const size_t minus = rem.find('-');
      if (minus != std::string::npos) {
        sign = '-';
        off = rem.substr(minus + 1);
      }
    }
    if (!sign) {
      *out = ani_percentile;
      return true;
    }
    double offset;
    if (!rust_parse_f64(off, &offset)) return false;  // "Invalid ANI offset"
    if (sign == '+') {
      const double v = ani_percentile + offset / 100.0;
      *out = v < 1.0 ? v : 1.0;  // f64::min(1.0) (NaN -> 1.0)
    } else {
      const double v = ani_percentile - offset / 100.0;
      *out = v > 0.0 ? v : 0.0;  // f64::max(0.0)
    }
    return true;
  }
  double v;
  if (!rust_parse_f64(s, &v)) return false;
  *out = v > 1.0 ? v / 100.0 : v;
  return true;
}

// main.rs:3485-3492
int parse_scoring(const std::string& s) {
  if (s == "ani" || s == "identity") return SCORE_IDENTITY;
  if (s == "length") return SCORE_LENGTH;
  if (s == "length-ani" || s == "length-identity") return SCORE_LENGTH_IDENTITY;
  if (s == "matches") return SCORE_MATCHES;
  return SCORE_LOG_LENGTH_IDENTITY;
}

// Rust to_lowercase on the strings we care about: ASCII letters plus the literal "∞".
static std::string ascii_lower(const std::string& s) {
  std::string r = s;
  for (auto& c : r)
    if (c >= 'A' && c <= 'Z') c = (char)(c - 'A' + 'a');
  return r;
}
static bool rust_parse_usize(const std::string& s, uint64_t* out) { return rust_parse_u64(s, out); }

// main.rs:244-293.  Limits: 0 == None.
bool parse_filter_mode(const std::string& mode, int* fmode, uint64_t* per_query,
                       uint64_t* per_target) {
  const std::string INF = "\xE2\x88\x9E";  // "∞"
  std::string lower = ascii_lower(mode);
  auto set = [&](int m, uint64_t q, uint64_t t) {
    *fmode = m;
    *per_query = q;
    *per_target = t;
    return true;
  };
  if (lower == "1:1") return set(ONE_TO_ONE, 1, 1);
  if (lower == "1" || lower == "1:" + INF || lower == "1:infinity" || lower == "1:many")
    return set(ONE_TO_MANY, 1, 0);
  if (lower == INF + ":1" || lower == "infinity:1" || lower == "many:1")
    return set(MANY_TO_MANY, 0, 1);
  if (lower == "many:many" || lower == INF + ":" + INF || lower == "infinity:infinity" ||
      lower == "many" || lower == INF || lower == "infinity" || lower == "-1" || lower == "-1:-1")
    return set(MANY_TO_MANY, 0, 0);
  if (lower.find(':') != std::string::npos) {
    std::vector<std::string> parts;
    size_t start = 0;
    while (true) {
      size_t p = lower.find(':', start);
      if (p == std::string::npos) {
        parts.push_back(lower.substr(start));
        break;
      }
      parts.push_back(lower.substr(start, p - start));
      start = p + 1;
    }
    if (parts.size() == 2) {
      auto side = [&](const std::string& p) -> uint64_t {
        if (p == INF || p == "infinity" || p == "many" || p == "-1") return 0;
        uint64_t v;
        if (!rust_parse_usize(p, &v)) return 0;  // .ok()
        return v > 0 ? v : 0;                    // .filter(|&x| x > 0)
      };
      uint64_t pq = side(parts[0]), pt = side(parts[1]);
      int m = MANY_TO_MANY;
      if (pq == 1 && pt == 1)
        m = ONE_TO_ONE;
      else if (pq == 1 && pt == 0)
        m = ONE_TO_MANY;
      return set(m, pq, pt);
    }
    return set(ONE_TO_ONE, 1, 1);
  }
  uint64_t n;
  if (rust_parse_usize(mode, &n)) {
    if (n == 0) return false;  // reference: std::process::exit(1)
    return set(ONE_TO_MANY, n, 0);
  }
  return set(ONE_TO_ONE, 1, 1);
}

// pansn.rs:176-193
uint64_t round_nice(uint64_t v) {
  if (v == 0) return 0;
  uint64_t step = v <= 500 ? 50 : v <= 1000 ? 100 : v <= 3000 ? 200 : 500;
  return std::max((v + step / 2) / step * step, step);
}
// pansn.rs:207-225
void clamp_scaffold_params(uint64_t user_jump, uint64_t user_mass, bool have_avg, uint64_t avg,
                           bool adaptive, uint64_t* jump, uint64_t* mass) {
  *jump = user_jump;
  *mass = user_mass;
  if (!adaptive || !have_avg || avg == 0) return;
  auto sat_mul = [](uint64_t a, uint64_t b) {
    unsigned __int128 p = (unsigned __int128)a * b;
    return p > UINT64_MAX ? UINT64_MAX : (uint64_t)p;
  };
  *jump = std::min(user_jump, sat_mul(avg, 10));
  *mass = round_nice(std::min(user_mass, sat_mul(avg, 3) / 5));
}

// ---------------------------------------------------------------------------------------
// paf_filter.rs
// ---------------------------------------------------------------------------------------
static std::vector<std::string> split_tabs(const std::string& line) {
  std::vector<std::string> f;
  size_t start = 0;
  while (true) {
    size_t p = line.find('\t', start);
    if (p == std::string::npos) {
      f.push_back(line.substr(start));
      break;
    }
    f.push_back(line.substr(start, p - start));
    start = p + 1;
  }
  return f;
}

// BufRead::lines(): split on '\n', drop the '\n' and a preceding '\r'; a trailing empty
// piece after the last '\n' is not a line.
static bool next_line(std::istream& in, std::string* line) {
  if (!std::getline(in, *line)) return false;
  // the '\r' goes only together with a '\n' (std's Lines::next: `if buf.ends_with('\n') { pop; if buf.ends_with('\r') { pop } }`):
  // a last line that ends in '\r' without a newline keeps it
  if (!in.eof() && !line->empty() && line->back() == '\r') line->pop_back();
  return true;
}

// paf_filter.rs:298-373 (one line)
bool PafFilter::parse_paf_line(const std::string& line, size_t rank, RecordMeta* out) {
  std::vector<std::string> fields = split_tabs(line);
  if (fields.size() < 11) return false;  // :302-304
  auto u64_or = [](const std::string& s, uint64_t dflt) {
    uint64_t v;
    return rust_parse_u64(s, &v) ? v : dflt;
  };
  RecordMeta m;
  m.rank = rank;
  m.query_name = fields[0];
  m.query_start = u64_or(fields[2], 0);
  m.query_end = u64_or(fields[3], 0);
  m.strand = fields[4] == "+" ? '+' : '-';
  m.target_name = fields[5];
  m.target_start = u64_or(fields[7], 0);
  m.target_end = u64_or(fields[8], 0);
  uint64_t matches = u64_or(fields[9], 0);
  uint64_t block_length = u64_or(fields[10], 1);
  uint64_t alignment_length = block_length;
  double identity = (double)matches / (double)std::max<uint64_t>(alignment_length, 1);
  uint64_t exact_matches = matches;
  for (size_t i = 11; i < fields.size(); ++i) {
    const std::string& f = fields[i];
    if (f.rfind("dv:f:", 0) == 0) {
      double div;
      if (rust_parse_f64(f.substr(5), &div)) identity = 1.0 - div;
    } else if (f.rfind("cg:Z:", 0) == 0) {
      uint64_t cm, cx, ci, cd;
      if (parse_cigar_counts(f.substr(5), &cm, &cx, &ci, &cd)) {
        if (cm > 0) {
          exact_matches = cm;
          identity = (double)cm / (double)std::max<uint64_t>(alignment_length, 1);
        }
      }
    }
  }
  m.block_length = block_length;
  m.identity = identity;
  m.matches = exact_matches;
  m.alignment_length = alignment_length;
  m.has_chain_id = false;
  m.chain_status = ST_UNASSIGNED;
  *out = m;
  return true;
}

// paf_filter.rs:292-376 (plain-text input only; bgzf input, paf.rs:10-28, is host I/O and
// not restated here)
std::vector<RecordMeta> PafFilter::extract_metadata(const std::string& path) const {
  std::ifstream in(path, std::ios::binary);
  if (!in) throw std::runtime_error("cannot open " + path);
  std::vector<RecordMeta> md;
  std::string line;
  size_t rank = 0;
  while (next_line(in, &line)) {
    RecordMeta m;
    if (parse_paf_line(line, rank, &m)) md.push_back(std::move(m));
    ++rank;
  }
  return md;
}

// paf_filter.rs:1022-1030 : prefix up to and including the LAST '#', else the whole name
static std::string extract_genome_prefix_last(const std::string& name) {
  size_t p = name.rfind('#');
  return p == std::string::npos ? name : name.substr(0, p + 1);
}

// paf_filter.rs:972-1123
std::vector<RecordMeta> PafFilter::apply_plane_sweep_to_mappings(
    const std::vector<RecordMeta>& mappings) const {
  if (mappings.size() <= 1) return mappings;  // :973-975
  std::vector<PlaneSweepMapping> psm(mappings.size());
  for (size_t i = 0; i < mappings.size(); ++i)
    psm[i] = {i, mappings[i].query_start, mappings[i].query_end, mappings[i].target_start,
              mappings[i].target_end, mappings[i].identity, 0};
  const double thr = config.overlap_threshold;
  uint64_t query_limit, target_limit;  // :1004-1014
  switch (config.mapping_filter_mode) {
    case ONE_TO_ONE:
      query_limit = 1;
      target_limit = 1;
      break;
    case ONE_TO_MANY:
      query_limit = config.mapping_max_per_query ? config.mapping_max_per_query : 1;
      target_limit = config.mapping_max_per_target ? config.mapping_max_per_target : K_INF;
      break;
    default:
      query_limit = config.mapping_max_per_query ? config.mapping_max_per_query : K_INF;
      target_limit = config.mapping_max_per_target ? config.mapping_max_per_target : K_INF;
  }
  using Key = std::pair<std::string, std::string>;
  IndexMap<Key, std::vector<size_t>> genome_pair_groups;  // :1037-1046
  for (size_t i = 0; i < mappings.size(); ++i)
    genome_pair_groups
        .entry(Key(extract_genome_prefix_last(mappings[i].query_name),
                   extract_genome_prefix_last(mappings[i].target_name)))
        .push_back(i);
  std::vector<size_t> all_kept;
  for (auto& gp : genome_pair_groups.items) {
    const std::vector<size_t>& gidx = gp.second;
    // query axis :1055-1076
    std::vector<size_t> query_kept_order;
    std::unordered_set<size_t> query_kept_set;
    IndexMap<std::string, std::vector<size_t>> by_query;
    for (size_t idx : gidx) by_query.entry(mappings[idx].query_name).push_back(idx);
    for (auto& g : by_query.items) {
      std::vector<PlaneSweepMapping> qm;
      for (size_t i : g.second) qm.push_back(psm[i]);
      for (size_t k : plane_sweep_query(qm, query_limit, thr, config.scoring_function))
        if (query_kept_set.insert(g.second[k]).second) query_kept_order.push_back(g.second[k]);
    }
    // target axis :1079-1100
    std::unordered_set<size_t> target_kept_set;
    IndexMap<std::string, std::vector<size_t>> by_target;
    for (size_t idx : gidx) by_target.entry(mappings[idx].target_name).push_back(idx);
    for (auto& g : by_target.items) {
      std::vector<PlaneSweepMapping> tm;
      for (size_t i : g.second) tm.push_back(psm[i]);
      for (size_t k : plane_sweep_target(tm, target_limit, thr, config.scoring_function))
        target_kept_set.insert(g.second[k]);
    }
    // intersection :1105-1111
    std::vector<size_t> intersect;
    for (size_t i : query_kept_order)
      if (target_kept_set.count(i)) intersect.push_back(i);
    std::sort(intersect.begin(), intersect.end());
    all_kept.insert(all_kept.end(), intersect.begin(), intersect.end());
  }
  std::vector<RecordMeta> result;
  result.reserve(all_kept.size());
  for (size_t idx : all_kept) result.push_back(mappings[idx]);
  return result;
}

// paf_filter.rs:750-933
std::vector<MergedChain> PafFilter::merge_mappings_into_chains(const std::vector<RecordMeta>& md,
                                                               uint64_t max_gap) const {
  using Key = std::tuple<std::string, std::string, char>;
  IndexMap<Key, std::vector<std::pair<size_t, size_t>>> groups;  // (rank, idx)  :761-770
  for (size_t idx = 0; idx < md.size(); ++idx)
    groups.entry(Key(md[idx].query_name, md[idx].target_name, md[idx].strand))
        .push_back({md[idx].rank, idx});
  std::vector<MergedChain> all_chains;
  for (auto& g : groups.items) {
    const std::string& query = std::get<0>(g.first);
    const std::string& target = std::get<1>(g.first);
    const char strand = std::get<2>(g.first);
    std::vector<std::pair<size_t, size_t>> sorted = g.second;
    std::stable_sort(sorted.begin(), sorted.end(),  // :777
                     [&](const std::pair<size_t, size_t>& a, const std::pair<size_t, size_t>& b) {
                       return md[a.second].query_start < md[b.second].query_start;
                     });
    const size_t n = sorted.size();
    std::vector<uint64_t> best_pred_score(n, UINT64_MAX);
    std::vector<int64_t> best_pred_idx(n, -1);
    for (size_t i = 0; i < n; ++i) {  // :784-851
      const RecordMeta& mi = md[sorted[i].second];
      uint64_t search_bound = mi.query_end + max_gap;
      int64_t best_j = -1;
      uint64_t best_score = UINT64_MAX;
      for (size_t j = i + 1; j < n; ++j) {
        const RecordMeta& mj = md[sorted[j].second];
        if (mj.query_start > search_bound) break;
        uint64_t q_gap;
        if (mj.query_start >= mi.query_end) {
          q_gap = mj.query_start - mi.query_end;
        } else {
          uint64_t ov = mi.query_end - mj.query_start;
          q_gap = ov <= max_gap / 5 ? ov : max_gap + 1;
        }
        uint64_t r_gap;
        if (strand == '+') {
          if (mj.target_start >= mi.target_end) {
            r_gap = mj.target_start - mi.target_end;
          } else {
            uint64_t ov = mi.target_end - mj.target_start;
            r_gap = ov <= max_gap / 5 ? ov : max_gap + 1;
          }
        } else if (mi.target_start >= mj.target_end) {
          r_gap = mi.target_start - mj.target_end;
        } else {
          uint64_t ov = mj.target_end - mi.target_start;
          r_gap = ov <= max_gap / 5 ? ov : max_gap + 1;
        }
        if (q_gap <= max_gap && r_gap <= max_gap) {
          uint64_t dist_sq = q_gap * q_gap + r_gap * r_gap;  // wrapping, as release Rust
          if (dist_sq < best_score && dist_sq < best_pred_score[j]) {
            best_score = dist_sq;
            best_j = (int64_t)j;
          }
        }
      }
      if (best_j >= 0) {
        best_pred_score[(size_t)best_j] = best_score;
        best_pred_idx[(size_t)best_j] = (int64_t)i;
      }
    }
    UnionFind uf(n);  // :854-859
    for (size_t j = 0; j < n; ++j)
      if (best_pred_idx[j] >= 0) uf.unite((size_t)best_pred_idx[j], j);
    for (auto& set_indices : uf.get_sets()) {  // :862-929
      if (set_indices.empty()) continue;
      uint64_t q_min = UINT64_MAX, q_max = 0, t_min = UINT64_MAX, t_max = 0;
      uint64_t sum_matches = 0, sum_block_lengths = 0;
      std::vector<size_t> member_ranks;
      for (size_t si : set_indices) {
        const RecordMeta& m = md[sorted[si].second];
        q_min = std::min(q_min, m.query_start);
        q_max = std::max(q_max, m.query_end);
        t_min = std::min(t_min, m.target_start);
        t_max = std::max(t_max, m.target_end);
        member_ranks.push_back(sorted[si].first);
        sum_matches += m.matches;
        sum_block_lengths += m.block_length;
      }
      uint64_t total_length = q_max - q_min;
      uint64_t gap_length = total_length > sum_block_lengths ? total_length - sum_block_lengths : 0;
      double log_compressed_gap = gap_length > 0 ? std::max(std::log((double)gap_length), 0.0) : 0.0;
      double effective_length = (double)sum_block_lengths + log_compressed_gap;
      double weighted_identity = effective_length > 0.0 ? (double)sum_matches / effective_length : 0.0;
      all_chains.push_back({query, target, q_min, q_max, t_min, t_max, strand, total_length,
                            weighted_identity, sum_matches, sum_block_lengths, member_ranks});
    }
  }
  return all_chains;
}

// paf_filter.rs:1126-1146
std::vector<MergedChain> PafFilter::apply_scaffold_plane_sweep(std::vector<MergedChain> chains) const {
  if (chains.size() <= 1) return chains;
  std::vector<ChainView> views;
  views.reserve(chains.size());
  for (const auto& c : chains)
    views.push_back({c.query_name, c.target_name, c.query_start, c.query_end, c.target_start,
                     c.target_end, c.weighted_identity});
  std::vector<size_t> kept =
      plane_sweep_scaffolds(views, config.scaffold_filter_mode, config.scaffold_max_per_query,
                            config.scaffold_max_per_target, config.scaffold_overlap_threshold,
                            config.scoring_function);
  std::vector<MergedChain> out;
  for (size_t i : kept) out.push_back(chains[i]);
  return out;
}

// Test-infrastructure switch (oracle_capi: orc_set_fast_inversion): step 4b through a bucket index instead of the reference's
// chains x reverse-mappings loop.  Off by default; every parity test of the suite runs the literal loop.
bool g_fast_inversion = false;

// paf_filter.rs:379-747
std::unordered_map<size_t, RecordMeta> PafFilter::apply_filters(std::vector<RecordMeta> metadata) const {
  std::unordered_map<size_t, RecordMeta> result;
  static const bool timing = getenv("ORC_TIMING") != nullptr;  // phase times on stderr (full-size runs)
  auto tprev = std::chrono::steady_clock::now();
  auto lap = [&](const char* what) {
    if (!timing) return;
    const auto now = std::chrono::steady_clock::now();
    fprintf(stderr, "[oracle] %-28s %8.2f s\n", what, std::chrono::duration<double>(now - tprev).count());
    tprev = now;
  };
  // 1. retain :384-388 (NaN identity fails `>=`)
  {
    std::vector<RecordMeta> kept;
    kept.reserve(metadata.size());
    for (auto& m : metadata)
      if (m.block_length >= config.min_block_length &&
          (config.keep_self || m.query_name != m.target_name) && m.identity >= config.min_identity)
        kept.push_back(std::move(m));
    metadata.swap(kept);
  }
  const std::vector<RecordMeta> all_original = metadata;  // :391
  metadata = apply_plane_sweep_to_mappings(metadata);     // :400
  lap("mapping plane sweep");
  if (config.scaffold_gap == 0) {                         // :409-434
    for (auto& m : metadata) result.emplace(m.rank, m);
    return result;
  }
  std::vector<MergedChain> merged = merge_mappings_into_chains(metadata, config.scaffold_gap);  // :441
  lap("merge into chains");
  std::vector<MergedChain> filtered_chains;  // :449-455
  for (auto& c : merged)
    if (c.total_length >= config.min_scaffold_length &&
        c.weighted_identity >= config.min_scaffold_identity)
      filtered_chains.push_back(c);
  std::unordered_set<size_t> pre_sweep_scaffold_members;  // :471-476
  for (const auto& c : filtered_chains)
    for (size_t r : c.member_indices) pre_sweep_scaffold_members.insert(r);
  filtered_chains = apply_scaffold_plane_sweep(filtered_chains);  // :478
  lap("scaffold plane sweep");

  std::unordered_map<size_t, const RecordMeta*> rank_to_meta;
  for (const auto& m : all_original) rank_to_meta[m.rank] = &m;

  if (config.scaffolds_only) {  // :486-513
    for (size_t ci = 0; ci < filtered_chains.size(); ++ci) {
      std::string chain_id = "chain_" + std::to_string(ci + 1);
      for (size_t r : filtered_chains[ci].member_indices) {
        auto it = rank_to_meta.find(r);
        if (it == rank_to_meta.end()) continue;
        RecordMeta sm = *it->second;
        sm.chain_status = ST_SCAFFOLD;
        sm.has_chain_id = true;
        sm.chain_id = chain_id;
        result[r] = sm;  // HashMap::insert overwrites
      }
    }
    return result;
  }

  // Step 4 :517-528
  std::unordered_set<size_t> anchor_ranks;
  std::unordered_map<size_t, std::string> rank_to_chain_id;
  for (size_t ci = 0; ci < filtered_chains.size(); ++ci) {
    std::string chain_id = "chain_" + std::to_string(ci + 1);
    for (size_t r : filtered_chains[ci].member_indices) {
      anchor_ranks.insert(r);
      rank_to_chain_id[r] = chain_id;
    }
  }
  // Step 4b :535-597
  const uint64_t max_diagonal_distance = config.scaffold_gap;
  std::map<std::pair<std::string, std::string>, std::vector<size_t>> reverse_by_chr_pair;
  for (size_t idx = 0; idx < all_original.size(); ++idx)
    if (all_original[idx].strand == '-')
      reverse_by_chr_pair[{all_original[idx].query_name, all_original[idx].target_name}].push_back(idx);
  if (!g_fast_inversion) {
    for (size_t ci = 0; ci < filtered_chains.size(); ++ci) {
      const MergedChain& chain = filtered_chains[ci];
      if (chain.strand != '+') continue;
      std::string chain_id = "chain_" + std::to_string(ci + 1);
      int64_t diagonal_offset = (int64_t)chain.target_start - (int64_t)chain.query_start;
      auto it = reverse_by_chr_pair.find({chain.query_name, chain.target_name});
      if (it == reverse_by_chr_pair.end()) continue;
      for (size_t idx : it->second) {
        const RecordMeta& mapping = all_original[idx];
        if (anchor_ranks.count(mapping.rank)) continue;
        uint64_t ext_start = chain.query_start > max_diagonal_distance
                                 ? chain.query_start - max_diagonal_distance
                                 : 0;  // saturating_sub
        uint64_t ext_end = chain.query_end > UINT64_MAX - max_diagonal_distance
                               ? UINT64_MAX
                               : chain.query_end + max_diagonal_distance;  // saturating_add
        if (mapping.query_end < ext_start || mapping.query_start > ext_end) continue;
        uint64_t q_center = (mapping.query_start + mapping.query_end) / 2;
        uint64_t t_center = (mapping.target_start + mapping.target_end) / 2;
        int64_t dev = (int64_t)t_center - (int64_t)q_center - diagonal_offset;
        uint64_t deviation = dev < 0 ? (uint64_t)0 - (uint64_t)dev : (uint64_t)dev;  // unsigned_abs
        double pd = (double)deviation / 1.4142135623730951;  // std::f64::consts::SQRT_2
        uint64_t perpendicular = pd >= 18446744073709551616.0 ? UINT64_MAX : (uint64_t)pd;
        if (perpendicular <= max_diagonal_distance) {
          anchor_ranks.insert(mapping.rank);
          rank_to_chain_id[mapping.rank] = chain_id;
        }
      }
    }
  } else {
    // NOT the reference's loop: the same result through an index, for record sets where chains x reverse mappings of a pair
    // (1.4 * 10^6 x 10^6 on BASELINE.json configs[2]) makes the literal loop above take hours.  In that loop a reverse
    // mapping joins the FIRST '+' chain in `filtered_chains` order whose window and diagonal tests it passes (later chains
    // find it in anchor_ranks), and the tests depend on nothing but the chain and the mapping.  So: per chromosome pair the
    // '+' chains are registered in every 65,536-bp bucket their extended query window touches; a mapping looks at the
    // buckets its own query span touches, applies the reference's tests to those chains and keeps the lowest chain index.
    // tests/test_oracle_fast_cpu.py holds this against the literal loop.
    const uint64_t B = 65536;
    struct PairIndex {
      std::unordered_map<uint64_t, std::vector<size_t>> buckets;
      std::vector<size_t> everywhere;  // chains whose window touches more than 2^16 buckets: tested against every mapping
    };
    std::map<std::pair<std::string, std::string>, PairIndex> index;
    for (size_t ci = 0; ci < filtered_chains.size(); ++ci) {
      const MergedChain& chain = filtered_chains[ci];
      if (chain.strand != '+') continue;
      auto it = reverse_by_chr_pair.find({chain.query_name, chain.target_name});
      if (it == reverse_by_chr_pair.end()) continue;
      const uint64_t ext_start = chain.query_start > max_diagonal_distance ? chain.query_start - max_diagonal_distance : 0;
      const uint64_t ext_end = chain.query_end > UINT64_MAX - max_diagonal_distance ? UINT64_MAX
                                                                                    : chain.query_end + max_diagonal_distance;
      PairIndex& pi = index[{chain.query_name, chain.target_name}];
      if (ext_end / B - ext_start / B > 65536) {
        pi.everywhere.push_back(ci);
        continue;
      }
      for (uint64_t b = ext_start / B;; ++b) {
        pi.buckets[b].push_back(ci);
        if (b == ext_end / B) break;
      }
    }
    for (auto& kv : reverse_by_chr_pair) {
      auto pit = index.find(kv.first);
      if (pit == index.end()) continue;
      for (size_t idx : kv.second) {
        const RecordMeta& mapping = all_original[idx];
        if (anchor_ranks.count(mapping.rank)) continue;  // members of kept chains (the only anchors before this step)
        size_t best = SIZE_MAX;
        auto test = [&](size_t ci) {
          if (ci >= best) return;
          const MergedChain& chain = filtered_chains[ci];
          const int64_t diagonal_offset = (int64_t)chain.target_start - (int64_t)chain.query_start;
          const uint64_t ext_start = chain.query_start > max_diagonal_distance ? chain.query_start - max_diagonal_distance : 0;
          const uint64_t ext_end = chain.query_end > UINT64_MAX - max_diagonal_distance ? UINT64_MAX
                                                                                        : chain.query_end + max_diagonal_distance;
          if (mapping.query_end < ext_start || mapping.query_start > ext_end) return;
          const uint64_t q_center = (mapping.query_start + mapping.query_end) / 2;
          const uint64_t t_center = (mapping.target_start + mapping.target_end) / 2;
          const int64_t dev = (int64_t)t_center - (int64_t)q_center - diagonal_offset;
          const uint64_t deviation = dev < 0 ? (uint64_t)0 - (uint64_t)dev : (uint64_t)dev;
          const double pd = (double)deviation / 1.4142135623730951;
          const uint64_t perpendicular = pd >= 18446744073709551616.0 ? UINT64_MAX : (uint64_t)pd;
          if (perpendicular <= max_diagonal_distance) best = ci;
        };
        for (size_t ci : pit->second.everywhere) test(ci);
        // a window and a span that overlap share a bucket; a mapping with end < start (malformed) overlaps nothing the literal
        // tests would accept beyond what its two coordinates' buckets hold... they are visited as a range all the same
        const uint64_t lo_q = std::min(mapping.query_start, mapping.query_end), hi_q = std::max(mapping.query_start, mapping.query_end);
        if (hi_q / B - lo_q / B > 65536) {
          for (auto& bk : pit->second.buckets)
            for (size_t ci : bk.second) test(ci);
        } else {
          for (uint64_t b = lo_q / B;; ++b) {
            auto bit = pit->second.buckets.find(b);
            if (bit != pit->second.buckets.end())
              for (size_t ci : bit->second) test(ci);
            if (b == hi_q / B) break;
          }
        }
        if (best != SIZE_MAX) {
          anchor_ranks.insert(mapping.rank);
          rank_to_chain_id[mapping.rank] = "chain_" + std::to_string(best + 1);
        }
      }
    }
  }
  lap("anchors + inversion capture");
  // :601-604
  std::unordered_set<size_t> filtered_scaffold_members;
  for (size_t r : pre_sweep_scaffold_members)
    if (!anchor_ranks.count(r)) filtered_scaffold_members.insert(r);

  // Step 5 :614-732
  using Key = std::pair<std::string, std::string>;
  IndexMap<Key, std::vector<size_t>> mappings_by_chr_pair;
  for (size_t idx = 0; idx < all_original.size(); ++idx)
    mappings_by_chr_pair.entry(Key(all_original[idx].query_name, all_original[idx].target_name))
        .push_back(idx);
  for (auto& kv : mappings_by_chr_pair.items)
    std::stable_sort(kv.second.begin(), kv.second.end(), [&](size_t a, size_t b) {
      return all_original[a].query_start < all_original[b].query_start;
    });
  // anchors per chr pair.  Reference iterates a HashSet here (non-deterministic order,
  // :637-644); ascending input index is the instance this oracle fixes.
  std::map<Key, std::vector<size_t>> anchors_by_chr_pair;
  for (size_t idx = 0; idx < all_original.size(); ++idx)
    if (anchor_ranks.count(all_original[idx].rank))
      anchors_by_chr_pair[Key(all_original[idx].query_name, all_original[idx].target_name)]
          .push_back(idx);
  const uint64_t max_deviation = config.scaffold_max_deviation;
  struct AnchorBuckets {
    bool built = false;
    uint64_t width = 1;
    std::unordered_map<uint64_t, std::vector<size_t>> map;
  };
  std::map<Key, AnchorBuckets> anchor_buckets;  // indexed evaluation only (g_fast_inversion)
  for (auto& kv : mappings_by_chr_pair.items) {
    auto ait = anchors_by_chr_pair.find(kv.first);
    if (ait == anchors_by_chr_pair.end() || ait->second.empty()) continue;  // :658-660
    const std::vector<size_t>& chr_anchors = ait->second;
    for (size_t midx : kv.second) {
      const RecordMeta& mapping = all_original[midx];
      if (anchor_ranks.count(mapping.rank)) {  // :666-674
        RecordMeta am = mapping;
        auto cit = rank_to_chain_id.find(mapping.rank);
        if (cit != rank_to_chain_id.end()) {
          am.has_chain_id = true;
          am.chain_id = cit->second;
        }
        am.chain_status = ST_SCAFFOLD;
        result[mapping.rank] = am;
      } else if (filtered_scaffold_members.count(mapping.rank)) {  // :675-678
        continue;
      } else if (max_deviation > 0) {  // :679-729
        uint64_t mq = (mapping.query_start + mapping.query_end) / 2;
        uint64_t mt = (mapping.target_start + mapping.target_end) / 2;
        uint64_t min_distance = UINT64_MAX;
        bool have_closest = false;
        size_t closest_anchor_rank = 0;
        auto look_at = [&](size_t aidx) -> bool {  // the body of the reference's loop; true = stop (`break`)
          const RecordMeta& anchor = all_original[aidx];
          uint64_t aq = (anchor.query_start + anchor.query_end) / 2;
          int64_t dq = (int64_t)mq - (int64_t)aq;
          uint64_t q_diff = dq < 0 ? (uint64_t)0 - (uint64_t)dq : (uint64_t)dq;
          if (q_diff > max_deviation) return false;
          uint64_t at = (anchor.target_start + anchor.target_end) / 2;
          int64_t dt = (int64_t)mt - (int64_t)at;
          uint64_t t_diff = dt < 0 ? (uint64_t)0 - (uint64_t)dt : (uint64_t)dt;
          double dd = std::sqrt((double)(q_diff * q_diff + t_diff * t_diff));
          uint64_t distance = dd >= 18446744073709551616.0 ? UINT64_MAX : (uint64_t)dd;
          if (distance < min_distance) {
            min_distance = distance;
            closest_anchor_rank = anchor.rank;
            have_closest = true;
          }
          return min_distance <= max_deviation;
        };
        if (!g_fast_inversion) {
          for (size_t aidx : chr_anchors)
            if (look_at(aidx)) break;
        } else {
          // NOT the reference's loop (same switch and same reason as step 4b above: mappings x anchors of a pair is 10^7 x
          // 10^5 on BASELINE.json configs[2]).  The loop stops at the first anchor, in ascending input index, within the
          // distance; anchors further than max_deviation in the query centre are skipped by the loop itself.  So: the pair's
          // anchors are bucketed by query centre (bucket = max_deviation), the buckets the mapping's centre +- max_deviation
          // touches are collected, and the same loop body runs over those anchors in ascending input index.  When no anchor
          // is within the distance the loop's only other effect (min_distance, closest) is not observable.
          auto& bk = anchor_buckets[kv.first];
          if (!bk.built) {
            bk.built = true;
            bk.width = max_deviation ? max_deviation : 1;
            for (size_t aidx : chr_anchors) {
              const RecordMeta& a = all_original[aidx];
              bk.map[((a.query_start + a.query_end) / 2) / bk.width].push_back(aidx);  // ascending aidx inside a bucket
            }
          }
          std::vector<size_t> cand;
          const uint64_t b_lo = (mq > max_deviation ? mq - max_deviation : 0) / bk.width;
          const uint64_t b_hi = (mq > UINT64_MAX - max_deviation ? UINT64_MAX : mq + max_deviation) / bk.width;
          for (uint64_t b = b_lo;; ++b) {
            auto it = bk.map.find(b);
            if (it != bk.map.end()) cand.insert(cand.end(), it->second.begin(), it->second.end());
            if (b == b_hi) break;
          }
          std::sort(cand.begin(), cand.end());
          for (size_t aidx : cand)
            if (look_at(aidx)) break;
        }
        if (min_distance <= max_deviation) {
          RecordMeta rm = mapping;
          if (have_closest) {
            auto cit = rank_to_chain_id.find(closest_anchor_rank);
            if (cit != rank_to_chain_id.end()) {
              rm.has_chain_id = true;
              rm.chain_id = cit->second;
            }
          }
          rm.chain_status = ST_RESCUED;
          result[mapping.rank] = rm;
        }
      }
    }
  }
  lap("rescue + result");
  return result;
}

// paf_filter.rs:1689-1726
void PafFilter::write_filtered_output(const std::string& in_path, const std::string& out_path,
                                      const std::unordered_map<size_t, RecordMeta>& passing) const {
  std::ifstream in(in_path, std::ios::binary);
  if (!in) throw std::runtime_error("cannot open " + in_path);
  std::ofstream out(out_path, std::ios::binary | std::ios::trunc);
  if (!out) throw std::runtime_error("cannot create " + out_path);
  std::string line;
  size_t rank = 0;
  while (next_line(in, &line)) {
    auto it = passing.find(rank);
    if (it != passing.end()) {
      const RecordMeta& meta = it->second;
      out << line;
      if (meta.has_chain_id) out << "\tch:Z:" << meta.chain_id;
      const char* st = meta.chain_status == ST_SCAFFOLD  ? "scaffold"
                       : meta.chain_status == ST_RESCUED ? "rescued"
                                                         : "unassigned";
      out << "\tst:Z:" << st << "\n";
    }
    ++rank;
  }
}

// paf_filter.rs:278-289
void PafFilter::filter_paf(const std::string& in, const std::string& out) const {
  std::vector<RecordMeta> md = extract_metadata(in);
  auto passing = apply_filters(std::move(md));
  write_filtered_output(in, out, passing);
}

// ---------------------------------------------------------------------------------------
// ANI pre-pass (src/main.rs:296-688)
// ---------------------------------------------------------------------------------------
bool parse_ani_method(const std::string& method_str, AniMethod* out) {  // main.rs:296-330
  const std::string lower = ascii_lower(method_str);
  if (lower == "all") {
    out->kind = ANI_ALL;
    return true;
  }
  if (lower == "orthogonal" || lower == "1:1") {
    out->kind = ANI_ORTHOGONAL;
    return true;
  }
  if (lower.empty() || lower[0] != 'n') return false;
  const std::string rest = lower.substr(1);
  std::vector<std::string> parts;
  for (size_t s0 = 0;;) {
    const size_t d = rest.find('-', s0);
    if (d == std::string::npos) {
      parts.push_back(rest.substr(s0));
      break;
    }
    parts.push_back(rest.substr(s0, d - s0));
    s0 = d + 1;
  }
  double pct;
  if (!rust_parse_f64(parts[0], &pct)) return false;
  if (!(pct > 0.0 && pct <= 100.0)) return false;
  int sort = NSORT_IDENTITY;
  if (parts.size() > 1) {
    if (parts[1] == "length") sort = NSORT_LENGTH;
    else if (parts[1] == "identity") sort = NSORT_IDENTITY;
    else if (parts[1] == "score") sort = NSORT_SCORE;
    else return false;
  }
  out->kind = ANI_NPERCENTILE;
  out->percentile = pct;
  out->sort = sort;
  return true;
}

namespace {

std::string genome_prefix_last(const std::string& name) {  // main.rs:416-425
  const size_t p = name.rfind('#');
  return p == std::string::npos ? name : name.substr(0, p + 1);
}

struct AniLine {
  std::string query_genome, target_genome;
  double matches, block_length, identity;
  uint64_t query_length, target_length;
};

// The per-line part shared by both passes (main.rs:405-446 and 531-586).  False = line skipped.
bool parse_ani_line(const std::string& line, std::vector<std::string>* fields, AniLine* a) {
  if (line.empty() || line[0] == '#') return false;
  fields->clear();
  for (size_t s0 = 0;;) {
    const size_t t = line.find('\t', s0);
    if (t == std::string::npos) {
      fields->push_back(line.substr(s0));
      break;
    }
    fields->push_back(line.substr(s0, t - s0));
    s0 = t + 1;
  }
  const auto& f = *fields;
  if (f.size() < 11) return false;
  a->query_genome = genome_prefix_last(f[0]);
  a->target_genome = genome_prefix_last(f[5]);
  if (a->query_genome == a->target_genome) return false;
  if (!rust_parse_u64(f[1], &a->query_length)) a->query_length = 0;
  if (!rust_parse_u64(f[6], &a->target_length)) a->target_length = 0;
  double matches, block_len;
  if (!rust_parse_f64(f[9], &matches)) matches = 0.0;
  if (!rust_parse_f64(f[10], &block_len)) block_len = 1.0;
  double final_matches = matches;
  for (size_t k = 11; k < f.size(); ++k) {
    if (f[k].compare(0, 5, "dv:f:") == 0) {
      double div;
      if (rust_parse_f64(f[k].substr(5), &div)) {
        final_matches = (1.0 - div) * block_len;
        break;
      }
    }
  }
  a->matches = final_matches;
  a->block_length = block_len;
  a->identity = final_matches / (block_len > 1.0 ? block_len : 1.0);  // block_len.max(1.0): NaN -> 1.0
  return true;
}

using PairSums = std::map<std::pair<std::string, std::string>, std::pair<double, double>>;

void add_pair(PairSums* pairs, const AniLine& a) {
  auto key = a.query_genome < a.target_genome ? std::make_pair(a.query_genome, a.target_genome)
                                              : std::make_pair(a.target_genome, a.query_genome);
  auto& e = (*pairs)[key];
  e.first += a.matches;
  e.second += a.block_length;
}

double median_pair_ani(const PairSums& pairs) {  // main.rs:461-498 / 660-686
  std::vector<double> v;
  for (const auto& kv : pairs) v.push_back(kv.second.second > 0.0 ? kv.second.first / kv.second.second : 0.0);
  for (double x : v)
    if (x != x) throw std::runtime_error("NaN in ANI values (the reference panics in partial_cmp().unwrap())");
  std::sort(v.begin(), v.end());
  const size_t mid = v.size() / 2;
  return (v.size() % 2 == 0 && v.size() > 1) ? (v[mid - 1] + v[mid]) / 2.0 : v[mid];
}

std::vector<std::string> read_lines_lossless(const std::string& path) {
  std::ifstream in(path, std::ios::binary);
  if (!in) throw std::runtime_error("cannot open " + path);
  std::vector<std::string> lines;
  std::string line;
  while (std::getline(in, line)) {
    if (!in.eof() && !line.empty() && line.back() == '\r') line.pop_back();  // only with its '\n', see next_line
    lines.push_back(line);
  }
  return lines;
}

double ani_n_percentile(const std::string& path, double percentile, int sort_method) {  // main.rs:500-688
  std::vector<AniLine> al;
  std::map<std::string, uint64_t> genome_sizes;  // key -> first-seen length (entry().or_insert)
  std::vector<std::string> fields;
  for (const std::string& line : read_lines_lossless(path)) {
    AniLine a;
    if (!parse_ani_line(line, &fields, &a)) continue;
    auto last_part = [](const std::string& n) {
      const size_t p = n.rfind('#');
      return p == std::string::npos ? n : n.substr(p + 1);
    };
    genome_sizes.emplace(a.query_genome + last_part(fields[0]), a.query_length);
    genome_sizes.emplace(a.target_genome + last_part(fields[5]), a.target_length);
    al.push_back(a);
  }
  if (al.empty()) return 0.0;
  auto key_of = [&](const AniLine& a) {
    if (sort_method == NSORT_LENGTH) return a.block_length;
    if (sort_method == NSORT_IDENTITY) return a.identity;
    double l = std::log(a.block_length);
    if (!(l > 1.0)) l = 1.0;  // ln().max(1.0); NaN -> 1.0
    return a.identity * l;
  };
  for (const auto& a : al) {
    const double k = key_of(a);
    if (k != k) throw std::runtime_error("NaN sort key (the reference panics in partial_cmp().unwrap())");
  }
  std::stable_sort(al.begin(), al.end(), [&](const AniLine& x, const AniLine& y) { return key_of(x) > key_of(y); });
  double total_genome_size = 0.0;
  for (const auto& kv : genome_sizes) total_genome_size += (double)kv.second;
  const double n_threshold = total_genome_size * (percentile / 100.0);
  double cumulative = 0.0;
  PairSums pairs;
  for (const auto& a : al) {
    cumulative += a.block_length;
    add_pair(&pairs, a);
    if (cumulative >= n_threshold) break;
  }
  return median_pair_ani(pairs);
}

}  // namespace

double calculate_ani_stats(const std::string& input_path, const AniMethod& method) {  // main.rs:334-498
  std::string final_input = input_path;
  std::string tmp;
  if (method.kind == ANI_ORTHOGONAL) {  // main.rs:343-382: best 1:1 mappings, >= 1 kb, scored by matches
    FilterConfig c;
    c.min_block_length = 1000;
    c.mapping_filter_mode = ONE_TO_ONE;
    c.mapping_max_per_query = 1;
    c.mapping_max_per_target = 1;
    c.scaffold_filter_mode = ONE_TO_ONE;
    c.scaffold_max_per_query = 1;
    c.scaffold_max_per_target = 1;
    c.overlap_threshold = 0.95;
    c.scaffold_gap = 10000;
    c.min_scaffold_length = 0;
    c.scaffold_overlap_threshold = 0.95;
    c.scaffold_max_deviation = 0;
    c.scoring_function = SCORE_MATCHES;
    c.min_identity = 0.0;
    c.min_scaffold_identity = 0.0;
    char name[] = "/tmp/orc_ani_XXXXXX";
    const int fd = mkstemp(name);
    if (fd < 0) throw std::runtime_error("mkstemp failed");
    close(fd);
    tmp = name;
    PafFilter(c).filter_paf(input_path, tmp);
    final_input = tmp;
  } else if (method.kind == ANI_NPERCENTILE) {
    return ani_n_percentile(input_path, method.percentile, method.sort);
  }
  PairSums pairs;
  std::vector<std::string> fields;
  for (const std::string& line : read_lines_lossless(final_input)) {
    AniLine a;
    if (parse_ani_line(line, &fields, &a)) add_pair(&pairs, a);
  }
  if (!tmp.empty()) std::remove(tmp.c_str());
  if (pairs.empty()) return 0.0;
  return median_pair_ani(pairs);
}

// ---- .1aln record derivation (src/unified_filter.rs:83-142) ---------------------------------------------------------
namespace {
// char::is_whitespace = Unicode White_Space: U+0009-000D, U+0020, U+0085, U+00A0, U+1680, U+2000-200A, U+2028, U+2029,
// U+202F, U+205F, U+3000.  Returns the byte length of the white-space character at s[i], or 0.
size_t ws_len(const std::string& s, size_t i) {
  const unsigned char c = (unsigned char)s[i];
  if ((c >= 0x09 && c <= 0x0d) || c == 0x20) return 1;
  if (c == 0xc2 && i + 1 < s.size()) {
    const unsigned char d = (unsigned char)s[i + 1];
    return (d == 0x85 || d == 0xa0) ? 2 : 0;
  }
  if (i + 2 < s.size()) {
    const unsigned char d = (unsigned char)s[i + 1], e = (unsigned char)s[i + 2];
    if (c == 0xe1 && d == 0x9a && e == 0x80) return 3;  // U+1680
    if (c == 0xe2 && d == 0x80 && ((e >= 0x80 && e <= 0x8a) || e == 0xa8 || e == 0xa9 || e == 0xaf)) return 3;
    if (c == 0xe2 && d == 0x81 && e == 0x9f) return 3;  // U+205F
    if (c == 0xe3 && d == 0x80 && e == 0x80) return 3;  // U+3000
  }
  return 0;
}
}  // namespace

std::string first_word_or_all(const std::string& s) {
  size_t i = 0;
  while (i < s.size()) {  // split_whitespace skips leading white space
    const size_t w = ws_len(s, i);
    if (!w) break;
    i += w;
  }
  if (i >= s.size()) return s;  // no word at all: unwrap_or(&full)
  size_t j = i;
  while (j < s.size() && !ws_len(s, j)) ++j;
  return s.substr(i, j - i);
}

std::vector<RecordMeta> records_from_1aln(const std::vector<AlnRecord>& alns) {
  std::vector<RecordMeta> out;
  out.reserve(alns.size());
  size_t rank = 0;
  for (const AlnRecord& a : alns) {
    RecordMeta m;
    m.rank = rank++;
    m.query_name = first_word_or_all(a.query_name);    // :83-87
    m.target_name = first_word_or_all(a.target_name);  // :88-92
    const uint64_t query_span = a.query_end - a.query_start;     // :107 (wrapping in release Rust)
    const uint64_t target_span = a.target_end - a.target_start;  // :108
    m.block_length = query_span + target_span;                   // :112
    m.matches = a.matches;                                       // :115
    m.identity = query_span > 0 ? (double)m.matches / (double)query_span : 0.0;  // :119-123
    m.query_start = a.query_start;
    m.query_end = a.query_end;
    m.target_start = a.target_start;
    m.target_end = a.target_end;
    m.alignment_length = m.block_length;
    m.strand = a.strand;
    out.push_back(m);
  }
  return out;
}

// ---- tree sparsification (src/tree_filter.rs) -------------------------------------------------------------------------
namespace {
inline uint64_t rotl64(uint64_t x, int b) { return (x << b) | (x >> (64 - b)); }
struct Sip13 {  // core::hash::sip::SipHasher13 with k0 = k1 = 0 (DefaultHasher::new())
  uint64_t v0 = 0x736f6d6570736575ull, v1 = 0x646f72616e646f6dull, v2 = 0x6c7967656e657261ull, v3 = 0x7465646279746573ull;
  uint64_t tail = 0;
  int ntail = 0;
  uint64_t length = 0;
  void round() {
    v0 += v1; v1 = rotl64(v1, 13); v1 ^= v0; v0 = rotl64(v0, 32);
    v2 += v3; v3 = rotl64(v3, 16); v3 ^= v2;
    v0 += v3; v3 = rotl64(v3, 21); v3 ^= v0;
    v2 += v1; v1 = rotl64(v1, 17); v1 ^= v2; v2 = rotl64(v2, 32);
  }
  void write(const unsigned char* p, size_t n) {
    for (size_t i = 0; i < n; ++i) {
      tail |= (uint64_t)p[i] << (8 * ntail);
      ++length;
      if (++ntail == 8) {
        v3 ^= tail;
        round();  // c_rounds = 1
        v0 ^= tail;
        tail = 0;
        ntail = 0;
      }
    }
  }
  uint64_t finish() {
    const uint64_t b = ((length & 0xff) << 56) | tail;
    v3 ^= b;
    round();
    v0 ^= b;
    v2 ^= 0xff;
    round(); round(); round();  // d_rounds = 3
    return v0 ^ v1 ^ v2 ^ v3;
  }
};
std::string tree_genome_prefix(const std::string& name) {  // src/tree_filter.rs:15-24
  const size_t p1 = name.find('#');
  if (p1 == std::string::npos) return name;
  const size_t p2 = name.find('#', p1 + 1);
  return name.substr(0, p1) + "#" + (p2 == std::string::npos ? name.substr(p1 + 1) : name.substr(p1 + 1, p2 - p1 - 1)) + "#";
}
bool parse_u64_rust(const std::string& s, uint64_t* out) {  // str::parse::<u64>: optional '+', digits, no overflow
  size_t i = 0;
  if (i < s.size() && s[i] == '+') ++i;
  if (i >= s.size()) return false;
  uint64_t v = 0;
  for (; i < s.size(); ++i) {
    if (s[i] < '0' || s[i] > '9') return false;
    const uint64_t d = (uint64_t)(s[i] - '0');
    if (v > (UINT64_MAX - d) / 10) return false;
    v = v * 10 + d;
  }
  *out = v;
  return true;
}
}  // namespace

uint64_t default_hash_str_pair(const std::string& a, const std::string& b) {
  Sip13 h;  // impl Hash for str: write(bytes) then write_u8(0xff)
  const unsigned char ff = 0xff;
  h.write(reinterpret_cast<const unsigned char*>(a.data()), a.size());
  h.write(&ff, 1);
  h.write(reinterpret_cast<const unsigned char*>(b.data()), b.size());
  h.write(&ff, 1);
  return h.finish();
}

std::vector<std::string> tree_filter_paf_lines(const std::vector<std::string>& lines, size_t k_nearest, size_t k_farthest,
                                               double random_fraction) {
  struct Aln {
    std::string qg, tg;
    uint64_t matches, block;
    size_t line;
  };
  std::vector<Aln> alns;
  for (size_t li = 0; li < lines.size(); ++li) {  // src/tree_filter.rs:222-252
    const std::string& line = lines[li];
    if (line.empty() || line[0] == '#') continue;
    std::vector<std::string> f;
    size_t s0 = 0;
    while (true) {
      const size_t t = line.find('\t', s0);
      f.push_back(line.substr(s0, t == std::string::npos ? std::string::npos : t - s0));
      if (t == std::string::npos) break;
      s0 = t + 1;
    }
    if (f.size() < 11) continue;
    Aln a;
    a.qg = tree_genome_prefix(f[0]);
    a.tg = tree_genome_prefix(f[5]);
    if (!parse_u64_rust(f[9], &a.matches)) a.matches = 0;
    if (!parse_u64_rust(f[10], &a.block)) a.block = 1;
    a.line = li;
    alns.push_back(a);
  }
  // build_identity_matrix (:39-75): canonical (sorted) genome pair -> (sum matches, sum block length) as f64
  std::map<std::pair<std::string, std::string>, std::pair<double, double>> sums;
  for (const Aln& a : alns) {
    if (a.qg == a.tg) continue;
    auto key = a.qg < a.tg ? std::make_pair(a.qg, a.tg) : std::make_pair(a.tg, a.qg);
    auto& e = sums[key];
    e.first += (double)a.matches;
    e.second += (double)a.block;
  }
  std::map<std::pair<std::string, std::string>, double> identity;
  for (const auto& kv : sums) identity[kv.first] = kv.second.second > 0.0 ? kv.second.first / kv.second.second : 0.0;
  // select_tree_pairs (:79-160)
  std::set<std::string> genomes;
  for (const auto& kv : identity) {
    genomes.insert(kv.first.first);
    genomes.insert(kv.first.second);
  }
  std::set<std::pair<std::string, std::string>> selected;
  for (const std::string& g : genomes) {
    std::vector<std::pair<std::string, double>> nb;  // std::map order = neighbour prefix ascending (the tie order chosen here)
    for (const auto& kv : identity) {
      if (kv.first.first == g)
        nb.emplace_back(kv.first.second, kv.second);
      else if (kv.first.second == g)
        nb.emplace_back(kv.first.first, kv.second);
    }
    std::stable_sort(nb.begin(), nb.end(), [](const auto& x, const auto& y) { return x.second > y.second; });  // :110
    auto add = [&](const std::string& other) {
      selected.insert(g < other ? std::make_pair(g, other) : std::make_pair(other, g));
    };
    for (size_t k = 0; k < nb.size() && k < k_nearest; ++k) add(nb[k].first);  // :113-120
    if (k_farthest > 0) {                                                       // :123-133
      std::reverse(nb.begin(), nb.end());
      for (size_t k = 0; k < nb.size() && k < k_farthest; ++k) add(nb[k].first);
    }
  }
  if (random_fraction > 0.0) {  // :137-156
    const double scaled = random_fraction * 18446744073709551615.0;  // u64::MAX as f64 = 2^64
    const uint64_t threshold = scaled >= 18446744073709551615.0 ? UINT64_MAX : (scaled <= 0.0 ? 0 : (uint64_t)scaled);  // `as u64` saturates
    for (const auto& kv : identity)
      if (default_hash_str_pair(kv.first.first, kv.first.second) <= threshold) selected.insert(kv.first);
  }
  // filter_tree_based (:172-200) + the writer (:277-282)
  std::vector<std::string> out;
  for (const Aln& a : alns) {
    if (a.qg == a.tg) continue;
    auto key = a.qg < a.tg ? std::make_pair(a.qg, a.tg) : std::make_pair(a.tg, a.qg);
    if (selected.count(key)) out.push_back(lines[a.line]);
  }
  return out;
}

}  // namespace orc
