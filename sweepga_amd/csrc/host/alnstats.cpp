// alnstats (src/bin/alnstats.rs): per-file and per-genome-pair statistics of a PAF -- mappings, bases, identity, coverage
// of every query genome by every target genome -- and the comparison of two files.  Host code: the text is cut into
// line-aligned slices, every thread folds its slice into small hash tables keyed by views into the text, and the slices
// are merged in file order (sizes: last writer wins; genome pairs: first appearance kept, which is the order this
// implementation gives the reference's HashMap-ordered per-pair list, :51-57).
#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <string_view>
#include <thread>
#include <unordered_map>
#include <unordered_set>
#include <vector>

#include "../../../include/sweepga_gpu.h"
#include "host_internal.h"

namespace {

using sv = std::string_view;

struct PairKey {
  sv a, b;
  bool operator==(const PairKey& o) const { return a == o.a && b == o.b; }
};
struct PairHash {
  size_t operator()(const PairKey& k) const {
    const size_t h1 = std::hash<sv>{}(k.a), h2 = std::hash<sv>{}(k.b);
    return h1 ^ (h2 + 0x9e3779b97f4a7c15ull + (h1 << 6) + (h1 >> 2));
  }
};
struct SeqSize {
  uint64_t line, size;
};
struct PairSum {
  uint64_t first_line, bases, matches;
};

sv genome_prefix(sv s) {  // :94-100: up to and including the last '#'
  const size_t p = s.rfind('#');
  return p == sv::npos ? s : s.substr(0, p + 1);
}

bool parse_u64(sv s, uint64_t* out) {  // str::parse::<u64>: optional '+', digits, no overflow
  size_t i = 0;
  if (s.empty()) return false;
  if (s[0] == '+') i = 1;
  if (i >= s.size()) return false;
  uint64_t v = 0;
  for (; i < s.size(); ++i) {
    const unsigned d = (unsigned)(s[i] - '0');
    if (d > 9) return false;
    if (v > (UINT64_MAX - d) / 10) return false;
    v = v * 10 + d;
  }
  *out = v;
  return true;
}

const char* const FIELD_ERR[6] = {"Invalid query length", "Invalid query start", "Invalid query end",
                                  "Invalid target length", "Invalid match count", "Invalid block length"};

struct Part {
  uint64_t total_mappings = 0, total_bases = 0, total_matches = 0, self_mappings = 0, inter_chromosomal = 0, inter_genome = 0;
  std::unordered_map<sv, SeqSize> sizes;
  std::unordered_map<PairKey, PairSum, PairHash> pairs;
  std::unordered_set<PairKey, PairHash> chr_pairs;
  uint64_t err_line = UINT64_MAX;
  int err_field = -1;
  uint64_t lines = 0;
};

void fold_slice(const char* text, size_t begin, size_t end, bool last_slice, Part* P) {
  uint64_t line = 0;  // relative; made absolute by the merge through P->lines of the slices before
  for (size_t pos = begin; pos < end; ++line) {
    const void* nl = std::memchr(text + pos, '\n', end - pos);
    const size_t e = nl ? (size_t)(static_cast<const char*>(nl) - text) : end;
    size_t ll = e - pos;
    if ((nl || !last_slice) && ll && text[pos + ll - 1] == '\r') --ll;  // BufRead::lines: '\r' only with its '\n'
    const char* b = text + pos;
    pos = e + 1;
    sv f[11];
    int k = 0;
    const char* fb = b;
    const char* le = b + ll;
    for (const char* q = b; k < 11; ++q) {
      if (q == le || *q == '\t') {
        f[k++] = sv(fb, (size_t)(q - fb));
        fb = q + 1;
        if (q == le) break;
      }
    }
    if (k < 11) continue;  // :113-115
    uint64_t v[6];
    static const int IDX[6] = {1, 2, 3, 6, 9, 10};
    int bad = -1;
    for (int j = 0; j < 6 && bad < 0; ++j)
      if (!parse_u64(f[IDX[j]], &v[j])) bad = j;
    if (bad >= 0) {  // the reference stops at the first such line of the file
      P->err_line = line;
      P->err_field = bad;
      P->lines = line + 1;
      return;
    }
    const sv query = f[0], target = f[5];
    P->total_mappings += 1;
    const uint64_t mapping_len = v[2] - v[1];  // wrapping, as the release build
    P->total_bases += mapping_len;
    P->total_matches += v[4];
    P->sizes[query] = SeqSize{2 * line, v[0]};       // :131-132: the target's insert comes second
    P->sizes[target] = SeqSize{2 * line + 1, v[3]};
    const sv qg = genome_prefix(query), tg = genome_prefix(target);
    if (query == target) {
      P->self_mappings += 1;
    } else if (qg != tg) {
      P->inter_genome += 1;
      auto it = P->pairs.try_emplace(PairKey{qg, tg}, PairSum{line, 0, 0}).first;
      it->second.bases += mapping_len;
      it->second.matches += v[4];
    } else {
      P->inter_chromosomal += 1;
    }
    P->chr_pairs.insert(PairKey{query, target});
  }
  P->lines = line;
}

}  // namespace

struct swg_alnstats {
  swg_alnstats_summary sum{};
  std::vector<std::string> pair_q, pair_t;  // first-appearance order
  std::vector<uint64_t> pair_bases, pair_matches;
  std::vector<double> pair_cov;
};

namespace {

thread_local std::string g_err;
int stats_error(int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  std::vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  g_err = buf;
  return code;
}

int compute(const char* text, size_t len, int threads, swg_alnstats** out) {
  if (threads <= 0) {
    const unsigned hc = std::thread::hardware_concurrency();
    threads = hc ? (int)(hc > 64 ? 64 : hc) : 1;
  }
  if ((size_t)threads > len / 65536 + 1) threads = (int)(len / 65536 + 1);
  std::vector<size_t> cut(threads + 1, len);
  cut[0] = 0;
  for (int t = 1; t < threads; ++t) {
    size_t b = len / threads * t;
    if (b < cut[t - 1]) b = cut[t - 1];
    const void* nl = b < len ? std::memchr(text + b, '\n', len - b) : nullptr;
    cut[t] = nl ? (size_t)(static_cast<const char*>(nl) - text) + 1 : len;
  }
  std::vector<Part> parts(threads);
  {
    std::vector<std::thread> pool;
    for (int t = 1; t < threads; ++t)
      pool.emplace_back([&, t] { fold_slice(text, cut[t], cut[t + 1], cut[t + 1] == len, &parts[t]); });
    fold_slice(text, cut[0], cut[1], cut[1] == len, &parts[0]);
    for (auto& th : pool) th.join();
  }
  // merge in file order
  uint64_t base = 0;
  for (auto& P : parts) {
    if (P.err_field >= 0) return stats_error(SWG_ERR_INVALID, "%s (line %llu)", FIELD_ERR[P.err_field], (unsigned long long)(base + P.err_line + 1));
    base += P.lines;
  }
  auto* S = new swg_alnstats;
  swg_alnstats_summary& s = S->sum;
  std::unordered_map<sv, SeqSize> sizes;
  std::unordered_map<PairKey, PairSum, PairHash> pairs;
  std::unordered_set<PairKey, PairHash> chr_pairs;
  base = 0;
  for (auto& P : parts) {
    s.total_mappings += P.total_mappings;
    s.total_bases += P.total_bases;
    s.total_matches += P.total_matches;
    s.self_mappings += P.self_mappings;
    s.inter_chromosomal += P.inter_chromosomal;
    s.inter_genome += P.inter_genome;
    for (const auto& kv : P.sizes) sizes[kv.first] = kv.second;  // later slices overwrite: last writer wins
    for (const auto& kv : P.pairs) {
      auto it = pairs.try_emplace(kv.first, PairSum{base + kv.second.first_line, 0, 0}).first;  // earlier slices came first
      it->second.bases += kv.second.bases;
      it->second.matches += kv.second.matches;
    }
    chr_pairs.insert(P.chr_pairs.begin(), P.chr_pairs.end());
    base += P.lines;
  }
  s.chr_pair_count = chr_pairs.size();
  // calculate_coverage_stats, :42-73
  std::unordered_map<sv, uint64_t> genome_totals;
  for (const auto& kv : sizes) genome_totals[genome_prefix(kv.first)] += kv.second.size;
  std::vector<std::pair<uint64_t, PairKey>> order;
  order.reserve(pairs.size());
  for (const auto& kv : pairs) order.emplace_back(kv.second.first_line, kv.first);
  std::sort(order.begin(), order.end(), [](const auto& a, const auto& b) {
    return a.first != b.first ? a.first < b.first : false;  // one pair per line: first lines are distinct
  });
  double sum = 0.0;
  for (const auto& o : order) {
    const auto gt = genome_totals.find(o.second.a);
    if (gt == genome_totals.end()) continue;  // cannot happen: every query sequence has a size
    const PairSum& ps = pairs.at(o.second);
    const double cov = 100.0 * (double)ps.bases / (double)gt->second;
    S->pair_q.emplace_back(o.second.a);
    S->pair_t.emplace_back(o.second.b);
    S->pair_bases.push_back(ps.bases);
    S->pair_matches.push_back(ps.matches);
    S->pair_cov.push_back(cov);
    sum += cov;
    s.above_95_pct += cov > 95.0;
  }
  s.genome_pairs = S->pair_cov.size();
  s.avg_coverage = s.genome_pairs ? sum / (double)s.genome_pairs : 0.0;
  s.avg_identity = s.total_bases > 0 ? (double)s.total_matches / (double)s.total_bases : 0.0;  // :75-81
  *out = S;
  return SWG_OK;
}

// ---- the reference's text output -----------------------------------------------------------------------
std::string format_number(uint64_t n) {  // :305-315
  char d[32];
  const int k = std::snprintf(d, sizeof d, "%llu", (unsigned long long)n);
  std::string r;
  for (int i = 0; i < k; ++i) {
    if (i && (k - i) % 3 == 0) r += ',';
    r += d[i];
  }
  return r;
}
std::string format_signed(int64_t n) { return (n >= 0 ? "+" : "-") + format_number(n >= 0 ? (uint64_t)n : (uint64_t)(-n)); }  // :317-323
std::string fixed1(double v, bool plus) {  // {:.1} / {:+.1}
  if (std::isnan(v)) return "NaN";
  if (std::isinf(v)) return v < 0 ? "-inf" : (plus ? "+inf" : "inf");
  char buf[400];
  std::snprintf(buf, sizeof buf, plus ? "%+.1f" : "%.1f", v);
  return buf;
}
size_t n_chars(const std::string& s) {
  size_t n = 0;
  for (unsigned char c : s) n += (c & 0xc0) != 0x80;
  return n;
}
void right(std::string* o, const std::string& s, size_t w) {
  const size_t c = n_chars(s);
  if (c < w) o->append(w - c, ' ');
  *o += s;
}
void left(std::string* o, const std::string& s, size_t w) {
  const size_t c = n_chars(s);
  *o += s;
  if (c < w) o->append(w - c, ' ');
}
void row(std::string* o, const char* label, const std::string& value, size_t w) {
  *o += label;
  right(o, value, w);
  *o += '\n';
}
std::string ratio(uint64_t a, uint64_t b) { return std::to_string(a) + "/" + std::to_string(b); }

int hand_over(const std::string& s, char** out_text, uint64_t* out_len) {
  char* p = static_cast<char*>(std::malloc(s.size() + 1));
  if (!p) return stats_error(SWG_ERR_OOM, "out of host memory");
  std::memcpy(p, s.data(), s.size());
  p[s.size()] = 0;
  *out_text = p;
  *out_len = s.size();
  return SWG_OK;
}

void comparison(std::string* o, const char* label, uint64_t v1, uint64_t v2) {  // :286-303
  *o += "\n";
  *o += label;
  *o += ":\n  ";
  left(o, "Before", 30);
  *o += ' ';
  right(o, format_number(v1), 12);
  *o += "\n  ";
  left(o, "After", 30);
  *o += ' ';
  right(o, format_number(v2), 12);
  const int64_t diff = (int64_t)v2 - (int64_t)v1;
  const double pct = v1 > 0 ? 100.0 * (double)diff / (double)v1 : 0.0;
  *o += "\n  ";
  left(o, "Change", 30);
  *o += ' ';
  right(o, format_signed(diff), 12);
  *o += " (" + fixed1(pct, true) + "%)\n";
}
void two_rows(std::string* o, const char* title, const std::string& f1, const std::string& f2, const std::string& v1,
              const std::string& v2, const std::string* change) {
  *o += "\n";
  *o += title;
  *o += "\n  ";
  left(o, f1, 30);
  *o += ' ';
  right(o, v1, change ? 11 : 12);
  *o += change ? "%\n  " : "\n  ";
  left(o, f2, 30);
  *o += ' ';
  right(o, v2, change ? 11 : 12);
  *o += change ? "%\n" : "\n";
  if (change) {
    *o += "  ";
    left(o, "Change", 30);
    *o += ' ';
    right(o, *change, 10);
    *o += "%\n";
  }
}

}  // namespace

extern "C" {

int swg_alnstats_open_buffer(const char* text, uint64_t len, int threads, swg_alnstats** out) {
  if (out) *out = nullptr;
  if (!out || (!text && len)) return stats_error(SWG_ERR_INVALID, "swg_alnstats_open_buffer: NULL argument");
  try {
    return compute(text ? text : "", (size_t)len, threads, out);
  } catch (const std::bad_alloc&) {
    return stats_error(SWG_ERR_OOM, "out of host memory");
  }
}

int swg_alnstats_open(const char* path, int threads, swg_alnstats** out) {
  if (out) *out = nullptr;
  if (!path || !out) return stats_error(SWG_ERR_INVALID, "swg_alnstats_open: NULL argument");
  const char* text = nullptr;
  size_t len = 0;
  void* h = nullptr;
  const int rc = swg_host_text_load(path, threads, &text, &len, &h);
  if (rc != SWG_OK) return stats_error(rc, "Failed to open %s: %s", path, swg_paf_last_error());
  int r;
  try {
    r = compute(text, len, threads, out);
  } catch (const std::bad_alloc&) {
    r = stats_error(SWG_ERR_OOM, "out of host memory");
  }
  swg_host_text_release(h);
  return r;
}

void swg_alnstats_close(swg_alnstats* s) { delete s; }
const char* swg_alnstats_last_error(void) { return g_err.c_str(); }
const swg_alnstats_summary* swg_alnstats_get(const swg_alnstats* s) { return s ? &s->sum : nullptr; }

int swg_alnstats_pair(const swg_alnstats* s, uint64_t i, const char** q_genome, const char** t_genome, double* coverage,
                      uint64_t* bases, uint64_t* matches) {
  if (!s || i >= s->pair_cov.size()) return SWG_ERR_INVALID;
  if (q_genome) *q_genome = s->pair_q[i].c_str();
  if (t_genome) *t_genome = s->pair_t[i].c_str();
  if (coverage) *coverage = s->pair_cov[i];
  if (bases) *bases = s->pair_bases[i];
  if (matches) *matches = s->pair_matches[i];
  return SWG_OK;
}

int swg_alnstats_report(const swg_alnstats* s, const char* label, int detailed, char** out_text, uint64_t* out_len) {  // :166-228
  if (!s || !label || !out_text || !out_len) return stats_error(SWG_ERR_INVALID, "swg_alnstats_report: NULL argument");
  const swg_alnstats_summary& m = s->sum;
  std::string o = "\nStatistics for ";
  o += label;
  o += ":\n" + std::string(60, '=') + "\n";
  row(&o, "Total mappings:        ", format_number(m.total_mappings), 12);
  row(&o, "Total bases:           ", format_number(m.total_bases), 12);
  row(&o, "Average identity:      ", fixed1(m.avg_identity * 100.0, false), 11);
  o.insert(o.size() - 1, "%");
  row(&o, "Self mappings:         ", format_number(m.self_mappings), 12);
  row(&o, "Inter-chromosomal:     ", format_number(m.inter_chromosomal), 12);
  row(&o, "Inter-genome:          ", format_number(m.inter_genome), 12);
  row(&o, "Chromosome pairs:      ", format_number(m.chr_pair_count), 12);
  row(&o, "Genome pairs:          ", std::to_string(m.genome_pairs), 12);
  row(&o, "Average coverage:      ", fixed1(m.avg_coverage, false), 11);
  o.insert(o.size() - 1, "%");
  row(&o, "Pairs >95% coverage:   ", ratio(m.above_95_pct, m.genome_pairs), 12);
  if (detailed && !s->pair_cov.empty()) {
    const size_t np = s->pair_cov.size();
    if (np > 1)
      for (double c : s->pair_cov)
        if (std::isnan(c))  // the reference panics here (partial_cmp().unwrap() on NaN, :204)
          return stats_error(SWG_ERR_INVALID, "a genome pair has NaN coverage (0 bases over a genome of size 0): the reference panics");
    o += "\nPer-genome-pair statistics:\n" + std::string(60, '-') + "\n";
    std::vector<size_t> idx(np);
    for (size_t i = 0; i < np; ++i) idx[i] = i;
    std::stable_sort(idx.begin(), idx.end(), [&](size_t a, size_t b) { return s->pair_cov[a] > s->pair_cov[b]; });  // :204
    for (size_t i : idx) {
      auto trimmed = [](std::string g) {
        while (!g.empty() && g.back() == '#') g.pop_back();
        return g;
      };
      const double identity = s->pair_bases[i] > 0 ? (double)s->pair_matches[i] / (double)s->pair_bases[i] * 100.0 : 0.0;
      left(&o, trimmed(s->pair_q[i]), 20);
      o += " -> ";
      left(&o, trimmed(s->pair_t[i]), 20);
      o += ' ';
      right(&o, fixed1(s->pair_cov[i], false), 6);
      o += "% cov, ";
      right(&o, fixed1(identity, false), 6);
      o += "% id, ";
      right(&o, format_number(s->pair_bases[i]), 10);
      o += " bp\n";
    }
  }
  return hand_over(o, out_text, out_len);
}

int swg_alnstats_compare(const swg_alnstats* a, const swg_alnstats* b, const char* file1, const char* file2, char** out_text,
                         uint64_t* out_len) {  // :230-284
  if (!a || !b || !file1 || !file2 || !out_text || !out_len) return stats_error(SWG_ERR_INVALID, "swg_alnstats_compare: NULL argument");
  const swg_alnstats_summary &x = a->sum, &y = b->sum;
  std::string o = "\nComparison: ";
  o += file1;
  o += " vs ";
  o += file2;
  o += "\n" + std::string(60, '=') + "\n";
  comparison(&o, "Mappings", x.total_mappings, y.total_mappings);
  comparison(&o, "Total bases", x.total_bases, y.total_bases);
  std::string ch = fixed1((y.avg_identity - x.avg_identity) * 100.0, true);
  two_rows(&o, "Average identity:", file1, file2, fixed1(x.avg_identity * 100.0, false), fixed1(y.avg_identity * 100.0, false), &ch);
  comparison(&o, "Inter-chromosomal", x.inter_chromosomal, y.inter_chromosomal);
  comparison(&o, "Chromosome pairs", x.chr_pair_count, y.chr_pair_count);
  ch = fixed1(y.avg_coverage - x.avg_coverage, true);
  two_rows(&o, "Average genome pair coverage:", file1, file2, fixed1(x.avg_coverage, false), fixed1(y.avg_coverage, false), &ch);
  two_rows(&o, "Genome pairs with >95% coverage:", file1, file2, ratio(x.above_95_pct, x.genome_pairs),
           ratio(y.above_95_pct, y.genome_pairs), nullptr);
  return hand_over(o, out_text, out_len);
}

}  // extern "C"
