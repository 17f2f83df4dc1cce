// TEST INFRASTRUCTURE ONLY -- CPU restatement of the reference's `alnstats` binary (src/bin/alnstats.rs), line by line,
// single-threaded.  The checker of sweepga_amd/bin/alnstats; never linked into or called by the product.
//
//   alnstats-ref <file1> [file2] [-d|--detailed]
//
// Parity: UNPINNED -- the reference holds no test or golden output for alnstats; tests/test_alnstats_cpu.py pins this
// restatement with hand-computed vectors.  Where the reference iterates a HashMap (the order of the per-pair list:
// src/bin/alnstats.rs:51-57, which decides the summation order of the average and the order of equal coverages in the
// detailed table) this restatement fixes the instance "first appearance of the genome pair in the file".
#include <cinttypes>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <map>
#include <set>
#include <string>
#include <unordered_map>
#include <vector>

namespace {

struct Stats {  // AlignmentStats, :28-39
  uint64_t total_mappings = 0, total_bases = 0, total_matches = 0, self_mappings = 0, inter_chromosomal = 0, inter_genome = 0,
           chr_pair_count = 0;
  std::vector<std::pair<std::string, std::string>> pair_order;  // first appearance
  std::map<std::pair<std::string, std::string>, uint64_t> pair_bases, pair_matches;
  std::vector<std::string> seq_order;
  std::unordered_map<std::string, uint64_t> genome_sizes;
};

struct Coverage {  // CoverageStats, :84-90
  double avg_coverage = 0.0;
  size_t genome_pairs = 0, above_95 = 0;
  std::vector<std::tuple<std::string, std::string, double, uint64_t>> per_pair;
};

std::string genome_prefix(const std::string& s) {  // :94-100
  const size_t p = s.rfind('#');
  return p == std::string::npos ? s : s.substr(0, p + 1);
}

bool rust_u64(const std::string& s, uint64_t* out) {  // str::parse::<u64>
  size_t i = 0;
  if (s.empty()) return false;
  if (s[0] == '+') i = 1;
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

[[noreturn]] void fail(const std::string& what) {  // anyhow: "Error: <context>" on stderr, exit status 1
  std::fprintf(stderr, "Error: %s\n", what.c_str());
  std::exit(1);
}

Stats parse_paf(const std::string& path) {  // :103-164 (plain text; the reference also reads .gz)
  std::ifstream in(path, std::ios::binary);
  if (!in) fail("Failed to open " + path);
  Stats st;
  std::set<std::pair<std::string, std::string>> chr_pairs;
  std::string line;
  while (std::getline(in, line)) {
    if (!in.eof() && !line.empty() && line.back() == '\r') line.pop_back();  // BufRead::lines
    std::vector<std::string> f;
    size_t b = 0;
    for (;;) {
      const size_t p = line.find('\t', b);
      f.push_back(line.substr(b, p == std::string::npos ? std::string::npos : p - b));
      if (p == std::string::npos) break;
      b = p + 1;
    }
    if (f.size() < 11) continue;  // :113-115
    uint64_t query_len, query_start, query_end, target_len, matches, block_len;
    if (!rust_u64(f[1], &query_len)) fail("Invalid query length");
    if (!rust_u64(f[2], &query_start)) fail("Invalid query start");
    if (!rust_u64(f[3], &query_end)) fail("Invalid query end");
    if (!rust_u64(f[6], &target_len)) fail("Invalid target length");
    if (!rust_u64(f[9], &matches)) fail("Invalid match count");
    if (!rust_u64(f[10], &block_len)) fail("Invalid block length");
    const std::string &query = f[0], &target = f[5];
    st.total_mappings += 1;
    const uint64_t mapping_len = query_end - query_start;  // release build: wrapping
    st.total_bases += mapping_len;
    st.total_matches += matches;
    for (const auto& kv : {std::make_pair(query, query_len), std::make_pair(target, target_len)}) {  // :131-132, last writer wins
      if (!st.genome_sizes.count(kv.first)) st.seq_order.push_back(kv.first);
      st.genome_sizes[kv.first] = kv.second;
    }
    const std::string qg = genome_prefix(query), tg = genome_prefix(target);
    if (query == target) {
      st.self_mappings += 1;
    } else if (qg != tg) {
      st.inter_genome += 1;
      const auto pair = std::make_pair(qg, tg);
      if (!st.pair_bases.count(pair)) st.pair_order.push_back(pair);
      st.pair_bases[pair] += mapping_len;
      st.pair_matches[pair] += matches;
    } else {
      st.inter_chromosomal += 1;
    }
    chr_pairs.insert({query, target});
  }
  st.chr_pair_count = chr_pairs.size();
  return st;
}

Coverage coverage_of(const Stats& st) {  // :42-73
  std::unordered_map<std::string, uint64_t> genome_totals;
  for (const auto& s : st.seq_order) genome_totals[genome_prefix(s)] += st.genome_sizes.at(s);
  Coverage c;
  for (const auto& pair : st.pair_order) {
    auto it = genome_totals.find(pair.first);
    if (it == genome_totals.end()) continue;
    const uint64_t bases = st.pair_bases.at(pair);
    c.per_pair.emplace_back(pair.first, pair.second, 100.0 * (double)bases / (double)it->second, bases);
  }
  if (!c.per_pair.empty()) {
    double sum = 0.0;  // Iterator::sum::<f64>() folds from 0.0
    for (const auto& p : c.per_pair) sum += std::get<2>(p);
    c.avg_coverage = sum / (double)c.per_pair.size();
  }
  for (const auto& p : c.per_pair) c.above_95 += std::get<2>(p) > 95.0;
  c.genome_pairs = c.per_pair.size();
  return c;
}

double avg_identity(const Stats& st) { return st.total_bases > 0 ? (double)st.total_matches / (double)st.total_bases : 0.0; }  // :75-81

std::string format_number(uint64_t n) {  // :305-315
  const std::string s = std::to_string(n);
  std::string r;
  for (size_t i = 0; i < s.size(); ++i) {
    if (i > 0 && (s.size() - i) % 3 == 0) r += ',';
    r += s[i];
  }
  return r;
}
std::string format_signed(int64_t n) {  // :317-323
  return n >= 0 ? "+" + format_number((uint64_t)n) : "-" + format_number((uint64_t)(-n));
}
// {:.1} / {:+.1} of an f64 (Display: "NaN", "inf", "-inf"; the sign flag gives "+inf", never "+NaN")
std::string f1(double v, bool plus = false) {
  if (std::isnan(v)) return "NaN";
  if (std::isinf(v)) return v < 0 ? "-inf" : (plus ? "+inf" : "inf");
  char buf[512];
  std::snprintf(buf, sizeof buf, plus ? "%+.1f" : "%.1f", v);
  return buf;
}
size_t chars(const std::string& s) {  // width counts chars, not bytes
  size_t n = 0;
  for (unsigned char c : s) n += (c & 0xc0) != 0x80;
  return n;
}
std::string padl(const std::string& s, size_t w) { return chars(s) >= w ? s : std::string(w - chars(s), ' ') + s; }  // {:>w}
std::string padr(const std::string& s, size_t w) { return chars(s) >= w ? s : s + std::string(w - chars(s), ' '); }  // {:w}
std::string trim_hashes(std::string s) {
  while (!s.empty() && s.back() == '#') s.pop_back();
  return s;
}

void print_stats(const std::string& path, const Stats& st, bool detailed) {  // :166-228
  const Coverage cov = coverage_of(st);
  std::printf("\nStatistics for %s:\n", path.c_str());
  std::printf("%s\n", std::string(60, '=').c_str());
  std::printf("Total mappings:        %s\n", padl(format_number(st.total_mappings), 12).c_str());
  std::printf("Total bases:           %s\n", padl(format_number(st.total_bases), 12).c_str());
  std::printf("Average identity:      %s%%\n", padl(f1(avg_identity(st) * 100.0), 11).c_str());
  std::printf("Self mappings:         %s\n", padl(format_number(st.self_mappings), 12).c_str());
  std::printf("Inter-chromosomal:     %s\n", padl(format_number(st.inter_chromosomal), 12).c_str());
  std::printf("Inter-genome:          %s\n", padl(format_number(st.inter_genome), 12).c_str());
  std::printf("Chromosome pairs:      %s\n", padl(format_number(st.chr_pair_count), 12).c_str());
  std::printf("Genome pairs:          %s\n", padl(std::to_string(cov.genome_pairs), 12).c_str());
  std::printf("Average coverage:      %s%%\n", padl(f1(cov.avg_coverage), 11).c_str());
  std::printf("Pairs >95%% coverage:   %s\n", padl(std::to_string(cov.above_95) + "/" + std::to_string(cov.genome_pairs), 12).c_str());
  if (detailed && !cov.per_pair.empty()) {
    std::printf("\nPer-genome-pair statistics:\n");
    std::printf("%s\n", std::string(60, '-').c_str());
    auto pairs = cov.per_pair;
    if (pairs.size() > 1)
      for (const auto& p : pairs)
        if (std::isnan(std::get<2>(p))) {  // partial_cmp().unwrap() on a NaN coverage (0 bases over a 0-length genome)
          std::fflush(stdout);
          std::fprintf(stderr, "thread 'main' panicked: called `Option::unwrap()` on a `None` value\n");
          std::exit(101);
        }
    // sort_by(|a, b| b.2.partial_cmp(&a.2).unwrap()): stable, coverage descending
    for (size_t i = 1; i < pairs.size(); ++i)
      for (size_t j = i; j > 0 && std::get<2>(pairs[j]) > std::get<2>(pairs[j - 1]); --j) std::swap(pairs[j], pairs[j - 1]);
    for (const auto& [q, t, c, bases] : pairs) {
      const uint64_t m = st.pair_matches.count({q, t}) ? st.pair_matches.at({q, t}) : 0;
      const double identity = bases > 0 ? (double)m / (double)bases * 100.0 : 0.0;
      std::printf("%s -> %s %s%% cov, %s%% id, %s bp\n", padr(trim_hashes(q), 20).c_str(), padr(trim_hashes(t), 20).c_str(),
                  padl(f1(c), 6).c_str(), padl(f1(identity), 6).c_str(), padl(format_number(bases), 10).c_str());
    }
  }
}

void print_comparison(const char* label, uint64_t v1, uint64_t v2) {  // :286-303
  std::printf("\n%s:\n", label);
  std::printf("  %s %s\n", padr("Before", 30).c_str(), padl(format_number(v1), 12).c_str());
  std::printf("  %s %s\n", padr("After", 30).c_str(), padl(format_number(v2), 12).c_str());
  const int64_t diff = (int64_t)v2 - (int64_t)v1;
  const double pct = v1 > 0 ? 100.0 * (double)diff / (double)v1 : 0.0;
  std::printf("  %s %s (%s%%)\n", padr("Change", 30).c_str(), padl(format_signed(diff), 12).c_str(), f1(pct, true).c_str());
}

void compare_stats(const std::string& f1n, const std::string& f2n, const Stats& a, const Stats& b) {  // :230-284
  const Coverage c1 = coverage_of(a), c2 = coverage_of(b);
  const double i1 = avg_identity(a), i2 = avg_identity(b);
  std::printf("\nComparison: %s vs %s\n", f1n.c_str(), f2n.c_str());
  std::printf("%s\n", std::string(60, '=').c_str());
  print_comparison("Mappings", a.total_mappings, b.total_mappings);
  print_comparison("Total bases", a.total_bases, b.total_bases);
  std::printf("\nAverage identity:\n");
  std::printf("  %s %s%%\n", padr(f1n, 30).c_str(), padl(f1(i1 * 100.0), 11).c_str());
  std::printf("  %s %s%%\n", padr(f2n, 30).c_str(), padl(f1(i2 * 100.0), 11).c_str());
  std::printf("  %s %s%%\n", padr("Change", 30).c_str(), padl(f1((i2 - i1) * 100.0, true), 10).c_str());
  print_comparison("Inter-chromosomal", a.inter_chromosomal, b.inter_chromosomal);
  print_comparison("Chromosome pairs", a.chr_pair_count, b.chr_pair_count);
  std::printf("\nAverage genome pair coverage:\n");
  std::printf("  %s %s%%\n", padr(f1n, 30).c_str(), padl(f1(c1.avg_coverage), 11).c_str());
  std::printf("  %s %s%%\n", padr(f2n, 30).c_str(), padl(f1(c2.avg_coverage), 11).c_str());
  std::printf("  %s %s%%\n", padr("Change", 30).c_str(), padl(f1(c2.avg_coverage - c1.avg_coverage, true), 10).c_str());
  std::printf("\nGenome pairs with >95%% coverage:\n");
  std::printf("  %s %s\n", padr(f1n, 30).c_str(), padl(std::to_string(c1.above_95) + "/" + std::to_string(c1.genome_pairs), 12).c_str());
  std::printf("  %s %s\n", padr(f2n, 30).c_str(), padl(std::to_string(c2.above_95) + "/" + std::to_string(c2.genome_pairs), 12).c_str());
}

}  // namespace

int main(int argc, char** argv) {  // :325-341
  std::vector<std::string> files;
  bool detailed = false;
  for (int i = 1; i < argc; ++i) {
    const std::string a = argv[i];
    if (a == "-d" || a == "--detailed") detailed = true;
    else files.push_back(a);
  }
  if (files.empty() || files.size() > 2) {
    std::fprintf(stderr, "usage: alnstats-ref <file1> [file2] [-d]\n");
    return 2;
  }
  const Stats s1 = parse_paf(files[0]);
  if (files.size() == 2) {
    const Stats s2 = parse_paf(files[1]);
    compare_stats(files[0], files[1], s1, s2);
  } else {
    print_stats(files[0], s1, detailed);
  }
  return 0;
}
