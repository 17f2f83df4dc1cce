// sweepga-gpu: the reference's filter-path command line (`sweepga <paf> --output-file out.paf ...`) with
// PafFilter::apply_filters running on an MI355X through libsweepga_gpu.so.
//
// Host side only, written in C++ because the reference's host is compiled code (Rust; no Rust toolchain in
// this image):
//   flags and defaults ........ src/cli.rs:204-288; flag -> FilterConfig mapping src/main.rs:3477-3568, 3590-3619
//   parse_filter_mode ......... src/main.rs:244-293        parse_metric_number / parse_identity_value ... src/cli.rs:26-130
//   extract_metadata .......... src/paf_filter.rs:292-376  parse_cigar_counts ... src/paf.rs:32-64
//   name interning ............ src/sequence_index.rs:7-31 genome prefixes ... src/paf_filter.rs:1022-1030,
//                                                          src/plane_sweep_scaffold.rs:13-22
//   write_filtered_output ..... src/paf_filter.rs:1689-1726
// Everything between "records parsed" and "per-record status + chain id" is one swg_filter() call.
// There is no CPU filter in this binary: without a GPU it exits with the library's error.
#include <cerrno>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <string_view>
#include <unordered_map>
#include <vector>

#include "../../../include/sweepga_gpu.h"

namespace {

[[noreturn]] void die(int code, const std::string& msg) {
  std::fprintf(stderr, "sweepga-gpu: %s\n", msg.c_str());
  std::exit(code);
}

// ---- Rust-compatible scalar parsers -----------------------------------------------------------------
bool parse_u64(std::string_view s, uint64_t* out) {  // str::parse::<u64>
  size_t i = 0;
  if (s.empty()) return false;
  if (s[0] == '+') i = 1;
  if (i >= s.size()) return false;
  uint64_t v = 0;
  for (; i < s.size(); ++i) {
    const char c = s[i];
    if (c < '0' || c > '9') return false;
    const uint64_t d = (uint64_t)(c - '0');
    if (v > (UINT64_MAX - d) / 10) return false;
    v = v * 10 + d;
  }
  *out = v;
  return true;
}
bool parse_f64(std::string_view sv, double* out) {  // str::parse::<f64>: no whitespace, no hex floats
  if (sv.empty()) return false;
  for (char c : sv)
    if (c == 'x' || c == 'X' || c == ' ' || c == '\t' || c == '\n' || c == '(') return false;
  std::string s(sv);
  char* e = nullptr;
  const double v = std::strtod(s.c_str(), &e);
  if (e == s.c_str() || *e != '\0') return false;
  *out = v;
  return true;
}
bool parse_metric_number(const std::string& s, uint64_t* out) {  // cli.rs:26-61
  if (s.empty()) return false;
  std::string num = s;
  char suffix = 0;
  const char last = s.back();
  if ((last >= 'a' && last <= 'z') || (last >= 'A' && last <= 'Z')) {
    suffix = last;
    num = s.substr(0, s.size() - 1);
  }
  double base;
  if (!parse_f64(num, &base)) return false;
  double mult = 1.0;
  switch (suffix) {
    case 0: break;
    case 'k': case 'K': mult = 1e3; break;
    case 'm': case 'M': mult = 1e6; break;
    case 'g': case 'G': mult = 1e9; break;
    default: return false;
  }
  const double r = base * mult;
  if (r > (double)UINT64_MAX) return false;
  *out = !(r > 0.0) ? 0 : (r >= 18446744073709551616.0 ? UINT64_MAX : (uint64_t)r);
  return true;
}
bool parse_identity_value(const std::string& s, double* out) {  // cli.rs:76-130, numeric forms
  std::string lower = s;
  for (auto& c : lower)
    if (c >= 'A' && c <= 'Z') c = (char)(c - 'A' + 'a');
  if (lower.rfind("ani", 0) == 0) return false;  // needs the ANI pre-pass (main.rs:334-688), out of scope
  double v;
  if (!parse_f64(s, &v)) return false;
  *out = v > 1.0 ? v / 100.0 : v;
  return true;
}
int parse_scoring(const std::string& s) {  // main.rs:3485-3492
  if (s == "ani" || s == "identity") return SWG_SCORE_IDENTITY;
  if (s == "length") return SWG_SCORE_LENGTH;
  if (s == "length-ani" || s == "length-identity") return SWG_SCORE_LENGTH_IDENTITY;
  if (s == "matches") return SWG_SCORE_MATCHES;
  return SWG_SCORE_LOG_LENGTH_IDENTITY;
}
// main.rs:244-293.  Returns false for a bare "0" (the reference exits the process there).
bool parse_filter_mode(const std::string& mode, int32_t* fmode, uint64_t* pq, uint64_t* pt) {
  const std::string INF = "\xE2\x88\x9E";
  std::string lower = mode;
  for (auto& c : lower)
    if (c >= 'A' && c <= 'Z') c = (char)(c - 'A' + 'a');
  auto set = [&](int m, uint64_t q, uint64_t t) {
    *fmode = m;
    *pq = q;
    *pt = t;
    return true;
  };
  if (lower == "1:1") return set(SWG_MODE_ONE_TO_ONE, 1, 1);
  if (lower == "1" || lower == "1:" + INF || lower == "1:infinity" || lower == "1:many") return set(SWG_MODE_ONE_TO_MANY, 1, 0);
  if (lower == INF + ":1" || lower == "infinity:1" || lower == "many:1") return set(SWG_MODE_MANY_TO_MANY, 0, 1);
  if (lower == "many:many" || lower == INF + ":" + INF || lower == "infinity:infinity" || lower == "many" || lower == INF ||
      lower == "infinity" || lower == "-1" || lower == "-1:-1")
    return set(SWG_MODE_MANY_TO_MANY, 0, 0);
  const size_t colon = lower.find(':');
  if (colon != std::string::npos) {
    if (lower.find(':', colon + 1) != std::string::npos) return set(SWG_MODE_ONE_TO_ONE, 1, 1);  // parts.len() != 2
    auto side = [&](const std::string& p) -> uint64_t {
      if (p == INF || p == "infinity" || p == "many" || p == "-1") return 0;
      uint64_t v;
      return parse_u64(p, &v) && v > 0 ? v : 0;
    };
    const uint64_t q = side(lower.substr(0, colon)), t = side(lower.substr(colon + 1));
    const int m = (q == 1 && t == 1) ? SWG_MODE_ONE_TO_ONE : (q == 1 && t == 0) ? SWG_MODE_ONE_TO_MANY : SWG_MODE_MANY_TO_MANY;
    return set(m, q, t);
  }
  uint64_t n;
  if (parse_u64(mode, &n)) {
    if (n == 0) return false;
    return set(SWG_MODE_ONE_TO_MANY, n, 0);
  }
  return set(SWG_MODE_ONE_TO_ONE, 1, 1);
}
// paf.rs:32-64: sum of '=' lengths; false on a number parse error
bool cigar_matches(std::string_view cigar, uint64_t* matches) {
  uint64_t m = 0, cur = 0;
  bool have = false, overflow = false;
  for (char ch : cigar) {
    if (ch >= '0' && ch <= '9') {
      const uint64_t d = (uint64_t)(ch - '0');
      if (cur > (UINT64_MAX - d) / 10) overflow = true;
      cur = cur * 10 + d;
      have = true;
    } else {
      if (!have || overflow) return false;
      if (ch == '=') m += cur;
      cur = 0;
      have = false;
      overflow = false;
    }
  }
  *matches = m;
  return true;
}

// ---- sequence index ---------------------------------------------------------------------------------------
struct SequenceIndex {
  std::unordered_map<std::string, uint32_t> ids;
  std::vector<std::string> names;
  uint32_t get_or_insert(std::string_view nm) {
    auto it = ids.find(std::string(nm));
    if (it != ids.end()) return it->second;
    const uint32_t id = (uint32_t)names.size();
    names.emplace_back(nm);
    ids.emplace(names.back(), id);
    return id;
  }
};
std::string prefix_last(const std::string& n) {  // paf_filter.rs:1022-1030
  const size_t p = n.rfind('#');
  return p == std::string::npos ? n : n.substr(0, p + 1);
}
std::string prefix_two(const std::string& n) {  // plane_sweep_scaffold.rs:13-22
  const size_t p1 = n.find('#');
  if (p1 == std::string::npos) return n;
  const size_t p2 = n.find('#', p1 + 1);
  return n.substr(0, p1) + "#" + (p2 == std::string::npos ? n.substr(p1 + 1) : n.substr(p1 + 1, p2 - p1 - 1)) + "#";
}
uint32_t genome_table(const SequenceIndex& idx, std::string (*fn)(const std::string&), std::vector<uint32_t>* out) {
  std::unordered_map<std::string, uint32_t> g;
  out->resize(idx.names.empty() ? 1 : idx.names.size(), 0);
  for (size_t i = 0; i < idx.names.size(); ++i) {
    auto it = g.emplace(fn(idx.names[i]), (uint32_t)g.size()).first;
    (*out)[i] = it->second;
  }
  return g.empty() ? 1u : (uint32_t)g.size();
}

struct Columns {
  std::vector<uint32_t> q_id, t_id, qs, qe, ts, te, matches, block;
  std::vector<double> identity;
  std::vector<uint8_t> strand;
  std::vector<uint64_t> rank;  // line index of each record
};

uint32_t narrow(uint64_t v, const char* what, size_t line) {
  if (v > 0xffffffffull)
    die(2, std::string(what) + " >= 2^32 on line " + std::to_string(line + 1) + " is not supported by the GPU layout");
  return (uint32_t)v;
}

}  // namespace

int main(int argc, char** argv) {
  std::string input, output_file;
  std::string num_mappings = "many:many", scoring = "log-length-ani", min_identity = "0";
  std::string scaffold_filter = "many:many", min_scaffold_identity = "0";
  double overlap = 0.95, scaffold_overlap = 0.5;
  uint64_t scaffold_jump = 50000, scaffold_mass = 10000, scaffold_dist = 0, block_length = 0;
  bool keep_self = false, no_filter = false, scaffolds_only = false, quiet = false;
  int device = 0;
  for (int i = 1; i < argc; ++i) {
    std::string a = argv[i], val;
    const size_t eq = a.find('=');
    const bool has_eq = a.rfind("--", 0) == 0 && eq != std::string::npos;
    if (has_eq) {
      val = a.substr(eq + 1);
      a = a.substr(0, eq);
    }
    auto value = [&]() -> std::string {
      if (has_eq) return val;
      if (i + 1 >= argc) die(2, "missing value for " + a);
      return argv[++i];
    };
    if (a == "--output-file" || a == "-o") output_file = value();
    else if (a == "--num-mappings") num_mappings = value();
    else if (a == "--overlap") overlap = std::strtod(value().c_str(), nullptr);
    else if (a == "--scoring") scoring = value();
    else if (a == "--min-aln-identity") min_identity = value();
    else if (a == "--min-aln-length") { if (!parse_metric_number(value(), &block_length)) die(2, "bad --min-aln-length"); }
    else if (a == "--self") keep_self = true;
    else if (a == "--no-filter") no_filter = true;
    else if (a == "--scaffold-jump") { if (!parse_metric_number(value(), &scaffold_jump)) die(2, "bad --scaffold-jump"); }
    else if (a == "--scaffold-mass") { if (!parse_metric_number(value(), &scaffold_mass)) die(2, "bad --scaffold-mass"); }
    else if (a == "--scaffold-filter") scaffold_filter = value();
    else if (a == "--scaffold-overlap") scaffold_overlap = std::strtod(value().c_str(), nullptr);
    else if (a == "--scaffold-dist") { if (!parse_metric_number(value(), &scaffold_dist)) die(2, "bad --scaffold-dist"); }
    else if (a == "--min-scaffold-identity") min_scaffold_identity = value();
    else if (a == "--scaffolds-only") scaffolds_only = true;
    else if (a == "--device") device = std::atoi(value().c_str());
    else if (a == "--quiet") quiet = true;
    else if (a == "--no-adaptive-scaffolds" || a == "--paf") { /* no effect for PAF input (main.rs:3515-3527) */ }
    else if (a == "--threads" || a == "-t") (void)value();
    else if (a == "--help" || a == "-h") {
      std::puts("usage: sweepga-gpu <in.paf> [--output-file out.paf] [--num-mappings M] [--overlap F] [--scoring S]\n"
                "         [--min-aln-identity I] [--min-aln-length N] [--self] [--no-filter] [--scaffold-jump N]\n"
                "         [--scaffold-mass N] [--scaffold-filter M] [--scaffold-overlap F] [--scaffold-dist N]\n"
                "         [--min-scaffold-identity I] [--scaffolds-only] [--device D] [--quiet]\n"
                "Filter path of pangenome/sweepga on an MI355X (libsweepga_gpu.so).  No CPU fallback.");
      return 0;
    } else if (a.rfind("-", 0) == 0 && a != "-") die(2, "unknown flag " + a);
    else input = a;
  }
  if (input.empty()) die(2, "usage: sweepga-gpu <in.paf> [--output-file out.paf] [filter flags]   (--help)");

  swg_config cfg{};
  if (!parse_filter_mode(num_mappings, &cfg.mapping_filter_mode, &cfg.mapping_max_per_query, &cfg.mapping_max_per_target)) return 1;
  if (!parse_filter_mode(scaffold_filter, &cfg.scaffold_filter_mode, &cfg.scaffold_max_per_query, &cfg.scaffold_max_per_target)) return 1;
  cfg.scoring_function = parse_scoring(scoring);
  cfg.min_block_length = block_length;
  cfg.overlap_threshold = overlap;
  cfg.scaffold_gap = scaffold_jump;
  cfg.min_scaffold_length = scaffold_mass;
  cfg.scaffold_overlap_threshold = scaffold_overlap;
  cfg.scaffold_max_deviation = scaffold_dist;
  if (!parse_identity_value(min_identity, &cfg.min_identity)) die(2, "bad --min-aln-identity (aniN presets need the ANI pre-pass)");
  if (min_scaffold_identity.empty()) cfg.min_scaffold_identity = cfg.min_identity;
  else if (!parse_identity_value(min_scaffold_identity, &cfg.min_scaffold_identity)) die(2, "bad --min-scaffold-identity");
  cfg.keep_self = keep_self;
  cfg.scaffolds_only = scaffolds_only;

  // ---- read the whole input; line table (BufRead::lines: split on '\n', strip one '\r')
  using clk = std::chrono::steady_clock;
  const auto t0 = clk::now();
  std::string text;
  {
    FILE* f = std::fopen(input.c_str(), "rb");
    if (!f) die(2, "cannot open " + input + ": " + std::strerror(errno));
    char buf[1 << 16];
    size_t n;
    while ((n = std::fread(buf, 1, sizeof buf, f)) > 0) text.append(buf, n);
    std::fclose(f);
  }
  std::vector<std::pair<size_t, size_t>> lines;  // (offset, length) without "\n" / "\r\n"
  for (size_t pos = 0; pos < text.size();) {
    const void* nl = std::memchr(text.data() + pos, '\n', text.size() - pos);
    const size_t end = nl ? (size_t)((const char*)nl - text.data()) : text.size();
    size_t len = end - pos;
    if (len && text[pos + len - 1] == '\r') --len;
    lines.emplace_back(pos, len);
    pos = end + 1;
  }
  FILE* out = output_file.empty() ? stdout : std::fopen(output_file.c_str(), "wb");
  if (!out) die(2, "cannot create " + output_file + ": " + std::strerror(errno));
  if (no_filter) {  // main.rs:3461-3470
    for (auto& ln : lines) {
      std::fwrite(text.data() + ln.first, 1, ln.second, out);
      std::fputc('\n', out);
    }
    if (out != stdout) std::fclose(out);
    return 0;
  }

  // ---- extract_metadata (paf_filter.rs:298-373)
  SequenceIndex idx;
  Columns c;
  std::vector<std::string_view> f;
  for (size_t li = 0; li < lines.size(); ++li) {
    const std::string_view line(text.data() + lines[li].first, lines[li].second);
    f.clear();
    for (size_t s = 0;;) {
      const size_t t = line.find('\t', s);
      if (t == std::string_view::npos) {
        f.push_back(line.substr(s));
        break;
      }
      f.push_back(line.substr(s, t - s));
      s = t + 1;
    }
    if (f.size() < 11) continue;
    auto u64_or = [](std::string_view s, uint64_t d) {
      uint64_t v;
      return parse_u64(s, &v) ? v : d;
    };
    uint64_t matches = u64_or(f[9], 0);
    const uint64_t block = u64_or(f[10], 1);
    const double denom = (double)(block > 1 ? block : 1);
    double identity = (double)matches / denom;
    for (size_t k = 11; k < f.size(); ++k) {
      if (f[k].substr(0, 5) == "dv:f:") {
        double dv;
        if (parse_f64(f[k].substr(5), &dv)) identity = 1.0 - dv;
      } else if (f[k].substr(0, 5) == "cg:Z:") {
        uint64_t cm;
        if (cigar_matches(f[k].substr(5), &cm) && cm > 0) {
          matches = cm;
          identity = (double)cm / denom;
        }
      }
    }
    c.q_id.push_back(idx.get_or_insert(f[0]));
    c.t_id.push_back(idx.get_or_insert(f[5]));
    c.qs.push_back(narrow(u64_or(f[2], 0), "query_start", li));
    c.qe.push_back(narrow(u64_or(f[3], 0), "query_end", li));
    c.ts.push_back(narrow(u64_or(f[7], 0), "target_start", li));
    c.te.push_back(narrow(u64_or(f[8], 0), "target_end", li));
    c.matches.push_back(narrow(matches, "matches", li));
    c.block.push_back(narrow(block, "block_length", li));
    c.identity.push_back(identity);
    c.strand.push_back(f[4] == "+" ? 0 : 1);
    c.rank.push_back(li);
  }
  const auto t1 = clk::now();
  const uint64_t n = c.rank.size();
  std::vector<uint32_t> g_last, g_two;
  const uint32_t n_last = genome_table(idx, prefix_last, &g_last), n_two = genome_table(idx, prefix_two, &g_two);

  // ---- apply_filters on the GPU
  std::vector<uint8_t> status(n ? n : 1, 0);
  std::vector<uint32_t> chain(n ? n : 1, 0);
  swg_stats st{};
  if (n) {
    swg_ctx* ctx = nullptr;
    int rc = swg_create(device, &ctx);
    if (rc != SWG_OK) die(3, std::string("no usable GPU: ") + swg_last_error(nullptr));
    swg_records r{};
    r.n = n;
    r.q_id = c.q_id.data();
    r.t_id = c.t_id.data();
    r.q_start = c.qs.data();
    r.q_end = c.qe.data();
    r.t_start = c.ts.data();
    r.t_end = c.te.data();
    r.identity = c.identity.data();
    r.matches = c.matches.data();
    r.block_len = c.block.data();
    r.strand = c.strand.data();
    r.n_seq = (uint32_t)(idx.names.empty() ? 1 : idx.names.size());
    r.seq_genome_last = g_last.data();
    r.n_genome_last = n_last;
    r.seq_genome_two = g_two.data();
    r.n_genome_two = n_two;
    rc = swg_filter(ctx, &r, &cfg, status.data(), chain.data(), &st);
    if (rc != SWG_OK) die(3, std::string("filter failed: ") + swg_last_error(ctx));
    swg_destroy(ctx);
  }
  const auto t2 = clk::now();

  // ---- write_filtered_output (paf_filter.rs:1689-1726): input order, original bytes + tags
  static const char* TAG[4] = {"", "scaffold", "rescued", "unassigned"};
  std::string buf;
  buf.reserve(1 << 20);
  uint64_t kept = 0;
  for (uint64_t k = 0; k < n; ++k) {
    if (!status[k]) continue;
    ++kept;
    const auto& ln = lines[c.rank[k]];
    buf.append(text.data() + ln.first, ln.second);
    if (chain[k]) {
      buf += "\tch:Z:chain_";
      buf += std::to_string(chain[k]);
    }
    buf += "\tst:Z:";
    buf += TAG[status[k] & 3];
    buf += '\n';
    if (buf.size() > (1 << 20) - 4096) {
      std::fwrite(buf.data(), 1, buf.size(), out);
      buf.clear();
    }
  }
  std::fwrite(buf.data(), 1, buf.size(), out);
  if (out != stdout) std::fclose(out);
  const auto t3 = clk::now();
  if (!quiet) {
    auto ms = [](clk::time_point a, clk::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
    std::fprintf(stderr,
                 "[sweepga-gpu] %llu records -> %llu kept | parse %.1f ms, filter %.1f ms (device %.1f, h2d %.1f, d2h %.1f), write %.1f ms\n",
                 (unsigned long long)n, (unsigned long long)kept, ms(t0, t1), ms(t1, t2), st.device_ms, st.h2d_ms, st.d2h_ms,
                 ms(t2, t3));
  }
  return 0;
}
