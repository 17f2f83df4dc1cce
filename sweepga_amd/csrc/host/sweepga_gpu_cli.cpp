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
#include <unistd.h>

#include <cerrno>
#include <sys/mman.h>
#include <sys/stat.h>

#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <string_view>
#include <thread>
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
bool parse_int(const std::string& t, int* out) {  // decimal digits only
  if (t.empty() || t.size() > 9 || t.find_first_not_of("0123456789") != std::string::npos) return false;
  *out = std::atoi(t.c_str());
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
// --sparsify (src/knn_graph.rs:59-160, src/main.rs:3494-3509).  `none`, `all`, a bare fraction and `random:<f>` have no
// effect on the PAF path (the filter never reads FilterConfig.sparsity).  `tree:` / `knn:` make the reference run
// tree_filter::apply_tree_filter_to_paf on the input BEFORE the filter (src/main.rs:3640-3688).  0 = fine (no effect),
// 1 = a strategy that is "not valid for post-alignment PAF/1aln filtering", 2 = unparsable, 3 = tree sampling (parameters
// returned through the pointers).
int check_sparsify(const std::string& v, unsigned long* tree_near = nullptr, unsigned long* tree_far = nullptr, double* tree_rand = nullptr) {
  auto frac_ok = [](const std::string& t, bool open_top) {
    char* e = nullptr;
    const double f = std::strtod(t.c_str(), &e);
    if (t.empty() || e == t.c_str() || *e) return false;
    return f > 0.0 && (open_top ? f < 1.0 : f <= 1.0);
  };
  {
    char* e = nullptr;
    const double f = std::strtod(v.c_str(), &e);
    if (!v.empty() && e != v.c_str() && !*e) return (f > 0.0 && f <= 1.0) ? 0 : 2;
  }
  if (v == "none" || v == "all") return 0;
  if (v == "auto") return 1;
  if (v.rfind("random:", 0) == 0) return frac_ok(v.substr(7), false) ? 0 : 2;
  if (v.rfind("giant:", 0) == 0 || v.rfind("connectivity:", 0) == 0) return frac_ok(v.substr(v.find(':') + 1), true) ? 1 : 2;
  if (v.rfind("wfmash:", 0) == 0) return (v.substr(7) == "auto" || frac_ok(v.substr(7), false)) ? 1 : 2;
  if (v.rfind("tree:", 0) == 0 || v.rfind("knn:", 0) == 0) {
    const std::string body = v.substr(v.find(':') + 1);
    unsigned long kn = 0, kf = 0;
    double rf = 0.0;
    int parts = 0;
    for (size_t s0 = 0; s0 <= body.size(); ++parts) {
      const size_t c = body.find(':', s0);
      const std::string tok = body.substr(s0, c == std::string::npos ? std::string::npos : c - s0);
      char* e = nullptr;
      if (parts < 2) {
        if (tok.empty() || tok.find_first_not_of("0123456789") != std::string::npos) return 2;
        (parts == 0 ? kn : kf) = std::strtoul(tok.c_str(), &e, 10);
      } else if (parts == 2) {
        rf = std::strtod(tok.c_str(), &e);
        if (tok.empty() || e == tok.c_str() || *e) return 2;
      } else {
        return 2;
      }
      if (c == std::string::npos) {
        ++parts;
        break;
      }
      s0 = c + 1;
    }
    if (parts > 3 || (kn == 0 && kf == 0) || rf < 0.0 || rf > 1.0) return 2;
    if (tree_near) *tree_near = kn;
    if (tree_far) *tree_far = kf;
    if (tree_rand) *tree_rand = rf;
    return 3;
  }
  return 2;
}

}  // namespace

int main(int argc, char** argv) {
  const auto t_main = std::chrono::steady_clock::now();
  std::string input, output_file;
  std::string num_mappings = "many:many", scoring = "log-length-ani", min_identity = "0";
  std::string scaffold_filter = "many:many", min_scaffold_identity = "0", ani_method_s = "n100";
  double overlap = 0.95, scaffold_overlap = 0.5;
  uint64_t scaffold_jump = 50000, scaffold_mass = 10000, scaffold_dist = 0, block_length = 0;
  bool keep_self = false, no_filter = false, scaffolds_only = false, quiet = false;
  int device = 0, threads = 0;
  std::vector<int> devices;
  std::string bad_sparsify, tree_sparsify;
  unsigned long tree_near = 0, tree_far = 0;
  double tree_rand = 0.0;
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
    else if (a == "--overlap") { if (!parse_f64(value(), &overlap)) die(2, "invalid value for --overlap"); }  // clap rejects it too
    else if (a == "--scoring") scoring = value();
    else if (a == "--min-aln-identity") min_identity = value();
    else if (a == "--min-aln-length") { if (!parse_metric_number(value(), &block_length)) die(2, "bad --min-aln-length"); }
    else if (a == "--self") keep_self = true;
    else if (a == "--no-filter") no_filter = true;
    else if (a == "--scaffold-jump") { if (!parse_metric_number(value(), &scaffold_jump)) die(2, "bad --scaffold-jump"); }
    else if (a == "--scaffold-mass") { if (!parse_metric_number(value(), &scaffold_mass)) die(2, "bad --scaffold-mass"); }
    else if (a == "--scaffold-filter") scaffold_filter = value();
    else if (a == "--scaffold-overlap") { if (!parse_f64(value(), &scaffold_overlap)) die(2, "invalid value for --scaffold-overlap"); }
    else if (a == "--scaffold-dist") { if (!parse_metric_number(value(), &scaffold_dist)) die(2, "bad --scaffold-dist"); }
    else if (a == "--min-scaffold-identity") min_scaffold_identity = value();
    else if (a == "--scaffolds-only") scaffolds_only = true;
    else if (a == "--ani-method") ani_method_s = value();
    else if (a == "--sparsify") {
      const std::string v = value();
      const int rc = check_sparsify(v, &tree_near, &tree_far, &tree_rand);
      if (rc == 2) die(2, "invalid value for --sparsify");
      if (rc == 1) bad_sparsify = v;  // reported after the --no-filter shortcut, as in main.rs:3461-3509
      tree_sparsify = rc == 3 ? v : std::string();
    }
    else if (a == "--device") { if (!parse_int(value(), &device) || device < 0) die(2, "invalid value for --device"); }
    else if (a == "--devices") {  // comma-separated: shard the genome pairs over several GPUs of the node
      const std::string v = value();
      for (size_t s0 = 0; s0 <= v.size();) {
        const size_t c = v.find(',', s0);
        const std::string tok = v.substr(s0, c == std::string::npos ? std::string::npos : c - s0);
        if (tok.empty() || tok.find_first_not_of("0123456789") != std::string::npos) die(2, "bad --devices");
        devices.push_back(std::atoi(tok.c_str()));
        if (c == std::string::npos) break;
        s0 = c + 1;
      }
    }
    else if (a == "--quiet") quiet = true;
    else if (a == "--no-adaptive-scaffolds" || a == "--paf") { /* no effect for PAF input (main.rs:3515-3527) */ }
    else if (a == "--threads" || a == "-t") { if (!parse_int(value(), &threads) || threads < 0) die(2, "invalid value for --threads"); }
    else if (a == "--help" || a == "-h") {
      std::puts("usage: sweepga-gpu <in.paf> [--output-file out.paf] [--num-mappings M] [--overlap F] [--scoring S]\n"
                "         [--min-aln-identity I] [--min-aln-length N] [--self] [--no-filter] [--scaffold-jump N]\n"
                "         [--scaffold-mass N] [--scaffold-filter M] [--scaffold-overlap F] [--scaffold-dist N]\n"
                "         [--min-scaffold-identity I] [--scaffolds-only] [--ani-method M]\n"
                "         [--device D | --devices D0,D1,...] [--threads T] [--quiet]\n"
                "Filter path of pangenome/sweepga on an MI355X (libsweepga_gpu.so).  No CPU fallback.");
      return 0;
    } else if (a.rfind("-", 0) == 0 && a != "-") die(2, "unknown flag " + a);
    else input = a;
  }
  if (input.empty()) die(2, "usage: sweepga-gpu <in.paf> [--output-file out.paf] [filter flags]   (--help)");

  if (!no_filter && !bad_sparsify.empty()) die(1, "--sparsify '" + bad_sparsify + "' is not valid for post-alignment PAF/1aln filtering");
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
  cfg.keep_self = keep_self;
  cfg.scaffolds_only = scaffolds_only;
  // identity thresholds: plain numbers are checked now, "aniN" forms after the ANI pre-pass (main.rs:3571-3595)
  auto lower = [](std::string v) {
    for (auto& c : v)
      if (c >= 'A' && c <= 'Z') c = (char)(c - 'A' + 'a');
    return v;
  };
  const bool need_ani = lower(min_identity).find("ani") != std::string::npos || lower(min_scaffold_identity).find("ani") != std::string::npos;
  auto set_identities = [&](double ani_percentile) {
    if (swg_parse_identity_value(min_identity.c_str(), ani_percentile, &cfg.min_identity) != SWG_OK)
      die(2, std::string("bad --min-aln-identity: ") + swg_paf_last_error());
    if (min_scaffold_identity.empty()) cfg.min_scaffold_identity = cfg.min_identity;
    else if (swg_parse_identity_value(min_scaffold_identity.c_str(), ani_percentile, &cfg.min_scaffold_identity) != SWG_OK)
      die(2, std::string("bad --min-scaffold-identity: ") + swg_paf_last_error());
  };
  if (!need_ani) set_identities(-1.0);

  // ---- HIP start-up (~0.1 s), device memory for an input of this size and the library's code objects (swg_warmup) are
  // prepared on a thread of their own while the host threads read and parse the input
  if (devices.empty()) devices.push_back(device);
  uint64_t records_hint = 0;
  {
    // lines in the file ~ file size / the average line length of its first 256 KB (real PAFs with cg:Z: / tp:A: tags run
    // 150-250 bytes a line, bare ones ~90: a fixed guess over-reserves device memory for the whole run); compressed inputs
    // are not guessed
    struct stat sb;
    const bool gz = input.size() > 3 && (input.rfind(".gz") == input.size() - 3 || input.rfind(".bgz") == input.size() - 4);
    if (input != "-" && !gz && stat(input.c_str(), &sb) == 0 && S_ISREG(sb.st_mode) && sb.st_size > 0) {
      double per_line = 200.0;
      if (FILE* f = fopen(input.c_str(), "rb")) {
        std::vector<char> head(size_t(256) << 10);
        const size_t got = fread(head.data(), 1, head.size(), f);
        fclose(f);
        size_t nl = 0, last = 0;
        for (size_t i = 0; i < got; ++i)
          if (head[i] == '\n') {
            ++nl;
            last = i + 1;
          }
        if (nl) per_line = (double)last / (double)nl;
      }
      records_hint = (uint64_t)((double)sb.st_size / per_line * 1.02) + 1;
    }
  }
  std::vector<swg_ctx*> ctxs;
  int init_rc = SWG_OK;
  std::string init_err;
  double create_ms = 0.0, warm_ms = 0.0;
  std::thread gpu_init([&] {
    if (no_filter) return;
    for (int d : devices) {
      swg_ctx* c = nullptr;
      const auto c0 = std::chrono::steady_clock::now();
      init_rc = swg_create(d, &c);
      create_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - c0).count();
      if (init_rc != SWG_OK) {
        init_err = swg_last_error(nullptr);  // thread-local: read it on this thread
        return;
      }
      ctxs.push_back(c);
      const auto w0 = std::chrono::steady_clock::now();
      // best effort: a failure here (e.g. no room for the speculative reservation) shows up again, with its own message, in
      // the filter call, which sizes itself; SWG_DEBUG prints this one too
      if (swg_warmup(c, records_hint / devices.size(), 4096, cfg.scaffold_gap != 0) != SWG_OK && getenv("SWG_DEBUG"))
        fprintf(stderr, "[sweepga-gpu] warm-up on device %d failed (ignored): %s\n", d, swg_last_error(c));
      warm_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - w0).count();
    }
  });

  // ---- open_paf_input + extract_metadata (paf_filter.rs:292-376), multi-threaded in libsweepga_gpu.so
  using clk = std::chrono::steady_clock;
  const auto t0 = clk::now();
  swg_paf* paf = nullptr;
  // the parse leaves a few hardware threads to the device start-up that runs beside it (the HIP runtime's own threads and
  // the warm-up: with every hardware thread parsing, context creation took 0.4-1.0 s instead of 0.1 s at 10^8 lines)
  // and no more than 64: the parse of 10^8 lines takes 0.62 s on 64 threads of the GPU box's 256-thread host, 0.91 s on 128
  // and 1.29 s on 248 (first-touch page faults of ~7 GB of columns in one address space: more threads get in each other's way)
  int parse_threads = threads;
  if (parse_threads <= 0 && !no_filter) {
    const unsigned hc = std::thread::hardware_concurrency();
    if (hc > 16) parse_threads = (int)std::min(hc - 8, 64u);
  }
  if (swg_paf_open(input.c_str(), parse_threads, &paf) != SWG_OK) {
    const std::string msg = swg_paf_last_error();
    gpu_init.join();
    die(2, msg);
  }
  const std::string out_path = output_file.empty() ? "-" : output_file;
  const swg_records* r = swg_paf_records(paf);
  uint64_t n = r->n;
  if (no_filter) {  // main.rs:3461-3473: every line, newline-normalised, ALWAYS to stdout (--output-file is not consulted)
    const char* text;
    uint64_t len;
    swg_paf_text(paf, &text, &len);
    FILE* out = stdout;
    for (uint64_t pos = 0; pos < len;) {
      const void* nl = std::memchr(text + pos, '\n', len - pos);
      const uint64_t end = nl ? (uint64_t)((const char*)nl - text) : len;
      uint64_t ll = end - pos;
      if (nl && ll && text[pos + ll - 1] == '\r') --ll;
      std::fwrite(text + pos, 1, ll, out);
      std::fputc('\n', out);
      pos = end + 1;
    }
    swg_paf_close(paf);
    gpu_init.join();
    return 0;
  }
  const auto t1 = clk::now();

  // ---- ANI pre-pass over the input when a threshold asks for it
  gpu_init.join();
  const auto t1b = clk::now();  // what the device start-up took beyond the read
  if ((n || need_ani) && init_rc != SWG_OK) die(3, "no usable GPU: " + init_err);
  swg_ctx* ctx = ctxs.empty() ? nullptr : ctxs[0];
  double ani_percentile = -1.0, ani_ms = 0.0;
  if (need_ani) {
    int kind = SWG_ANI_NPERCENTILE, nsort = SWG_NSORT_IDENTITY;
    double pct = 50.0;
    if (!swg_parse_ani_method(ani_method_s.c_str(), &kind, &pct, &nsort)) {  // unknown method: n50-identity
      kind = SWG_ANI_NPERCENTILE;
      pct = 50.0;
      nsort = SWG_NSORT_IDENTITY;
    }
    const auto ta = clk::now();
    if (swg_paf_ani_stats(ctx, paf, kind, pct, nsort, threads, &ani_percentile) != SWG_OK) die(3, std::string("ANI pre-pass failed: ") + swg_paf_last_error());
    ani_ms = std::chrono::duration<double, std::milli>(clk::now() - ta).count();
  }
  if (need_ani) set_identities(ani_percentile);
  if (need_ani && !quiet) std::fprintf(stderr, "[sweepga-gpu] ANI pre-pass (%s): median %.6f in %.1f ms\n", ani_method_s.c_str(), ani_percentile, ani_ms);

  // ---- tree sparsification of the input (src/main.rs:3640-3688): the filter then runs on the surviving lines, whose ranks
  // are their positions in the sparsified text (the reference filters the temporary tree-filtered file)
  if (!tree_sparsify.empty()) {
    const char* text;
    uint64_t len;
    swg_paf_text(paf, &text, &len);
    char* kept_text = nullptr;
    uint64_t kept_len = 0;
    if (swg_paf_tree_filter(text, len, tree_near, tree_far, tree_rand, &kept_text, &kept_len) != SWG_OK) die(2, "tree sparsification failed");
    swg_paf_close(paf);
    paf = nullptr;
    if (swg_paf_open_buffer(kept_text, kept_len, threads, &paf) != SWG_OK) die(2, swg_paf_last_error());
    swg_free(kept_text);
    r = swg_paf_records(paf);
    n = r->n;
  }

  // ---- apply_filters on the GPU
  // result columns: uninitialised storage (the filter writes every entry; zero-filling 0.5 GB on this thread first cost
  // ~85 ms per 10^8 records), 2 MB-aligned and marked for huge pages like the ingest's columns
  auto alloc_col = [&](size_t bytes) -> void* {
    void* p = nullptr;
    const size_t two_mb = size_t(2) << 20;
    if (bytes >= (size_t(8) << 20) && posix_memalign(&p, two_mb, (bytes + two_mb - 1) & ~(two_mb - 1)) == 0)
      madvise(p, bytes, MADV_HUGEPAGE);
    else
      p = std::calloc(bytes ? bytes : 1, 1);
    if (!p) die(3, "out of host memory for the result columns");
    return p;
  };
  struct View8 { uint8_t* p; uint8_t* data() const { return p; } } status{static_cast<uint8_t*>(alloc_col(n ? n : 1))};
  struct View32 { uint32_t* p; uint32_t* data() const { return p; } } chain{static_cast<uint32_t*>(alloc_col((n ? n : 1) * sizeof(uint32_t)))};
  swg_stats st{};
  if (n) {
    // no dv:f: override anywhere: identity = matches / max(block length, 1) for every record, which the device evaluates
    // itself -- the column (8 of 47 bytes per record) stays on the host
    swg_records rr = *r;
    if (swg_paf_identity_is_derived(paf)) rr.identity = nullptr;
    // The result columns are not cleared (posix_memalign above): every filter path must write every entry.  SWG_DEBUG checks
    // that it does: the columns start as 0xff (no status, no chain number of a real result has that byte pattern in every
    // byte: statuses are 0..3, chain numbers stay below 2^31) and none of it may be left.
    const bool poison = std::getenv("SWG_DEBUG") != nullptr;
    if (poison) {
      std::memset(status.data(), 0xff, n);
      std::memset(chain.data(), 0xff, n * sizeof(uint32_t));
    }
    const int rc = ctxs.size() > 1 ? swg_filter_multi(ctxs.data(), (int)ctxs.size(), &rr, &cfg, status.data(), chain.data(), &st)
                                   : swg_filter(ctx, &rr, &cfg, status.data(), chain.data(), &st);
    if (rc != SWG_OK) die(3, std::string("filter failed: ") + swg_last_error(ctx));
    if (poison) {
      for (uint64_t i = 0; i < n; ++i)
        if (status.data()[i] == 0xff || chain.data()[i] == 0xffffffffu)
          die(3, "internal: the filter left record " + std::to_string(i) + " of the result columns unwritten");
    }
  }
  const auto t2 = clk::now();

  // ---- write_filtered_output (paf_filter.rs:1689-1726): input order, original bytes + tags
  uint64_t kept = 0;
  if (swg_paf_write(paf, out_path.c_str(), status.data(), chain.data(), threads, &kept) != SWG_OK) die(2, swg_paf_last_error());
  const auto t3 = clk::now();
  if (!quiet) {
    auto ms = [](clk::time_point a, clk::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
    double load_ms, parse_ms;
    swg_paf_timing(paf, &load_ms, &parse_ms);
    std::fprintf(stderr,
                 "[sweepga-gpu] %llu records -> %llu kept | read %.1f ms (load %.1f, parse %.1f), filter %.1f ms (device %.1f, h2d %.1f, "
                 "d2h %.1f), write %.1f ms | device start-up %.1f ms beside the read (create %.1f, warm-up %.1f), %.1f ms waited for\n",
                 (unsigned long long)n, (unsigned long long)kept, ms(t0, t1), load_ms, parse_ms, ms(t1b, t2), st.device_ms, st.h2d_ms,
                 st.d2h_ms, ms(t2, t3), create_ms + warm_ms, create_ms, warm_ms, ms(t1, t1b));
  }
  if (!quiet) {
    auto ms = [](clk::time_point a, clk::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
    std::fprintf(stderr, "[sweepga-gpu] main() %.1f ms\n", ms(t_main, clk::now()));
  }
  // The output is written and closed.  Unmapping a multi-GB input, freeing the columns and the HIP runtime's
  // exit handlers cost ~0.1 s that buy nothing in a process about to end: leave all of it to the kernel.
  std::fflush(stdout);
  std::fflush(stderr);
  (void)paf;
  (void)ctxs;
  _exit(0);
}
