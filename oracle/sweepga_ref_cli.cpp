// TEST INFRASTRUCTURE ONLY -- `sweepga-ref`: the CPU oracle behind the reference's own
// filter-path command line (`sweepga <paf> --output-file out.paf ...`), so that the inline
// PAFs of the reference's binary-invoking tests can be replayed.  Flag names and defaults
// follow src/cli.rs:204-288; flag -> FilterConfig mapping follows src/main.rs:3477-3568,
// 3590-3619, 3689-3691.
#include <vector>
#include <unistd.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <string>

#include "sweepga_oracle.h"

using namespace orc;

static void die(const std::string& msg) {
  std::fprintf(stderr, "sweepga-ref: %s\n", msg.c_str());
  std::exit(2);
}

static // --sparsify (src/knn_graph.rs:59-160, src/main.rs:3494-3509).  `none`, `all`, a bare fraction and `random:<f>` have no
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

int main(int argc, char** argv) {
  std::string input, output_file;
  std::string num_mappings = "many:many", scoring = "log-length-ani", min_identity = "0";
  std::string scaffold_filter = "many:many", min_scaffold_identity = "0", ani_method_s = "n100";
  double overlap = 0.95, scaffold_overlap = 0.5;
  uint64_t scaffold_jump = 50000, scaffold_mass = 10000, scaffold_dist = 0;
  bool have_block_length = false;
  uint64_t block_length = 0;
  bool keep_self = false, no_filter = false, scaffolds_only = false;
  std::string bad_sparsify, tree_sparsify;
  unsigned long tree_near = 0, tree_far = 0;
  double tree_rand = 0.0;

  auto need = [&](int& i) -> std::string {
    if (i + 1 >= argc) die(std::string("missing value for ") + argv[i]);
    return argv[++i];
  };
  for (int i = 1; i < argc; ++i) {
    std::string a = argv[i];
    std::string val;
    size_t eq = a.find('=');
    bool has_eq = a.rfind("--", 0) == 0 && eq != std::string::npos;
    if (has_eq) {
      val = a.substr(eq + 1);
      a = a.substr(0, eq);
    }
    auto value = [&]() { return has_eq ? val : need(i); };
    if (a == "--output-file" || a == "-o") output_file = value();
    else if (a == "--num-mappings") num_mappings = value();
    else if (a == "--overlap") overlap = std::strtod(value().c_str(), nullptr);
    else if (a == "--scoring") scoring = value();
    else if (a == "--min-aln-identity") min_identity = value();
    else if (a == "--min-aln-length") {
      if (!parse_metric_number(value(), &block_length)) die("bad --min-aln-length");
      have_block_length = true;
    } else if (a == "--self") keep_self = true;
    else if (a == "--no-filter") no_filter = true;
    else if (a == "--scaffold-jump") {
      if (!parse_metric_number(value(), &scaffold_jump)) die("bad --scaffold-jump");
    } else if (a == "--scaffold-mass") {
      if (!parse_metric_number(value(), &scaffold_mass)) die("bad --scaffold-mass");
    } else if (a == "--scaffold-filter") scaffold_filter = value();
    else if (a == "--scaffold-overlap") scaffold_overlap = std::strtod(value().c_str(), nullptr);
    else if (a == "--scaffold-dist") {
      if (!parse_metric_number(value(), &scaffold_dist)) die("bad --scaffold-dist");
    } else if (a == "--min-scaffold-identity") min_scaffold_identity = value();
    else if (a == "--scaffolds-only") scaffolds_only = true;
    else if (a == "--ani-method") ani_method_s = value();
    else if (a == "--sparsify") {
      const std::string v = value();
      const int rc = check_sparsify(v, &tree_near, &tree_far, &tree_rand);
      if (rc == 2) die("invalid value for --sparsify");
      if (rc == 1) bad_sparsify = v;  // reported after the --no-filter shortcut, as in main.rs:3461-3509
      tree_sparsify = rc == 3 ? v : std::string();
    }
    else if (a == "--no-adaptive-scaffolds" || a == "--quiet" || a == "--paf") { /* no effect here */ }
    else if (a == "--threads" || a == "-t") (void)value();
    else if (a.rfind("-", 0) == 0 && a != "-") die("unknown flag " + a);
    else input = a;
  }
  if (input.empty()) die("usage: sweepga-ref <in.paf> [--output-file out.paf] [filter flags]");

  std::string out_path = output_file.empty() ? "/dev/stdout" : output_file;
  if (no_filter) {  // main.rs:3461-3473: always to stdout, --output-file is not consulted
    std::ifstream in(input, std::ios::binary);
    std::ofstream out("/dev/stdout", std::ios::binary);
    std::string line;
    while (std::getline(in, line)) {
      if (!in.eof() && !line.empty() && line.back() == '\r') line.pop_back();
      out << line << "\n";
    }
    return 0;
  }

  if (!bad_sparsify.empty()) {
    std::fprintf(stderr, "sweepga-ref: --sparsify '%s' is not valid for post-alignment PAF/1aln filtering\n", bad_sparsify.c_str());
    return 1;
  }
  FilterConfig cfg;
  int mode;
  uint64_t pq, pt;
  if (!parse_filter_mode(num_mappings, &mode, &pq, &pt)) return 1;
  cfg.mapping_filter_mode = mode;
  cfg.mapping_max_per_query = pq;
  cfg.mapping_max_per_target = pt;
  if (!parse_filter_mode(scaffold_filter, &mode, &pq, &pt)) return 1;
  cfg.scaffold_filter_mode = mode;
  cfg.scaffold_max_per_query = pq;
  cfg.scaffold_max_per_target = pt;
  cfg.scoring_function = parse_scoring(scoring);
  // clamp_scaffold_params is a no-op for PAF input (no .fai for a PAF): main.rs:3515-3527
  cfg.min_block_length = have_block_length ? block_length : 0;
  cfg.overlap_threshold = overlap;
  cfg.scaffold_gap = scaffold_jump;
  cfg.min_scaffold_length = scaffold_mass;
  cfg.scaffold_overlap_threshold = scaffold_overlap;
  cfg.scaffold_max_deviation = scaffold_dist;
  // main.rs:3571-3595: ANI pre-pass only when a threshold mentions "ani"
  AniMethod ani_method;
  if (!parse_ani_method(ani_method_s, &ani_method)) {
    ani_method.kind = ANI_NPERCENTILE;
    ani_method.percentile = 50.0;
    ani_method.sort = NSORT_IDENTITY;
  }
  auto lower = [](std::string v) {
    for (auto& c : v)
      if (c >= 'A' && c <= 'Z') c = (char)(c - 'A' + 'a');
    return v;
  };
  double ani_percentile = -1.0;
  try {
    if (lower(min_identity).find("ani") != std::string::npos || lower(min_scaffold_identity).find("ani") != std::string::npos)
      ani_percentile = calculate_ani_stats(input, ani_method);
  } catch (const std::exception& e) {
    die(e.what());
  }
  if (!parse_identity_value(min_identity, &cfg.min_identity, ani_percentile)) die("bad --min-aln-identity");
  if (min_scaffold_identity.empty())
    cfg.min_scaffold_identity = cfg.min_identity;
  else if (!parse_identity_value(min_scaffold_identity, &cfg.min_scaffold_identity, ani_percentile))
    die("bad --min-scaffold-identity");
  cfg.keep_self = keep_self;  // || no_filter, handled above
  cfg.scaffolds_only = scaffolds_only;

  try {
    std::string filter_input = input;
    std::string tree_tmp;
    if (!tree_sparsify.empty()) {  // main.rs:3640-3688: the filter runs on the tree-filtered temporary file
      std::vector<std::string> lines;
      {
        std::ifstream in(input, std::ios::binary);
        if (!in) die("cannot open " + input);
        std::string line;
        while (std::getline(in, line)) {
          if (!in.eof() && !line.empty() && line.back() == '\r') line.pop_back();
          lines.push_back(line);
        }
      }
      const std::vector<std::string> kept = tree_filter_paf_lines(lines, tree_near, tree_far, tree_rand);
      tree_tmp = (output_file.empty() ? std::string("/tmp/sweepga_ref") : output_file) + ".tree." + std::to_string((long)getpid()) + ".paf";
      std::ofstream out(tree_tmp, std::ios::binary);
      for (const std::string& l : kept) out << l << "\n";
      out.close();
      filter_input = tree_tmp;
    }
    PafFilter(cfg).filter_paf(filter_input, out_path);
    if (!tree_tmp.empty()) std::remove(tree_tmp.c_str());
  } catch (const std::exception& e) {
    die(e.what());
  }
  return 0;
}
