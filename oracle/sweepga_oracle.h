// TEST INFRASTRUCTURE ONLY -- CPU oracle for the sweepga filter path.
//
// This header belongs to oracle/: a literal CPU restatement of the reference's
// filter (pangenome/sweepga, src/paf_filter.rs, src/plane_sweep_exact.rs,
// src/plane_sweep_scaffold.rs, src/union_find.rs, src/plane_sweep_core.rs).
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use
// it, and only as the checker.  The product (sweepga_amd/, include/) never
// includes, links or calls anything from here.
//
// Parity status: the reference is Rust and cannot be compiled in this image
// (no cargo/rustc), so there is no oracle/_ref.  The restatement is pinned by
// the reference's own known-answer tests (tests/test_oracle_kat.py transcribes
// them with file:line citations).
#pragma once
#include <cstddef>
#include <cstdint>
#include <string>
#include <unordered_map>
#include <vector>

namespace orc {

// src/filter_types.rs:8-14
enum Scoring : int {
  SCORE_IDENTITY = 0,
  SCORE_LENGTH = 1,
  SCORE_LENGTH_IDENTITY = 2,
  SCORE_LOG_LENGTH_IDENTITY = 3,
  SCORE_MATCHES = 4,
};

// src/filter_types.rs:18-22
enum FilterMode : int { ONE_TO_ONE = 0, ONE_TO_MANY = 1, MANY_TO_MANY = 2 };

// src/mapping.rs:82-86
enum ChainStatus : int { ST_SCAFFOLD = 1, ST_RESCUED = 2, ST_UNASSIGNED = 3 };

constexpr uint64_t K_INF = UINT64_MAX;  // usize::MAX

// src/plane_sweep_exact.rs:10-18
struct PlaneSweepMapping {
  size_t idx;
  uint64_t query_start, query_end, target_start, target_end;
  double identity;
  uint8_t flags;  // bit0 discard, bit1 overlapped
};

double score_with_function(const PlaneSweepMapping& m, int scoring);

std::vector<size_t> plane_sweep_query(std::vector<PlaneSweepMapping>& m, uint64_t keep,
                                      double thr, int scoring);
std::vector<size_t> plane_sweep_target(std::vector<PlaneSweepMapping>& m, uint64_t keep,
                                       double thr, int scoring);
std::vector<size_t> plane_sweep_both(std::vector<PlaneSweepMapping>& m, uint64_t qkeep,
                                     uint64_t tkeep, double thr, int scoring);

// src/plane_sweep_scaffold.rs:25-33 (ScaffoldLike)
struct ChainView {
  std::string query_name, target_name;
  uint64_t query_start, query_end, target_start, target_end;
  double identity;
};
// max_per_* : 0 == None
std::vector<size_t> plane_sweep_scaffolds(const std::vector<ChainView>& chains, int mode,
                                          uint64_t max_per_query, uint64_t max_per_target,
                                          double thr, int scoring);

// src/union_find.rs
struct UnionFind {
  std::vector<size_t> parent, rank;
  explicit UnionFind(size_t n);
  size_t find(size_t x);
  void unite(size_t x, size_t y);
  std::vector<std::vector<size_t>> get_sets();
};

// src/plane_sweep_core.rs:8-14
struct Interval {
  size_t idx;
  uint32_t begin, end;
  double score;
  uint32_t flags;
};
std::vector<size_t> plane_sweep_core(std::vector<Interval>& iv, uint64_t max_to_keep, double thr);

// src/paf_filter.rs:20-49 (only the fields the filter reads) + keep_self/scaffolds_only
struct FilterConfig {
  uint64_t min_block_length = 0;
  int mapping_filter_mode = MANY_TO_MANY;
  uint64_t mapping_max_per_query = 0;   // 0 == None
  uint64_t mapping_max_per_target = 0;  // 0 == None
  int scaffold_filter_mode = MANY_TO_MANY;
  uint64_t scaffold_max_per_query = 0;
  uint64_t scaffold_max_per_target = 0;
  double overlap_threshold = 0.95;
  uint64_t scaffold_gap = 50000;
  uint64_t min_scaffold_length = 10000;
  double scaffold_overlap_threshold = 0.5;
  uint64_t scaffold_max_deviation = 0;
  int scoring_function = SCORE_LOG_LENGTH_IDENTITY;
  double min_identity = 0.0;
  double min_scaffold_identity = 0.0;
  bool keep_self = false;
  bool scaffolds_only = false;
};

// src/paf_filter.rs:54-71
struct RecordMeta {
  size_t rank = 0;
  std::string query_name, target_name;
  uint64_t query_start = 0, query_end = 0, target_start = 0, target_end = 0;
  uint64_t block_length = 0;
  double identity = 0.0;
  uint64_t matches = 0;
  uint64_t alignment_length = 0;
  char strand = '+';
  bool has_chain_id = false;
  std::string chain_id;
  int chain_status = ST_UNASSIGNED;
};

// A decoded .1aln alignment as fastga-rs' AlnReader::read_alignment hands it to extract_1aln_metadata
// (src/unified_filter.rs:67-82: names already looked up in id_to_name, or the raw field when the id is unknown).
// The decoder itself (fastga-rs 0.1.2 / onecode 0.1.0) is an un-vendored dependency: parity of DECODING is unpinned.
struct AlnRecord {
  std::string query_name, target_name;  // full FASTA headers
  uint64_t query_start = 0, query_end = 0, target_start = 0, target_end = 0;
  uint64_t matches = 0;
  char strand = '+';
};
// src/unified_filter.rs:83-142: the RecordMeta of every alignment (rank = position)
std::vector<RecordMeta> records_from_1aln(const std::vector<AlnRecord>& alns);
// str::split_whitespace().next().unwrap_or(full) (src/unified_filter.rs:83-92)
std::string first_word_or_all(const std::string& s);

// ---- tree sparsification of a PAF (src/tree_filter.rs:13-285), run by the reference BEFORE the filter when
// --sparsify tree:<near>[:<far>[:<random>]] is given (src/main.rs:3640-3688).  Returns the kept lines (input order,
// each without its line terminator).  The reference breaks identity ties among a genome's neighbours by HashMap
// iteration order (arbitrary); here ties fall to the neighbour's genome prefix in ascending byte order -- one admissible
// instance.  Parity unpinned: the reference holds no test vector for this pass beyond extract_genome_prefix.
std::vector<std::string> tree_filter_paf_lines(const std::vector<std::string>& lines, size_t k_nearest, size_t k_farthest,
                                               double random_fraction);
// std::collections::hash_map::DefaultHasher (SipHash-1-3, keys 0/0) over `a.hash(); b.hash()` of two strs
// (src/tree_filter.rs:142-147)
uint64_t default_hash_str_pair(const std::string& a, const std::string& b);

// src/paf_filter.rs:142-155
struct MergedChain {
  std::string query_name, target_name;
  uint64_t query_start, query_end, target_start, target_end;
  char strand;
  uint64_t total_length;
  double weighted_identity;
  uint64_t sum_matches, sum_block_lengths;
  std::vector<size_t> member_indices;  // ranks
};

extern bool g_fast_inversion;  // see sweepga_oracle.cpp (step 4b through an index; full-size checks only)

struct PafFilter {
  FilterConfig config;
  explicit PafFilter(const FilterConfig& c) : config(c) {}
  // src/paf_filter.rs:292-376
  std::vector<RecordMeta> extract_metadata(const std::string& path) const;
  static bool parse_paf_line(const std::string& line, size_t rank, RecordMeta* out);
  // src/paf_filter.rs:379-747 ; result keyed by rank
  std::unordered_map<size_t, RecordMeta> apply_filters(std::vector<RecordMeta> metadata) const;
  // src/paf_filter.rs:750-933
  std::vector<MergedChain> merge_mappings_into_chains(const std::vector<RecordMeta>& md,
                                                      uint64_t max_gap) const;
  // src/paf_filter.rs:972-1123
  std::vector<RecordMeta> apply_plane_sweep_to_mappings(const std::vector<RecordMeta>& m) const;
  // src/paf_filter.rs:1126-1146
  std::vector<MergedChain> apply_scaffold_plane_sweep(std::vector<MergedChain> chains) const;
  // src/paf_filter.rs:1689-1726
  void write_filtered_output(const std::string& in, const std::string& out,
                             const std::unordered_map<size_t, RecordMeta>& passing) const;
  // src/paf_filter.rs:278-289
  void filter_paf(const std::string& in, const std::string& out) const;
};

// src/main.rs:244-293 ; returns false where the reference calls process::exit(1) (bare "0")
bool parse_filter_mode(const std::string& mode, int* fmode, uint64_t* per_query,
                       uint64_t* per_target);
// src/cli.rs:26-61
bool parse_metric_number(const std::string& s, uint64_t* out);
// src/cli.rs:76-130.  `ani_percentile` < 0 means None ("aniN" forms then fail, as in the reference).
bool parse_identity_value(const std::string& s, double* out, double ani_percentile = -1.0);

// ---- ANI pre-pass (src/main.rs:172-188, 296-688) ----
enum AniMethodKind { ANI_ALL = 0, ANI_ORTHOGONAL = 1, ANI_NPERCENTILE = 2 };
enum NSort { NSORT_LENGTH = 0, NSORT_IDENTITY = 1, NSORT_SCORE = 2 };
struct AniMethod {
  int kind = ANI_NPERCENTILE;
  double percentile = 50.0;
  int sort = NSORT_IDENTITY;
};
// src/main.rs:296-330 ; false = None
bool parse_ani_method(const std::string& s, AniMethod* out);
// src/main.rs:334-498 (All / Orthogonal) and 500-688 (N-percentile).  Returns the median per-genome-pair ANI.
double calculate_ani_stats(const std::string& input_path, const AniMethod& method);
// src/main.rs:3485-3492
int parse_scoring(const std::string& s);
// src/pansn.rs:176-225
uint64_t round_nice(uint64_t v);
void clamp_scaffold_params(uint64_t user_jump, uint64_t user_mass, bool have_avg, uint64_t avg,
                           bool adaptive, uint64_t* jump, uint64_t* mass);
// src/paf.rs:32-64 ; returns false on a number parse error
bool parse_cigar_counts(const std::string& cigar, uint64_t* m, uint64_t* x, uint64_t* i,
                        uint64_t* d);

}  // namespace orc
