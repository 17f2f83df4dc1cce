/* sweepga_gpu.h -- C ABI of the MI355X plane-sweep / scaffold filter.
 *
 * Drop-in boundary for sweepga's filter path (`sweepga <paf> --output-file ...`).  The
 * reference has no FFI layer; the seam this library replaces is
 *
 *     PafFilter::apply_filters(&self, Vec<RecordMeta>) -> Result<HashMap<usize, RecordMeta>>
 *                                                         (src/paf_filter.rs:379-382)
 *
 * called from PafFilter::filter_paf (src/paf_filter.rs:283), unified_filter::filter_file
 * (src/unified_filter.rs:316) and examples/compare_filter_outcomes.rs:62-63.  The host keeps
 * CLI parsing, PAF/.1aln I/O and name -> id interning; everything between "records parsed" and
 * "per-record status + chain id" runs in hand-written gfx950 kernels.
 *
 * Conventions
 *   - plain C types only; the caller owns every buffer it passes; the library owns device
 *     memory, streams and staging inside swg_ctx;
 *   - every call returns SWG_OK (0) or a negative SWG_ERR_* code; swg_last_error() gives text;
 *   - no exceptions, aborts or allocations cross the boundary;
 *   - one swg_ctx is used by one host thread at a time; contexts are independent (one per GPU);
 *   - there is NO CPU fallback: without a usable HIP device every compute call fails with
 *     SWG_ERR_NO_DEVICE.
 */
#ifndef SWEEPGA_GPU_H
#define SWEEPGA_GPU_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SWG_ABI_VERSION 1

/* error codes */
#define SWG_OK 0
#define SWG_ERR_INVALID (-1)     /* bad argument */
#define SWG_ERR_NO_DEVICE (-2)   /* no HIP device / device init failed */
#define SWG_ERR_HIP (-3)         /* a HIP runtime call or kernel failed */
#define SWG_ERR_OOM (-4)         /* device or host allocation failed */
#define SWG_ERR_RANGE (-5)       /* mapped stretch of a sequence >= 2^32 bases, or composite sort key wider than 64 bits */
#define SWG_ERR_UNSUPPORTED (-6) /* valid in the reference, not implemented here (documented) */

/* ScoringFunction, src/filter_types.rs:8-14 */
#define SWG_SCORE_IDENTITY 0
#define SWG_SCORE_LENGTH 1
#define SWG_SCORE_LENGTH_IDENTITY 2
#define SWG_SCORE_LOG_LENGTH_IDENTITY 3
#define SWG_SCORE_MATCHES 4

/* FilterMode, src/filter_types.rs:18-22 */
#define SWG_MODE_ONE_TO_ONE 0
#define SWG_MODE_ONE_TO_MANY 1
#define SWG_MODE_MANY_TO_MANY 2

/* per-record result, ChainStatus of src/mapping.rs:82-86 plus "dropped" */
#define SWG_ST_DROPPED 0
#define SWG_ST_SCAFFOLD 1   /* st:Z:scaffold   */
#define SWG_ST_RESCUED 2    /* st:Z:rescued    */
#define SWG_ST_UNASSIGNED 3 /* st:Z:unassigned */

/* usize::MAX of the reference ("keep all") for the k arguments of the sweep entry points */
#define SWG_K_INF UINT64_MAX

typedef struct swg_ctx swg_ctx;

/* The fields of FilterConfig (src/paf_filter.rs:20-49) that the filter reads, plus the two
 * PafFilter builder switches (src/paf_filter.rs:267-275).  Limits: 0 means None. */
typedef struct swg_config {
  uint64_t min_block_length;         /* --min-aln-length        */
  int32_t mapping_filter_mode;       /* --num-mappings (mode)   */
  uint64_t mapping_max_per_query;    /*   per-query limit, 0 = None  */
  uint64_t mapping_max_per_target;   /*   per-target limit, 0 = None */
  int32_t scaffold_filter_mode;      /* --scaffold-filter       */
  uint64_t scaffold_max_per_query;
  uint64_t scaffold_max_per_target;
  double overlap_threshold;          /* --overlap               */
  uint64_t scaffold_gap;             /* --scaffold-jump; 0 disables scaffolding */
  uint64_t min_scaffold_length;      /* --scaffold-mass         */
  double scaffold_overlap_threshold; /* --scaffold-overlap      */
  uint64_t scaffold_max_deviation;   /* --scaffold-dist         */
  int32_t scoring_function;          /* --scoring               */
  double min_identity;               /* --min-aln-identity      */
  double min_scaffold_identity;      /* --min-scaffold-identity */
  int32_t keep_self;                 /* --self                  */
  int32_t scaffolds_only;            /* --scaffolds-only        */
} swg_config;

/* Column (SoA) form of Vec<RecordMeta> (src/paf_filter.rs:54-71), one entry per parsed PAF /
 * .1aln record in input (rank) order.  Sequence names are interned by the host into one id
 * space shared by queries and targets (the reference's SequenceIndex, src/sequence_index.rs:7-31);
 * the two per-sequence tables give each sequence its genome under the two prefix rules the
 * reference uses.  Pointers are host pointers for swg_filter and device pointers for
 * swg_filter_device. */
typedef struct swg_records {
  uint64_t n;                /* number of records */
  const uint32_t* q_id;      /* [n] sequence id of the query name  */
  const uint32_t* t_id;      /* [n] sequence id of the target name */
  const uint32_t* q_start;   /* [n] */
  const uint32_t* q_end;     /* [n] */
  const uint32_t* t_start;   /* [n] */
  const uint32_t* t_end;     /* [n] */
  const double* identity;    /* [n] RecordMeta.identity; may be NULL = matches / max(block_len, 1) for every record, which
                                is what extract_metadata computes when no dv:f: tag overrides it (src/paf_filter.rs:322):
                                evaluated on the device (one IEEE division, same bits), 8 of 47 bytes per record less over PCIe */
  const uint32_t* matches;   /* [n] RecordMeta.matches */
  const uint32_t* block_len; /* [n] RecordMeta.block_length */
  const uint8_t* strand;     /* [n] 0 = '+', 1 = '-' */
  uint32_t n_seq;            /* number of sequence ids */
  const uint32_t* seq_genome_last; /* [n_seq] genome id, prefix = up to the LAST '#'
                                      (src/paf_filter.rs:1022-1030) */
  uint32_t n_genome_last;
  const uint32_t* seq_genome_two;  /* [n_seq] genome id, prefix = first two '#' parts
                                      (src/plane_sweep_scaffold.rs:13-22) */
  uint32_t n_genome_two;
} swg_records;

/* Per-call timing/statistics (optional, may be NULL). */
typedef struct swg_stats {
  uint64_t n_in;          /* records in */
  uint64_t n_retained;    /* after the step-1 retain (src/paf_filter.rs:384-388) */
  uint64_t n_swept;       /* after the mapping plane sweep */
  uint64_t n_chains;      /* chains built */
  uint64_t n_chains_kept; /* chains after span/identity filter and scaffold sweep */
  uint64_t n_out;         /* records kept */
  double device_ms;       /* GPU time of the call measured with HIP events on the ctx stream */
  double h2d_ms, d2h_ms;  /* copies (swg_filter only) */
} swg_stats;

/* ---- context ------------------------------------------------------------------------- */
int swg_abi_version(void);
/* device: HIP device ordinal.  On failure *out is NULL and the code says why. */
int swg_create(int device, swg_ctx** out);
void swg_destroy(swg_ctx* ctx);
/* Text of the last error on this context (or of the last failed swg_create if ctx is NULL). */
const char* swg_last_error(const swg_ctx* ctx);
/* The HIP stream (hipStream_t) all work of this context is enqueued on. */
void* swg_stream(swg_ctx* ctx);
int swg_synchronize(swg_ctx* ctx);

/* ---- the filter: PafFilter::apply_filters (src/paf_filter.rs:379-747) ----------------- */
/* status_out[n]: SWG_ST_*; chain_out[n]: N of "ch:Z:chain_N", 0 = no ch:Z: tag.
 * Host buffers in, host buffers out (copies included).  When the records are grouped by query genome (an aligner writes its
 * PAF query by query) and there are millions of them, the call is streamed: ranges of whole query genomes -- closed under
 * genome pairs, the filter's independent units -- are uploaded while their predecessors are filtered, and each range's
 * results are copied back as soon as it is done (SWG_STREAM=0 switches that off; results are identical either way). */
int swg_filter(swg_ctx* ctx, const swg_records* rec, const swg_config* cfg, uint8_t* status_out,
               uint32_t* chain_out, swg_stats* stats);
/* The ranges such a call would use (host code, no GPU): bounds_out[0 .. *n_chunks] are record indices, ranges of at least
 * target_records records cut where the query genome changes.  *n_chunks = 0: the records are not grouped by query genome
 * (or the reference's two genome-prefix rules partition the sequences differently) -- the call runs in one piece. */
int swg_stream_plan(const swg_records* rec, uint64_t target_records, uint64_t* bounds_out, uint64_t bounds_capacity,
                    uint64_t* n_chunks);
/* Same with every pointer of `rec`, status_out and chain_out in device memory of ctx's GPU.
 * Asynchronous on swg_stream(ctx) except for the small read-backs the pipeline needs. */
int swg_filter_device(swg_ctx* ctx, const swg_records* rec, const swg_config* cfg,
                      uint8_t* status_out, uint32_t* chain_out, swg_stats* stats);

/* Vec<RecordMeta> with the reference's own field widths (query_start .. block_length are u64, src/paf_filter.rs:58-62).
 * The device layout stays 32-bit: every coordinate is rebased to the smallest coordinate its sequence has anywhere in the
 * record set (query and target appearances alike).  apply_filters only ever uses differences, orders and midpoints of
 * positions on one sequence, so the results are those of the unrebased records; what has to fit 32 bits is the stretch of
 * each sequence that mappings touch (and every matches / block_length value).  A sequence touched over 2^32 bases or more is
 * rebased per sweep segment instead -- query coordinates to the smallest one of their (query sequence, genome of the target)
 * segment, target coordinates likewise (src/paf_filter.rs:1037-1100: no step of apply_filters compares coordinates across
 * those segments) -- so that only the stretch touched by the mappings against ONE genome has to fit; beyond that
 * SWG_ERR_RANGE.  start <= end is assumed, as in any PAF.  swg_filter64: host pointers, rebased by host threads, then swg_filter (no extra PCIe bytes);
 * swg_filter_device64: device pointers, rebased by two kernels, then the same pipeline as swg_filter_device. */
typedef struct swg_records64 {
  uint64_t n;
  const uint32_t* q_id;
  const uint32_t* t_id;
  const uint64_t* q_start;
  const uint64_t* q_end;
  const uint64_t* t_start;
  const uint64_t* t_end;
  const double* identity;
  const uint64_t* matches;
  const uint64_t* block_len;
  const uint8_t* strand;
  uint32_t n_seq;
  const uint32_t* seq_genome_last;
  uint32_t n_genome_last;
  const uint32_t* seq_genome_two;
  uint32_t n_genome_two;
} swg_records64;
int swg_filter64(swg_ctx* ctx, const swg_records64* rec, const swg_config* cfg, uint8_t* status_out, uint32_t* chain_out,
                 swg_stats* stats);
int swg_filter_device64(swg_ctx* ctx, const swg_records64* rec, const swg_config* cfg, uint8_t* status_out,
                        uint32_t* chain_out, swg_stats* stats);

/* ---- lower public seams of the reference, exercised by its tests ---------------------- */
/* plane_sweep_query / plane_sweep_target / plane_sweep_both (src/plane_sweep_exact.rs:268,
 * 355, 436) on ONE segment of n mappings given as host arrays.  axis: 0 query, 1 target,
 * 2 both.  keep_out[i] = 1 iff index i is in the returned Vec<usize>.  u64 coordinates as in
 * PlaneSweepMapping: values >= 2^32 are handled by shrinking the stretches no interval covers (exact for a
 * sweep; the reference's u64::MAX test runs this way); SWG_ERR_RANGE only if the covered span of an axis
 * itself does not fit 32 bits.  The same holds for swg_plane_sweep_scaffolds. */
int swg_plane_sweep(swg_ctx* ctx, int axis, uint64_t n, const uint64_t* q_start,
                    const uint64_t* q_end, const uint64_t* t_start, const uint64_t* t_end,
                    const double* identity, uint64_t k_query, uint64_t k_target,
                    double overlap_threshold, int scoring, uint8_t* keep_out);

/* plane_sweep_scaffolds (src/plane_sweep_scaffold.rs:47-94) on n chains; q_id/t_id are
 * sequence ids, seq_genome_two as in swg_records.  order_out receives the kept indices in the
 * reference's output order, *n_kept their number. */
int swg_plane_sweep_scaffolds(swg_ctx* ctx, uint64_t n, const uint32_t* q_id, const uint32_t* t_id,
                              uint32_t n_seq, const uint32_t* seq_genome_two, uint32_t n_genome_two,
                              const uint64_t* q_start, const uint64_t* q_end,
                              const uint64_t* t_start, const uint64_t* t_end, const double* identity,
                              int mode, uint64_t max_per_query, uint64_t max_per_target,
                              double overlap_threshold, int scoring, uint64_t* order_out,
                              uint64_t* n_kept);

/* merge_mappings_into_chains (src/paf_filter.rs:750-933) on n records (no retain, no sweep).
 * chain_of[i] = index of record i's chain in the reference's all_chains order; per-chain
 * outputs hold *n_chains entries (buffers sized n). */
int swg_merge_chains(swg_ctx* ctx, const swg_records* rec, uint64_t max_gap, uint32_t* chain_of,
                     uint32_t* c_q_start, uint32_t* c_q_end, uint32_t* c_t_start,
                     uint32_t* c_t_end, double* c_weighted_identity, uint64_t* n_chains);

/* UnionFind::get_sets (src/union_find.rs:52-63) after union(xs[e], ys[e]) for e = 0..m-1:
 * set_of[i] = position of i's set in get_sets() order (ascending smallest... see DESIGN.md). */
int swg_union_find_sets(swg_ctx* ctx, uint64_t n, uint64_t m, const uint32_t* xs,
                        const uint32_t* ys, uint32_t* set_of, uint64_t* n_sets);

/* f64::ln as the reference evaluates it (glibc log) for n host doubles, computed on the GPU. */
int swg_log(swg_ctx* ctx, uint64_t n, const double* x, double* y);
/* ln(first + i * stride) for i in [0, n), compared on the device against nothing: returns the
 * values so a test can compare them with the host libm. */
int swg_log_range(swg_ctx* ctx, uint64_t first, uint64_t stride, uint64_t n, double* y);

/* ---- per-kernel timing (HIP events on swg_stream) ---------------------------------------- */
/* When enabled, every kernel launch of later calls is bracketed by HIP events on the context's
 * stream and its elapsed time accumulated per kernel name.  Costs two event records per launch. */
int swg_profile_enable(swg_ctx* ctx, int on);
/* Restricts the bracketing to the launches of ONE kernel (its name as swg_profile_get reports it; NULL or "" = every
 * launch again): a timed region can then carry HIP events around the kernel it wants the duration of -- two event records
 * per launch of that kernel -- without the ~200 event records per call that bracketing every launch costs (about 1 ms per
 * call on a pipeline of ~100 launches). */
int swg_profile_select(swg_ctx* ctx, const char* kernel_name);
int swg_profile_reset(swg_ctx* ctx);
/* Number of distinct kernel names seen since the last reset. */
int swg_profile_count(swg_ctx* ctx);
/* Entry i: name (owned by ctx, valid until the next reset), launches, summed milliseconds. */
int swg_profile_get(swg_ctx* ctx, int i, const char** name, uint64_t* launches, double* total_ms);
/* Elements worked on, summed over entry i's launches, for kernels that run on sub-problems of the call (the radix sort
 * passes: pairs sorted); 0 for kernels that always run over the call's whole record set. */
int swg_profile_units(swg_ctx* ctx, int i, uint64_t* units);

/* Device scratch of this context: capacity of the arena (kept between calls, grown on demand) and the high-water
 * mark of the last call.  A call whose scratch does not fit grows the arena and runs once more, so a host that
 * knows its sizes can avoid that by one warm-up call or by swg_reserve(). */
int swg_memory_info(const swg_ctx* ctx, uint64_t* arena_capacity, uint64_t* arena_peak_last_call);
int swg_reserve(swg_ctx* ctx, uint64_t arena_bytes);
/* What a context's first swg_filter call would otherwise pay for inside the call (the reference has no counterpart: it has
 * no device): scratch and staging memory for about n_records_hint records (0 = none) and the library's code objects on the
 * device (one small built-in filter call).  Meant to run on its own host thread while the input is read and parsed. */
int swg_warmup(swg_ctx* ctx, uint64_t n_records_hint, uint32_t n_seq_hint, int with_scaffold);

/* ---- several devices of one node ---------------------------------------------------------------
 * swg_filter over n_ctx contexts (one per device, created by the caller).  Records grouped by query genome: ranges of whole
 * query genomes are dealt to the contexts by size (longest first), every context streams its ranges as swg_filter does --
 * slices of the caller's columns in, slices of the caller's result arrays out, no host-side copy of the record set.
 * Otherwise: records are partitioned by genome pair (first-two-'#'-parts prefix) on host threads, pairs are bin-packed
 * onto the contexts by mapping count, every context filters its part on its own host thread.  Either way chain numbers are
 * made global again on the host (the reference numbers kept chains genome pair by genome pair in first-appearance
 * order, src/paf_filter.rs:517-521).  No collective.  Falls
 * back to ctxs[0] alone when the two genome-prefix rules of the reference partition the sequences differently.
 * Results are identical to swg_filter(ctxs[0], ...).  Errors are reported on ctxs[0]. */
int swg_filter_multi(swg_ctx* const* ctxs, int n_ctx, const swg_records* records, const swg_config* cfg, uint8_t* status_out,
                     uint32_t* chain_out, swg_stats* stats);
int swg_filter_multi64(swg_ctx* const* ctxs, int n_ctx, const swg_records64* records, const swg_config* cfg, uint8_t* status_out,
                       uint32_t* chain_out, swg_stats* stats); /* swg_filter64's rebasing, then the same */

/* ---- PAF ingest / egress (host side; no GPU needed for open/write) ----------------------------------------
 * The reference reads the PAF twice (extract_metadata, then write_filtered_output re-reads it).  A swg_paf
 * handle keeps the mapped text, the SoA columns swg_filter() takes and each record's (offset,length), so the
 * writer does not parse again.  `threads` <= 0 means "all host cores".  Errors: negative code, message from
 * swg_paf_last_error() (thread-local).
 */
typedef struct swg_paf swg_paf;
/* open_paf_input (src/paf.rs:10-30: .gz/.bgz -> BGZF, blocks inflated in parallel; "-" = stdin) +
 * PafFilter::extract_metadata (src/paf_filter.rs:292-376) + SequenceIndex (src/sequence_index.rs:7-31). */
int swg_paf_open(const char* path, int threads, swg_paf** out);
/* same, over PAF text already in memory (copied) */
int swg_paf_open_buffer(const char* text, uint64_t len, int threads, swg_paf** out);
void swg_paf_close(swg_paf* p);
/* records in input order; pointers are owned by the handle.  A file with a coordinate, matches or block length >= 2^32
 * is parsed once more into 64-bit columns and rebased per sequence as swg_filter64 does: the coordinate columns are
 * then relative to swg_paf_seq_offsets()[sequence id] (NULL for a file that needed no rebasing -- and for one with a sequence
 * touched over 2^32 bases or more, whose columns are relative to one constant per sweep segment, see swg_records64). */
const swg_records* swg_paf_records(const swg_paf* p);
/* 1 when every record's identity is matches / max(block_len, 1) -- no dv:f: tag had the last word on any line: a caller may
 * then pass identity = NULL in the records it hands to the filter and save the column's trip over PCIe. */
int swg_paf_identity_is_derived(const swg_paf* p);
const uint64_t* swg_paf_seq_offsets(const swg_paf* p);
/* ... a file with a sequence touched over 2^32 bases or more: [n] what was taken off every RECORD's query (axis 0) or target
 * (axis 1) coordinates (one constant per sweep segment, see swg_records64); NULL otherwise. */
const uint64_t* swg_paf_record_offsets(const swg_paf* p, int axis);
/* physical line count (incl. skipped lines) and each record's rank = 0-based line index (src/paf_filter.rs:298) */
uint64_t swg_paf_num_lines(const swg_paf* p);
const uint64_t* swg_paf_ranks(const swg_paf* p);
uint32_t swg_paf_num_sequences(const swg_paf* p);
const char* swg_paf_sequence_name(const swg_paf* p, uint32_t id);
void swg_paf_timing(const swg_paf* p, double* load_ms, double* parse_ms);
int swg_paf_text(const swg_paf* p, const char** text, uint64_t* len);
/* PafFilter::write_filtered_output (src/paf_filter.rs:1689-1726): records with status != 0, input order,
 * original bytes + "\tch:Z:chain_<N>" (chain != 0) + "\tst:Z:<status>".  out_path "-" = stdout. */
int swg_paf_write(const swg_paf* p, const char* out_path, const uint8_t* status, const uint32_t* chain, int threads,
                  uint64_t* n_written);
/* PafFilter::filter_paf (src/paf_filter.rs:278-289): open -> swg_filter -> write.
 * timing_ms (optional) = {load, parse, filter (incl. PCIe), write}. */
int swg_filter_paf(swg_ctx* ctx, const char* in_path, const char* out_path, const swg_config* cfg, int threads,
                   swg_stats* stats, double timing_ms[4]);
const char* swg_paf_last_error(void);

/* ---- .1aln front end: record derivation (src/unified_filter.rs:21-154) ----------------------------------------
 * The reference reads .1aln through fastga-rs (AlnReader::open / get_all_seq_names / read_alignment, call sites
 * src/unified_filter.rs:27-36, 67), an un-vendored dependency (fastga-rs 0.1.2 @5216a15, onecode 0.1.0 @5fa1e93,
 * Cargo.lock:617-619, 1192-1194): the DECODER stays in the Rust host.  What crosses the boundary is the decoded
 * alignment as that reader returns it; the library derives the RecordMeta columns exactly as extract_1aln_metadata does:
 *   names cut at the first white space after skipping leading white space, the whole header when it has no word
 *     (split_whitespace().next().unwrap_or(full), :83-92; white space = Unicode White_Space)
 *   block_length = (query_end - query_start) + (target_end - target_start)   (:107-112, wrapping u64)
 *   identity = matches / query_span as f64, 0.0 when the span is 0           (:119-123)
 *   rank = position of the alignment in the file                              (:63, :142)
 * and interns the names like the PAF path.  swg_filter() over swg_aln_records() is filter_file's .1aln branch
 * (src/unified_filter.rs:310-317); the host writes the passing alignments itself (write_1aln_filtered, :158-190: the
 * ranks with status != 0).  Values >= 2^32: coordinates are rebased per sequence as in swg_filter64 (offsets from
 * swg_aln_seq_offsets, NULL when nothing was rebased); SWG_ERR_RANGE if a sequence's mapped stretch, a block length or a
 * match count still does not fit 32 bits (same limit as the PAF path). */
typedef struct swg_aln_input {
  uint64_t n;
  const char* const* query_name;   /* [n] NUL-terminated: id_to_name[aln.query_name], or the raw field (:71-82) */
  const char* const* target_name;  /* [n] */
  const uint64_t* query_start;     /* [n] aln.query_start as u64 ... */
  const uint64_t* query_end;
  const uint64_t* target_start;
  const uint64_t* target_end;
  const uint64_t* matches;         /* [n] aln.matches as u64 (:115) */
  const char* strand;              /* [n] aln.strand: '+', anything else counts as '-' */
} swg_aln_input;
typedef struct swg_aln swg_aln;
int swg_aln_open(const swg_aln_input* in, swg_aln** out);
void swg_aln_close(swg_aln* a);
/* records in file order (rank k = record k); pointers are owned by the handle */
const swg_records* swg_aln_records(const swg_aln* a);
const uint64_t* swg_aln_seq_offsets(const swg_aln* a);
const uint64_t* swg_aln_record_offsets(const swg_aln* a, int axis);  /* (as swg_paf_record_offsets) */
uint32_t swg_aln_num_sequences(const swg_aln* a);
const char* swg_aln_sequence_name(const swg_aln* a, uint32_t id); /* the name after the first-word cut */

/* ---- tree sparsification of the PAF before the filter (--sparsify tree:<near>[:<far>[:<random>]] / knn:...) ---------
 * tree_filter::apply_tree_filter_to_paf (src/tree_filter.rs:205-285), which the reference runs on the input before
 * PafFilter::filter_paf (src/main.rs:3640-3688): per unordered pair of genomes (first two '#' parts) identity =
 * sum(matches) / sum(block length); every genome keeps its k_nearest best and k_farthest worst neighbours, plus every pair
 * whose DefaultHasher (SipHash-1-3) value is <= random_fraction * 2^64; the lines of the kept pairs survive (input order,
 * "\n" ends).  Identity ties fall to the neighbour's prefix in ascending order (the reference's order is arbitrary
 * there).  Host code.  *out_text is allocated by the library: release it with swg_free(). */
int swg_paf_tree_filter(const char* text, uint64_t len, uint64_t k_nearest, uint64_t k_farthest, double random_fraction,
                        char** out_text, uint64_t* out_len);
void swg_free(void* p);

/* ---- alnstats (src/bin/alnstats.rs): statistics of a PAF and the comparison of two ------------------------------------
 * parse_paf (:103-164) over host threads: lines with fewer than 11 fields are skipped; a line whose columns 2, 3, 4, 7,
 * 10 or 11 do not parse as u64 ends the run as in the reference (SWG_ERR_INVALID, "Invalid query length" ... in
 * swg_alnstats_last_error()).  mapping length = query_end - query_start; genome = name up to the last '#' (:94-100);
 * a genome's size = sum of the last-seen lengths of its sequences; coverage of (query genome, target genome) =
 * 100 * bases / size(query genome) over the inter-genome lines (:42-73).  The reference lists the pairs in HashMap
 * order; here they come in order of first appearance in the file (that fixes the summation order of avg_coverage and
 * the order of equal coverages in the detailed table).  swg_alnstats_report / _compare produce the exact text of
 * print_stats (:166-228) / compare_stats (:230-284); release it with swg_free().  Host code, no GPU. */
typedef struct swg_alnstats swg_alnstats;
typedef struct swg_alnstats_summary {
  uint64_t total_mappings, total_bases, total_matches, self_mappings, inter_chromosomal, inter_genome, chr_pair_count;
  uint64_t genome_pairs, above_95_pct;
  double avg_identity; /* total_matches / total_bases, 0 when there are no bases (:75-81) */
  double avg_coverage;
} swg_alnstats_summary;
int swg_alnstats_open(const char* path, int threads, swg_alnstats** out); /* plain / .gz / .bgz / "-" like swg_paf_open */
int swg_alnstats_open_buffer(const char* text, uint64_t len, int threads, swg_alnstats** out);
void swg_alnstats_close(swg_alnstats* s);
const swg_alnstats_summary* swg_alnstats_get(const swg_alnstats* s);
/* pair i < genome_pairs; the genome strings keep their trailing '#' and are owned by the handle */
int swg_alnstats_pair(const swg_alnstats* s, uint64_t i, const char** q_genome, const char** t_genome, double* coverage,
                      uint64_t* bases, uint64_t* matches);
int swg_alnstats_report(const swg_alnstats* s, const char* label, int detailed, char** out_text, uint64_t* out_len);
int swg_alnstats_compare(const swg_alnstats* a, const swg_alnstats* b, const char* file1, const char* file2,
                         char** out_text, uint64_t* out_len);
const char* swg_alnstats_last_error(void);

/* ---- ANI pre-pass for "aniN" identity thresholds (src/main.rs:296-688, src/cli.rs:76-130) -------------------
 * calculate_ani_stats: median over genome pairs (last-'#' prefixes, unordered) of Σmatches / Σblock_len, over
 *   SWG_ANI_ALL         every inter-genome line                                   main.rs:339-342, 392-498
 *   SWG_ANI_ORTHOGONAL  the survivors of a fixed 1:1 / >= 1 kb / matches-scored filter   main.rs:343-382
 *   SWG_ANI_NPERCENTILE lines taken in descending length / identity / identity*max(ln len,1) order (stable)
 *                       until their block lengths cover `percentile` % of the total sequence size   main.rs:500-688
 * Host threads parse the ANI view of each line (f64 matches/block length, first valid dv:f:, sequence lengths);
 * the GPU does the key sort, the prefix cut and the per-pair sums (each pair summed in reference order, so the
 * f64 sums are bit-identical); the 1:1 filter of ORTHOGONAL is swg_filter.  The reference panics on NaN keys
 * (partial_cmp().unwrap()); here that is SWG_ERR_INVALID.  Non-integral block lengths: SWG_ERR_UNSUPPORTED. */
enum { SWG_ANI_ALL = 0, SWG_ANI_ORTHOGONAL = 1, SWG_ANI_NPERCENTILE = 2 };
enum { SWG_NSORT_LENGTH = 0, SWG_NSORT_IDENTITY = 1, SWG_NSORT_SCORE = 2 };
typedef struct swg_ani_input {
  uint64_t n;               /* records of the swg_paf handle */
  const uint8_t* eligible;  /* [n] line takes part (not '#'-led, genomes differ)         main.rs:406-432 */
  const uint32_t* pair;     /* [n] unordered genome-pair id < n_pairs */
  uint64_t n_pairs;
  const double* matches;    /* [n] final_matches: column 10, or (1-dv)*block_len          main.rs:435-446 */
  const double* block_len;  /* [n] column 11 as f64 (default 1.0) */
  double total_genome_size; /* Σ first-seen length of every sequence on an eligible line  main.rs:560-572, 626 */
} swg_ani_input;
/* parse_ani_method (main.rs:296-330): returns 1 and fills the outputs, 0 for None */
int swg_parse_ani_method(const char* s, int* kind, double* percentile, int* sort);
/* parse_identity_value (cli.rs:76-130): ani_percentile < 0 means None.  Returns SWG_OK or SWG_ERR_INVALID. */
int swg_parse_identity_value(const char* s, double ani_percentile, double* out);
/* host side: the ANI view of the records (arrays owned by the handle, valid until swg_paf_close) */
int swg_paf_ani_input(swg_paf* p, int threads, swg_ani_input* out);
/* device side: median per-pair ANI.  `select` (optional, [n]) further restricts the lines (ORTHOGONAL survivors);
 * kind ALL/ORTHOGONAL = file order, no cut.  0.0 when no line takes part (main.rs:448-451, 604-607). */
int swg_ani_median(swg_ctx* ctx, const swg_ani_input* in, const uint8_t* select, int kind, double percentile, int sort,
                   double* ani50);
/* calculate_ani_stats over an open PAF (runs the ORTHOGONAL filter itself) */
int swg_paf_ani_stats(swg_ctx* ctx, swg_paf* p, int kind, double percentile, int sort, int threads, double* ani50);

#ifdef __cplusplus
}
#endif
#endif /* SWEEPGA_GPU_H */
