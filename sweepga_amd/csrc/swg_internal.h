// Internal declarations shared by the HIP translation units of libsweepga_gpu.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <string>
#include <vector>

#include "../../include/sweepga_gpu.h"

// ---- error plumbing ---------------------------------------------------------------------
struct swg_ctx {
  int device = -1;
  hipStream_t stream = nullptr;
  hipStream_t copy_stream = nullptr;  // H2D copies of the streamed host path (created on first use)
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  // bump arena for per-call scratch; grown (never shrunk) between calls
  char* arena = nullptr;
  size_t arena_cap = 0;
  size_t arena_off = 0;
  size_t arena_peak = 0;   // high-water mark of the current call (may exceed cap -> retry)
  bool arena_overflow = false;
  // device block holding the records and results of swg_filter() (host buffers in / out): kept between calls, grown on
  // demand, so a host that filters file after file pays no hipMalloc / hipFree in steady state
  char* io_block = nullptr;
  size_t io_cap = 0;
  // pinned staging ring of swg_filter_gathered (records picked out of the caller's columns chunk by chunk on their way to the
  // device): RING_SLOTS slots of ring_slot_bytes, one event per slot (the slot's last copy); allocated on first use
  char* ring = nullptr;
  size_t ring_slot_bytes = 0;
  hipEvent_t ring_ev[4] = {nullptr, nullptr, nullptr, nullptr};
  uint32_t* narrow_host = nullptr;  // swg_filter64: the rebased 32-bit columns (host side, malloc), released by swg_narrow_release
  size_t narrow_cap = 0;            //   in words
  // pinned host scratch for small read-backs
  uint64_t* h_scalars = nullptr;  // 64 x u64
  std::string err;
  int num_cu = 256;
  uint64_t n_readbacks = 0;  // swg_read_scalars calls (each one a stream synchronisation), for SWG_DEBUG
  // record slots that prepare writes only if the device-side probe of the input order says so (*flag != 0), for the all-members
  // gathers of the scaffold stage (swg_filter.hip)
  const struct swg_key_ends* call_probe_slots = nullptr;
  const uint32_t* call_probe_flag = nullptr;
  const uint32_t* call_group32 = nullptr;  // the running call's (query, target, strand) group of every record, when prepare wrote it
  int sort_drop_level = 0;   // raised when a sort on a truncated key met runs too long to order in the gather (swg_radix_drop_bits)
  uint64_t sort_drop_n = 0;  // ... by a call over this many records: a call of a very different size starts from level 0 again
  uint64_t pair_fallback_n[2] = {0, 0};   // the pair-resident scaffold stage handed a call of this many records back on a condition found
  int pair_fallback_skips[2] = {0, 0};    // calls that did not try since (every 16th does: the next input of that size may be another kind)
  int pair_fallback_count[2] = {0, 0};    //   on the device (deep long units, dense LDS batches, a degenerate record), so many times in a
                                          //   row: from the second time on, calls of about that size do not try it (swg_filter.hip).  [0]: the
                                          //   attempt that takes an unlimited mapping sweep as the identity, [1]: the one behind a real sweep
  uint64_t seg_sweep_deep_n = 0;  // the segment-resident k = 1 sweep met deep data on an axis of this many records: axes of about
                                  //   that size go straight to the tile kernels (swg_seg_sweep_k1)
  // per-kernel profiler (swg_profile_*)
  bool prof_on = false;
  std::string prof_only;  // non-empty: only launches with this label are bracketed (swg_profile_select)
  struct prof_pending { int name; hipEvent_t a, b; };
  struct prof_entry { std::string name; uint64_t launches = 0; double ms = 0.0; uint64_t units = 0; };
  std::vector<prof_pending> prof_pending_list;
  std::vector<hipEvent_t> prof_free_events;
  std::vector<prof_entry> prof_entries;
};

// RAII bracket around one kernel launch (or a short run of launches of one kind).
struct swg_prof_scope {
  swg_ctx* ctx;
  int name = -1;
  hipEvent_t a = nullptr, b = nullptr;
  // units: elements this launch works on (0 = unknown: the whole record set of the call)
  swg_prof_scope(swg_ctx* c, const char* kernel_name, uint64_t units = 0);
  ~swg_prof_scope();
};
// Resolves pending event pairs into the per-name table (synchronises the stream).
int swg_prof_collect(swg_ctx* ctx);

extern thread_local std::string swg_create_error;

int swg_set_error(swg_ctx* ctx, int code, const char* fmt, ...);

#define SWG_HIP(ctx, call)                                                                      \
  do {                                                                                          \
    hipError_t e_ = (call);                                                                     \
    if (e_ != hipSuccess)                                                                       \
      return swg_set_error((ctx), e_ == hipErrorOutOfMemory ? SWG_ERR_OOM : SWG_ERR_HIP,        \
                           "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__,    \
                           __LINE__);                                                           \
  } while (0)

#define SWG_TRY(expr)          \
  do {                         \
    int rc_ = (expr);          \
    if (rc_ != SWG_OK) return rc_; \
  } while (0)

#define SWG_KERNEL_CHECK(ctx) SWG_HIP((ctx), hipGetLastError())
// Launch bracket: `SWG_LAUNCH(ctx, "name", kernel<<<grid, block, 0, stream>>>(args));`
#define SWG_LAUNCH(ctx, name, ...)        \
  do {                                    \
    swg_prof_scope prof_scope_(ctx, name); \
    __VA_ARGS__;                          \
  } while (0)
// the same with the number of elements the launch works on (kernels launched on sub-problems, e.g. the sort passes)
#define SWG_LAUNCH_N(ctx, name, units, ...)        \
  do {                                             \
    swg_prof_scope prof_scope_(ctx, name, units);  \
    __VA_ARGS__;                                   \
  } while (0)

// ---- arena --------------------------------------------------------------------------------
// Allocation never fails inside a pipeline: when the arena is too small the pointer returned is
// NULL-safe garbage-free (nullptr) and arena_overflow is set; pipelines are written as
// "plan, then run": run_with_arena() executes the pipeline body, and if it overflowed, grows the
// arena to the recorded peak and runs it again.
void* swg_arena_alloc(swg_ctx* ctx, size_t bytes);
template <class T>
static inline T* swg_alloc(swg_ctx* ctx, size_t n) {
  return reinterpret_cast<T*>(swg_arena_alloc(ctx, n * sizeof(T)));
}
void swg_arena_reset(swg_ctx* ctx);
int swg_arena_reserve(swg_ctx* ctx, size_t bytes);
struct swg_arena_mark {
  size_t off;
};
static inline swg_arena_mark swg_arena_save(swg_ctx* ctx) { return {ctx->arena_off}; }
static inline void swg_arena_restore(swg_ctx* ctx, swg_arena_mark m) { ctx->arena_off = m.off; }

// Runs body(ctx) with a fresh arena; on overflow grows the arena and runs it once more.
template <class F>
static inline int swg_run_with_arena(swg_ctx* ctx, F&& body) {
  for (int attempt = 0; attempt < 8; ++attempt) {
    swg_arena_reset(ctx);
    int rc = body();
    if (!ctx->arena_overflow) return rc;
    // overflow: the body saw a nullptr and bailed out with SWG_ERR_OOM before using it
    size_t need = ctx->arena_peak + (ctx->arena_peak >> 1) + (size_t(1) << 20);
    if (need < 2 * ctx->arena_cap) need = 2 * ctx->arena_cap;
    int rc2 = swg_arena_reserve(ctx, need);
    if (rc2 != SWG_OK) return rc2;
  }
  return swg_set_error(ctx, SWG_ERR_OOM, "scratch arena still too small after growing 8 times");
}
#define SWG_CHECK_ARENA(ctx)                                          \
  do {                                                                \
    if ((ctx)->arena_overflow) return SWG_ERR_OOM;                    \
  } while (0)

// ---- primitives (swg_sort.hip) ---------------------------------------------------------------
// Exclusive prefix sum of n u32 values, in place allowed (out may equal in).  If total_out is
// non-null it receives the grand total (device pointer, u32... as u64).
int swg_exclusive_scan_u32(swg_ctx* ctx, const uint32_t* in, uint32_t* out, uint64_t n,
                           uint64_t* d_total_out);
// Inclusive running maximum of n u32 values (in place allowed).
int swg_inclusive_max_scan_u32(swg_ctx* ctx, const uint32_t* in, uint32_t* out, uint64_t n);
// Stream compaction from byte flags (non-zero = set) without a position array:
//   swg_flags_count  : per-4096-tile counts + their exclusive scan (tile_off, from the arena), total -> *d_total
//   swg_flags_compact: list[rank of i among set flags] = i, ascending
struct swg_flag_scan {
  const uint8_t* flags;
  uint64_t n;
  uint32_t* tile_off;
};
int swg_flags_count(swg_ctx* ctx, const uint8_t* flags, uint64_t n, swg_flag_scan* fs, uint64_t* d_total);
int swg_flags_compact(swg_ctx* ctx, const swg_flag_scan& fs, uint32_t* list);
int swg_inclusive_max_scan_u64(swg_ctx* ctx, const uint64_t* in, uint64_t* out, uint64_t n);
int swg_inclusive_sum_scan_u64(swg_ctx* ctx, const uint64_t* in, uint64_t* out, uint64_t n);
// Stable LSD radix sort of (key, value) pairs on bits [begin_bit, end_bit) of the key.
// *keys / *vals hold the input; the *_alt buffers are scratch of the same size.  Passes ping-pong
// between the two pairs of buffers and the POINTERS are swapped so that on return *keys / *vals
// address the sorted data (no copy-back pass).
// `prehist` (optional): the digit histograms of every pass, [SWG_RADIX_MAX_PASSES][SWG_RADIX_BINS] device words over the
// same bit range, already accumulated by the kernel that wrote the keys (swg_radix_hist_add below) -- the sort then skips
// its own pass over the keys.
int swg_radix_sort_pairs(swg_ctx* ctx, uint64_t** keys, uint32_t** vals, uint64_t** keys_alt, uint32_t** vals_alt,
                         uint64_t n, int begin_bit, int end_bit, uint32_t* prehist = nullptr);
// The same sort over the key bits [0, key_bits) with 8-byte elements after the first pass: pass 1 reads the (key, value)
// pairs and writes packed words ((key >> 8) << val_bits) | value; the later passes move 16 bytes per element instead of 24.
// Needs key_bits - 8 + val_bits <= 64 and every value < 2^val_bits.  `keys` (n u64) is overwritten, `scratch` is n u64 of
// scratch; *packed_out is whichever of the two holds the sorted packed words.  Returns SWG_ERR_UNSUPPORTED when the shape
// does not qualify (the caller then uses swg_radix_sort_pairs).
// vals == nullptr: the values are the identity (element i carries i), no array is read.
int swg_radix_sort_packed(swg_ctx* ctx, uint64_t* keys, const uint32_t* vals, uint64_t* scratch, uint64_t n, int key_bits,
                          int val_bits, uint32_t* prehist, uint64_t** packed_out);
// whether swg_radix_sort_packed will take this shape (so that the caller can leave an identity value array unwritten)
bool swg_radix_sort_packed_applies(uint64_t n, int key_bits, int val_bits);
constexpr int SWG_RADIX_BINS = 512;  // row stride of the digit histograms: digits are 8 bits wide, 9 in some packed passes
constexpr int SWG_RADIX_MAX_PASSES = 8;
// The digits of a sort: pass p takes `bits[p]` key bits from bit `shift[p]` on.
struct swg_radix_plan {
  int npasses;
  uint8_t shift[SWG_RADIX_MAX_PASSES];
  uint8_t bits[SWG_RADIX_MAX_PASSES];
};
// swg_radix_sort_pairs: 8-bit digits over [begin_bit, end_bit) (npasses > 8: the three-kernel fallback takes the sort).
static inline swg_radix_plan swg_radix_plan_pairs(int begin_bit, int end_bit) {
  swg_radix_plan pl{};
  const int np = (end_bit - begin_bit + 7) / 8;
  pl.npasses = np > SWG_RADIX_MAX_PASSES ? 0 : np;
  for (int p = 0; p < pl.npasses; ++p) {
    pl.shift[p] = (uint8_t)(begin_bit + 8 * p);
    pl.bits[p] = (uint8_t)(end_bit - (begin_bit + 8 * p) < 8 ? end_bit - (begin_bit + 8 * p) : 8);
  }
  return pl;
}
// swg_radix_sort_packed: the first pass takes the key's low 8 bits (the packed word drops exactly those); the rest is cut
// into 9-bit digits where that saves a pass (34 remaining bits: 9 + 9 + 8 + 8 instead of five passes), else into 8-bit ones.
// SWG_SORT_BITS8=1 (test knob): 8-bit digits everywhere.
swg_radix_plan swg_radix_plan_packed(int key_bits);
// Sort on a truncated key: words (k << val_bits) | value over all `sorted_bits` bits of k, plain 8-byte passes
// (swg_sort.hip has the story).  swg_radix_drop_bits: how many low bits of a key of key_bits (the lowest `low_bits` of which
// are the caller's to order afterwards) to leave out, 0 = take the packed sort.
swg_radix_plan swg_radix_plan_words(int sorted_bits);
int swg_radix_drop_bits(uint64_t n, int key_bits, int low_bits, int val_bits, int level, int dmax0 = 10);
int swg_radix_sort_words(swg_ctx* ctx, uint64_t* words, uint64_t* scratch, uint64_t n, int sorted_bits, int val_bits,
                         uint32_t* prehist, uint64_t** out);
constexpr int SWG_RUN_HALO = 64;  // a gather orders runs of equal truncated keys of up to this many elements (+1)
#ifdef __HIPCC__
// Accumulates one key of every lane of the wavefront into the work-group's LDS histograms h[pass][bin] (zero them first,
// flush them with swg_radix_hist_flush).  High digits are usually the same for a whole wavefront (segment bits): one add
// then instead of 64 serialised LDS atomics on one bin.
__device__ __forceinline__ void swg_radix_hist_zero(uint32_t (*h)[SWG_RADIX_BINS], int npasses) {
  for (int p = 0; p < npasses; ++p)
    for (int b = threadIdx.x; b < SWG_RADIX_BINS; b += blockDim.x) h[p][b] = 0;
}
__device__ __forceinline__ void swg_radix_hist_add(uint32_t (*h)[SWG_RADIX_BINS], uint64_t k, bool valid, const swg_radix_plan& pl) {
  const uint64_t vmask = __ballot(valid);
  // the plan's two byte arrays as two 64-bit scalars: pass p's shift and width come out by shifting, the loop stays rolled
  // (indexing the arrays sends the plan through scratch; unrolling all eight passes doubles the scalar work of kernels that
  // are bound by the scalar unit)
  uint64_t sh64, b64;
  __builtin_memcpy(&sh64, pl.shift, 8);
  __builtin_memcpy(&b64, pl.bits, 8);
  static_assert(SWG_RADIX_MAX_PASSES == 8, "one byte per pass in a 64-bit word");
#pragma unroll 1
  for (int p = 0; p < pl.npasses; ++p) {
    const int shift = (int)((sh64 >> (8 * p)) & 0xffu), bits = (int)((b64 >> (8 * p)) & 0xffu);
    const uint32_t d = (uint32_t)(k >> shift) & ((1u << bits) - 1u);
    const uint32_t d0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)d);
    const bool uniform = __ballot(valid && d != d0) == 0 && (vmask & 1ull);
    if (uniform) {
      if ((threadIdx.x & 63) == 0) atomicAdd(&h[p][d0], (uint32_t)__popcll(vmask));
    } else if (valid) {
      atomicAdd(&h[p][d], 1u);
    }
  }
}
__device__ __forceinline__ void swg_radix_hist_flush(uint32_t (*h)[SWG_RADIX_BINS], int npasses, uint32_t* ghist) {
  for (int p = 0; p < npasses; ++p)
    for (int b = threadIdx.x; b < SWG_RADIX_BINS; b += blockDim.x) {
      const uint32_t c = h[p][b];
      if (c) atomicAdd(&ghist[p * SWG_RADIX_BINS + b], c);
    }
}
#endif
// Copies `count` u64 scalars from device to host (pinned), synchronising the stream.
int swg_narrow_coords(swg_ctx* ctx, uint64_t n, const uint64_t* s0, const uint64_t* e0, uint32_t* out_s, uint32_t* out_e,
                      const char* axis);
// swg_records64 (host pointers) -> 32-bit columns in ctx->narrow_host, coordinates rebased per sequence (host/rebase.h)
int swg_rebase_host(swg_ctx* ctx, const swg_records64* rec, const swg_config* cfg, swg_records* out);
void swg_narrow_release(swg_ctx* ctx);  // after the call that used them (keeps at most 256 MB with the context)
int swg_read_scalars(swg_ctx* ctx, const uint64_t* d_src, uint64_t* h_dst, int count);
// apply_filters over the records idx[0 .. m) (ascending) of the caller's host columns: gathered chunk by chunk into a pinned ring,
// uploaded on the copy stream behind the gathering, filtered, results into status_sub / chain_sub [m] (swg_filter.hip; the shards
// of swg_filter_multi when the input is not grouped by query genome)
int swg_filter_gathered(swg_ctx* ctx, const swg_records* rec, const uint32_t* idx, uint64_t m, const swg_config* cfg, uint8_t* status_sub,
                        uint32_t* chain_sub, swg_stats* stats, int threads);

static inline int swg_bits_for(uint64_t max_value) {  // bits needed to represent max_value
  int b = 0;
  while (max_value) {
    ++b;
    max_value >>= 1;
  }
  return b;
}

// ---- sweep (swg_sweep.hip) ---------------------------------------------------------------------
// One axis of the plane sweep over n intervals that all live in one index space [0, n):
//   seg[i]    dense-ish segment id (u64 composite already reduced to < 2^seg_bits)
//   start/end axis coordinates; score_key[i] = sortable score (smaller = better); tie-break = i
//   alive[i]  (optional) 0 = interval does not take part
//   k         mappings_to_keep (SWG_K_INF = unbounded), thr = overlap threshold
// Output keep[i] = 1 iff interval i is returned by plane_sweep_query/target on its segment
// (src/plane_sweep_exact.rs:268-433); keep[i] = 0 for !alive.
// XCD-aware block index: workgroups are dealt round-robin over the 8 XCDs (b and b + 8 share an L2), so the blocks one
// XCD runs are spread over the whole grid.  Kernels that gather by original record index (neighbouring sorted
// positions point into the same region of the input when the input is grouped) want neighbouring blocks on the SAME
// L2 instead: logical block = a contiguous eighth of the grid per XCD.  Bijective for any grid size; a speed choice
// only (placement is not a contract).
__device__ __forceinline__ uint32_t swg_xcd_block(uint32_t bid, uint32_t nwg) {
  const uint32_t q = nwg >> 3, r = nwg & 7u, x = bid & 7u;
  return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (bid >> 3);
}

// Score key and the four coordinates of a record in one 32-byte slot = one memory sector: the sweep's post-sort gather
// costs one random sector per begin, and the begins can be sorted as packed 8-byte words (the low 8 bits of the start,
// which the packed word drops after the first pass, come back from here).
struct __attribute__((aligned(32))) swg_key_ends {
  uint64_t key;
  uint32_t start[2];  // [0] query start, [1] target start
  uint32_t end[2];    // [0] query end, [1] target end
  uint32_t pad[2];    // with a scaffold stage: [0] matches, [1] block length (else 0)
};

struct swg_axis_input {
  uint64_t n;
  const uint64_t* seg = nullptr;  // [n] explicit segment ids, or nullptr: computed on the fly as
                                  //   seg_a[i] * seg_mul + (seg_table ? seg_table[seg_b[i]] : seg_b[i])
  const uint32_t* seg_a = nullptr;
  const uint32_t* seg_b = nullptr;
  const uint32_t* seg_table = nullptr;
  uint32_t seg_mul = 0;
  int seg_bits;              // ids < 2^seg_bits
  const uint32_t* start;     // [n]
  const uint32_t* end;       // [n]
  int pos_bits;              // coordinates < 2^pos_bits
  const uint64_t* score_key; // [n]
  const uint8_t* alive;      // [n] or nullptr
  const uint8_t* and_with = nullptr;  // optional: keep[i] &= and_with[i] (intersection with another axis' result)
  const swg_key_ends* packed = nullptr;  // optional: replaces score_key / end in the post-sort gather
  int packed_end = 0;                    //   which end of `packed` is this axis' end
  uint32_t* sorted_idx_out = nullptr;    // optional [n]: the record indices in (segment, start, index) order, dead first --
  int* sorted_idx_valid = nullptr;       //   *valid = 1 when the begins were sorted (not for k = inf without zero lengths)
  // optional: the input is grouped by (seg_a, seg_b) pair and these are its runs (device array of {first record, length}); the
  // begins are then sorted segment by segment in LDS (swg_segsort.hip) instead of by the radix sort.  Needs score_key / end as
  // plain columns (packed == nullptr), the live records per run and their total.
  const void* seg_runs = nullptr;
  uint32_t n_seg_runs = 0;
  const uint32_t* seg_run_alive = nullptr;
  uint64_t n_alive = 0;
};
int swg_sweep_axis(swg_ctx* ctx, const swg_axis_input& in, uint64_t k, double thr, uint8_t* keep);
int swg_seg_run_alive(swg_ctx* ctx, const void* runs, uint32_t n_runs, const uint8_t* alive, uint32_t* run_alive);
// what swg_seg_sort_begins leaves behind for the streaming sweep over its output (device pointers into the arena: valid until the
// caller restores its mark)
struct swg_seg_plan_view {
  int valid = 0;
  const uint32_t *seg_a = nullptr, *seg_e = nullptr, *class_list = nullptr;
  uint32_t* counters = nullptr;
  uint32_t n_runs = 0, ncls[4] = {0, 0, 0, 0};
  uint64_t n_dead = 0;
};
// tile_flag (zero before, or nullptr): the sort settles the lone intervals itself -- single[record] = 1 for those that are kept --
// and flags every other begin's sorted position for the tile kernels (SegSortArgs)
int swg_seg_sort_begins(swg_ctx* ctx, const swg_axis_input& in, uint64_t* S, uint32_t* I, uint32_t* E, uint64_t* KEY, uint64_t* tile_xf,
                        uint32_t ntilesf, uint8_t* single, int* done, swg_seg_plan_view* view = nullptr, uint8_t* tile_flag = nullptr);
int swg_seg_stream_sweep_k1(swg_ctx* ctx, const swg_seg_plan_view& v, const uint64_t* S, const uint32_t* I, const uint32_t* E, const uint64_t* KEY,
                            int pos_bits, double thr, const uint8_t* and_with, uint8_t* keep, uint64_t n, int* done);
int swg_seg_sweep_k1(swg_ctx* ctx, const swg_axis_input& in, double thr, uint8_t* keep, uint64_t* S, uint32_t* I, uint32_t* E, uint64_t* KEY,
                     uint64_t* tile_xf, uint8_t* single, uint64_t* nb_left, int* outcome);
// Both axes with k = inf in one pass; *done = 0 when zero-length intervals exist (then the per-axis calls are needed).
int swg_kinf_both(swg_ctx* ctx, uint64_t n, const uint32_t* qs, const uint32_t* qe, const uint32_t* ts, const uint32_t* te,
                  const uint8_t* alive, uint8_t* keep, int* done);

// score keys: key[i] = order-preserving transform of -score so that smaller key = better
// (src/plane_sweep_exact.rs:29-86, 183-193); length is always q_end - q_start.
int swg_score_keys(swg_ctx* ctx, uint64_t n, const uint32_t* q_start, const uint32_t* q_end,
                   const double* identity, int scoring, uint64_t* key_out);
// Step-1 retain (src/paf_filter.rs:384-388), score keys and the two scalars the pipeline needs, in one pass over
// the records: scalars[0] = max coordinate, scalars[1] = number of retained records (device u64, pre-zeroed).
int swg_prepare(swg_ctx* ctx, const swg_records* r, const swg_config* cfg, uint8_t* alive, swg_key_ends* key_ends, bool with_keys,
                unsigned long long* scalars, uint32_t* group32 = nullptr, uint32_t* probe_flag = nullptr, uint64_t* score_out = nullptr);
