// The filter pipeline: PafFilter::apply_filters (src/paf_filter.rs:379-747) on the device, and
// the host-array entry points of the C ABI.
#include <chrono>
#include <cstdio>
#include <cstring>
#include <vector>

#include <algorithm>

#include "swg_internal.h"
#include "swg_log.h"
#include "swg_pipeline.h"
#include "swg_scaffold_internal.h"
#include "host/rebase.h"
#include "host/threads.h"

namespace {

// scratch bytes per record reserved up front (the scaffold figure had to follow the input-order probe's slot buffer, 32 B per
// record: with 256 the first call of a context overflowed and ran twice -- seen as 1.25 launches per call in a 4-call profile)
constexpr size_t SWG_ARENA_B_SWEEP = 96, SWG_ARENA_B_SCAFFOLD = 296;

constexpr int EW = 256;
inline unsigned nblk(uint64_t n) { return (unsigned)((n + EW - 1) / EW); }

// 16 records per thread (16-byte flag load, 16-byte status store, four 16-byte chain stores)
__global__ __launch_bounds__(EW) void unassigned_status_kernel(uint64_t n, const uint8_t* __restrict__ keep,
                                                               uint8_t* __restrict__ status,
                                                               uint32_t* __restrict__ chain, int aligned) {
  const uint64_t i0 = ((uint64_t)blockIdx.x * EW + threadIdx.x) * 16;
  if (i0 >= n) return;
  if (aligned && i0 + 16 <= n) {
    const uint4 k = *reinterpret_cast<const uint4*>(keep + i0);
    auto f = [](uint32_t w) { return (((((w & 0x7f7f7f7fu) + 0x7f7f7f7fu) | w) >> 7) & 0x01010101u) * (uint32_t)SWG_ST_UNASSIGNED; };
    *reinterpret_cast<uint4*>(status + i0) = make_uint4(f(k.x), f(k.y), f(k.z), f(k.w));
    const uint4 z = make_uint4(0, 0, 0, 0);
#pragma unroll
    for (int j = 0; j < 4; ++j) *reinterpret_cast<uint4*>(chain + i0 + 4 * j) = z;
    return;
  }
  for (uint64_t i = i0; i < i0 + 16 && i < n; ++i) {
    status[i] = keep[i] ? SWG_ST_UNASSIGNED : SWG_ST_DROPPED;
    chain[i] = 0;
  }
}

// number of non-zero bytes (grid-stride, one atomic per wave of a small grid)
__global__ __launch_bounds__(EW) void count_nonzero_kernel(uint64_t n, const uint8_t* __restrict__ v,
                                                           unsigned long long* __restrict__ out) {
  uint32_t cnt = 0;
  for (uint64_t i = (uint64_t)blockIdx.x * EW + threadIdx.x; i < n; i += (uint64_t)gridDim.x * EW) cnt += v[i] ? 1 : 0;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) cnt += __shfl_down(cnt, o, 64);
  if ((threadIdx.x & 63) == 0 && cnt) atomicAdd(out, (unsigned long long)cnt);
}

__global__ __launch_bounds__(EW) void log_kernel(uint64_t n, const double* __restrict__ x, double* __restrict__ y) {
  uint64_t i = (uint64_t)blockIdx.x * EW + threadIdx.x;
  if (i < n) y[i] = swg_log_glibc(x[i]);
}
__global__ __launch_bounds__(EW) void log_range_kernel(uint64_t first, uint64_t stride, uint64_t n,
                                                       double* __restrict__ y) {
  uint64_t i = (uint64_t)blockIdx.x * EW + threadIdx.x;
  if (i < n) y[i] = swg_log_glibc((double)(first + i * stride));
}

// ---- 64-bit records: per-sequence rebasing (host/rebase.h has the argument and the host version) ------------------
// lo[s] = smallest coordinate sequence s has anywhere.  Grouped inputs name one query (and often one target) across a
// whole wavefront: one atomic per wavefront then, one per lane otherwise.
__device__ __forceinline__ void seq_min_atomic(unsigned long long* lo, uint32_t id, unsigned long long v, bool in) {
  const uint32_t id0 = __shfl(id, __ffsll((long long)__ballot(in)) - 1, 64);
  if (__all(!in || id == id0)) {
    unsigned long long m = in ? v : ~0ull;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const unsigned long long x = __shfl_xor(m, o, 64);
      m = x < m ? x : m;
    }
    if ((threadIdx.x & 63) == 0 && m != ~0ull) atomicMin(lo + id0, m);
  } else if (in) {
    atomicMin(lo + id, v);
  }
}
__global__ __launch_bounds__(EW) void seq_lo_kernel(uint64_t n, const uint32_t* __restrict__ q_id, const uint32_t* __restrict__ t_id,
                                                    const uint64_t* __restrict__ qs, const uint64_t* __restrict__ qe,
                                                    const uint64_t* __restrict__ ts, const uint64_t* __restrict__ te,
                                                    uint32_t n_seq, unsigned long long* __restrict__ lo,
                                                    unsigned long long* __restrict__ bad) {
  const uint64_t i = (uint64_t)blockIdx.x * EW + threadIdx.x;
  bool in = i < n;
  if (__ballot(in) == 0) return;
  uint32_t q = 0, t = 0;
  unsigned long long a = 0, b = 0;
  if (in) {
    q = q_id[i];
    t = t_id[i];
    if (q >= n_seq || t >= n_seq) {  // the ids index the table: reported like the host version does (field 6), never followed
      atomicMin(bad, ((unsigned long long)(i + 1) << 3) | 6u);
      in = false;
      q = t = 0;
    } else {
      a = qs[i] < qe[i] ? qs[i] : qe[i];
      b = ts[i] < te[i] ? ts[i] : te[i];
    }
  }
  seq_min_atomic(lo, q, a, in);
  seq_min_atomic(lo, t, b, in);
}
// bad[0] = 1 + smallest record index whose rebased value does not fit 32 bits (0: none), bad[1] = its field
__global__ __launch_bounds__(EW) void rebase_kernel(uint64_t n, const uint32_t* __restrict__ q_id, const uint32_t* __restrict__ t_id,
                                                    const uint64_t* __restrict__ qs, const uint64_t* __restrict__ qe,
                                                    const uint64_t* __restrict__ ts, const uint64_t* __restrict__ te,
                                                    const uint64_t* __restrict__ matches, const uint64_t* __restrict__ block,
                                                    uint32_t n_seq, const unsigned long long* __restrict__ lo,
                                                    uint32_t* __restrict__ o_qs, uint32_t* __restrict__ o_qe,
                                                    uint32_t* __restrict__ o_ts, uint32_t* __restrict__ o_te,
                                                    uint32_t* __restrict__ o_m, uint32_t* __restrict__ o_b,
                                                    unsigned long long* __restrict__ bad) {
  const uint64_t i = (uint64_t)blockIdx.x * EW + threadIdx.x;
  if (i >= n) return;
  if (q_id[i] >= n_seq || t_id[i] >= n_seq) return;  // reported by seq_lo_kernel
  const unsigned long long oq = lo[q_id[i]], ot = lo[t_id[i]];
  const unsigned long long v[6] = {qs[i] - oq, qe[i] - oq, ts[i] - ot, te[i] - ot, matches[i], block[i]};
  o_qs[i] = (uint32_t)v[0];
  o_qe[i] = (uint32_t)v[1];
  o_ts[i] = (uint32_t)v[2];
  o_te[i] = (uint32_t)v[3];
  o_m[i] = (uint32_t)v[4];
  o_b[i] = (uint32_t)v[5];
  int f = -1;
#pragma unroll
  for (int k = 5; k >= 0; --k)
    if (v[k] >> 32) f = k;
  if (f >= 0) atomicMin(bad, ((unsigned long long)(i + 1) << 3) | (unsigned)f);
}

// the same with one constant per sweep segment and axis (host/rebase.h, columns_by_axis): lo_q[(q, genome(t))], lo_t[(t, genome(q))]
// (only behind seq_lo_kernel: the ids are known to be in range; the genome table's entries are the caller's, checked here)
__global__ __launch_bounds__(EW) void axis_lo_kernel(uint64_t n, const uint32_t* __restrict__ q_id, const uint32_t* __restrict__ t_id,
                                                     const uint64_t* __restrict__ qs, const uint64_t* __restrict__ qe,
                                                     const uint64_t* __restrict__ ts, const uint64_t* __restrict__ te,
                                                     const uint32_t* __restrict__ seq_genome, uint32_t n_genome,
                                                     unsigned long long* __restrict__ lo_q, unsigned long long* __restrict__ lo_t) {
  const uint64_t i = (uint64_t)blockIdx.x * EW + threadIdx.x;
  const bool in = i < n;
  if (__ballot(in) == 0) return;
  uint32_t cq = 0, ct = 0;
  unsigned long long a = 0, b = 0;
  bool ok = in;
  if (in) {
    const uint32_t q = q_id[i], t = t_id[i], gq = seq_genome[q], gt = seq_genome[t];
    ok = gq < n_genome && gt < n_genome;
    if (ok) {
      cq = q * n_genome + gt;
      ct = t * n_genome + gq;
      a = qs[i] < qe[i] ? qs[i] : qe[i];
      b = ts[i] < te[i] ? ts[i] : te[i];
    }
  }
  seq_min_atomic(lo_q, cq, a, ok);
  seq_min_atomic(lo_t, ct, b, ok);
}
__global__ __launch_bounds__(EW) void axis_rebase_kernel(uint64_t n, const uint32_t* __restrict__ q_id, const uint32_t* __restrict__ t_id,
                                                         const uint64_t* __restrict__ qs, const uint64_t* __restrict__ qe,
                                                         const uint64_t* __restrict__ ts, const uint64_t* __restrict__ te,
                                                         const uint64_t* __restrict__ matches, const uint64_t* __restrict__ block,
                                                         const uint32_t* __restrict__ seq_genome, uint32_t n_genome,
                                                         const unsigned long long* __restrict__ lo_q, const unsigned long long* __restrict__ lo_t,
                                                         uint32_t* __restrict__ o_qs, uint32_t* __restrict__ o_qe, uint32_t* __restrict__ o_ts,
                                                         uint32_t* __restrict__ o_te, unsigned long long* __restrict__ bad) {
  const uint64_t i = (uint64_t)blockIdx.x * EW + threadIdx.x;
  if (i >= n) return;
  const uint32_t q = q_id[i], t = t_id[i], gq = seq_genome[q], gt = seq_genome[t];
  if (gq >= n_genome || gt >= n_genome) {
    atomicMin(bad, ((unsigned long long)(i + 1) << 3) | 6u);
    return;
  }
  const unsigned long long oq = lo_q[(size_t)q * n_genome + gt], ot = lo_t[(size_t)t * n_genome + gq];
  const unsigned long long v[6] = {qs[i] - oq, qe[i] - oq, ts[i] - ot, te[i] - ot, matches[i], block[i]};  // (the last two: rebase_kernel wrote them)
  o_qs[i] = (uint32_t)v[0];
  o_qe[i] = (uint32_t)v[1];
  o_ts[i] = (uint32_t)v[2];
  o_te[i] = (uint32_t)v[3];
  int f = -1;
#pragma unroll
  for (int k = 5; k >= 0; --k)
    if (v[k] >> 32) f = k;
  if (f >= 0) atomicMin(bad, ((unsigned long long)(i + 1) << 3) | (unsigned)f);
}

__global__ void call_begin_kernel() {}

void limits_from_mode(int mode, uint64_t max_q, uint64_t max_t, uint64_t* kq, uint64_t* kt) {
  // src/paf_filter.rs:1004-1014
  switch (mode) {
    case SWG_MODE_ONE_TO_ONE:
      *kq = 1;
      *kt = 1;
      break;
    case SWG_MODE_ONE_TO_MANY:
      *kq = max_q ? max_q : 1;
      *kt = max_t ? max_t : SWG_K_INF;
      break;
    default:
      *kq = max_q ? max_q : SWG_K_INF;
      *kt = max_t ? max_t : SWG_K_INF;
  }
}

}  // namespace

// ---- mapping-level sweep ----------------------------------------------------------------------
int swg_mapping_sweep(swg_ctx* ctx, const swg_records* r, const swg_config* cfg, const uint8_t* alive,
                      const swg_key_ends* key_ends, int pos_bits, uint8_t* keep, uint32_t* q_order, int* q_order_valid,
                      const void* pair_runs, uint32_t n_pair_runs, const uint64_t* score_key, uint64_t n_alive) {
  const uint64_t n = r->n;
  uint64_t kq, kt;
  limits_from_mode(cfg->mapping_filter_mode, cfg->mapping_max_per_query, cfg->mapping_max_per_target, &kq, &kt);
  if (kq == SWG_K_INF && kt == SWG_K_INF) {  // no limit on either axis: one pass, unless zero-length intervals exist
    int done = 0;
    if (q_order_valid) *q_order_valid = 0;
    SWG_TRY(swg_kinf_both(ctx, n, r->q_start, r->q_end, r->t_start, r->t_end, alive, keep, &done));
    if (done) return SWG_OK;
  }
  swg_arena_mark mark = swg_arena_save(ctx);
  uint8_t* keep_q = swg_alloc<uint8_t>(ctx, n);
  SWG_CHECK_ARENA(ctx);
  // segment ids of the mapping-level sweep (src/paf_filter.rs:1037-1100), computed where they are used:
  //   query axis : (query sequence, genome of the target)   target axis : (target sequence, genome of the query)
  swg_axis_input ax;
  ax.n = n;
  ax.seg_bits = swg_bits_for((uint64_t)r->n_seq * r->n_genome_last);  // ids + 1 <= n_seq * n_genome
  ax.seg_mul = r->n_genome_last;
  ax.seg_table = r->seq_genome_last;
  ax.pos_bits = pos_bits;
  ax.score_key = nullptr;
  ax.packed = key_ends;
  ax.alive = alive;
  ax.seg_a = r->q_id;
  ax.seg_b = r->t_id;
  ax.start = r->q_start;
  ax.end = r->q_end;
  ax.packed_end = 0;
  ax.sorted_idx_out = q_order;
  ax.sorted_idx_valid = q_order ? q_order_valid : nullptr;
  if (pair_runs && score_key && !key_ends && !q_order) {
    // the input is grouped by (query, target) pair: a segment of either axis is a handful of whole runs, sorted in LDS
    uint32_t* run_alive = swg_alloc<uint32_t>(ctx, n_pair_runs);
    SWG_CHECK_ARENA(ctx);
    SWG_TRY(swg_seg_run_alive(ctx, pair_runs, n_pair_runs, alive, run_alive));
    ax.seg_runs = pair_runs;
    ax.n_seg_runs = n_pair_runs;
    ax.seg_run_alive = run_alive;
    ax.n_alive = n_alive;
    ax.score_key = score_key;
    ax.packed = nullptr;
  }
  SWG_TRY(swg_sweep_axis(ctx, ax, kq, cfg->overlap_threshold, keep_q));
  ax.sorted_idx_out = nullptr;
  ax.sorted_idx_valid = nullptr;
  ax.seg_a = r->t_id;
  ax.seg_b = r->q_id;
  ax.start = r->t_start;
  ax.end = r->t_end;
  ax.packed_end = 1;
  ax.and_with = keep_q;  // intersection of the two axes, src/paf_filter.rs:1105-1111
  SWG_TRY(swg_sweep_axis(ctx, ax, kt, cfg->overlap_threshold, keep));
  swg_arena_restore(ctx, mark);
  return SWG_OK;
}

// ---- apply_filters on device-resident records ---------------------------------------------------
static int filter_device_body(swg_ctx* ctx, const swg_records* r, const swg_config* cfg, uint8_t* status_out,
                              uint32_t* chain_out, swg_stats* stats) {
  const uint64_t n = r->n;
  hipStream_t st = ctx->stream;
  if (stats) {
    stats->n_in = n;
    stats->n_retained = stats->n_swept = stats->n_chains = stats->n_chains_kept = stats->n_out = 0;
  }
  if (n == 0) return SWG_OK;
  {
    // SWG_CALL_MARKER=1 (tools/profile_round.sh): an empty launch opens every call, so that a kernel trace can be cut into calls
    // whatever path each of them takes (tools/pmc_traffic.py)
    static const bool marker = getenv("SWG_CALL_MARKER") != nullptr;
    if (marker) {
      SWG_LAUNCH(ctx, "call_begin", call_begin_kernel<<<1, 64, 0, st>>>());
      SWG_KERNEL_CHECK(ctx);
    }
  }
  // the per-call pointers the stages below find in the context do not outlive this call (the seams -- swg_merge_chains ... --
  // run the same stages without them)
  struct CallScope {
    swg_ctx* c;
    ~CallScope() {
      c->call_probe_slots = nullptr;
      c->call_probe_flag = nullptr;
      c->call_group32 = nullptr;
    }
  } call_scope{ctx};
  // Records grouped by chromosome pair and a scaffold stage: the pair-resident stage (swg_pair.hip).  Its plan -- the pairs of
  // the input -- is made first.  Without limits in the mapping-level sweep the stage evaluates step 1 itself and takes the
  // unlimited sweep as the identity; it leaves the call to the stages below when it meets a retained record that an unlimited
  // sweep would drop (zero length), or anything else it does not cover.  Behind a mapping sweep it takes the sweep's flags
  // (further down), and the sweep is told that nobody will ask for its sorted order.
  swg_scaf::PairPlan pair_plan;
  // (a context whose calls of about this size came back from the pair-resident stage on the device's word twice in a row --
  // deep long units, dense LDS batches, degenerate records: the whole stage run for nothing each time -- stops trying; a call of
  // another size, or one the stage finishes, starts afresh.  ADVICE round 5.)
  auto about_n = [&](int kind) { return n >= ctx->pair_fallback_n[kind] / 2 && n <= ctx->pair_fallback_n[kind] * 2; };
  for (int kind = 0; kind < 2; ++kind)
    if (ctx->pair_fallback_count[kind] && !about_n(kind)) ctx->pair_fallback_count[kind] = 0;
  // (off: not tried -- but every 16th call of that size is, for the next input of that size may be of another kind; a hand-over
  // then keeps the verdict, a finished call clears it)
  auto off = [&](int kind) {
    if (ctx->pair_fallback_count[kind] < 2) return false;
    if (++ctx->pair_fallback_skips[kind] % 16 == 0) return false;
    return true;
  };
  const bool identity_try_off = off(0);  // (the attempt before any sweep)
  const bool pair_stage_off = off(1);    // (the attempt behind the sweep: then neither is made)
  auto pair_stage = [&](const uint8_t* alive_in, const uint8_t* member_in, bool assumed_identity, int* taken) -> int {
    SWG_TRY(swg_scaf::scaffold_stage_pairs(ctx, r, cfg, alive_in, member_in, assumed_identity, status_out, chain_out, stats, taken, &pair_plan));
    const int kind = assumed_identity ? 0 : 1;
    if (*taken) {
      ctx->pair_fallback_count[kind] = 0;
    } else {  // (a valid plan and no result: handed back)
      ctx->pair_fallback_count[kind] = ctx->pair_fallback_count[kind] && about_n(kind) ? ctx->pair_fallback_count[kind] + 1 : 1;
      ctx->pair_fallback_n[kind] = n;
    }
    return SWG_OK;
  };
  {
    uint64_t kq1, kt1;
    limits_from_mode(cfg->mapping_filter_mode, cfg->mapping_max_per_query, cfg->mapping_max_per_target, &kq1, &kt1);
    const bool sweeps1 = kq1 != SWG_K_INF || kt1 != SWG_K_INF;
    // (the plan also serves a mapping sweep with limits: its axes sort their begins segment by segment over the plan's runs)
    // (... of a large input: the small ones' plan goes through the hash grouping, which only the scaffold stage reads)
    if ((cfg->scaffold_gap != 0 && !pair_stage_off) || (sweeps1 && n > 65536)) SWG_TRY(swg_scaf::pair_plan(ctx, r, cfg, &pair_plan));
  }
  {
    uint64_t kq1, kt1;
    limits_from_mode(cfg->mapping_filter_mode, cfg->mapping_max_per_query, cfg->mapping_max_per_target, &kq1, &kt1);
    if (pair_plan.valid && !pair_stage_off && !identity_try_off && cfg->scaffold_gap != 0 && kq1 == SWG_K_INF && kt1 == SWG_K_INF) {
      int taken = 0;
      SWG_TRY(pair_stage(nullptr, nullptr, true, &taken));
      if (taken) return SWG_OK;
    } else if (!pair_plan.valid && pair_plan.not_grouped && !pair_stage_off && !identity_try_off && cfg->scaffold_gap != 0 && kq1 == SWG_K_INF &&
               kt1 == SWG_K_INF) {
      // a large input whose pairs are interleaved (one query after the other with the targets mixed: what wfmash writes): grouped
      // on the device -- a stable sort of the record indices by pair, one gather into a pair-major copy -- and the pair-resident
      // stage over the copy (swg_pair.hip, pair_group_records)
      const swg_arena_mark gm = swg_arena_save(ctx);
      swg_records copy;
      uint32_t* perm = nullptr;
      int ok = 0;
      SWG_TRY(swg_scaf::pair_group_records(ctx, r, &copy, &perm, &ok));
      if (ok) {
        swg_scaf::PairPlan plan2;
        SWG_TRY(swg_scaf::pair_plan(ctx, &copy, cfg, &plan2));
        if (plan2.valid) {
          plan2.orig = perm;
          uint8_t* st2 = swg_alloc<uint8_t>(ctx, n);
          uint32_t* ch2 = swg_alloc<uint32_t>(ctx, n);
          SWG_CHECK_ARENA(ctx);
          int taken = 0;
          SWG_TRY(swg_scaf::scaffold_stage_pairs(ctx, &copy, cfg, nullptr, nullptr, true, st2, ch2, stats, &taken, &plan2));
          if (taken) {
            ctx->pair_fallback_count[0] = 0;
            SWG_TRY(swg_scaf::pair_ungroup_results(ctx, n, perm, st2, ch2, status_out, chain_out));
            return SWG_OK;
          }
        }
        // (handed back, or the copy's pairs are too many or too small for the stage as well: remembered like any other hand-over)
        ctx->pair_fallback_count[0] = ctx->pair_fallback_count[0] && about_n(0) ? ctx->pair_fallback_count[0] + 1 : 1;
        ctx->pair_fallback_n[0] = n;
      }
      swg_arena_restore(ctx, gm);
    }
  }
  uint8_t* alive = swg_alloc<uint8_t>(ctx, n);
  uint8_t* keep1 = swg_alloc<uint8_t>(ctx, n);
  // The 32-byte record slots (both starts, both ends, score key, matches, block length).  The mapping-level sweep sorts and
  // ranks by them.  With both limits infinite (the CLI defaults) nothing sweeps and the score keys are not computed; the
  // scaffold stage's gather after its first sort can still take a record's six columns from ONE slot instead of six
  // scattered 4-byte reads, each of which moves a 128-byte line from L2.  Whether that pays depends on where those reads hit:
  // the gather walks the records pair by pair, so with ~10^4 records per sequence pair (S-pan) a pair's columns sit in L2 and
  // the slots only move the cost (gather_all_words 3.3 -> 2.4 ms, prepare 0.8 -> 1.6 ms, 6.4 GB more traffic), while one pair
  // of 10^7 records (S-big1) has no such locality (1.20 -> 0.31 ms for 0.02 ms more in prepare).  Chosen by the records per
  // POSSIBLE sequence pair, which is all that is known before the first kernel; SWG_SLOTS=1 / 0 force either (test knobs).
  uint64_t kq0, kt0;
  limits_from_mode(cfg->mapping_filter_mode, cfg->mapping_max_per_query, cfg->mapping_max_per_target, &kq0, &kt0);
  const bool sweeps = !(kq0 == SWG_K_INF && kt0 == SWG_K_INF);
  static const int slots_knob = getenv("SWG_SLOTS") ? atoi(getenv("SWG_SLOTS")) : -1;
  const uint64_t pairs_ub = (uint64_t)r->n_seq * r->n_seq ? (uint64_t)r->n_seq * r->n_seq : 1;
  const bool deep_pairs = slots_knob >= 0 ? slots_knob != 0 : n / pairs_ub >= (uint64_t(1) << 17);
  // a sweep over a pair-grouped input (the plan's runs; not the small inputs grouped through the hash table) sorts its begins
  // segment by segment in LDS and reads plain columns: the score keys as 8 bytes per record instead of the 32-byte slots
  const bool seg_sweep = sweeps && pair_plan.valid && !pair_plan.by_hash;
  uint64_t* score_col = seg_sweep ? swg_alloc<uint64_t>(ctx, n) : nullptr;
  swg_key_ends* key_ends = ((sweeps && !seg_sweep) || (cfg->scaffold_gap != 0 && deep_pairs)) ? swg_alloc<swg_key_ends>(ctx, n) : nullptr;
  // Shallow pairs, nothing sweeps, a scaffold stage follows: whether the slots pay depends on the ORDER of the input -- records
  // grouped by sequence pair (what an aligner writes) keep a pair's columns in L2, any other order does not (S-pan shuffled:
  // 28.8 ms with the column gathers).  That is probed on the device (input_order_probe_kernel); prepare writes the slots and the
  // all-members gathers read them only if the probe says "not grouped".  Nothing else may read these slots.
  swg_key_ends* probe_slots = nullptr;
  uint32_t* probe_flag = nullptr;
  if (!key_ends && cfg->scaffold_gap != 0 && slots_knob < 0 && n >= 65536 && !(pair_plan.valid && !pair_stage_off)) {  // (a valid plan: grouped, and the stage is the pair-resident one)
    probe_slots = swg_alloc<swg_key_ends>(ctx, n);
    probe_flag = swg_alloc<uint32_t>(ctx, 1);
  }
  ctx->call_probe_slots = probe_slots;
  ctx->call_probe_flag = probe_flag;
  unsigned long long* scalars = swg_alloc<unsigned long long>(ctx, 8);
  SWG_CHECK_ARENA(ctx);
  SWG_HIP(ctx, hipMemsetAsync(scalars, 0, 8 * sizeof(unsigned long long), st));
  // behind a mapping sweep the scaffold stage's first sort only orders the (query, target, strand) groups (the query axis'
  // order has the rest): prepare leaves the group of every record as one 4-byte value for it
  // (neither when the pair-resident stage is going to run: it sorts inside its pairs)
  uint32_t* group32 = (sweeps && cfg->scaffold_gap != 0 && !(pair_plan.valid && !pair_stage_off) && (uint64_t)r->n_seq * r->n_seq * 2 < (uint64_t(1) << 32)) ? swg_alloc<uint32_t>(ctx, n) : nullptr;
  SWG_CHECK_ARENA(ctx);
  ctx->call_group32 = group32;
  SWG_TRY(swg_prepare(ctx, r, cfg, alive, key_ends ? key_ends : probe_slots, sweeps, scalars, group32, probe_flag, score_col));
  uint64_t h[3];
  SWG_TRY(swg_read_scalars(ctx, reinterpret_cast<uint64_t*>(scalars), h, 3));
  const int pos_bits = swg_bits_for(h[0]) ? swg_bits_for(h[0]) : 1;
  if (stats) stats->n_retained = h[1];
  // Both limits infinite and no retained record of zero length: the mapping-level sweep keeps exactly the retained
  // records (every interval with start < end survives an unlimited sweep, plane_sweep_exact.rs:219-228), so it is not run
  // and the scaffold stage is told that members == retained records (same array).
  const bool sweep_is_identity = kq0 == SWG_K_INF && kt0 == SWG_K_INF && h[2] == 0;
  if (sweep_is_identity) keep1 = alive;

  // with scaffolding on, the query axis' sorted order is kept: sort A of the chaining is the same order refined by
  // (target sequence, strand)
  uint32_t* q_order = cfg->scaffold_gap != 0 && !(pair_plan.valid && sweeps) ? swg_alloc<uint32_t>(ctx, n) : nullptr;
  SWG_CHECK_ARENA(ctx);
  int q_order_valid = 0;
  if (!sweep_is_identity)
    // (the slots carry score keys only when a sweep with limits was expected: `sweeps`.  Without one they were filled for the
    // scaffold stage's gathers, every key zero -- an unlimited sweep over zero-length records reads none, and gets none)
    SWG_TRY(swg_mapping_sweep(ctx, r, cfg, alive, sweeps ? key_ends : nullptr, pos_bits, keep1, q_order, &q_order_valid, seg_sweep ? pair_plan.runs : nullptr,
                              pair_plan.n_runs, score_col, h[1]));

  if (cfg->scaffold_gap == 0) {  // src/paf_filter.rs:409-434
    const uintptr_t ptrs = reinterpret_cast<uintptr_t>(keep1) | reinterpret_cast<uintptr_t>(status_out) | reinterpret_cast<uintptr_t>(chain_out);
    SWG_LAUNCH(ctx, "unassigned_status", unassigned_status_kernel<<<nblk((n + 15) / 16), EW, 0, st>>>(n, keep1, status_out, chain_out,
                                                                                            (ptrs & 15) == 0));
    SWG_KERNEL_CHECK(ctx);
    if (stats) {
      SWG_LAUNCH(ctx, "count_nonzero", count_nonzero_kernel<<<ctx->num_cu * 4, EW, 0, st>>>(n, keep1, scalars + 3));
      SWG_KERNEL_CHECK(ctx);
      uint64_t c;
      SWG_TRY(swg_read_scalars(ctx, reinterpret_cast<uint64_t*>(scalars + 3), &c, 1));
      stats->n_swept = stats->n_out = c;
    }
    return SWG_OK;
  }
  if (!sweep_is_identity && pair_plan.valid && !pair_stage_off) {  // behind a mapping sweep: with the sweep's flags (members = the records it kept)
    int taken = 0;
    SWG_TRY(pair_stage(alive, keep1, false, &taken));
    if (taken) return SWG_OK;
  }
  return swg_scaffold_stage(ctx, r, cfg, alive, keep1, pos_bits, status_out, chain_out, stats, q_order_valid ? q_order : nullptr, h[1],
                            key_ends);
}

static int validate(swg_ctx* ctx, const swg_records* r, const swg_config* cfg) {
  if (!ctx) return SWG_ERR_INVALID;
  if (!r || !cfg) return swg_set_error(ctx, SWG_ERR_INVALID, "records/config is NULL");
  if (r->n >= (uint64_t(1) << 31)) return swg_set_error(ctx, SWG_ERR_RANGE, "more than 2^31-1 records");
  // (identity may be NULL: then it is matches / max(block_len, 1), computed where it is read -- see swg_records)
  if (r->n && (!r->q_id || !r->t_id || !r->q_start || !r->q_end || !r->t_start || !r->t_end ||
               !r->matches || !r->block_len || !r->strand || !r->seq_genome_last || !r->seq_genome_two))
    return swg_set_error(ctx, SWG_ERR_INVALID, "a record column is NULL");
  if (r->n && (r->n_seq == 0 || r->n_genome_last == 0 || r->n_genome_two == 0))
    return swg_set_error(ctx, SWG_ERR_INVALID, "n_seq / n_genome_* must be > 0");
  if (cfg->scoring_function < 0 || cfg->scoring_function > 4)
    return swg_set_error(ctx, SWG_ERR_INVALID, "bad scoring_function");
  if (cfg->mapping_filter_mode < 0 || cfg->mapping_filter_mode > 2 || cfg->scaffold_filter_mode < 0 ||
      cfg->scaffold_filter_mode > 2)
    return swg_set_error(ctx, SWG_ERR_INVALID, "bad filter mode");
  return SWG_OK;
}

// Scratch high-water marks measured on the 10^8 workload (round 3): 82 B/record for the sweep-only pipeline (32-byte record
// slots, packed sort), 204-223 B/record with the scaffold stage.  Reserving that up front avoids the grow-and-rerun path on a
// context's first call (and, for the streamed host path, any re-allocation between its ranges).
int swg_filter_reserve_arena(swg_ctx* ctx, uint64_t n, const swg_records* rec, const swg_config* cfg, bool wide) {
  // deep sequence pairs (the shape test of filter_device_body): the candidate arrays of the wavefront-per-element lists, 56 B per
  // record, come on top (the S-big1 profiles showed the first call of a fresh context running twice: 1.25 launches per call)
  const uint64_t pairs_ub = (uint64_t)rec->n_seq * rec->n_seq ? (uint64_t)rec->n_seq * rec->n_seq : 1;
  const size_t deep_extra = (cfg->scaffold_gap != 0 && n / pairs_ub >= (uint64_t(1) << 17)) ? 64 : 0;
  size_t want = (size_t)n * ((cfg->scaffold_gap == 0 ? SWG_ARENA_B_SWEEP : SWG_ARENA_B_SCAFFOLD) + deep_extra + (wide ? 24 : 0)) +
                (size_t(8) << 20) + (wide ? (size_t)rec->n_seq * 8 : 0);
  if (cfg->scaffold_gap != 0) {
    // the scaffold stage keeps two genome-pair tables (first appearance of a pair under either prefix rule): dense
    // G x G up to 2^14 genomes, else hashed over the pairs that occur (names without '#': every sequence its own genome)
    for (const uint64_t g : {(uint64_t)rec->n_genome_last, (uint64_t)rec->n_genome_two})
      want += g * g <= (uint64_t(1) << 28) ? (size_t)(g * g) * sizeof(uint32_t) : (size_t)n * 4 * 12;
  }
  if (ctx->arena_cap < want) {
    size_t free_b = 0, total_b = 0;
    const bool known = hipMemGetInfo(&free_b, &total_b) == hipSuccess;
    // never ask for more than what is free (plus what the old arena gives back); the retry path still covers the rest
    const size_t room = known ? free_b + ctx->arena_cap : want;
    SWG_TRY(swg_arena_reserve(ctx, want < room ? want : (room > (size_t(64) << 20) ? room - (size_t(64) << 20) : want)));
  }
  return SWG_OK;
}

// rec64 != NULL: rec's six 32-bit columns are produced from rec64's inside the arena before the pipeline runs
static int filter_device_any(swg_ctx* ctx, const swg_records* rec, const swg_records64* rec64, const swg_config* cfg,
                             uint8_t* status_out, uint32_t* chain_out, swg_stats* stats);

extern "C" int swg_filter_device(swg_ctx* ctx, const swg_records* rec, const swg_config* cfg,
                                 uint8_t* status_out, uint32_t* chain_out, swg_stats* stats) {
  SWG_TRY(validate(ctx, rec, cfg));
  return filter_device_any(ctx, rec, nullptr, cfg, status_out, chain_out, stats);
}

// Which of the value columns a flag set reads at all.  matches / block_len / identity feed (1) the step-1 identity floor and the
// block-length floor, (2) the score keys of a mapping sweep with limits, (3) a chain's weighted identity: the identity floor of
// the span / identity filter and the scores of a scaffold sweep with limits.  The CLI defaults use none of them: with a DERIVED
// identity (no identity column from the caller: matches / max(block_len, 1)) the host paths then send neither matches nor
// block_len (25 instead of 33 bytes per record through the link), and the device derives an "identity" from whatever the two
// unsent columns hold -- finite and non-negative whatever it is (u32 / max(u32, 1)), which a floor of zero or less passes; a
// chain's weighted identity likewise (paf_filter.rs:896-913).  A caller's own identity column is always sent: it may hold
// anything, and a negative or NaN value fails even a floor of zero.  SWG_POISON=1 fills unsent columns
// with 0xff bytes, so that a test sees a reader that should not be there.
void swg_value_columns_needed(const swg_config* cfg, bool* identity_value, bool* weighted_identity) {
  uint64_t kq, kt;
  limits_from_mode(cfg->mapping_filter_mode, cfg->mapping_max_per_query, cfg->mapping_max_per_target, &kq, &kt);
  const bool sweeps = kq != SWG_K_INF || kt != SWG_K_INF;
  uint64_t sq, sk;
  if (cfg->scaffold_filter_mode == SWG_MODE_ONE_TO_ONE) {
    sq = sk = 1;
  } else {
    sq = cfg->scaffold_max_per_query ? cfg->scaffold_max_per_query : SWG_K_INF;
    sk = cfg->scaffold_max_per_target ? cfg->scaffold_max_per_target : SWG_K_INF;
  }
  const bool limited = cfg->scaffold_gap != 0 && (sq != SWG_K_INF || sk != SWG_K_INF);
  static const bool send_all = getenv("SWG_SEND_ALL") != nullptr;  // (test knob: every column whatever the flag set)
  *identity_value = send_all || sweeps || !(cfg->min_identity <= 0.0);
  *weighted_identity = send_all || (cfg->scaffold_gap != 0 && (limited || !(cfg->min_scaffold_identity <= 0.0)));
}

static swg_records narrow_view(const swg_records64* r) {  // everything but the six wide columns
  swg_records v{};
  v.n = r->n;
  v.q_id = r->q_id;
  v.t_id = r->t_id;
  v.identity = r->identity;
  v.strand = r->strand;
  v.n_seq = r->n_seq;
  v.seq_genome_last = r->seq_genome_last;
  v.n_genome_last = r->n_genome_last;
  v.seq_genome_two = r->seq_genome_two;
  v.n_genome_two = r->n_genome_two;
  return v;
}
static int validate64(swg_ctx* ctx, const swg_records64* r, const swg_config* cfg) {
  if (!ctx) return SWG_ERR_INVALID;
  if (!r || !cfg) return swg_set_error(ctx, SWG_ERR_INVALID, "records/config is NULL");
  swg_records v = narrow_view(r);
  static const uint32_t dummy = 0;  // the wide columns are checked here, the rest by validate()
  v.q_start = v.q_end = v.t_start = v.t_end = v.matches = v.block_len = &dummy;
  if (r->n && (!r->q_start || !r->q_end || !r->t_start || !r->t_end || !r->matches || !r->block_len))
    return swg_set_error(ctx, SWG_ERR_INVALID, "a record column is NULL");
  return validate(ctx, &v, cfg);
}

extern "C" int swg_filter_device64(swg_ctx* ctx, const swg_records64* rec, const swg_config* cfg, uint8_t* status_out,
                                   uint32_t* chain_out, swg_stats* stats) {
  SWG_TRY(validate64(ctx, rec, cfg));
  const swg_records v = narrow_view(rec);
  return filter_device_any(ctx, &v, rec, cfg, status_out, chain_out, stats);
}

static int filter_device_any(swg_ctx* ctx, const swg_records* rec, const swg_records64* rec64, const swg_config* cfg,
                             uint8_t* status_out, uint32_t* chain_out, swg_stats* stats) {
  if (rec->n && (!status_out || !chain_out)) return swg_set_error(ctx, SWG_ERR_INVALID, "output buffer is NULL");
  SWG_HIP(ctx, hipSetDevice(ctx->device));
  SWG_TRY(swg_filter_reserve_arena(ctx, rec->n, rec, cfg, rec64 != nullptr));
  // sorts on truncated keys (swg_radix_drop_bits): a context that met runs too long for them stops trying -- for inputs of
  // about the size that failed (a host filtering the same kind of file again and again pays the failed attempt once)
  if (ctx->sort_drop_level > 0 && (rec->n < ctx->sort_drop_n / 2 || rec->n > 2 * ctx->sort_drop_n)) ctx->sort_drop_level = 0;
  const int drop_level0 = ctx->sort_drop_level;
  const uint64_t readbacks0 = ctx->n_readbacks;
  SWG_HIP(ctx, hipEventRecord(ctx->ev0, ctx->stream));
  int rc = swg_run_with_arena(ctx, [&]() -> int {
    if (!rec64 || rec->n == 0) return filter_device_body(ctx, rec, cfg, status_out, chain_out, stats);
    const uint64_t n = rec->n;
    hipStream_t st = ctx->stream;
    uint32_t* c[6];
    for (auto& p : c) p = swg_alloc<uint32_t>(ctx, n);
    unsigned long long* lo = swg_alloc<unsigned long long>(ctx, (size_t)rec->n_seq + 1);  // + the error word
    SWG_CHECK_ARENA(ctx);
    unsigned long long* bad = lo + rec->n_seq;
    SWG_HIP(ctx, hipMemsetAsync(lo, 0xff, ((size_t)rec->n_seq + 1) * sizeof(unsigned long long), st));
    SWG_LAUNCH(ctx, "seq_lo", seq_lo_kernel<<<nblk(n), EW, 0, st>>>(n, rec64->q_id, rec64->t_id, rec64->q_start, rec64->q_end,
                                                                   rec64->t_start, rec64->t_end, rec->n_seq, lo, bad));
    SWG_LAUNCH(ctx, "rebase", rebase_kernel<<<nblk(n), EW, 0, st>>>(n, rec64->q_id, rec64->t_id, rec64->q_start, rec64->q_end,
                                                                   rec64->t_start, rec64->t_end, rec64->matches, rec64->block_len,
                                                                   rec->n_seq, lo, c[0], c[1], c[2], c[3], c[4], c[5], bad));
    SWG_KERNEL_CHECK(ctx);
    uint64_t hb;
    SWG_TRY(swg_read_scalars(ctx, reinterpret_cast<uint64_t*>(bad), &hb, 1));
    if (hb != ~0ull && (hb & 7) < 4 && swg_rebase::axis_tables_fit(rec->n_seq, rec->n_genome_last)) {
      // a sequence touched over 2^32 bases or more: the constants per sweep segment -- (sequence, genome of the other side) --
      // instead (host/rebase.h, columns_by_axis)
      const size_t cells = (size_t)rec->n_seq * rec->n_genome_last;
      unsigned long long* lo2 = swg_alloc<unsigned long long>(ctx, 2 * cells + 1);
      SWG_CHECK_ARENA(ctx);
      unsigned long long* bad2 = lo2 + 2 * cells;
      SWG_HIP(ctx, hipMemsetAsync(lo2, 0xff, (2 * cells + 1) * sizeof(unsigned long long), st));
      SWG_LAUNCH(ctx, "axis_lo", axis_lo_kernel<<<nblk(n), EW, 0, st>>>(n, rec64->q_id, rec64->t_id, rec64->q_start, rec64->q_end, rec64->t_start,
                                                                     rec64->t_end, rec->seq_genome_last, rec->n_genome_last, lo2, lo2 + cells));
      SWG_LAUNCH(ctx, "axis_rebase", axis_rebase_kernel<<<nblk(n), EW, 0, st>>>(n, rec64->q_id, rec64->t_id, rec64->q_start, rec64->q_end,
                                                                             rec64->t_start, rec64->t_end, rec64->matches, rec64->block_len, rec->seq_genome_last,
                                                                             rec->n_genome_last, lo2, lo2 + cells, c[0], c[1], c[2], c[3], bad2));
      SWG_KERNEL_CHECK(ctx);
      SWG_TRY(swg_read_scalars(ctx, reinterpret_cast<uint64_t*>(bad2), &hb, 1));
    }
    if (hb != ~0ull) {
      static const char* const F[8] = {"query_start", "query_end", "target_start", "target_end", "matches", "block_length", "?", "?"};
      const int f = (int)(hb & 7);
      if (f == 6)
        return swg_set_error(ctx, SWG_ERR_INVALID, "record %llu: sequence id out of range", (unsigned long long)((hb >> 3) - 1));
      return swg_set_error(ctx, SWG_ERR_RANGE,
                           f >= 4 ? "record %llu: %s >= 2^32 is not supported"
                                  : "record %llu: the stretch of its sequence that the mappings against one genome touch spans 2^32 bases or "
                                    "more (%s): not supported by the 32-bit device layout",
                           (unsigned long long)((hb >> 3) - 1), F[f]);
    }
    swg_records r32 = *rec;
    r32.q_start = c[0];
    r32.q_end = c[1];
    r32.t_start = c[2];
    r32.t_end = c[3];
    r32.matches = c[4];
    r32.block_len = c[5];
    return filter_device_body(ctx, &r32, cfg, status_out, chain_out, stats);
  });
  if (rc != SWG_OK) return rc;
  if (ctx->sort_drop_level > drop_level0) ctx->sort_drop_n = rec->n;
  SWG_HIP(ctx, hipEventRecord(ctx->ev1, ctx->stream));
  {
    static const bool dbg = getenv("SWG_DEBUG") != nullptr;
    if (dbg) fprintf(stderr, "[swg] filter call over %llu records: %llu scalar read-backs (stream synchronisations)\n",
                     (unsigned long long)rec->n, (unsigned long long)(ctx->n_readbacks - readbacks0));
  }
  if (stats) {
    SWG_HIP(ctx, hipEventSynchronize(ctx->ev1));
    float ms = 0.f;
    SWG_HIP(ctx, hipEventElapsedTime(&ms, ctx->ev0, ctx->ev1));
    stats->device_ms = ms;
    stats->h2d_ms = stats->d2h_ms = 0.0;
  }
  return SWG_OK;
}

// bytes of the staging block swg_filter needs for n records over n_seq sequences (8 u32 columns + identity + strand +
// status + chain + the two genome tables, every column rounded up to 256 bytes)
static size_t io_block_bytes(uint64_t n, uint32_t n_seq) {
  const size_t col4 = ((n * 4 + 255) & ~size_t(255)), col8 = ((n * 8 + 255) & ~size_t(255)),
               col1 = ((n + 255) & ~size_t(255)), seqt = (((size_t)n_seq * 4 + 255) & ~size_t(255));
  return col4 * 8 + col8 + col1 * 2 + seqt * 2 + col4;
}
static int io_block_reserve(swg_ctx* ctx, size_t total) {
  if (ctx->io_cap >= total) return SWG_OK;
  if (ctx->io_block) {
    SWG_HIP(ctx, hipStreamSynchronize(ctx->stream));
    SWG_HIP(ctx, hipFree(ctx->io_block));
    ctx->io_block = nullptr;
    ctx->io_cap = 0;
  }
  void* p = nullptr;
  size_t got = total + (total >> 3);  // a little headroom: the next file is rarely exactly this size
  hipError_t e = hipMalloc(&p, got);
  if (e != hipSuccess) {
    got = total;
    e = hipMalloc(&p, got);
  }
  if (e != hipSuccess)
    return swg_set_error(ctx, SWG_ERR_OOM, "hipMalloc of %zu bytes for record staging failed: %s", total, hipGetErrorString(e));
  ctx->io_block = static_cast<char*>(p);
  ctx->io_cap = got;
  return SWG_OK;
}

int swg_io_block_reserve(swg_ctx* ctx, uint64_t n, uint32_t n_seq) { return io_block_reserve(ctx, io_block_bytes(n, n_seq)); }

// Host buffers in / out: stage through device copies, then the device entry point.  The staging block lives in the
// context (no allocation in steady state).  Only what the configuration reads crosses PCIe: `matches` and `strand` are
// scaffold-stage inputs (src/paf_filter.rs:875-894, 761-770) and the chain ids are all zero without scaffolding
// (:409-434), so with scaffold_gap == 0 they are neither uploaded nor downloaded; `block_len` is only needed by the
// scaffold stage or a non-zero --min-aln-length (32 instead of 47 B up, 1 instead of 5 B down per record).
extern "C" int swg_filter(swg_ctx* ctx, const swg_records* rec, const swg_config* cfg, uint8_t* status_out,
                          uint32_t* chain_out, swg_stats* stats) {
  SWG_TRY(validate(ctx, rec, cfg));
  const uint64_t n = rec->n;
  if (n == 0) {
    if (stats) *stats = swg_stats{};
    return SWG_OK;
  }
  if (!status_out || !chain_out) return swg_set_error(ctx, SWG_ERR_INVALID, "output buffer is NULL");
  {  // records grouped by query genome and enough of them: ranges uploaded while their predecessors are filtered
    int taken = 0;
    swg_ctx* one[1] = {ctx};
    const int src = swg_stream_try(one, 1, rec, cfg, status_out, chain_out, stats, &taken);
    if (src != SWG_OK || taken) return src;
  }
  SWG_HIP(ctx, hipSetDevice(ctx->device));
  hipStream_t st = ctx->stream;
  const bool scaffold = cfg->scaffold_gap != 0;
  const bool derived_identity = rec->identity == nullptr;  // matches / max(block_len, 1), evaluated on the device
  const size_t col4 = ((n * 4 + 255) & ~size_t(255)), col8 = ((n * 8 + 255) & ~size_t(255)),
               col1 = ((n + 255) & ~size_t(255)), seqt = (((size_t)rec->n_seq * 4 + 255) & ~size_t(255));
  SWG_TRY(io_block_reserve(ctx, io_block_bytes(n, rec->n_seq)));
  char* blk = ctx->io_block;
  size_t off = 0;
  auto take = [&](size_t bytes) {
    char* p = blk + off;
    off += bytes;
    return p;
  };
  swg_records d = *rec;
  hipEvent_t e0 = ctx->ev0, e1 = ctx->ev1;
  int rc = SWG_OK;
  float h2d = 0.f, d2h = 0.f;
  static const bool poison = getenv("SWG_POISON") != nullptr;
  auto up = [&](const void* src, size_t bytes, size_t slot, bool needed) -> void* {
    char* dst = take(slot);
    if (needed && rc == SWG_OK && hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, st) != hipSuccess)
      rc = swg_set_error(ctx, SWG_ERR_HIP, "H2D copy failed");
    if (!needed && poison) (void)hipMemsetAsync(dst, 0xff, slot, st);
    return dst;
  };
  bool id_value, wid_value;
  swg_value_columns_needed(cfg, &id_value, &wid_value);
  // the caller's own identity column always goes: whatever it holds (a dv:f: override may be negative or NaN) decides the
  // step-1 test even against a floor of zero -- only the DERIVED identity is known to pass it
  const bool send_identity = !derived_identity;
  const bool send_mb = wid_value || (derived_identity && id_value);     // matches and block_len
  (void)hipEventRecord(e0, st);
  d.q_id = (const uint32_t*)up(rec->q_id, n * 4, col4, true);
  d.t_id = (const uint32_t*)up(rec->t_id, n * 4, col4, true);
  d.q_start = (const uint32_t*)up(rec->q_start, n * 4, col4, true);
  d.q_end = (const uint32_t*)up(rec->q_end, n * 4, col4, true);
  d.t_start = (const uint32_t*)up(rec->t_start, n * 4, col4, true);
  d.t_end = (const uint32_t*)up(rec->t_end, n * 4, col4, true);
  d.matches = (const uint32_t*)up(rec->matches, n * 4, col4, send_mb);
  d.block_len = (const uint32_t*)up(rec->block_len, n * 4, col4, send_mb || cfg->min_block_length != 0);
  d.identity = (const double*)up(rec->identity, n * 8, col8, send_identity);
  if (!send_identity) d.identity = nullptr;  // (derived on the device: from the columns, or -- nobody reading it -- from anything)
  d.strand = (const uint8_t*)up(rec->strand, n, col1, scaffold);         // read by the scaffold stage only
  d.seq_genome_last = (const uint32_t*)up(rec->seq_genome_last, (size_t)rec->n_seq * 4, seqt, true);
  d.seq_genome_two = (const uint32_t*)up(rec->seq_genome_two, (size_t)rec->n_seq * 4, seqt, true);
  uint8_t* d_status = (uint8_t*)take(col1);
  uint32_t* d_chain = (uint32_t*)take(col4);
  (void)hipEventRecord(e1, st);
  if (rc == SWG_OK && hipEventSynchronize(e1) == hipSuccess) (void)hipEventElapsedTime(&h2d, e0, e1);
  swg_stats local{};
  if (rc == SWG_OK) rc = swg_filter_device(ctx, &d, cfg, d_status, d_chain, &local);
  if (rc == SWG_OK) {
    (void)hipEventRecord(e0, st);
    if (hipMemcpyAsync(status_out, d_status, n, hipMemcpyDeviceToHost, st) != hipSuccess ||
        (scaffold && hipMemcpyAsync(chain_out, d_chain, n * 4, hipMemcpyDeviceToHost, st) != hipSuccess))
      rc = swg_set_error(ctx, SWG_ERR_HIP, "D2H copy failed");
    (void)hipEventRecord(e1, st);
    if (!scaffold) std::memset(chain_out, 0, n * sizeof(uint32_t));  // no ch:Z: tags without scaffolding; overlaps the status copy
    if (hipStreamSynchronize(st) != hipSuccess && rc == SWG_OK)
      rc = swg_set_error(ctx, SWG_ERR_HIP, "stream synchronize failed: %s", hipGetErrorString(hipGetLastError()));
    if (rc == SWG_OK) (void)hipEventElapsedTime(&d2h, e0, e1);
  } else {
    (void)hipStreamSynchronize(st);
  }
  if (stats && rc == SWG_OK) {
    *stats = local;
    stats->h2d_ms = h2d;
    stats->d2h_ms = d2h;
  }
  return rc;
}

// ---- a shard of swg_filter_multi picked out of the caller's columns ("hipMemcpyAsync from pinned memory on per-
// device streams") ---------------------------------------------------------------------------------------------------------------
// The shard's records are idx[0 .. m) of the caller's host columns, ascending.  Round 2-5 copied them into pageable columns of
// their own first (host/shard_host.h scatter: ~50 fresh bytes per record, 0.41 s of page faults per 10^8 records) and uploaded
// those.  Here host threads gather chunk after chunk into the slots of a small pinned ring that stays with the context, every
// slot goes to the device on the copy stream as soon as it is full (only the columns the flag set reads), and a slot is
// gathered into again when its copy is through -- the gathering of chunk c + 1 runs beside the copy of chunk c.
namespace {
constexpr int RING_SLOTS = 3;
constexpr uint64_t RING_CHUNK_MAX = uint64_t(1) << 20;  // records per slot
inline uint64_t ring_chunk() {  // SWG_RING_CHUNK (test knob): fewer records per slot, so that a small input wraps around the ring
  static const uint64_t v = [] {
    const char* e = getenv("SWG_RING_CHUNK");
    const long long x = e ? atoll(e) : 0;
    return x >= 256 && (uint64_t)x < RING_CHUNK_MAX ? (uint64_t)x : RING_CHUNK_MAX;
  }();
  return v;
}
}
int swg_filter_gathered(swg_ctx* ctx, const swg_records* rec, const uint32_t* idx, uint64_t m, const swg_config* cfg, uint8_t* status_sub,
                        uint32_t* chain_sub, swg_stats* stats, int threads) {
  SWG_TRY(validate(ctx, rec, cfg));
  if (m == 0) {
    if (stats) *stats = swg_stats{};
    return SWG_OK;
  }
  if (m >= (uint64_t(1) << 31)) return swg_set_error(ctx, SWG_ERR_RANGE, "more than 2^31-1 records in a shard");
  SWG_HIP(ctx, hipSetDevice(ctx->device));
  hipStream_t st = ctx->stream;
  if (!ctx->copy_stream) SWG_HIP(ctx, hipStreamCreateWithFlags(&ctx->copy_stream, hipStreamNonBlocking));
  hipStream_t cs = ctx->copy_stream;
  const uint64_t RING_CHUNK = ring_chunk();
  const size_t slot_bytes = (size_t)RING_CHUNK_MAX * (8 * 4 + 8 + 1);
  if (!ctx->ring) {
    void* hp = nullptr;
    SWG_HIP(ctx, hipHostMalloc(&hp, slot_bytes * RING_SLOTS, hipHostMallocDefault));
    ctx->ring = static_cast<char*>(hp);
    ctx->ring_slot_bytes = slot_bytes;
    for (int k = 0; k < RING_SLOTS; ++k) SWG_HIP(ctx, hipEventCreateWithFlags(&ctx->ring_ev[k], hipEventDisableTiming));
  }
  const bool scaffold = cfg->scaffold_gap != 0;
  const bool derived_identity = rec->identity == nullptr;
  bool id_value, wid_value;
  swg_value_columns_needed(cfg, &id_value, &wid_value);
  const bool send_identity = !derived_identity;
  const bool send_mb = wid_value || (derived_identity && id_value);
  const bool send_block = send_mb || cfg->min_block_length != 0;
  // the device block: swg_filter's layout
  const size_t col4 = ((m * 4 + 255) & ~size_t(255)), col8 = ((m * 8 + 255) & ~size_t(255)), col1 = ((m + 255) & ~size_t(255)),
               seqt = (((size_t)rec->n_seq * 4 + 255) & ~size_t(255));
  SWG_TRY(io_block_reserve(ctx, io_block_bytes(m, rec->n_seq)));
  char* blk = ctx->io_block;
  size_t off = 0;
  auto take = [&](size_t bytes) {
    char* p = blk + off;
    off += bytes;
    return p;
  };
  char* d_col[10];  // q_id, t_id, q_start, q_end, t_start, t_end, matches, block_len (4 bytes), identity (8), strand (1)
  for (int c = 0; c < 8; ++c) d_col[c] = take(col4);
  d_col[8] = take(col8);
  d_col[9] = take(col1);
  char* d_gl = take(seqt);
  char* d_g2 = take(seqt);
  uint8_t* d_status = (uint8_t*)take(col1);
  uint32_t* d_chain = (uint32_t*)take(col4);
  const void* h_col[10] = {rec->q_id, rec->t_id, rec->q_start, rec->q_end, rec->t_start, rec->t_end, rec->matches, rec->block_len, rec->identity, rec->strand};
  const bool need[10] = {true, true, true, true, true, true, send_mb, send_block, send_identity, scaffold};
  static const bool poison = getenv("SWG_POISON") != nullptr;
  if (poison)
    for (int c = 6; c < 10; ++c)
      if (!need[c]) SWG_HIP(ctx, hipMemsetAsync(d_col[c], 0xff, c == 8 ? col8 : (c == 9 ? col1 : col4), cs));
  SWG_HIP(ctx, hipMemcpyAsync(d_gl, rec->seq_genome_last, (size_t)rec->n_seq * 4, hipMemcpyHostToDevice, cs));
  SWG_HIP(ctx, hipMemcpyAsync(d_g2, rec->seq_genome_two, (size_t)rec->n_seq * 4, hipMemcpyHostToDevice, cs));
  if (threads < 1) threads = 1;
  (void)hipEventRecord(ctx->ev0, cs);
  int rc = SWG_OK;
  const uint64_t n_chunks = (m + RING_CHUNK - 1) / RING_CHUNK;
  for (uint64_t ch = 0; ch < n_chunks && rc == SWG_OK; ++ch) {
    const int k = (int)(ch % RING_SLOTS);
    const uint64_t c0 = ch * RING_CHUNK, cnt = m - c0 < RING_CHUNK ? m - c0 : RING_CHUNK;
    if (ch >= (uint64_t)RING_SLOTS && hipEventSynchronize(ctx->ring_ev[k]) != hipSuccess) {  // the slot's last copy is through
      rc = swg_set_error(ctx, SWG_ERR_HIP, "waiting for a staging slot failed");
      break;
    }
    char* slot = ctx->ring + (size_t)k * slot_bytes;
    // the slot's columns side by side: RING_CHUNK * {4 x 8, 8, 1}
    char* s_col[10];
    for (int c = 0; c < 8; ++c) s_col[c] = slot + (size_t)c * RING_CHUNK * 4;
    s_col[8] = slot + (size_t)8 * RING_CHUNK * 4;
    s_col[9] = s_col[8] + (size_t)RING_CHUNK * 8;
    try {
      const int tt = (uint64_t)threads > cnt / 16384 + 1 ? (int)(cnt / 16384 + 1) : threads;
      swg_host::run(tt, [&](int t) {
        const uint64_t b = cnt * (uint64_t)t / tt, e = cnt * (uint64_t)(t + 1) / tt;
        const uint32_t* ix = idx + c0;
        for (int c = 0; c < 8; ++c) {
          if (!need[c]) continue;
          const uint32_t* src = static_cast<const uint32_t*>(h_col[c]);
          uint32_t* dst = reinterpret_cast<uint32_t*>(s_col[c]);
          for (uint64_t j = b; j < e; ++j) dst[j] = src[ix[j]];
        }
        if (need[8]) {
          const double* src = static_cast<const double*>(h_col[8]);
          double* dst = reinterpret_cast<double*>(s_col[8]);
          for (uint64_t j = b; j < e; ++j) dst[j] = src[ix[j]];
        }
        if (need[9]) {
          const uint8_t* src = static_cast<const uint8_t*>(h_col[9]);
          uint8_t* dst = reinterpret_cast<uint8_t*>(s_col[9]);
          for (uint64_t j = b; j < e; ++j) dst[j] = src[ix[j]];
        }
      });
    } catch (const std::system_error& e) {
      rc = swg_set_error(ctx, SWG_ERR_OOM, "cannot start host threads: %s", e.what());
      break;
    } catch (const std::bad_alloc&) {
      rc = swg_set_error(ctx, SWG_ERR_OOM, "out of host memory while staging a shard");
      break;
    }
    for (int c = 0; c < 10 && rc == SWG_OK; ++c) {
      if (!need[c]) continue;
      const size_t w = c == 8 ? 8 : (c == 9 ? 1 : 4);
      if (hipMemcpyAsync(d_col[c] + c0 * w, s_col[c], cnt * w, hipMemcpyHostToDevice, cs) != hipSuccess)
        rc = swg_set_error(ctx, SWG_ERR_HIP, "H2D copy of a staged chunk failed");
    }
    (void)hipEventRecord(ctx->ring_ev[k], cs);
  }
  (void)hipEventRecord(ctx->ev1, cs);
  float h2d = 0.f, d2h = 0.f;
  if (rc == SWG_OK && hipEventSynchronize(ctx->ev1) == hipSuccess) (void)hipEventElapsedTime(&h2d, ctx->ev0, ctx->ev1);  // (the columns are complete: the filter's stream may start)
  if (rc != SWG_OK) {
    (void)hipStreamSynchronize(cs);
    return rc;
  }
  swg_records d = *rec;
  d.n = m;
  d.q_id = (const uint32_t*)d_col[0]; d.t_id = (const uint32_t*)d_col[1]; d.q_start = (const uint32_t*)d_col[2]; d.q_end = (const uint32_t*)d_col[3];
  d.t_start = (const uint32_t*)d_col[4]; d.t_end = (const uint32_t*)d_col[5]; d.matches = (const uint32_t*)d_col[6]; d.block_len = (const uint32_t*)d_col[7];
  d.identity = send_identity ? (const double*)d_col[8] : nullptr;
  d.strand = (const uint8_t*)d_col[9];
  d.seq_genome_last = (const uint32_t*)d_gl;
  d.seq_genome_two = (const uint32_t*)d_g2;
  swg_stats local{};
  rc = swg_filter_device(ctx, &d, cfg, d_status, d_chain, &local);
  if (rc == SWG_OK) {
    (void)hipEventRecord(ctx->ev0, st);
    if (hipMemcpyAsync(status_sub, d_status, m, hipMemcpyDeviceToHost, st) != hipSuccess ||
        (scaffold && hipMemcpyAsync(chain_sub, d_chain, m * 4, hipMemcpyDeviceToHost, st) != hipSuccess))
      rc = swg_set_error(ctx, SWG_ERR_HIP, "D2H copy failed");
    (void)hipEventRecord(ctx->ev1, st);
    if (!scaffold) std::memset(chain_sub, 0, m * sizeof(uint32_t));
    if (hipStreamSynchronize(st) != hipSuccess && rc == SWG_OK) rc = swg_set_error(ctx, SWG_ERR_HIP, "stream synchronize failed");
    if (rc == SWG_OK) (void)hipEventElapsedTime(&d2h, ctx->ev0, ctx->ev1);
  } else {
    (void)hipStreamSynchronize(st);
  }
  if (stats && rc == SWG_OK) {
    *stats = local;
    stats->h2d_ms = h2d;
    stats->d2h_ms = d2h;
  }
  return rc;
}

// RecordMeta's own widths from host memory: rebased by host threads into 32-bit columns kept in the context, then swg_filter
// (the bytes crossing PCIe are those of the 32-bit layout).
int swg_rebase_host(swg_ctx* ctx, const swg_records64* rec, const swg_config* cfg, swg_records* out) {
  SWG_TRY(validate64(ctx, rec, cfg));
  swg_records& v = *out;
  v = narrow_view(rec);
  const uint64_t n = rec->n;
  if (n == 0) return SWG_OK;
  // six uninitialised 32-bit columns (every word is written by the threaded pass below); see swg_narrow_release
  if (ctx->narrow_cap < 6 * n) {
    std::free(ctx->narrow_host);
    ctx->narrow_cap = 0;
    ctx->narrow_host = static_cast<uint32_t*>(std::malloc(6 * n * sizeof(uint32_t)));
    if (!ctx->narrow_host) return swg_set_error(ctx, SWG_ERR_OOM, "out of host memory for the 32-bit columns");
    ctx->narrow_cap = 6 * n;
  }
  uint32_t* h = ctx->narrow_host;
  const uint64_t* const c64[6] = {rec->q_start, rec->q_end, rec->t_start, rec->t_end, rec->matches, rec->block_len};
  uint32_t* const c32[6] = {h, h + n, h + 2 * n, h + 3 * n, h + 4 * n, h + 5 * n};
  unsigned hc = std::thread::hardware_concurrency();
  swg_rebase::Result rr;
  try {
    std::vector<uint64_t> lo(rec->n_seq);
    rr = swg_rebase::columns(n, rec->q_id, rec->t_id, c64, rec->n_seq, hc ? (int)(hc > 64 ? 64 : hc) : 1, c32, lo.data());
  } catch (const std::bad_alloc&) {  // also what a failed worker body turns into (host/threads.h)
    return swg_set_error(ctx, SWG_ERR_OOM, "out of host memory while rebasing %llu records", (unsigned long long)n);
  } catch (const std::system_error& e) {
    return swg_set_error(ctx, SWG_ERR_OOM, "cannot start host threads: %s", e.what());
  }
  if (!rr.ok && rr.bad_field == 6)
    return swg_set_error(ctx, SWG_ERR_INVALID, "record %llu: sequence id out of range", (unsigned long long)rr.bad_record);
  if (!rr.ok && rr.bad_field < 4 && swg_rebase::axis_tables_fit(rec->n_seq, rec->n_genome_last)) {
    // a sequence touched over 2^32 bases or more: the constants per sweep segment -- (sequence, genome of the other side) -- instead
    for (uint32_t s = 0; s < rec->n_seq; ++s)
      if (rec->seq_genome_last[s] >= rec->n_genome_last) return swg_set_error(ctx, SWG_ERR_INVALID, "seq_genome_last[%u] out of range", s);
    try {
      rr = swg_rebase::columns_by_axis(n, rec->q_id, rec->t_id, c64, rec->n_seq, rec->seq_genome_last, rec->n_genome_last,
                                       hc ? (int)(hc > 64 ? 64 : hc) : 1, c32);
    } catch (const std::bad_alloc&) {
      return swg_set_error(ctx, SWG_ERR_OOM, "out of host memory while rebasing %llu records", (unsigned long long)n);
    } catch (const std::system_error& e) {
      return swg_set_error(ctx, SWG_ERR_OOM, "cannot start host threads: %s", e.what());
    }
  }
  if (!rr.ok)
    return swg_set_error(ctx, SWG_ERR_RANGE,
                         rr.bad_field >= 4 ? "record %llu: %s >= 2^32 is not supported"
                                           : "record %llu: the stretch of its sequence that the mappings against one genome touch spans 2^32 "
                                             "bases or more (%s): not supported by the 32-bit device layout",
                         (unsigned long long)rr.bad_record, swg_rebase::field_name(rr.bad_field));
  v.q_start = c32[0];
  v.q_end = c32[1];
  v.t_start = c32[2];
  v.t_end = c32[3];
  v.matches = c32[4];
  v.block_len = c32[5];
  return SWG_OK;
}
extern "C" int swg_filter64(swg_ctx* ctx, const swg_records64* rec, const swg_config* cfg, uint8_t* status_out, uint32_t* chain_out,
                            swg_stats* stats) {
  swg_records v;
  SWG_TRY(swg_rebase_host(ctx, rec, cfg, &v));
  const int rc = swg_filter(ctx, &v, cfg, status_out, chain_out, stats);
  swg_narrow_release(ctx);
  return rc;
}
// The 32-bit host columns are only needed for the duration of one call: a small buffer stays with the context (a host that
// filters file after file re-uses it), anything beyond 256 MB (10^7 records) goes back to the system.
void swg_narrow_release(swg_ctx* ctx) {
  if (ctx->narrow_cap * sizeof(uint32_t) > (size_t(256) << 20)) {
    std::free(ctx->narrow_host);
    ctx->narrow_host = nullptr;
    ctx->narrow_cap = 0;
  }
}

// Everything a context's first swg_filter call would otherwise pay for inside the call: the scratch arena and the staging
// block for `n_records_hint` records (hipMalloc), and the library's code objects on the device (the first launch from every
// translation unit loads it) -- a 2,000-record filter call over a built-in record set, both flag families.  A host that
// starts this on its own thread while it reads and parses its input (sweepga-gpu does) finds a warm context afterwards.
extern "C" int swg_warmup(swg_ctx* ctx, uint64_t n_records_hint, uint32_t n_seq_hint, int with_scaffold) {
  if (!ctx) return SWG_ERR_INVALID;
  SWG_HIP(ctx, hipSetDevice(ctx->device));
  const bool dbg = getenv("SWG_DEBUG") != nullptr;
  const auto w0 = std::chrono::steady_clock::now();
  auto lap = [&](const char* what) {
    if (dbg) fprintf(stderr, "[swg] warm-up: %s at %.1f ms\n", what, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - w0).count());
  };
  if (n_records_hint) {
    // scratch: an input of millions of records grouped by query genome (what aligners write) is filtered in eight ranges
    // (csrc/swg_stream.hip) and needs scratch for one of them -- 25 GB of hipMalloc for 10^8 records took up to 1.3 s on the GPU
    // box; an input that turns out not to be grouped grows the arena inside its call instead
    const uint64_t scratch_records = n_records_hint >= (uint64_t(12) << 20) ? std::max<uint64_t>(n_records_hint / 6, uint64_t(6) << 20) : n_records_hint;
    size_t want = (size_t)scratch_records * (with_scaffold ? SWG_ARENA_B_SCAFFOLD : SWG_ARENA_B_SWEEP) + (size_t(8) << 20);
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) == hipSuccess && want + io_block_bytes(n_records_hint, n_seq_hint) > free_b / 2)
      want = 0;  // not the kind of input to guess about: let the call size itself
    if (want && ctx->arena_cap < want) SWG_TRY(swg_arena_reserve(ctx, want));
    if (want) SWG_TRY(io_block_reserve(ctx, io_block_bytes(n_records_hint, n_seq_hint ? n_seq_hint : 1)));
    lap("device memory reserved");
  }
  const uint32_t n = 2000;
  std::vector<uint32_t> q(n), t(n), qs(n), qe(n), ts(n), te(n), m(n), b(n), g(4);
  std::vector<double> id(n);
  std::vector<uint8_t> sd(n), status(n);
  std::vector<uint32_t> chain(n);
  uint64_t x = 0x9e3779b97f4a7c15ull;
  for (uint32_t i = 0; i < n; ++i) {
    x ^= x << 13; x ^= x >> 7; x ^= x << 17;
    q[i] = (uint32_t)(x & 1);
    t[i] = 2 + (uint32_t)((x >> 1) & 1);
    qs[i] = (uint32_t)((x >> 8) % 1000000);
    const uint32_t len = 200 + (uint32_t)((x >> 32) % 5000);
    qe[i] = qs[i] + len;
    ts[i] = qs[i] + (uint32_t)((x >> 48) % 3000);
    te[i] = ts[i] + len;
    b[i] = len;
    m[i] = len - len / 20;
    id[i] = (double)m[i] / (double)b[i];
    sd[i] = (uint8_t)((x >> 60) & 1);
  }
  for (uint32_t k = 0; k < 4; ++k) g[k] = k;
  swg_records r{};
  r.n = n;
  r.q_id = q.data(); r.t_id = t.data(); r.q_start = qs.data(); r.q_end = qe.data(); r.t_start = ts.data(); r.t_end = te.data();
  r.identity = id.data(); r.matches = m.data(); r.block_len = b.data(); r.strand = sd.data();
  r.n_seq = 4; r.seq_genome_last = g.data(); r.n_genome_last = 4; r.seq_genome_two = g.data(); r.n_genome_two = 4;
  swg_config c{};
  c.mapping_filter_mode = SWG_MODE_ONE_TO_ONE;
  c.scaffold_filter_mode = SWG_MODE_ONE_TO_ONE;
  c.overlap_threshold = 0.95;
  c.scaffold_overlap_threshold = 0.5;
  c.scoring_function = 3;
  c.scaffold_gap = with_scaffold ? 50000 : 0;
  c.min_scaffold_length = 1000;
  c.scaffold_max_deviation = with_scaffold ? 20000 : 0;
  SWG_TRY(swg_filter(ctx, &r, &c, status.data(), chain.data(), nullptr));
  lap("first small call");
  if (with_scaffold) {  // the unlimited-sweep kernels of the default flags
    c.mapping_filter_mode = SWG_MODE_MANY_TO_MANY;
    c.scaffold_filter_mode = SWG_MODE_MANY_TO_MANY;
    c.scaffold_max_deviation = 0;
    SWG_TRY(swg_filter(ctx, &r, &c, status.data(), chain.data(), nullptr));
  }
  lap("done");
  return SWG_OK;
}

// ---- plane_sweep_query / target / both on one segment of host arrays --------------------------------
extern "C" int swg_plane_sweep(swg_ctx* ctx, int axis, uint64_t n, const uint64_t* q_start, const uint64_t* q_end,
                               const uint64_t* t_start, const uint64_t* t_end, const double* identity,
                               uint64_t k_query, uint64_t k_target, double thr, int scoring, uint8_t* keep_out) {
  if (!ctx) return SWG_ERR_INVALID;
  if (axis < 0 || axis > 2) return swg_set_error(ctx, SWG_ERR_INVALID, "axis must be 0, 1 or 2");
  if (scoring < 0 || scoring > 4) return swg_set_error(ctx, SWG_ERR_INVALID, "bad scoring");
  if (k_query == 0 || k_target == 0) return swg_set_error(ctx, SWG_ERR_INVALID, "k must be >= 1");
  if (n == 0) return SWG_OK;
  if (!q_start || !q_end || !t_start || !t_end || !identity || !keep_out)
    return swg_set_error(ctx, SWG_ERR_INVALID, "NULL array");
  if (n >= (uint64_t(1) << 31)) return swg_set_error(ctx, SWG_ERR_RANGE, "too many mappings");
  SWG_HIP(ctx, hipSetDevice(ctx->device));
  std::vector<uint32_t> h(4 * n);
  uint32_t mx = 0;
  SWG_TRY(swg_narrow_coords(ctx, n, q_start, q_end, h.data(), h.data() + n, "query"));
  SWG_TRY(swg_narrow_coords(ctx, n, t_start, t_end, h.data() + 2 * n, h.data() + 3 * n, "target"));
  for (uint64_t i = 0; i < 4 * n; ++i)
    if (h[i] > mx) mx = h[i];
  const int pos_bits = swg_bits_for(mx) ? swg_bits_for(mx) : 1;
  hipStream_t st = ctx->stream;
  if (ctx->arena_cap == 0) SWG_TRY(swg_arena_reserve(ctx, (size_t)n * 200 + (size_t(8) << 20)));
  return swg_run_with_arena(ctx, [&]() -> int {
    uint32_t* d_c = swg_alloc<uint32_t>(ctx, 4 * n);
    double* d_id = swg_alloc<double>(ctx, n);
    uint64_t* d_key = swg_alloc<uint64_t>(ctx, n);
    uint64_t* d_seg = swg_alloc<uint64_t>(ctx, n);
    uint8_t* d_kq = swg_alloc<uint8_t>(ctx, n);
    uint8_t* d_kt = swg_alloc<uint8_t>(ctx, n);
    SWG_CHECK_ARENA(ctx);
    SWG_HIP(ctx, hipMemcpyAsync(d_c, h.data(), 4 * n * sizeof(uint32_t), hipMemcpyHostToDevice, st));
    SWG_HIP(ctx, hipMemcpyAsync(d_id, identity, n * sizeof(double), hipMemcpyHostToDevice, st));
    SWG_HIP(ctx, hipMemsetAsync(d_seg, 0, n * sizeof(uint64_t), st));
    SWG_TRY(swg_score_keys(ctx, n, d_c, d_c + n, d_id, scoring, d_key));
    swg_axis_input ax;
    ax.n = n;
    ax.seg = d_seg;
    ax.seg_bits = 1;
    ax.pos_bits = pos_bits;
    ax.score_key = d_key;
    ax.alive = nullptr;
    uint8_t* result = d_kq;
    if (axis == 0 || axis == 2) {
      ax.start = d_c;
      ax.end = d_c + n;
      SWG_TRY(swg_sweep_axis(ctx, ax, k_query, thr, d_kq));
    }
    if (axis == 1 || axis == 2) {
      ax.start = d_c + 2 * n;
      ax.end = d_c + 3 * n;
      ax.alive = axis == 2 ? d_kq : nullptr;  // plane_sweep_both: target sweep over the query survivors
      SWG_TRY(swg_sweep_axis(ctx, ax, k_target, thr, d_kt));
      result = d_kt;
    }
    SWG_HIP(ctx, hipMemcpyAsync(keep_out, result, n, hipMemcpyDeviceToHost, st));
    SWG_HIP(ctx, hipStreamSynchronize(st));
    return SWG_OK;
  });
}

extern "C" int swg_log(swg_ctx* ctx, uint64_t n, const double* x, double* y) {
  if (!ctx) return SWG_ERR_INVALID;
  if (n == 0) return SWG_OK;
  if (!x || !y) return swg_set_error(ctx, SWG_ERR_INVALID, "NULL array");
  SWG_HIP(ctx, hipSetDevice(ctx->device));
  if (ctx->arena_cap == 0) SWG_TRY(swg_arena_reserve(ctx, (size_t)n * 16 + (size_t(8) << 20)));
  return swg_run_with_arena(ctx, [&]() -> int {
    double* dx = swg_alloc<double>(ctx, n);
    double* dy = swg_alloc<double>(ctx, n);
    SWG_CHECK_ARENA(ctx);
    SWG_HIP(ctx, hipMemcpyAsync(dx, x, n * 8, hipMemcpyHostToDevice, ctx->stream));
    SWG_LAUNCH(ctx, "log", log_kernel<<<nblk(n), EW, 0, ctx->stream>>>(n, dx, dy));
    SWG_KERNEL_CHECK(ctx);
    SWG_HIP(ctx, hipMemcpyAsync(y, dy, n * 8, hipMemcpyDeviceToHost, ctx->stream));
    SWG_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return SWG_OK;
  });
}

extern "C" int swg_log_range(swg_ctx* ctx, uint64_t first, uint64_t stride, uint64_t n, double* y) {
  if (!ctx) return SWG_ERR_INVALID;
  if (n == 0) return SWG_OK;
  if (!y) return swg_set_error(ctx, SWG_ERR_INVALID, "NULL array");
  SWG_HIP(ctx, hipSetDevice(ctx->device));
  if (ctx->arena_cap == 0) SWG_TRY(swg_arena_reserve(ctx, (size_t)n * 8 + (size_t(8) << 20)));
  return swg_run_with_arena(ctx, [&]() -> int {
    double* dy = swg_alloc<double>(ctx, n);
    SWG_CHECK_ARENA(ctx);
    SWG_LAUNCH(ctx, "log_range", log_range_kernel<<<nblk(n), EW, 0, ctx->stream>>>(first, stride, n, dy));
    SWG_KERNEL_CHECK(ctx);
    SWG_HIP(ctx, hipMemcpyAsync(y, dy, n * 8, hipMemcpyDeviceToHost, ctx->stream));
    SWG_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return SWG_OK;
  });
}
