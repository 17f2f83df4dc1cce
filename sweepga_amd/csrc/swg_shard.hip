// One record set over several devices of a node: host-side partition (threads, csrc/host/shard_host.h),
// one swg_filter per context on its own host thread, host-side merge (threads).  No collective: genome pairs are independent units of the
// filter -- every sweep segment (src/paf_filter.rs:1037-1100), chain group (:761-770), scaffold chromosome pair
// (src/plane_sweep_scaffold.rs:116-130) and rescue pair (src/paf_filter.rs:625-629) nests inside one pair.
//
// Chain numbers are global in the reference: kept chains are numbered genome pair by genome pair, in the order
// the pairs first appear (src/paf_filter.rs:517-521 over plane_sweep_scaffolds' output order).  After the
// gather every shard-local number is shifted by the number of kept chains of all pairs that appear earlier.
// That is exact when the reference's two genome-prefix rules (last '#' vs first two '#' parts) induce the same
// partition of the sequences; otherwise the call falls back to one device.
#include <algorithm>
#include <cstdlib>
#include <thread>
#include <new>
#include <system_error>
#include <vector>

#include "host/shard_host.h"
#include "host/stream_plan.h"
#include "swg_internal.h"
#include "swg_pipeline.h"

using swg_streamed::same_partition;

extern "C" int swg_filter_multi(swg_ctx* const* ctxs, int n_ctx, const swg_records* r, const swg_config* cfg, uint8_t* status_out,
                                uint32_t* chain_out, swg_stats* stats) {
  if (!ctxs || n_ctx < 1 || !ctxs[0]) return SWG_ERR_INVALID;
  swg_ctx* ctx0 = ctxs[0];
  if (!r || !cfg) return swg_set_error(ctx0, SWG_ERR_INVALID, "records/config is NULL");
  for (int d = 0; d < n_ctx; ++d)
    if (!ctxs[d]) return swg_set_error(ctx0, SWG_ERR_INVALID, "context %d is NULL", d);
  const uint64_t n = r->n;
  if (n_ctx == 1 || n == 0 || !r->seq_genome_last || !r->seq_genome_two || !same_partition(r))
    return swg_filter(ctx0, r, cfg, status_out, chain_out, stats);
  if (n >= (uint64_t(1) << 31)) return swg_set_error(ctx0, SWG_ERR_RANGE, "more than 2^31-1 records");
  if (!r->q_id || !r->t_id || !r->q_start || !r->q_end || !r->t_start || !r->t_end || !r->matches || !r->block_len ||
      !r->strand || !status_out || !chain_out)   // (identity may be NULL: derived on the device, see swg_records)
    return swg_set_error(ctx0, SWG_ERR_INVALID, "a record column or an output buffer is NULL");
  // Records grouped by query genome (what an aligner writes): ranges of whole query genomes go to the devices as slices of
  // the caller's own columns, uploads overlapped with the filter, results straight into the caller's arrays -- no host-side
  // scatter or merge (csrc/swg_stream.hip).  Otherwise: the scatter path below.
  {
    if (cfg->scoring_function < 0 || cfg->scoring_function > 4 || cfg->mapping_filter_mode < 0 || cfg->mapping_filter_mode > 2 ||
        cfg->scaffold_filter_mode < 0 || cfg->scaffold_filter_mode > 2)
      return swg_set_error(ctx0, SWG_ERR_INVALID, "bad scoring_function / filter mode");
    int taken = 0;
    const int rc = swg_stream_try(ctxs, n_ctx, r, cfg, status_out, chain_out, stats, &taken);
    if (rc != SWG_OK || taken) return rc;
  }
  // ---- plan + the shards' index lists on host threads (csrc/host/shard_host.h); the records themselves go from the caller's
  // columns to every device through its context's pinned ring (swg_filter_gathered).  SWG_MULTI_SCATTER=1: rounds 2-5's way --
  // every shard copied into host columns of its own first (a comparison knob)
  static const bool copy_first = getenv("SWG_MULTI_SCATTER") != nullptr;
  swg_shard::Plan P;
  std::vector<swg_shard::Shard> sh;
  try {
    if (!swg_shard::make_plan(*r, *cfg, n_ctx, swg_shard::default_threads(n), &P))
      return swg_set_error(ctx0, SWG_ERR_INVALID, "sequence id out of range (record %llu)", (unsigned long long)P.bad_record);
    if (copy_first) swg_shard::scatter(*r, P, &sh); else swg_shard::scatter_indices(P, &sh);
  } catch (const std::bad_alloc&) {
    return swg_set_error(ctx0, SWG_ERR_OOM, "out of host memory while sharding %llu records", (unsigned long long)n);
  } catch (const std::system_error& e) {
    return swg_set_error(ctx0, SWG_ERR_OOM, "cannot start host threads: %s", e.what());
  }

  // ---- one host thread per device: filter the shard, keep the results
  try {
    swg_shard::run(n_ctx, [&](int s) {
      swg_shard::Shard& S = sh[s];
      if (S.m == 0) return;
      if (copy_first) {
        const swg_records sub = S.view(*r);
        S.rc = swg_filter(ctxs[s], &sub, cfg, S.status.data(), S.chain.data(), &S.stats);
      } else {
        // (host threads of this device's gathering: the machine's threads shared out among the devices, at most 8 each)
        unsigned hc = std::thread::hardware_concurrency();
        int per = hc ? (int)(hc / (unsigned)n_ctx) : 1;
        per = per < 1 ? 1 : (per > 8 ? 8 : per);
        S.rc = swg_filter_gathered(ctxs[s], r, S.idx.data(), S.m, cfg, S.status.data(), S.chain.data(), &S.stats, per);
      }
    });
  } catch (const std::system_error& e) {
    return swg_set_error(ctx0, SWG_ERR_OOM, "cannot start host threads: %s", e.what());
  }
  for (int s = 0; s < n_ctx; ++s)
    if (sh[s].rc != SWG_OK) {
      const std::string msg = swg_last_error(ctxs[s]);
      return swg_set_error(ctx0, sh[s].rc, "shard %d of %d: %s", s, n_ctx, msg.c_str());
    }

  // ---- merge (threads): record order, global chain numbers
  try {
    swg_shard::merge(P, sh, status_out, chain_out);
  } catch (const std::bad_alloc&) {
    return swg_set_error(ctx0, SWG_ERR_OOM, "out of host memory while merging the shards");
  } catch (const std::system_error& e) {
    return swg_set_error(ctx0, SWG_ERR_OOM, "cannot start host threads: %s", e.what());
  }
  if (stats) {
    *stats = swg_stats{};
    stats->n_in = n;
    for (int s = 0; s < n_ctx; ++s) {
      const swg_stats& a = sh[s].stats;
      stats->n_retained += a.n_retained;
      stats->n_swept += a.n_swept;
      stats->n_chains += a.n_chains;
      stats->n_chains_kept += a.n_chains_kept;
      stats->n_out += a.n_out;
      stats->device_ms = std::max(stats->device_ms, a.device_ms);  // devices run concurrently
      stats->h2d_ms = std::max(stats->h2d_ms, a.h2d_ms);
      stats->d2h_ms = std::max(stats->d2h_ms, a.d2h_ms);
    }
  }
  return SWG_OK;
}

// RecordMeta's own widths: rebased on the host first (ctxs[0] keeps the 32-bit columns), then the same sharding.
extern "C" int swg_filter_multi64(swg_ctx* const* ctxs, int n_ctx, const swg_records64* r, const swg_config* cfg, uint8_t* status_out,
                                  uint32_t* chain_out, swg_stats* stats) {
  if (!ctxs || n_ctx < 1 || !ctxs[0]) return SWG_ERR_INVALID;
  swg_records v;
  SWG_TRY(swg_rebase_host(ctxs[0], r, cfg, &v));
  const int rc = swg_filter_multi(ctxs, n_ctx, &v, cfg, status_out, chain_out, stats);
  swg_narrow_release(ctxs[0]);
  return rc;
}
