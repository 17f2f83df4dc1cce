// Streamed host path of swg_filter / swg_filter_multi: the record set is cut into ranges of whole query genomes
// (csrc/host/stream_plan.h) and range k + 1 crosses PCIe while range k is being filtered.
//
//   per device   one uploader thread issues the H2D copies of the device's ranges in order on a copy stream (a slice of each
//                of the caller's columns -> the same slice of the device's columns; an event per range), the calling thread
//                runs swg_filter_device range by range on the context's stream (each waits for its range's event) and copies
//                the range's status / chain slice straight back into the caller's output arrays.  No host-side scatter, no
//                merge, no staging copy of the records: what SURVEY.md 8(e) asks of the partitioning ("one contiguous SoA
//                slice per device, hipMemcpyAsync ... on per-device streams") without the 0.4 s per 10^8 records that
//                materialising shard columns on the host costs (csrc/host/shard_host.h, still the path for inputs that are
//                not grouped by query genome).
//   chain ids    local to a range; global number = local + kept chains of all earlier ranges (stream_plan.h has the
//                argument).  One device: added on the device before the range is copied back; several: on host threads after
//                the last device has finished (only then are all counts known).
//
// The source columns are pageable: the HIP runtime stages them through its own pinned buffers, measured at the PCIe rate
// (4.7 GB in 72 ms = 65 GB/s, profiles/README.md), so a second, library-owned pinned ring would only add a copy.
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstring>
#include <string>
#include <mutex>
#include <thread>
#include <vector>

#include "host/stream_plan.h"
#include "swg_internal.h"
#include "swg_pipeline.h"

namespace {

constexpr int EW = 256;

__global__ __launch_bounds__(EW) void chain_shift_kernel(uint64_t n, uint32_t* __restrict__ chain, uint32_t base) {
  const uint64_t i = (uint64_t)blockIdx.x * EW + threadIdx.x;
  if (i < n) {
    const uint32_t c = chain[i];
    if (c) chain[i] = c + base;
  }
}

struct DeviceRun {
  swg_ctx* ctx = nullptr;
  std::vector<int> mine;            // chunk indices, ascending
  std::vector<uint64_t> local_off;  // [mine.size() + 1] offsets of the chunks in the device's columns
  std::vector<swg_stats> cstats;    // per chunk of `mine`
  int rc = SWG_OK;
  double h2d_ms = 0.0, d2h_ms = 0.0, device_ms = 0.0;
};

// Everything one device does.  `renumber_here`: chain numbers are made global on the device (single-device runs).
void run_device(DeviceRun& D, const swg_records* r, const swg_config* cfg, const std::vector<swg_streamed::Chunk>& chunks,
                uint8_t* status_out, uint32_t* chain_out, bool renumber_here) {
  swg_ctx* ctx = D.ctx;
  const size_t nc = D.mine.size();
  D.cstats.assign(nc, swg_stats{});
  if (nc == 0) return;
  auto fail = [&](int rc) { D.rc = rc; };
  if (hipSetDevice(ctx->device) != hipSuccess) return fail(swg_set_error(ctx, SWG_ERR_HIP, "hipSetDevice failed"));
  const bool scaffold = cfg->scaffold_gap != 0;
  bool id_value, wid_value;
  swg_value_columns_needed(cfg, &id_value, &wid_value);  // (swg_filter.hip: the CLI defaults read none of the value columns)
  const bool send_identity = r->identity != nullptr;  // (the caller's own column: always -- see swg_value_columns_needed)
  const bool need_matches = wid_value || (!r->identity && id_value);
  const bool need_block = need_matches || cfg->min_block_length != 0;
  static const bool poison = getenv("SWG_POISON") != nullptr;
  uint64_t m = 0, longest = 0;
  D.local_off.assign(nc + 1, 0);
  for (size_t j = 0; j < nc; ++j) {
    const swg_streamed::Chunk& c = chunks[(size_t)D.mine[j]];
    m = (m + 63) & ~uint64_t(63);  // every range starts on a 64-record boundary of the device's columns (aligned slices)
    D.local_off[j] = m;
    m += c.hi - c.lo;
    longest = std::max(longest, c.hi - c.lo);
  }
  D.local_off[nc] = m;
  // device memory: the columns of all of this device's ranges (staging block of the context) + scratch for the longest one
  const bool dbg = getenv("SWG_DEBUG") != nullptr;
  const auto r0 = std::chrono::steady_clock::now();
  auto since = [&]() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - r0).count(); };
  if (int rc = swg_io_block_reserve(ctx, m, r->n_seq); rc != SWG_OK) return fail(rc);
  if (int rc = swg_filter_reserve_arena(ctx, longest, r, cfg, false); rc != SWG_OK) return fail(rc);
  if (dbg) fprintf(stderr, "[swg] device %d: memory for %llu records in %zu ranges (longest %llu) reserved at %.1f ms\n", ctx->device,
                   (unsigned long long)m, nc, (unsigned long long)longest, since());
  if (!ctx->copy_stream && hipStreamCreateWithFlags(&ctx->copy_stream, hipStreamNonBlocking) != hipSuccess)
    return fail(swg_set_error(ctx, SWG_ERR_HIP, "hipStreamCreate (copy stream) failed"));
  const size_t col4 = ((m * 4 + 255) & ~size_t(255)), col8 = ((m * 8 + 255) & ~size_t(255)), col1 = ((m + 255) & ~size_t(255)),
               seqt = (((size_t)r->n_seq * 4 + 255) & ~size_t(255));
  char* blk = ctx->io_block;
  size_t off = 0;
  auto take = [&](size_t bytes) {
    char* p = blk + off;
    off += bytes;
    return p;
  };
  // (same layout as swg_filter's block: every column is reserved whether this flag set uploads it or not)
  uint32_t* d_c4[8];
  for (auto& p : d_c4) p = reinterpret_cast<uint32_t*>(take(col4));  // q_id t_id q_start q_end t_start t_end matches block_len
  double* d_identity = reinterpret_cast<double*>(take(col8));
  uint8_t* d_strand = reinterpret_cast<uint8_t*>(take(col1));
  uint32_t* d_gl = reinterpret_cast<uint32_t*>(take(seqt));
  uint32_t* d_g2 = reinterpret_cast<uint32_t*>(take(seqt));
  uint8_t* d_status = reinterpret_cast<uint8_t*>(take(col1));
  uint32_t* d_chain = reinterpret_cast<uint32_t*>(take(col4));
  const uint32_t* const h_c4[8] = {r->q_id, r->t_id, r->q_start, r->q_end, r->t_start, r->t_end, r->matches, r->block_len};
  const bool want4[8] = {true, true, true, true, true, true, need_matches, need_block};

  std::vector<hipEvent_t> up_ev(nc, nullptr);
  hipEvent_t t_first = nullptr, t_last = nullptr;
  bool ev_ok = hipEventCreate(&t_first) == hipSuccess && hipEventCreate(&t_last) == hipSuccess;
  for (auto& e : up_ev) ev_ok = ev_ok && hipEventCreateWithFlags(&e, hipEventDisableTiming) == hipSuccess;
  auto destroy_events = [&]() {
    for (auto e : up_ev)
      if (e) (void)hipEventDestroy(e);
    if (t_first) (void)hipEventDestroy(t_first);
    if (t_last) (void)hipEventDestroy(t_last);
  };
  if (!ev_ok) {
    destroy_events();
    return fail(swg_set_error(ctx, SWG_ERR_HIP, "hipEventCreate failed"));
  }

  // ---- the uploader: ranges in order on the copy stream; `issued` = ranges whose event has been recorded
  std::mutex mu;
  std::condition_variable cv;
  size_t issued = 0;
  bool up_failed = false;
  std::atomic<bool> stop{false};
  hipError_t up_err = hipSuccess;  // what the uploader's failing call returned (HIP's last error is per thread: the filtering
                                   // thread cannot ask for it); written under mu next to up_failed
  auto uploader = [&]() {
    hipError_t e = hipSetDevice(ctx->device);
    auto step = [&](hipError_t x) {  // runs the call only while everything before it succeeded; keeps the first failure
      if (e == hipSuccess) e = x;
      return e == hipSuccess;
    };
    hipStream_t cs = ctx->copy_stream;
    if (e == hipSuccess) step(hipEventRecord(t_first, cs));
    if (e == hipSuccess) step(hipMemcpyAsync(d_gl, r->seq_genome_last, (size_t)r->n_seq * 4, hipMemcpyHostToDevice, cs));
    if (e == hipSuccess) step(hipMemcpyAsync(d_g2, r->seq_genome_two, (size_t)r->n_seq * 4, hipMemcpyHostToDevice, cs));
    for (size_t j = 0; j < nc && e == hipSuccess && !stop.load(std::memory_order_relaxed); ++j) {
      const swg_streamed::Chunk& c = chunks[(size_t)D.mine[j]];
      const uint64_t len = c.hi - c.lo, lo = c.lo, lo_d = D.local_off[j];
      for (int k = 0; k < 8 && e == hipSuccess; ++k)
        if (want4[k]) step(hipMemcpyAsync(d_c4[k] + lo_d, h_c4[k] + lo, len * 4, hipMemcpyHostToDevice, cs));
      if (e == hipSuccess && send_identity) step(hipMemcpyAsync(d_identity + lo_d, r->identity + lo, len * 8, hipMemcpyHostToDevice, cs));
      if (e == hipSuccess && poison) {  // (test knob: unsent columns full of 0xff bytes)
        for (int k = 6; k < 8 && e == hipSuccess; ++k)
          if (!want4[k]) step(hipMemsetAsync(d_c4[k] + lo_d, 0xff, len * 4, cs));
        if (e == hipSuccess && !send_identity) step(hipMemsetAsync(d_identity + lo_d, 0xff, len * 8, cs));
      }
      if (e == hipSuccess && scaffold) step(hipMemcpyAsync(d_strand + lo_d, r->strand + lo, len, hipMemcpyHostToDevice, cs));
      if (e == hipSuccess && j + 1 == nc) step(hipEventRecord(t_last, cs));
      if (e == hipSuccess) step(hipEventRecord(up_ev[j], cs));
      {
        std::lock_guard<std::mutex> g(mu);
        if (e == hipSuccess) {
          issued = j + 1;
        } else {
          up_failed = true;
          up_err = e;
        }
      }
      cv.notify_all();
    }
    if (e != hipSuccess) {
      {
        std::lock_guard<std::mutex> g(mu);
        up_failed = true;
        up_err = e;
      }
      cv.notify_all();
    }
  };
  std::thread up_thread;
  try {
    up_thread = std::thread(uploader);
  } catch (const std::system_error&) {
    uploader();  // no thread to be had: upload first, then filter (no overlap, same result)
  }

  // ---- the filter, range by range
  swg_records d = *r;
  d.seq_genome_last = d_gl;
  d.seq_genome_two = d_g2;
  uint32_t base = 0;
  int rc = SWG_OK;
  for (size_t j = 0; j < nc && rc == SWG_OK; ++j) {
    {
      std::unique_lock<std::mutex> g(mu);
      cv.wait(g, [&] { return issued > j || up_failed; });
      if (issued <= j) {
        rc = swg_set_error(ctx, SWG_ERR_HIP, "H2D copy of range %zu failed: %s", j, hipGetErrorString(up_err));
        break;
      }
    }
    if (hipStreamWaitEvent(ctx->stream, up_ev[j], 0) != hipSuccess) {
      rc = swg_set_error(ctx, SWG_ERR_HIP, "hipStreamWaitEvent failed");
      break;
    }
    const swg_streamed::Chunk& c = chunks[(size_t)D.mine[j]];
    const uint64_t len = c.hi - c.lo, lo_d = D.local_off[j];
    d.n = len;
    d.q_id = d_c4[0] + lo_d;
    d.t_id = d_c4[1] + lo_d;
    d.q_start = d_c4[2] + lo_d;
    d.q_end = d_c4[3] + lo_d;
    d.t_start = d_c4[4] + lo_d;
    d.t_end = d_c4[5] + lo_d;
    d.matches = d_c4[6] + lo_d;
    d.block_len = d_c4[7] + lo_d;
    d.identity = send_identity ? d_identity + lo_d : nullptr;  // (derived on the device: from the columns, or -- nobody reading it -- from anything)
    d.strand = d_strand + lo_d;
    swg_stats& st = D.cstats[j];
    const double waited = since();
    rc = swg_filter_device(ctx, &d, cfg, d_status + lo_d, d_chain + lo_d, &st);  // (with stats: returns when the range is done)
    if (rc != SWG_OK) break;
    if (dbg) fprintf(stderr, "[swg] device %d range %zu (%llu records): uploaded by %.1f ms, filtered by %.1f ms (device %.2f ms)\n", ctx->device, j,
                     (unsigned long long)len, waited, since(), st.device_ms);
    D.device_ms += st.device_ms;
    if (scaffold && renumber_here && base)
      SWG_LAUNCH(ctx, "chain_shift", chain_shift_kernel<<<(unsigned)((len + EW - 1) / EW), EW, 0, ctx->stream>>>(len, d_chain + lo_d, base));
    base += (uint32_t)st.n_chains_kept;
    const auto t0 = std::chrono::steady_clock::now();
    if (hipMemcpyAsync(status_out + c.lo, d_status + lo_d, len, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess ||
        (scaffold && hipMemcpyAsync(chain_out + c.lo, d_chain + lo_d, len * 4, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess) ||
        hipStreamSynchronize(ctx->stream) != hipSuccess) {
      rc = swg_set_error(ctx, SWG_ERR_HIP, "D2H copy failed: %s", hipGetErrorString(hipGetLastError()));
      break;
    }
    if (!scaffold) std::memset(chain_out + c.lo, 0, len * sizeof(uint32_t));  // no ch:Z: tags without scaffolding
    D.d2h_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  }
  stop.store(true);
  if (up_thread.joinable()) up_thread.join();
  (void)hipStreamSynchronize(ctx->copy_stream);
  if (rc == SWG_OK) {
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, t_first, t_last) == hipSuccess) D.h2d_ms = ms;
  }
  destroy_events();
  D.rc = rc;
}

}  // namespace

// 0 = the monolithic path should be taken (input not grouped by query genome, too small, switched off, ...)
int swg_stream_try(swg_ctx* const* ctxs, int n_ctx, const swg_records* r, const swg_config* cfg, uint8_t* status_out,
                   uint32_t* chain_out, swg_stats* stats, int* taken) {
  *taken = 0;
  const char* knob = getenv("SWG_STREAM");              // "0": never.  (read per call: tests switch it inside one process)
  const char* chunk_knob = getenv("SWG_STREAM_CHUNK");  // records per range (tests); also lifts the size threshold
  if (knob && knob[0] == '0') return SWG_OK;
  const uint64_t n = r->n;
  uint64_t target = chunk_knob ? strtoull(chunk_knob, nullptr, 10) : 0;
  if (!target) {
    // A range costs ~2.5 ms of launches and read-backs on top of its share of the kernels (~0.2 ms per 10^6 records) and its
    // upload takes ~0.75 ms per 10^6 records: only ranges of more than ~5 M records are filtered faster than the next one
    // arrives (measured: 10^7 records in five ranges of 2 M took 62 ms against 45-55 ms in one piece; 10^8 in 16 ranges of
    // 6.25 M took 67 ms of device time and the call became compute-bound; in 8 ranges it is copy-bound, 85 against 102 ms).
    // So: at least 6 M records per range, eight ranges per device when there are enough records, and no streaming at all
    // below two such ranges per device.
    // Several devices: the ranges also replace the host-side scatter and merge of the record set (0.4 s per 10^8 records,
    // csrc/host/shard_host.h), which is worth a few ranges' overhead at any size from a million records per device up.
    constexpr uint64_t MIN_RANGE = 6u << 20;
    if (n_ctx == 1) {
      if (n < 2 * MIN_RANGE) return SWG_OK;
      target = std::max<uint64_t>(n / 8, MIN_RANGE);
    } else {
      if (n < (uint64_t(1) << 20) * (uint64_t)n_ctx) return SWG_OK;
      target = std::max<uint64_t>(n / (uint64_t)(8 * n_ctx), uint64_t(1) << 20);
    }
  }
  if (!r->seq_genome_last || !r->seq_genome_two || !swg_streamed::same_partition(r)) return SWG_OK;
  std::vector<swg_streamed::Chunk> chunks;
  std::vector<std::vector<int>> per;
  try {
    unsigned hc = std::thread::hardware_concurrency();
    if (!swg_streamed::plan(n, r->q_id, r->seq_genome_two, r->n_seq, r->n_genome_two, target, hc ? (int)std::min(hc, 32u) : 1, &chunks))
      return SWG_OK;
    if (chunks.size() < 2 || (int)chunks.size() < n_ctx) return SWG_OK;
    per = swg_streamed::assign(chunks, n_ctx);
  } catch (const std::bad_alloc&) {
    return swg_set_error(ctxs[0], SWG_ERR_OOM, "out of host memory while planning the streamed upload");
  }
  *taken = 1;
  const bool dbg = getenv("SWG_DEBUG") != nullptr;
  const auto w0 = std::chrono::steady_clock::now();
  auto lap = [&](const char* what) {
    if (dbg) fprintf(stderr, "[swg] streamed call: %s at %.1f ms\n", what, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - w0).count());
  };
  if (dbg) fprintf(stderr, "[swg] streamed call: %zu ranges over %d device(s), target %llu records\n", chunks.size(), n_ctx, (unsigned long long)target);
  std::vector<DeviceRun> runs((size_t)n_ctx);
  for (int d = 0; d < n_ctx; ++d) {
    runs[(size_t)d].ctx = ctxs[d];
    runs[(size_t)d].mine = per[(size_t)d];
  }
  const bool one = n_ctx == 1;
  try {
    swg_host::run(n_ctx, [&](int d) { run_device(runs[(size_t)d], r, cfg, chunks, status_out, chain_out, one); });
  } catch (const std::bad_alloc&) {
    return swg_set_error(ctxs[0], SWG_ERR_OOM, "out of host memory in the streamed filter");
  }
  lap("devices done");
  for (int d = 0; d < n_ctx; ++d)
    if (runs[(size_t)d].rc != SWG_OK) {
      if (d == 0) return runs[0].rc;
      const std::string msg = swg_last_error(ctxs[d]);
      return swg_set_error(ctxs[0], runs[(size_t)d].rc, "device %d of %d: %s", d, n_ctx, msg.c_str());
    }
  // per-range statistics back in range order
  std::vector<swg_stats> cs(chunks.size());
  for (int d = 0; d < n_ctx; ++d)
    for (size_t j = 0; j < runs[(size_t)d].mine.size(); ++j) cs[(size_t)runs[(size_t)d].mine[j]] = runs[(size_t)d].cstats[j];
  if (!one && cfg->scaffold_gap != 0) {
    // global chain numbers on host threads: range k's numbers are shifted by the kept chains of ranges 0..k-1
    std::vector<uint32_t> base(chunks.size(), 0);
    for (size_t k = 1; k < chunks.size(); ++k) base[k] = base[k - 1] + (uint32_t)cs[k - 1].n_chains_kept;
    unsigned hc = std::thread::hardware_concurrency();
    const int T = hc ? (int)std::min(hc, 32u) : 1;
    try {
      swg_host::run(T, [&](int t) {
        for (size_t k = 1; k < chunks.size(); ++k) {
          if (!base[k]) continue;
          const uint64_t len = chunks[k].hi - chunks[k].lo;
          const uint64_t a = chunks[k].lo + len / (uint64_t)T * (uint64_t)t + std::min<uint64_t>((uint64_t)t, len % (uint64_t)T);
          const uint64_t b = chunks[k].lo + len / (uint64_t)T * (uint64_t)(t + 1) + std::min<uint64_t>((uint64_t)(t + 1), len % (uint64_t)T);
          for (uint64_t i = a; i < b; ++i)
            if (chain_out[i]) chain_out[i] += base[k];
        }
      });
    } catch (const std::bad_alloc&) {
      return swg_set_error(ctxs[0], SWG_ERR_OOM, "out of host memory while renumbering");
    }
  }
  if (stats) {
    *stats = swg_stats{};
    stats->n_in = n;
    for (const swg_stats& a : cs) {
      stats->n_retained += a.n_retained;
      stats->n_swept += a.n_swept;
      stats->n_chains += a.n_chains;
      stats->n_chains_kept += a.n_chains_kept;
      stats->n_out += a.n_out;
    }
    for (const DeviceRun& D : runs) {  // devices run concurrently
      stats->device_ms = std::max(stats->device_ms, D.device_ms);
      stats->h2d_ms = std::max(stats->h2d_ms, D.h2d_ms);
      stats->d2h_ms = std::max(stats->d2h_ms, D.d2h_ms);
    }
  }
  return SWG_OK;
}

// The plan by itself (host code, no GPU): bounds_out receives n_chunks + 1 record indices (capacity `cap` entries).
// *n_chunks = 0: the records are not grouped by query genome (or the two prefix rules disagree) and the call would run as one
// piece.  Exists so that the partitioning can be tested and inspected without a device.
extern "C" int swg_stream_plan(const swg_records* r, uint64_t target_records, uint64_t* bounds_out, uint64_t cap, uint64_t* n_chunks) {
  if (!r || !n_chunks) return SWG_ERR_INVALID;
  *n_chunks = 0;
  if (r->n == 0) return SWG_OK;
  if (!r->q_id || !r->seq_genome_last || !r->seq_genome_two || r->n_seq == 0) return SWG_ERR_INVALID;
  try {
    if (!swg_streamed::same_partition(r)) return SWG_OK;
    std::vector<swg_streamed::Chunk> chunks;
    unsigned hc = std::thread::hardware_concurrency();
    if (!swg_streamed::plan(r->n, r->q_id, r->seq_genome_two, r->n_seq, r->n_genome_two, target_records ? target_records : 1,
                          hc ? (int)std::min(hc, 32u) : 1, &chunks))
      return SWG_OK;
    if (bounds_out) {
      if (cap < chunks.size() + 1) return SWG_ERR_INVALID;
      for (size_t k = 0; k < chunks.size(); ++k) bounds_out[k] = chunks[k].lo;
      bounds_out[chunks.size()] = chunks.back().hi;
    }
    *n_chunks = chunks.size();
  } catch (const std::bad_alloc&) {
    return SWG_ERR_OOM;
  }
  return SWG_OK;
}
