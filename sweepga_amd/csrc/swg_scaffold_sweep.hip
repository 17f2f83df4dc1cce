// Scaffold stage, part 3 (src/plane_sweep_scaffold.rs:47-251): plane_sweep_both per chromosome pair over the span /
// identity-filtered chains on the sweep kernels, then chain_N numbering in plane_sweep_scaffolds' output order
// (genome pair -> chromosome pair -> index).
#include "swg_scaffold_internal.h"

namespace swg_scaf {
namespace {

// ---- scaffold sweep + numbering ------------------------------------------------------------------------------
__global__ __launch_bounds__(EW) void chain_seg_kernel(uint64_t nc, const uint32_t* __restrict__ C_qid,
                                                       const uint32_t* __restrict__ C_tid, uint32_t n_seq,
                                                       uint64_t* __restrict__ seg) {
  uint64_t c = (uint64_t)blockIdx.x * EW + threadIdx.x;
  if (c < nc) seg[c] = (uint64_t)C_qid[c] * n_seq + C_tid[c];
}
// score key + the four coordinates of a chain in one 32-byte slot, as prepare_kernel writes them for the mappings: the sweep
// then sorts the chains' begins as packed 8-byte words and gathers one sector per begin (swg_key_ends, swg_internal.h)
__global__ __launch_bounds__(EW) void chain_slots_kernel(uint64_t nc, const uint64_t* __restrict__ skey,
                                                         const uint32_t* __restrict__ qs, const uint32_t* __restrict__ qe,
                                                         const uint32_t* __restrict__ ts, const uint32_t* __restrict__ te,
                                                         swg_key_ends* __restrict__ slots) {
  uint64_t c = (uint64_t)blockIdx.x * EW + threadIdx.x;
  if (c >= nc) return;
  swg_key_ends ke;
  ke.key = skey[c];
  ke.start[0] = qs[c];
  ke.start[1] = ts[c];
  ke.end[0] = qe[c];
  ke.end[1] = te[c];
  ke.pad[0] = ke.pad[1] = 0;
  slots[c] = ke;
}
// after the stable sort of chains by chromosome pair: run heads
__global__ __launch_bounds__(EW) void run_flag_kernel(uint64_t nc, const uint64_t* __restrict__ sorted_seg,
                                                      uint32_t* __restrict__ flag) {
  uint64_t s = (uint64_t)blockIdx.x * EW + threadIdx.x;
  if (s < nc) flag[s] = (s == 0 || sorted_seg[s - 1] != sorted_seg[s]) ? 1u : 0u;
}
// Chains here are already the span/identity-filtered ones, in index order, and the sort by chromosome pair is
// stable: the first chain of a run is the pair's first appearance (plane_sweep_scaffold.rs:116-130 insertion
// order), and the genome pair's (first two '#' parts) first appearance is the minimum over its runs' heads.
__global__ __launch_bounds__(EW) void first_appearance_kernel(uint64_t nc, const uint32_t* __restrict__ sorted_c,
                                                              const uint32_t* __restrict__ run_excl,
                                                              const uint32_t* __restrict__ run_flag,
                                                              const uint32_t* __restrict__ C_qid,
                                                              const uint32_t* __restrict__ C_tid,
                                                              const uint32_t* __restrict__ seq_genome2,
                                                              uint32_t* __restrict__ pair_first,
                                                              PairTable gp2_first) {
  uint64_t s = (uint64_t)blockIdx.x * EW + threadIdx.x;
  if (s >= nc) return;
  if (run_flag[s]) {
    const uint32_t c = sorted_c[s];
    const uint32_t run = run_excl[s];
    pair_first[run] = c;
    atomicMin(pair_slot(gp2_first, seq_genome2[C_qid[c]], seq_genome2[C_tid[c]]), c);
  }
}
// Numbering without sorting the kept chains.  plane_sweep_scaffolds returns them genome pair by genome pair (first
// appearance), chromosome pair by chromosome pair (first appearance) inside, index order inside that
// (plane_sweep_scaffold.rs:116-130, 204-251).  The chains of a chromosome pair are one run of the seg-sorted order, in index
// order (stable sort), so: the RUNS are sorted by (genome pair's first chain, run's first chain) -- a few thousand keys instead
// of every kept chain (round 3 sorted all of them: seven 12-byte passes over 1.8*10^7 on S-pan) -- a run's base is the kept
// chains of the runs before it, and a chain adds its rank among the kept chains of its run.
__global__ __launch_bounds__(EW) void kept_sorted_kernel(uint64_t nc, const uint32_t* __restrict__ sorted_c,
                                                         const uint8_t* __restrict__ kept, const uint32_t* __restrict__ run_flag,
                                                         const uint32_t* __restrict__ run_excl, uint32_t* __restrict__ ks,
                                                         uint32_t* __restrict__ run_start) {
  uint64_t s = (uint64_t)blockIdx.x * EW + threadIdx.x;
  if (s >= nc) return;
  ks[s] = kept[sorted_c[s]] ? 1u : 0u;
  if (run_flag[s]) run_start[run_excl[s]] = (uint32_t)s;
}
__global__ __launch_bounds__(EW) void run_keys_kernel(uint32_t n_runs, const uint32_t* __restrict__ pair_first,
                                                      const uint32_t* __restrict__ run_start, const uint32_t* __restrict__ kex,
                                                      uint32_t nc, uint32_t nk, PairTable gp2_first,
                                                      const uint32_t* __restrict__ C_qid, const uint32_t* __restrict__ C_tid,
                                                      const uint32_t* __restrict__ seq_genome2, int c_bits,
                                                      uint64_t* __restrict__ key, uint32_t* __restrict__ val,
                                                      uint32_t* __restrict__ run_kept) {
  uint32_t r = blockIdx.x * EW + threadIdx.x;
  if (r >= n_runs) return;
  const uint32_t c = pair_first[r];
  const uint32_t g2 = pair_get(gp2_first, seq_genome2[C_qid[c]], seq_genome2[C_tid[c]]);
  key[r] = ((uint64_t)g2 << c_bits) | c;
  val[r] = r;
  const uint32_t b = kex[run_start[r]];
  const uint32_t e = r + 1 < n_runs ? kex[run_start[r + 1]] : nk;
  (void)nc;
  run_kept[r] = e - b;
}
__global__ __launch_bounds__(EW) void run_sizes_sorted_kernel(uint32_t n_runs, const uint32_t* __restrict__ r_sorted,
                                                              const uint32_t* __restrict__ run_kept, uint32_t* __restrict__ sizes) {
  uint32_t k = blockIdx.x * EW + threadIdx.x;
  if (k < n_runs) sizes[k] = run_kept[r_sorted[k]];
}
__global__ __launch_bounds__(EW) void run_base_kernel(uint32_t n_runs, const uint32_t* __restrict__ r_sorted,
                                                      const uint32_t* __restrict__ base_sorted, uint32_t* __restrict__ run_base) {
  uint32_t k = blockIdx.x * EW + threadIdx.x;
  if (k < n_runs) run_base[r_sorted[k]] = base_sorted[k];
}
__global__ __launch_bounds__(EW) void assign_numbers_runs_kernel(uint64_t nc, const uint32_t* __restrict__ sorted_c,
                                                                 const uint32_t* __restrict__ ks, const uint32_t* __restrict__ kex,
                                                                 const uint32_t* __restrict__ run_flag,
                                                                 const uint32_t* __restrict__ run_excl,
                                                                 const uint32_t* __restrict__ run_start,
                                                                 const uint32_t* __restrict__ run_base, uint32_t* __restrict__ C_num) {
  uint64_t s = (uint64_t)blockIdx.x * EW + threadIdx.x;
  if (s >= nc) return;
  const uint32_t c = sorted_c[s];
  uint32_t num = 0;
  if (ks[s]) {
    const uint32_t r = run_excl[s] + run_flag[s] - 1;
    num = run_base[r] + (kex[s] - kex[run_start[r]]) + 1;
  }
  C_num[c] = num;  // (every chain is written: no memset of the column)
}
}  // namespace

// plane_sweep_both (plane_sweep_exact.rs:436-461) over nc chains in segments (one per chromosome pair; seg[c] < 2^seg_bits),
// index order inside a segment = the reference's order of the pair's chains.  kept[c] = 1 iff chain c survives both axes.
int scaffold_sweep_segments(swg_ctx* ctx, uint64_t nc, const uint64_t* seg, int seg_bits, const uint32_t* qs, const uint32_t* qe,
                            const uint32_t* ts, const uint32_t* te, const double* wid, uint64_t kq, uint64_t kt, double thr, int scoring,
                            int pos_bits, uint8_t* kept, const void* runs, uint32_t n_runs) {
  hipStream_t st = ctx->stream;
  uint64_t* skey = swg_alloc<uint64_t>(ctx, nc);
  uint8_t* keep_q = swg_alloc<uint8_t>(ctx, nc);
  SWG_CHECK_ARENA(ctx);
  SWG_TRY(swg_score_keys(ctx, nc, qs, qe, wid, scoring, skey));
  swg_axis_input ax;
  ax.n = nc;
  ax.seg = seg;
  ax.seg_bits = seg_bits;
  ax.pos_bits = pos_bits;
  ax.score_key = skey;
  // runs: every segment is one stretch of the table (the pair-resident stage's chain table): the axes sort their begins segment
  // by segment in LDS (swg_segsort.hip) and read plain columns
  static const int seg_knob = getenv("SWG_SEG_SORT") ? atoi(getenv("SWG_SEG_SORT")) : -1;
  uint32_t* run_alive = nullptr;
  if (runs && n_runs && seg_knob != 0 && nc > 16384) {
    run_alive = swg_alloc<uint32_t>(ctx, n_runs);
    SWG_CHECK_ARENA(ctx);
    SWG_TRY(swg_seg_run_alive(ctx, runs, n_runs, nullptr, run_alive));
    ax.seg_runs = runs;
    ax.n_seg_runs = n_runs;
    ax.seg_run_alive = run_alive;
    ax.n_alive = nc;
  } else if (kq != SWG_K_INF || kt != SWG_K_INF) {  // a sorting sweep will run: the packed form of its inputs
    swg_key_ends* slots = swg_alloc<swg_key_ends>(ctx, nc);
    SWG_CHECK_ARENA(ctx);
    SWG_LAUNCH(ctx, "chain_slots", chain_slots_kernel<<<nblk(nc), EW, 0, st>>>(nc, skey, qs, qe, ts, te, slots));
    SWG_KERNEL_CHECK(ctx);
    ax.packed = slots;
  }
  ax.alive = nullptr;
  ax.start = qs;
  ax.end = qe;
  ax.packed_end = 0;
  SWG_TRY(swg_sweep_axis(ctx, ax, kq, thr, keep_q));
  ax.alive = keep_q;  // plane_sweep_both: the target sweep sees the query survivors only
  if (run_alive) {
    SWG_TRY(swg_seg_run_alive(ctx, runs, n_runs, keep_q, run_alive));
    ax.n_alive = ~0ull;  // (their number comes back with the segment plan)
  }
  ax.start = ts;
  ax.end = te;
  ax.packed_end = 1;
  SWG_TRY(swg_sweep_axis(ctx, ax, kt, thr, kept));
  return SWG_OK;
}

// plane_sweep_scaffolds (plane_sweep_scaffold.rs:47-251) + chain numbering.  Chains are given in the
// reference's all_chains order (their index is the plane sweep's tie-break `idx`).
// Outputs: C_kept[c] (u8), C_num[c] (1-based position in the reference's output Vec, 0 if dropped).
int scaffold_sweep_and_number(swg_ctx* ctx, const ChainTable& T, uint32_t n_seq, const uint32_t* seq_genome2,
                              uint32_t n_g2, int mode, uint64_t max_q, uint64_t max_t, double thr, int scoring,
                              int pos_bits, uint8_t* C_kept, uint32_t* C_num, uint64_t* n_kept_out) {
  hipStream_t st = ctx->stream;
  *n_kept_out = 0;
  if (T.nc == 0) return SWG_OK;
  uint64_t kq, kt;
  if (mode == SWG_MODE_ONE_TO_ONE) {
    kq = 1;
    kt = 1;
  } else {
    kq = max_q ? max_q : SWG_K_INF;
    kt = max_t ? max_t : SWG_K_INF;
  }
  // every chain of T takes part (the span / identity filter was applied when T was built); index order = all_chains order
  // restricted to them, which is all the plane sweep's index tie-break needs
  const uint64_t nc = T.nc;
  uint64_t* d_tot = swg_alloc<uint64_t>(ctx, 1);
  uint32_t *qid = T.qid, *tid = T.tid;
  uint64_t* seg = swg_alloc<uint64_t>(ctx, nc);
  uint8_t* kept = C_kept;
  uint32_t* num = C_num;
  SWG_CHECK_ARENA(ctx);
  SWG_LAUNCH(ctx, "chain_seg", chain_seg_kernel<<<nblk(nc), EW, 0, st>>>(nc, qid, tid, n_seq, seg));
  SWG_KERNEL_CHECK(ctx);
  const int seg_bits = swg_bits_for((uint64_t)n_seq * n_seq);
  SWG_TRY(scaffold_sweep_segments(ctx, nc, seg, seg_bits, T.qs, T.qe, T.ts, T.te, T.wid, kq, kt, thr, scoring, pos_bits,
                                  kept));

  // ---- numbering -------------------------------------------------------------------------------------
  uint64_t* seg_sorted = swg_alloc<uint64_t>(ctx, nc);
  uint64_t* seg_tmp = swg_alloc<uint64_t>(ctx, nc);
  uint32_t* c_sorted = swg_alloc<uint32_t>(ctx, nc);
  uint32_t* c_tmp = swg_alloc<uint32_t>(ctx, nc);
  uint32_t* run_flag = swg_alloc<uint32_t>(ctx, nc);
  uint32_t* run_excl = swg_alloc<uint32_t>(ctx, nc);
  uint32_t* pair_first = swg_alloc<uint32_t>(ctx, nc);
  SWG_CHECK_ARENA(ctx);
  PairTable gp2_first;
  SWG_TRY(pair_table_make(ctx, n_g2, nc, &gp2_first));  // pairs that occur <= chromosome-pair runs <= chains
  SWG_HIP(ctx, hipMemcpyAsync(seg_sorted, seg, nc * 8, hipMemcpyDeviceToDevice, st));
  SWG_LAUNCH(ctx, "iota", iota_u32_kernel<<<nblk(nc), EW, 0, st>>>(nc, c_sorted));
  SWG_KERNEL_CHECK(ctx);
  SWG_TRY(swg_radix_sort_pairs(ctx, &seg_sorted, &c_sorted, &seg_tmp, &c_tmp, nc, 0, seg_bits));
  SWG_LAUNCH(ctx, "run_flag", run_flag_kernel<<<nblk(nc), EW, 0, st>>>(nc, seg_sorted, run_flag));
  SWG_KERNEL_CHECK(ctx);
  SWG_TRY(swg_exclusive_scan_u32(ctx, run_flag, run_excl, nc, d_tot));  // total = number of chromosome-pair runs
  SWG_LAUNCH(ctx, "first_appearance", first_appearance_kernel<<<nblk(nc), EW, 0, st>>>(nc, c_sorted, run_excl, run_flag, qid, tid, seq_genome2,
                                                                           pair_first, gp2_first));
  SWG_KERNEL_CHECK(ctx);
  // kept flags in seg-sorted order and their prefix counts; run starts
  uint32_t* ks = swg_alloc<uint32_t>(ctx, nc);
  uint32_t* kex = swg_alloc<uint32_t>(ctx, nc);
  uint32_t* run_start = swg_alloc<uint32_t>(ctx, nc);
  uint64_t* d_nk = swg_alloc<uint64_t>(ctx, 1);  // (d_tot, d_nk adjacent allocations are not assumed: two read-backs in one call below)
  SWG_CHECK_ARENA(ctx);
  SWG_LAUNCH(ctx, "kept_sorted", kept_sorted_kernel<<<nblk(nc), EW, 0, st>>>(nc, c_sorted, kept, run_flag, run_excl, ks, run_start));
  SWG_KERNEL_CHECK(ctx);
  SWG_TRY(swg_exclusive_scan_u32(ctx, ks, kex, nc, d_nk));
  uint64_t n_runs = 0, nk = 0;
  SWG_TRY(swg_read_scalars(ctx, d_tot, &n_runs, 1));
  SWG_TRY(swg_read_scalars(ctx, d_nk, &nk, 1));
  *n_kept_out = nk;
  if (nk == 0) {
    SWG_HIP(ctx, hipMemsetAsync(num, 0, nc * sizeof(uint32_t), st));
    return SWG_OK;
  }
  {
    uint64_t* rkey = swg_alloc<uint64_t>(ctx, n_runs);
    uint64_t* rkey_tmp = swg_alloc<uint64_t>(ctx, n_runs);
    uint32_t* r_sorted = swg_alloc<uint32_t>(ctx, n_runs);
    uint32_t* r_tmp = swg_alloc<uint32_t>(ctx, n_runs);
    uint32_t* run_kept = swg_alloc<uint32_t>(ctx, n_runs);
    uint32_t* r_sizes = swg_alloc<uint32_t>(ctx, n_runs);
    uint32_t* run_base = swg_alloc<uint32_t>(ctx, n_runs);
    SWG_CHECK_ARENA(ctx);
    const int c_bits = swg_bits_for(nc) ? swg_bits_for(nc) : 1;
    const unsigned rb = nblk(n_runs);
    SWG_LAUNCH(ctx, "run_keys", run_keys_kernel<<<rb, EW, 0, st>>>((uint32_t)n_runs, pair_first, run_start, kex, (uint32_t)nc, (uint32_t)nk, gp2_first,
                                                        qid, tid, seq_genome2, c_bits, rkey, r_sorted, run_kept));
    SWG_KERNEL_CHECK(ctx);
    SWG_TRY(swg_radix_sort_pairs(ctx, &rkey, &r_sorted, &rkey_tmp, &r_tmp, n_runs, 0, 2 * c_bits));
    SWG_LAUNCH(ctx, "run_sizes_sorted", run_sizes_sorted_kernel<<<rb, EW, 0, st>>>((uint32_t)n_runs, r_sorted, run_kept, r_sizes));
    SWG_KERNEL_CHECK(ctx);
    SWG_TRY(swg_exclusive_scan_u32(ctx, r_sizes, r_sizes, n_runs, nullptr));
    SWG_LAUNCH(ctx, "run_base", run_base_kernel<<<rb, EW, 0, st>>>((uint32_t)n_runs, r_sorted, r_sizes, run_base));
    SWG_KERNEL_CHECK(ctx);
    SWG_LAUNCH(ctx, "assign_numbers", assign_numbers_runs_kernel<<<nblk(nc), EW, 0, st>>>(nc, c_sorted, ks, kex, run_flag, run_excl, run_start,
                                                                              run_base, num));
    SWG_KERNEL_CHECK(ctx);
  }
  return SWG_OK;
}

}  // namespace swg_scaf

using namespace swg_scaf;

// ---- public seams -----------------------------------------------------------------------------------------------------
extern "C" int swg_plane_sweep_scaffolds(swg_ctx* ctx, uint64_t n, const uint32_t* q_id, const uint32_t* t_id,
                                         uint32_t n_seq, const uint32_t* seq_genome_two, uint32_t n_genome_two,
                                         const uint64_t* q_start, const uint64_t* q_end, const uint64_t* t_start,
                                         const uint64_t* t_end, const double* identity, int mode,
                                         uint64_t max_per_query, uint64_t max_per_target, double thr, int scoring,
                                         uint64_t* order_out, uint64_t* n_kept) {
  if (!ctx) return SWG_ERR_INVALID;
  if (n_kept) *n_kept = 0;
  if (n == 0) return SWG_OK;
  if (!q_id || !t_id || !seq_genome_two || !q_start || !q_end || !t_start || !t_end || !identity || !order_out ||
      !n_kept)
    return swg_set_error(ctx, SWG_ERR_INVALID, "NULL array");
  if (mode < 0 || mode > 2 || scoring < 0 || scoring > 4 || n_seq == 0 || n_genome_two == 0)
    return swg_set_error(ctx, SWG_ERR_INVALID, "bad mode / scoring / table size");
  if (n >= (uint64_t(1) << 31)) return swg_set_error(ctx, SWG_ERR_RANGE, "too many chains");
  SWG_HIP(ctx, hipSetDevice(ctx->device));
  std::vector<uint32_t> h(4 * n);
  uint32_t mx = 0;
  SWG_TRY(swg_narrow_coords(ctx, n, q_start, q_end, h.data(), h.data() + n, "query"));
  SWG_TRY(swg_narrow_coords(ctx, n, t_start, t_end, h.data() + 2 * n, h.data() + 3 * n, "target"));
  for (uint64_t i = 0; i < 4 * n; ++i)
    if (h[i] > mx) mx = h[i];
  for (uint64_t i = 0; i < n; ++i)
    if (q_id[i] >= n_seq || t_id[i] >= n_seq) return swg_set_error(ctx, SWG_ERR_INVALID, "sequence id out of range");
  const int pos_bits = swg_bits_for(mx) ? swg_bits_for(mx) : 1;
  hipStream_t st = ctx->stream;
  if (ctx->arena_cap == 0) SWG_TRY(swg_arena_reserve(ctx, (size_t)n * 400 + (size_t(16) << 20)));
  std::vector<uint32_t> num(n);
  int rc = swg_run_with_arena(ctx, [&]() -> int {
    ChainTable T;
    T.nc = n;
    uint32_t* d_c = swg_alloc<uint32_t>(ctx, 4 * n);
    T.qid = swg_alloc<uint32_t>(ctx, n);
    T.tid = swg_alloc<uint32_t>(ctx, n);
    T.wid = swg_alloc<double>(ctx, n);
    uint32_t* d_g2 = swg_alloc<uint32_t>(ctx, n_seq);
    uint8_t* C_kept = swg_alloc<uint8_t>(ctx, n);
    uint32_t* C_num = swg_alloc<uint32_t>(ctx, n);
    SWG_CHECK_ARENA(ctx);
    T.qs = d_c;
    T.qe = d_c + n;
    T.ts = d_c + 2 * n;
    T.te = d_c + 3 * n;
    SWG_HIP(ctx, hipMemcpyAsync(d_c, h.data(), 4 * n * 4, hipMemcpyHostToDevice, st));
    SWG_HIP(ctx, hipMemcpyAsync(T.qid, q_id, n * 4, hipMemcpyHostToDevice, st));
    SWG_HIP(ctx, hipMemcpyAsync(T.tid, t_id, n * 4, hipMemcpyHostToDevice, st));
    SWG_HIP(ctx, hipMemcpyAsync(T.wid, identity, n * 8, hipMemcpyHostToDevice, st));
    SWG_HIP(ctx, hipMemcpyAsync(d_g2, seq_genome_two, (size_t)n_seq * 4, hipMemcpyHostToDevice, st));
    uint64_t nk = 0;
    SWG_TRY(scaffold_sweep_and_number(ctx, T, n_seq, d_g2, n_genome_two, mode, max_per_query, max_per_target, thr,
                                      scoring, pos_bits, C_kept, C_num, &nk));
    SWG_HIP(ctx, hipMemcpyAsync(num.data(), C_num, n * 4, hipMemcpyDeviceToHost, st));
    SWG_HIP(ctx, hipStreamSynchronize(st));
    *n_kept = nk;
    return SWG_OK;
  });
  if (rc != SWG_OK) return rc;
  for (uint64_t i = 0; i < n; ++i)
    if (num[i]) order_out[num[i] - 1] = i;
  return SWG_OK;
}
