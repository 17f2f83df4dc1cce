// UnionFind::get_sets seam (src/union_find.rs:52-63).
#include "swg_scaffold_internal.h"

using namespace swg_scaf;

// Connected components standing in for UnionFind::get_sets (src/union_find.rs:52-63).  Sets are returned in
// ascending order of their smallest member, members ascending -- identical to the reference's root order
// whenever every union(x, y) joins a fresh singleton y > x to x's set (how the filter uses it,
// paf_filter.rs:854-859); for arbitrary union orders the reference's order depends on union-by-rank history.
namespace {
__global__ __launch_bounds__(EW) void cc_hook_kernel(uint64_t m, const uint32_t* __restrict__ xs,
                                                     const uint32_t* __restrict__ ys, uint32_t* label,
                                                     uint32_t* __restrict__ changed) {
  uint64_t e = (uint64_t)blockIdx.x * EW + threadIdx.x;
  if (e >= m) return;
  const uint32_t lx = label[xs[e]], ly = label[ys[e]];
  if (lx == ly) return;
  const uint32_t lo = lx < ly ? lx : ly, hi = lx < ly ? ly : lx;
  atomicMin(&label[hi], lo);
  *changed = 1;
}
__global__ __launch_bounds__(EW) void cc_compress_kernel(uint64_t n, uint32_t* label, uint32_t* __restrict__ changed) {
  uint64_t i = (uint64_t)blockIdx.x * EW + threadIdx.x;
  if (i >= n) return;
  uint32_t l = label[i];
  uint32_t ll = label[l];
  if (ll != l) {
    label[i] = ll;
    *changed = 1;
  }
}
__global__ __launch_bounds__(EW) void cc_root_flag_kernel(uint64_t n, const uint32_t* __restrict__ label,
                                                          uint32_t* __restrict__ flag) {
  uint64_t i = (uint64_t)blockIdx.x * EW + threadIdx.x;
  if (i < n) flag[i] = label[i] == i ? 1u : 0u;
}
__global__ __launch_bounds__(EW) void cc_set_of_kernel(uint64_t n, const uint32_t* __restrict__ label,
                                                       const uint32_t* __restrict__ root_excl,
                                                       uint32_t* __restrict__ set_of) {
  uint64_t i = (uint64_t)blockIdx.x * EW + threadIdx.x;
  if (i < n) set_of[i] = root_excl[label[i]];
}
}  // namespace

extern "C" int swg_union_find_sets(swg_ctx* ctx, uint64_t n, uint64_t m, const uint32_t* xs, const uint32_t* ys,
                                   uint32_t* set_of, uint64_t* n_sets) {
  if (!ctx) return SWG_ERR_INVALID;
  if (!n_sets) return swg_set_error(ctx, SWG_ERR_INVALID, "NULL argument");
  *n_sets = 0;
  if (n == 0) return SWG_OK;
  if (!set_of || (m && (!xs || !ys))) return swg_set_error(ctx, SWG_ERR_INVALID, "NULL array");
  if (n >= (uint64_t(1) << 32)) return swg_set_error(ctx, SWG_ERR_RANGE, "too many elements");
  for (uint64_t e = 0; e < m; ++e)
    if (xs[e] >= n || ys[e] >= n) return swg_set_error(ctx, SWG_ERR_INVALID, "element out of range");
  SWG_HIP(ctx, hipSetDevice(ctx->device));
  hipStream_t st = ctx->stream;
  if (ctx->arena_cap == 0) SWG_TRY(swg_arena_reserve(ctx, (size_t)(n + m) * 32 + (size_t(8) << 20)));
  return swg_run_with_arena(ctx, [&]() -> int {
    uint32_t* label = swg_alloc<uint32_t>(ctx, n);
    uint32_t* flag = swg_alloc<uint32_t>(ctx, n);
    uint32_t* excl = swg_alloc<uint32_t>(ctx, n);
    uint32_t* d_set = swg_alloc<uint32_t>(ctx, n);
    uint32_t* dx = swg_alloc<uint32_t>(ctx, m + 1);
    uint32_t* dy = swg_alloc<uint32_t>(ctx, m + 1);
    uint32_t* changed = swg_alloc<uint32_t>(ctx, 2);
    uint64_t* d_tot = swg_alloc<uint64_t>(ctx, 1);
    SWG_CHECK_ARENA(ctx);
    if (m) {
      SWG_HIP(ctx, hipMemcpyAsync(dx, xs, m * 4, hipMemcpyHostToDevice, st));
      SWG_HIP(ctx, hipMemcpyAsync(dy, ys, m * 4, hipMemcpyHostToDevice, st));
    }
    SWG_LAUNCH(ctx, "iota", iota_u32_kernel<<<nblk(n), EW, 0, st>>>(n, label));
    SWG_KERNEL_CHECK(ctx);
    for (int round = 0; round < 100000 && m; ++round) {
      SWG_HIP(ctx, hipMemsetAsync(changed, 0, 8, st));
      SWG_LAUNCH(ctx, "cc_hook", cc_hook_kernel<<<nblk(m), EW, 0, st>>>(m, dx, dy, label, changed));
      SWG_KERNEL_CHECK(ctx);
      SWG_LAUNCH(ctx, "cc_compress", cc_compress_kernel<<<nblk(n), EW, 0, st>>>(n, label, changed));
      SWG_KERNEL_CHECK(ctx);
      uint64_t ch = 0;
      SWG_TRY(swg_read_scalars(ctx, reinterpret_cast<uint64_t*>(changed), &ch, 1));
      if ((uint32_t)ch == 0) break;
    }
    SWG_LAUNCH(ctx, "cc_root_flag", cc_root_flag_kernel<<<nblk(n), EW, 0, st>>>(n, label, flag));
    SWG_KERNEL_CHECK(ctx);
    SWG_TRY(swg_exclusive_scan_u32(ctx, flag, excl, n, d_tot));
    SWG_LAUNCH(ctx, "cc_set_of", cc_set_of_kernel<<<nblk(n), EW, 0, st>>>(n, label, excl, d_set));
    SWG_KERNEL_CHECK(ctx);
    uint64_t ns = 0;
    SWG_TRY(swg_read_scalars(ctx, d_tot, &ns, 1));
    SWG_HIP(ctx, hipMemcpyAsync(set_of, d_set, n * 4, hipMemcpyDeviceToHost, st));
    SWG_HIP(ctx, hipStreamSynchronize(st));
    *n_sets = ns;
    return SWG_OK;
  });
}
