// Scan and radix-sort primitives for gfx950 (wave64).
//
//  * swg_exclusive_scan_u32 : reduce / recurse / down-sweep scan, 4096 elements per workgroup.
//  * swg_radix_sort_pairs   : stable LSD radix sort of (u64 key, u32 value) pairs, 8-bit digits, one "onesweep" kernel per
//      digit (below); swg_radix_sort_packed: the same over 8-byte words ((key >> 8) << index bits | index) after the first
//      pass, with 9-bit digits where they save a pass.  Stability is what makes the multi-word sorts of the pipeline compose
//      and is what carries the reference's "ties fall to input order" rule (sort_by_key is stable,
//      src/plane_sweep_exact.rs:300, src/paf_filter.rs:777).
//      Fallback (more than 8 passes, or SWG_SORT_FALLBACK=1): per pass (1) per-tile digit histogram, (2) exclusive scan of
//      the digit-major histogram, (3) stable scatter: each tile re-reads its keys in 256-element rows, ranks every row with a
//      wavefront match (8 x 64-bit ballots) + cross-wave prefix in LDS, and writes the pair to its final slot.
#include <cstdlib>

#include "swg_internal.h"

namespace {

constexpr int SCAN_THREADS = 256;
constexpr int SCAN_ITEMS = 16;
constexpr int SCAN_TILE = SCAN_THREADS * SCAN_ITEMS;

// OP: 0 = add, 1 = max.  Identity is 0 for both.  T = uint32_t or uint64_t.
template <int OP, typename T>
__device__ __forceinline__ T scan_op(T a, T b) {
  return OP == 0 ? a + b : (a > b ? a : b);
}

template <int OP, typename T>
__device__ __forceinline__ T wave_inclusive_scan(T v, int lane) {
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    T t = __shfl_up(v, d, 64);
    if (lane >= d) v = scan_op<OP>(t, v);
  }
  return v;
}

// Block-wide scan of one value per thread (256 threads); returns the EXCLUSIVE prefix (identity for
// the first thread) and the block total through *total.
template <int OP, typename T>
__device__ __forceinline__ T block_exclusive_scan(T v, T* total, T* lds_wave, T* lds_prev) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  T inc = wave_inclusive_scan<OP>(v, lane);
  if (lane == 63) lds_wave[wave] = inc;
  lds_prev[threadIdx.x] = inc;
  __syncthreads();
  T base = 0, tot = 0;
#pragma unroll
  for (int w = 0; w < SCAN_THREADS / 64; ++w) {
    T s = lds_wave[w];
    if (w < wave) base = scan_op<OP>(base, s);
    tot = scan_op<OP>(tot, s);
  }
  const T prev_in_wave = lane ? lds_prev[threadIdx.x - 1] : (T)0;
  __syncthreads();
  *total = tot;
  return scan_op<OP>(base, prev_in_wave);
}

// Tile layout of the scans: 4096 elements per workgroup; wave w owns the contiguous 1024 elements
// [w*1024, (w+1)*1024) as SCAN_ROWS rows of 256, lane l holding 4 consecutive elements of each row
// (one 16-byte load per row for 32-bit elements -> every row is one fully coalesced 1 KB access).
constexpr int SCAN_VEC = 4;
constexpr int SCAN_ROWS = SCAN_ITEMS / SCAN_VEC;
constexpr int SCAN_WAVE_SPAN = 64 * SCAN_ITEMS;

template <typename T>
__device__ __forceinline__ void load_vec4(const T* in, uint64_t i, uint64_t n, bool aligned, T v[SCAN_VEC]) {
  if (aligned && i + SCAN_VEC <= n) {
    if constexpr (sizeof(T) == 4) {
      const uint4 q = *reinterpret_cast<const uint4*>(in + i);
      v[0] = q.x, v[1] = q.y, v[2] = q.z, v[3] = q.w;
    } else {
      const ulonglong2 a = *reinterpret_cast<const ulonglong2*>(in + i);
      const ulonglong2 b = *reinterpret_cast<const ulonglong2*>(in + i + 2);
      v[0] = a.x, v[1] = a.y, v[2] = b.x, v[3] = b.y;
    }
  } else {
#pragma unroll
    for (int j = 0; j < SCAN_VEC; ++j) v[j] = i + j < n ? in[i + j] : (T)0;
  }
}
template <typename T>
__device__ __forceinline__ void store_vec4(T* out, uint64_t i, uint64_t n, bool aligned, const T v[SCAN_VEC]) {
  if (aligned && i + SCAN_VEC <= n) {
    if constexpr (sizeof(T) == 4) {
      *reinterpret_cast<uint4*>(out + i) = make_uint4(v[0], v[1], v[2], v[3]);
    } else {
      *reinterpret_cast<ulonglong2*>(out + i) = make_ulonglong2(v[0], v[1]);
      *reinterpret_cast<ulonglong2*>(out + i + 2) = make_ulonglong2(v[2], v[3]);
    }
  } else {
#pragma unroll
    for (int j = 0; j < SCAN_VEC; ++j)
      if (i + j < n) out[i + j] = v[j];
  }
}

template <int OP, typename T>
__global__ __launch_bounds__(SCAN_THREADS) void scan_reduce_kernel(const T* __restrict__ in, T* __restrict__ block_sums,
                                                                    uint64_t n, int aligned) {
  __shared__ T lds_wave[SCAN_THREADS / 64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const uint64_t wbase = (uint64_t)blockIdx.x * SCAN_TILE + (uint64_t)wave * SCAN_WAVE_SPAN;
  T s = 0;
#pragma unroll
  for (int r = 0; r < SCAN_ROWS; ++r) {
    T v[SCAN_VEC];
    load_vec4(in, wbase + (uint64_t)r * 256 + (uint64_t)lane * SCAN_VEC, n, aligned != 0, v);
#pragma unroll
    for (int j = 0; j < SCAN_VEC; ++j) s = scan_op<OP>(s, v[j]);
  }
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) s = scan_op<OP>(s, (T)__shfl_xor(s, d, 64));
  if (lane == 0) lds_wave[wave] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    T t = 0;
#pragma unroll
    for (int w = 0; w < SCAN_THREADS / 64; ++w) t = scan_op<OP>(t, lds_wave[w]);
    block_sums[blockIdx.x] = t;
  }
}

// `in` and `out` may alias (in-place scan): a lane only ever writes elements it has read itself.
template <int OP, bool INCLUSIVE, typename T>
__global__ __launch_bounds__(SCAN_THREADS) void scan_down_kernel(const T* in, T* out, const T* __restrict__ block_offsets,
                                                                  uint64_t n, uint64_t* __restrict__ total_out,
                                                                  int write_total, int aligned) {
  __shared__ T lds_wave[SCAN_THREADS / 64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const uint64_t wbase = (uint64_t)blockIdx.x * SCAN_TILE + (uint64_t)wave * SCAN_WAVE_SPAN;
  T v[SCAN_ROWS][SCAN_VEC];
  T ex[SCAN_ROWS];  // exclusive prefix of the lane's 4-element group inside the wave's span
  T carry = 0;
#pragma unroll
  for (int r = 0; r < SCAN_ROWS; ++r) {
    load_vec4(in, wbase + (uint64_t)r * 256 + (uint64_t)lane * SCAN_VEC, n, aligned != 0, v[r]);
    T s = 0;
#pragma unroll
    for (int j = 0; j < SCAN_VEC; ++j) s = scan_op<OP>(s, v[r][j]);
    const T inc = wave_inclusive_scan<OP>(s, lane);
    T prev = (T)__shfl_up(inc, 1, 64);
    if (lane == 0) prev = 0;
    ex[r] = scan_op<OP>(carry, prev);
    carry = scan_op<OP>(carry, (T)__shfl(inc, 63, 64));
  }
  if (lane == 0) lds_wave[wave] = carry;
  __syncthreads();
  T base = block_offsets ? block_offsets[blockIdx.x] : (T)0;
  T tot = base;
#pragma unroll
  for (int w = 0; w < SCAN_THREADS / 64; ++w) {
    const T s = lds_wave[w];
    if (w < wave) base = scan_op<OP>(base, s);
    tot = scan_op<OP>(tot, s);
  }
#pragma unroll
  for (int r = 0; r < SCAN_ROWS; ++r) {
    T run = scan_op<OP>(base, ex[r]);
    T o[SCAN_VEC];
#pragma unroll
    for (int j = 0; j < SCAN_VEC; ++j) {
      if (INCLUSIVE) run = scan_op<OP>(run, v[r][j]);
      o[j] = run;
      if (!INCLUSIVE) run = scan_op<OP>(run, v[r][j]);
    }
    store_vec4(out, wbase + (uint64_t)r * 256 + (uint64_t)lane * SCAN_VEC, n, aligned != 0, o);
  }
  if (write_total && blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) *total_out = (uint64_t)tot;
}

template <int OP, bool INCLUSIVE, typename T>
int scan_impl(swg_ctx* ctx, const T* in, T* out, uint64_t n, uint64_t* d_total_out) {
  if (n == 0) {
    if (d_total_out) SWG_HIP(ctx, hipMemsetAsync(d_total_out, 0, sizeof(uint64_t), ctx->stream));
    return SWG_OK;
  }
  const int aligned = ((reinterpret_cast<uintptr_t>(in) | reinterpret_cast<uintptr_t>(out)) & 15) == 0;
  const uint64_t nb = (n + SCAN_TILE - 1) / SCAN_TILE;
  if (nb == 1) {
    SWG_LAUNCH(ctx, "scan_down", scan_down_kernel<OP, INCLUSIVE, T><<<1, SCAN_THREADS, 0, ctx->stream>>>(
                                     in, out, (const T*)nullptr, n, d_total_out, d_total_out ? 1 : 0, aligned));
    SWG_KERNEL_CHECK(ctx);
    return SWG_OK;
  }
  swg_arena_mark mark = swg_arena_save(ctx);
  T* sums = swg_alloc<T>(ctx, nb);
  SWG_CHECK_ARENA(ctx);
  SWG_LAUNCH(ctx, "scan_reduce", scan_reduce_kernel<OP, T><<<(unsigned)nb, SCAN_THREADS, 0, ctx->stream>>>(in, sums, n, aligned));
  SWG_KERNEL_CHECK(ctx);
  SWG_TRY((scan_impl<OP, false, T>(ctx, sums, sums, nb, nullptr)));  // block offsets are always exclusive
  SWG_LAUNCH(ctx, "scan_down", scan_down_kernel<OP, INCLUSIVE, T><<<(unsigned)nb, SCAN_THREADS, 0, ctx->stream>>>(
                                   in, out, (const T*)sums, n, d_total_out, d_total_out ? 1 : 0, aligned));
  SWG_KERNEL_CHECK(ctx);
  swg_arena_restore(ctx, mark);  // stream order keeps `sums` alive until the kernels above ran
  return SWG_OK;
}

// ---- stream compaction straight from byte flags ------------------------------------------------------
// Tile = 4096 flags; wave w owns 1024 of them, lane l 16 consecutive bytes (one 16-byte load).
__device__ __forceinline__ uint32_t nonzero_bytes(uint32_t w) {  // 0x80 in every byte of w that is != 0
  return (((w & 0x7f7f7f7fu) + 0x7f7f7f7fu) | w) & 0x80808080u;
}
__device__ __forceinline__ void load_flags16(const uint8_t* __restrict__ f, uint64_t i, uint64_t n, bool aligned, uint32_t nz[4]) {
  if (aligned && i + 16 <= n) {
    const uint4 q = *reinterpret_cast<const uint4*>(f + i);
    nz[0] = nonzero_bytes(q.x), nz[1] = nonzero_bytes(q.y), nz[2] = nonzero_bytes(q.z), nz[3] = nonzero_bytes(q.w);
  } else {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      uint32_t m = 0;
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        const uint64_t j = i + (uint64_t)k * 4 + b;
        if (j < n && f[j]) m |= 0x80u << (8 * b);
      }
      nz[k] = m;
    }
  }
}

__global__ __launch_bounds__(SCAN_THREADS) void flag_count_kernel(const uint8_t* __restrict__ f, uint64_t n, int aligned,
                                                                   uint32_t* __restrict__ tile_cnt) {
  __shared__ uint32_t lds_wave[SCAN_THREADS / 64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  uint32_t nz[4];
  load_flags16(f, (uint64_t)blockIdx.x * SCAN_TILE + (uint64_t)threadIdx.x * 16, n, aligned != 0, nz);
  uint32_t c = __popc(nz[0]) + __popc(nz[1]) + __popc(nz[2]) + __popc(nz[3]);
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) c += __shfl_xor(c, d, 64);
  if (lane == 0) lds_wave[wave] = c;
  __syncthreads();
  if (threadIdx.x == 0) tile_cnt[blockIdx.x] = lds_wave[0] + lds_wave[1] + lds_wave[2] + lds_wave[3];
}

// list[rank of i among the set flags] = i
__global__ __launch_bounds__(SCAN_THREADS) void flag_compact_kernel(const uint8_t* __restrict__ f, uint64_t n, int aligned,
                                                                     const uint32_t* __restrict__ tile_off,
                                                                     uint32_t* __restrict__ list) {
  __shared__ uint32_t lds_wave[SCAN_THREADS / 64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const uint64_t i0 = (uint64_t)blockIdx.x * SCAN_TILE + (uint64_t)threadIdx.x * 16;
  uint32_t nz[4];
  load_flags16(f, i0, n, aligned != 0, nz);
  const uint32_t c = __popc(nz[0]) + __popc(nz[1]) + __popc(nz[2]) + __popc(nz[3]);
  const uint32_t inc = wave_inclusive_scan<0>(c, lane);
  if (lane == 63) lds_wave[wave] = inc;
  __syncthreads();
  uint32_t pos = tile_off[blockIdx.x] + inc - c;
  for (int w = 0; w < wave; ++w) pos += lds_wave[w];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    uint32_t m = nz[k];
    while (m) {
      const int b = __builtin_ctz(m) >> 3;
      list[pos++] = (uint32_t)(i0 + (uint64_t)k * 4 + b);
      m &= m - 1;
    }
  }
}

}  // namespace

int swg_exclusive_scan_u32(swg_ctx* ctx, const uint32_t* in, uint32_t* out, uint64_t n, uint64_t* d_total_out) {
  return scan_impl<0, false, uint32_t>(ctx, in, out, n, d_total_out);
}
int swg_inclusive_max_scan_u32(swg_ctx* ctx, const uint32_t* in, uint32_t* out, uint64_t n) {
  return scan_impl<1, true, uint32_t>(ctx, in, out, n, nullptr);
}
int swg_inclusive_max_scan_u64(swg_ctx* ctx, const uint64_t* in, uint64_t* out, uint64_t n) {
  return scan_impl<1, true, uint64_t>(ctx, in, out, n, nullptr);
}

int swg_inclusive_sum_scan_u64(swg_ctx* ctx, const uint64_t* in, uint64_t* out, uint64_t n) {
  return scan_impl<0, true, uint64_t>(ctx, in, out, n, nullptr);
}

int swg_flags_count(swg_ctx* ctx, const uint8_t* flags, uint64_t n, swg_flag_scan* fs, uint64_t* d_total) {
  fs->flags = flags;
  fs->n = n;
  fs->tile_off = nullptr;
  if (n == 0) {
    if (d_total) SWG_HIP(ctx, hipMemsetAsync(d_total, 0, sizeof(uint64_t), ctx->stream));
    return SWG_OK;
  }
  const uint64_t nb = (n + SCAN_TILE - 1) / SCAN_TILE;
  fs->tile_off = swg_alloc<uint32_t>(ctx, nb);
  SWG_CHECK_ARENA(ctx);
  const int aligned = (reinterpret_cast<uintptr_t>(flags) & 15) == 0;
  SWG_LAUNCH(ctx, "flag_count", flag_count_kernel<<<(unsigned)nb, SCAN_THREADS, 0, ctx->stream>>>(flags, n, aligned, fs->tile_off));
  SWG_KERNEL_CHECK(ctx);
  return scan_impl<0, false, uint32_t>(ctx, fs->tile_off, fs->tile_off, nb, d_total);
}

int swg_flags_compact(swg_ctx* ctx, const swg_flag_scan& fs, uint32_t* list) {
  if (fs.n == 0) return SWG_OK;
  const uint64_t nb = (fs.n + SCAN_TILE - 1) / SCAN_TILE;
  const int aligned = (reinterpret_cast<uintptr_t>(fs.flags) & 15) == 0;
  SWG_LAUNCH(ctx, "flag_compact", flag_compact_kernel<<<(unsigned)nb, SCAN_THREADS, 0, ctx->stream>>>(fs.flags, fs.n, aligned,
                                                                                            fs.tile_off, list));
  SWG_KERNEL_CHECK(ctx);
  return SWG_OK;
}

// ---------------------------------------------------------------------------------------------
// radix sort
// ---------------------------------------------------------------------------------------------
namespace {

constexpr int RS_THREADS = 256;
constexpr int RS_ROWS = 16;  // rows of 256 elements per tile
constexpr int RS_TILE = RS_THREADS * RS_ROWS;
constexpr int RS_RADIX = 256;

__global__ __launch_bounds__(RS_THREADS) void rs_hist_kernel(const uint64_t* __restrict__ keys, uint64_t n,
                                                              int shift, uint32_t mask,
                                                              uint32_t* __restrict__ hist, uint32_t ntiles) {
  __shared__ uint32_t h[RS_RADIX];
  h[threadIdx.x] = 0;
  __syncthreads();
  const uint64_t base = (uint64_t)blockIdx.x * RS_TILE;
#pragma unroll 4
  for (int r = 0; r < RS_ROWS; ++r) {
    uint64_t i = base + (uint64_t)r * RS_THREADS + threadIdx.x;
    if (i < n) {
      uint32_t d = (uint32_t)(keys[i] >> shift) & mask;
      atomicAdd(&h[d], 1u);
    }
  }
  __syncthreads();
  hist[(uint64_t)threadIdx.x * ntiles + blockIdx.x] = h[threadIdx.x];
}

__global__ __launch_bounds__(RS_THREADS) void rs_scatter_kernel(
    const uint64_t* __restrict__ keys_in, const uint32_t* __restrict__ vals_in,
    uint64_t* __restrict__ keys_out, uint32_t* __restrict__ vals_out, uint64_t n, int shift, uint32_t mask,
    const uint32_t* __restrict__ offsets, uint32_t ntiles) {
  __shared__ uint32_t running[RS_RADIX];
  __shared__ uint32_t wcnt[RS_THREADS / 64][RS_RADIX];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  running[threadIdx.x] = offsets[(uint64_t)threadIdx.x * ntiles + blockIdx.x];
#pragma unroll
  for (int w = 0; w < RS_THREADS / 64; ++w) wcnt[w][threadIdx.x] = 0;
  __syncthreads();
  const uint64_t base = (uint64_t)blockIdx.x * RS_TILE;
  const uint64_t lt_mask = lane == 0 ? 0ull : (~0ull >> (64 - lane));
  for (int r = 0; r < RS_ROWS; ++r) {
    const uint64_t i = base + (uint64_t)r * RS_THREADS + threadIdx.x;
    const bool valid = i < n;
    uint64_t key = 0;
    uint32_t val = 0;
    if (valid) {
      key = keys_in[i];
      val = vals_in[i];
    }
    const uint32_t d = (uint32_t)(key >> shift) & mask;
    uint64_t peers = __ballot(valid);
#pragma unroll
    for (int b = 0; b < 8; ++b) {
      const bool bit = (d >> b) & 1u;
      const uint64_t m = __ballot(bit);
      peers &= bit ? m : ~m;
    }
    const uint32_t rank_in_wave = __popcll(peers & lt_mask);
    if (valid && rank_in_wave == 0) wcnt[wave][d] = __popcll(peers);
    __syncthreads();
    if (valid) {
      uint32_t off = running[d] + rank_in_wave;
#pragma unroll
      for (int w = 0; w < RS_THREADS / 64; ++w)
        if (w < wave) off += wcnt[w][d];
      keys_out[off] = key;
      vals_out[off] = val;
    }
    __syncthreads();
    {
      uint32_t add = 0;
#pragma unroll
      for (int w = 0; w < RS_THREADS / 64; ++w) {
        add += wcnt[w][threadIdx.x];
        wcnt[w][threadIdx.x] = 0;
      }
      running[threadIdx.x] += add;
    }
    __syncthreads();
  }
}

}  // namespace

// ---------------------------------------------------------------------------------------------
// "onesweep" pass: one kernel per digit, read once / write once.
//   * digit histograms of every pass come from one upfront sweep over the keys (or from the kernel that wrote them);
//   * a tile (8192 elements, 512 threads) ranks its keys with wavefront match + per-wave LDS counters, learns the
//     number of equal-digit keys in all earlier tiles by decoupled look-back over per-tile status
//     words (count | flag in one 32-bit word, agent-scope relaxed atomics: single-word hand-off),
//     reorders the tile in LDS and writes digit runs with consecutive lanes on consecutive addresses.
//   Tiles take their index from an atomic ticket so that every tile a block waits on has started.
// ---------------------------------------------------------------------------------------------
namespace {

constexpr int OS_THREADS = 256;
#ifndef SWG_OS_ITEMS
#define SWG_OS_ITEMS 16
#endif
constexpr int OS_ITEMS = SWG_OS_ITEMS;
constexpr int OS_TILE = OS_THREADS * OS_ITEMS;
static_assert(OS_ITEMS % 2 == 0, "rows are ranked in pairs");
constexpr int OS_WAVES = OS_THREADS / 64;
// (the constants above: the stand-alone histogram kernel)
// The pass kernels run 512-thread work-groups over 8192-element tiles (two per CU): the look-back is the longest wait in a
// tile's life (an agent-scope round trip per hop, ~1 us across the XCDs, over every tile that has not published its
// inclusive prefix yet -- as many as start during one look-back), and half as many tiles per pass halve that walk.
constexpr int PK_THREADS = 512;
constexpr int PK_ITEMS = 16;
constexpr int PK_TILE = PK_THREADS * PK_ITEMS;
constexpr int PK_WAVES = PK_THREADS / 64;
static_assert(PK_ITEMS % 2 == 0 && PK_THREADS >= RS_RADIX, "rows are ranked in pairs; one thread per digit");

// look-back word: flag in the top two bits, count below.  32-bit words hold prefixes < 2^30; inputs of 2^30 .. 2^32-1
// pairs use 64-bit words (same protocol, one relaxed 8-byte access instead of a 4-byte one).
template <typename ST>
struct os_word {
  static constexpr ST LOCAL = ST(1) << (sizeof(ST) * 8 - 2);
  static constexpr ST GLOBAL = ST(2) << (sizeof(ST) * 8 - 2);
  static constexpr ST MASK = LOCAL - 1;
};
constexpr int OS_MAX_PASSES = 8;

// Stable rank of a row of 64 digits inside its wavefront: the number of lower lanes holding the same digit (below) and the
// number of lanes holding it at all (peers).  Eight ballots narrow the peer mask one digit bit at a time; with the bit
// sign-extended over a word (v_bfe_i32) each half of the mask takes one three-operand bit op per ballot
// (v_bitop3: mask &= ~(ballot ^ bit); 4 vector instructions per digit bit), and the count of lower peers is the mbcnt pair
// over the final mask.  Two rows go through together: a ballot's SGPR result cannot feed the next vector instruction
// (2 wait states on gfx950), the second row's instructions fill those slots.
struct os_match {
  uint32_t lo, hi;
  __device__ __forceinline__ uint32_t below() const { return __builtin_amdgcn_mbcnt_hi(hi, __builtin_amdgcn_mbcnt_lo(lo, 0u)); }
  __device__ __forceinline__ uint32_t peers() const { return (uint32_t)__popc(lo) + (uint32_t)__popc(hi); }
};
__device__ __forceinline__ void os_match_step(os_match& a, uint32_t d, int b) {
  uint32_t nb = (uint32_t)__builtin_amdgcn_sbfe((int)d, (unsigned)b, 1u);  // 0 or ~0
  asm volatile("" : "+v"(nb));  // keeps the ballot's compare on nb itself (otherwise a shift + sign test of d is added)
  const uint64_t m = __ballot(nb != 0u);
  a.lo = __builtin_amdgcn_bitop3_b32(a.lo, (uint32_t)m, nb, 0x90);  // lo & ~(m ^ nb)
  a.hi = __builtin_amdgcn_bitop3_b32(a.hi, (uint32_t)(m >> 32), nb, 0x90);
}

// Two rows of the ranking phase.  Every lane reads its digit's running count of the wave, the lowest lane of each digit adds
// the row's peer count with a no-return LDS add: LDS operations of one wavefront execute in issue order, so the next row's
// reads see the add, and nothing waits on the read before the add is issued.
// Lanes past the end of the input carry the all-ones key: they rank behind every real element of the (last) tile, inflate
// only the count of the highest digit there, and land in staging slots >= tile_n that are never written out.
__device__ __forceinline__ void os_rank_rows(uint32_t* wave_cnt, uint32_t d0, uint32_t d1, uint32_t* r0, uint32_t* r1) {
  os_match a{~0u, ~0u}, b{~0u, ~0u};
  const uint32_t prev0 = __hip_atomic_load(wave_cnt + d0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#pragma unroll
  for (int bit = 0; bit < 8; ++bit) {
    os_match_step(a, d0, bit);
    os_match_step(b, d1, bit);
  }
  const uint32_t below0 = a.below(), below1 = b.below();
  if (below0 == 0) __hip_atomic_fetch_add(wave_cnt + d0, a.peers(), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  const uint32_t prev1 = __hip_atomic_load(wave_cnt + d1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  if (below1 == 0) __hip_atomic_fetch_add(wave_cnt + d1, b.peers(), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  *r0 = prev0 + below0;
  *r1 = prev1 + below1;
}

__global__ __launch_bounds__(OS_THREADS) void os_hist_kernel(const uint64_t* __restrict__ keys, uint64_t n, swg_radix_plan plan,
                                                              uint32_t* __restrict__ ghist) {
  __shared__ uint32_t h[OS_MAX_PASSES][SWG_RADIX_BINS];
  swg_radix_hist_zero(h, plan.npasses);
  __syncthreads();
  for (uint64_t base = (uint64_t)blockIdx.x * OS_TILE; base < n; base += (uint64_t)gridDim.x * OS_TILE) {
#pragma unroll 4
    for (int r = 0; r < OS_ITEMS; ++r) {
      const uint64_t i = base + (uint64_t)r * OS_THREADS + threadIdx.x;
      const bool valid = i < n;
      swg_radix_hist_add(h, valid ? keys[i] : 0ull, valid, plan);
    }
  }
  __syncthreads();
  swg_radix_hist_flush(h, plan.npasses, ghist);
}

// one block per pass: exclusive scan of its (up to 512) bins, in place
__global__ __launch_bounds__(SWG_RADIX_BINS) void os_scan_hist_kernel(uint32_t* __restrict__ ghist) {
  __shared__ uint32_t wsum[SWG_RADIX_BINS / 64];
  uint32_t* row = ghist + (size_t)blockIdx.x * SWG_RADIX_BINS;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const uint32_t v = row[threadIdx.x];
  const uint32_t inc = wave_inclusive_scan<0>(v, lane);
  if (lane == 63) wsum[wave] = inc;
  __syncthreads();
  uint32_t base = 0;
  for (int w = 0; w < wave; ++w) base += wsum[w];
  row[threadIdx.x] = base + inc - v;
}

// Decoupled look-back of one digit (thread = digit) of tile `tile`: publishes the tile's count, sums the counts of the tiles
// before it back to the nearest tile whose inclusive prefix is known, publishes its own inclusive prefix and returns the
// exclusive one.  OS_LOOKBACK predecessors are read per round trip (the loads are independent; an agent-scope load is ~1 us).
#ifndef SWG_OS_LOOKBACK
#define SWG_OS_LOOKBACK 4
#endif
constexpr int OS_LOOKBACK = SWG_OS_LOOKBACK;  // predecessors read per round trip
template <typename ST>
__device__ __forceinline__ ST os_lookback(ST* status, uint32_t tile, int tid, uint32_t tot, int bins = RS_RADIX) {
  using W = os_word<ST>;
  ST* my = status + (size_t)tile * bins + tid;
  if (tile == 0) {
    __hip_atomic_store(my, W::GLOBAL | (ST)tot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return 0;
  }
  __hip_atomic_store(my, W::LOCAL | (ST)tot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  ST excl = 0;
  uint32_t tt = tile - 1;  // the next tile to account for
  while (true) {
    ST sv[OS_LOOKBACK];
#pragma unroll
    for (int j = 0; j < OS_LOOKBACK; ++j) {
      const uint32_t t = tt >= (uint32_t)j ? tt - j : 0u;  // (tile 0 ends every walk: it only ever publishes GLOBAL)
      sv[j] = __hip_atomic_load(status + (size_t)t * bins + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    bool done = false;
    int used = 0;
#pragma unroll
    for (int j = 0; j < OS_LOOKBACK; ++j) {
      if (!done && used == j) {
        const ST f = sv[j] & ~W::MASK;
        if (f != 0) {
          excl += sv[j] & W::MASK;
          ++used;
          if (f == W::GLOBAL) done = true;
        }
      }
    }
    if (done) break;
    if (used == 0) __builtin_amdgcn_s_sleep(1);
    tt -= (uint32_t)used;
  }
  __hip_atomic_store(my, W::GLOBAL | (excl + (ST)tot), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return excl;
}

template <typename ST>
__global__ __launch_bounds__(PK_THREADS) void os_pass_kernel(const uint64_t* __restrict__ keys_in,
                                                              const uint32_t* __restrict__ vals_in,
                                                              uint64_t* __restrict__ keys_out,
                                                              uint32_t* __restrict__ vals_out, uint64_t n, int shift,
                                                              uint32_t mask, const uint32_t* __restrict__ gbase,
                                                              ST* status, uint32_t* ticket) {
  __shared__ uint64_t lkeys[PK_TILE];  // staging for the keys, then reused (as u32) for the values
  __shared__ uint32_t cnt[PK_WAVES][RS_RADIX];
  __shared__ uint32_t dst_base[RS_RADIX];
  __shared__ uint32_t lds_wave[PK_THREADS / 64];
  __shared__ uint32_t lds_prev[PK_THREADS];
  __shared__ uint32_t s_tile;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid == 0) s_tile = atomicAdd(ticket, 1u);
  const uint32_t gb = tid < RS_RADIX ? gbase[tid] : 0u;  // (early: its latency hides behind the ticket's)
  for (int i = tid; i < PK_WAVES * RS_RADIX; i += PK_THREADS) (&cnt[0][0])[i] = 0;
  __syncthreads();
  const uint32_t tile = s_tile;
  const uint64_t tile_base = (uint64_t)tile * PK_TILE;
  const uint64_t wbase = tile_base + (uint64_t)wave * (64 * PK_ITEMS);

  uint64_t key[PK_ITEMS];
  uint32_t val[PK_ITEMS];
  uint32_t rank[PK_ITEMS];
  if (tile_base + PK_TILE <= n) {  // every tile but the last: no bounds checks
#pragma unroll
    for (int r = 0; r < PK_ITEMS; ++r) {
      key[r] = keys_in[wbase + (uint64_t)r * 64 + lane];
      val[r] = vals_in[wbase + (uint64_t)r * 64 + lane];
    }
  } else {
#pragma unroll
    for (int r = 0; r < PK_ITEMS; ++r) {
      const uint64_t i = wbase + (uint64_t)r * 64 + lane;
      key[r] = i < n ? keys_in[i] : ~0ull;
      val[r] = i < n ? vals_in[i] : 0u;
    }
  }
  // ---- rank inside the wave (stable): match + per-wave running digit counters in LDS
#pragma unroll
  for (int r = 0; r < PK_ITEMS; r += 2)
    os_rank_rows(cnt[wave], (uint32_t)(key[r] >> shift) & mask, (uint32_t)(key[r + 1] >> shift) & mask, &rank[r], &rank[r + 1]);
  __syncthreads();
  // ---- per digit (thread = digit): tile count, wave offsets, tile-exclusive prefix, look-back
  {
    const bool dthread = tid < RS_RADIX;
    uint32_t c[PK_WAVES], tot = 0;
    if (dthread) {
#pragma unroll
      for (int w = 0; w < PK_WAVES; ++w) {
        c[w] = cnt[w][tid];
        cnt[w][tid] = tot;  // exclusive over waves
        tot += c[w];
      }
    }
    uint32_t block_total;
    const uint32_t ex = block_exclusive_scan<0>(tot, &block_total, lds_wave, lds_prev);  // (threads >= 256 only keep step)
    if (dthread) {
#pragma unroll
      for (int w = 0; w < PK_WAVES; ++w) cnt[w][tid] += ex;  // slot of the wave's first element of this digit in the tile
      const ST excl = os_lookback<ST>(status, tile, tid, tot);
      dst_base[tid] = gb + (uint32_t)excl - ex;
    }
  }
  __syncthreads();
  // ---- reorder the tile in LDS: keys first, then the values through the same buffer
  uint32_t pos[PK_ITEMS];
#pragma unroll
  for (int r = 0; r < PK_ITEMS; ++r) {
    const uint32_t d = (uint32_t)(key[r] >> shift) & mask;
    pos[r] = cnt[wave][d] + rank[r];
    lkeys[pos[r]] = key[r];
  }
  __syncthreads();
  const uint32_t tile_n = (uint32_t)((n - tile_base) < (uint64_t)PK_TILE ? (n - tile_base) : (uint64_t)PK_TILE);
  const bool full_tile = tile_n == (uint32_t)PK_TILE;
  uint32_t dst[PK_ITEMS];
  if (full_tile) {
#pragma unroll
    for (int r = 0; r < PK_ITEMS; ++r) {
      const uint32_t p = (uint32_t)r * PK_THREADS + tid;
      const uint64_t k = lkeys[p];
      dst[r] = dst_base[(uint32_t)(k >> shift) & mask] + p;
      keys_out[dst[r]] = k;
    }
  } else {
#pragma unroll
    for (int r = 0; r < PK_ITEMS; ++r) {
      const uint32_t p = (uint32_t)r * PK_THREADS + tid;
      dst[r] = 0;
      if (p < tile_n) {
        const uint64_t k = lkeys[p];
        dst[r] = dst_base[(uint32_t)(k >> shift) & mask] + p;
        keys_out[dst[r]] = k;
      }
    }
  }
  __syncthreads();
  uint32_t* lvals = reinterpret_cast<uint32_t*>(lkeys);
#pragma unroll
  for (int r = 0; r < PK_ITEMS; ++r) lvals[pos[r]] = val[r];
  __syncthreads();
  if (full_tile) {
#pragma unroll
    for (int r = 0; r < PK_ITEMS; ++r) vals_out[dst[r]] = lvals[(uint32_t)r * PK_THREADS + tid];
  } else {
#pragma unroll
    for (int r = 0; r < PK_ITEMS; ++r) {
      const uint32_t p = (uint32_t)r * PK_THREADS + tid;
      if (p < tile_n) vals_out[dst[r]] = lvals[p];
    }
  }
}

// The same pass over 8-byte packed words.  FIRST: the input is still the (key, value) pair, the digit is the key's lowest
// one, and what is written is ((key >> 8) << val_bits) | value; later passes read and write packed words, the digit of pass
// p sits at bit val_bits + 8 (p - 1).  One LDS staging round instead of two, 16 (20 for the first) bytes per element.
template <typename ST, bool FIRST>
__global__ __launch_bounds__(PK_THREADS) void os_pass_packed_kernel(const uint64_t* __restrict__ in,
                                                                     const uint32_t* __restrict__ vals_in,
                                                                     uint64_t* __restrict__ out, uint64_t n, int shift,
                                                                     uint32_t mask, int val_bits,
                                                                     const uint32_t* __restrict__ gbase, ST* status,
                                                                     uint32_t* ticket) {
  __shared__ uint64_t lkeys[PK_TILE];
  __shared__ uint32_t cnt[PK_WAVES][RS_RADIX];
  __shared__ uint32_t dst_base[RS_RADIX];
  __shared__ uint32_t lds_wave[PK_THREADS / 64];
  __shared__ uint32_t lds_prev[PK_THREADS];
  __shared__ uint32_t s_tile;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid == 0) s_tile = atomicAdd(ticket, 1u);
  const uint32_t gb = tid < RS_RADIX ? gbase[tid] : 0u;  // (early: its latency hides behind the ticket's)
  for (int i = tid; i < PK_WAVES * RS_RADIX; i += PK_THREADS) (&cnt[0][0])[i] = 0;
  __syncthreads();
  const uint32_t tile = s_tile;
  const uint64_t tile_base = (uint64_t)tile * PK_TILE;
  const uint64_t wbase = tile_base + (uint64_t)wave * (64 * PK_ITEMS);
  uint64_t key[PK_ITEMS];  // the word that is written
  uint32_t dig[PK_ITEMS];
  uint32_t rank[PK_ITEMS];
  if (tile_base + PK_TILE <= n) {  // every tile but the last: no bounds checks
    if (FIRST) {
      uint32_t v[PK_ITEMS];
      if (vals_in) {  // (one uniform branch around all the loads: a select per element would split them up)
#pragma unroll
        for (int r = 0; r < PK_ITEMS; ++r) v[r] = vals_in[wbase + (uint64_t)r * 64 + lane];
      } else {  // no array: the values are the identity
#pragma unroll
        for (int r = 0; r < PK_ITEMS; ++r) v[r] = (uint32_t)(wbase + (uint64_t)r * 64 + lane);
      }
#pragma unroll
      for (int r = 0; r < PK_ITEMS; ++r) {
        const uint64_t k = in[wbase + (uint64_t)r * 64 + lane];
        dig[r] = (uint32_t)k & mask;  // shift = 0
        key[r] = ((k >> 8) << val_bits) | v[r];
      }
    } else {
#pragma unroll
      for (int r = 0; r < PK_ITEMS; ++r) {
        key[r] = in[wbase + (uint64_t)r * 64 + lane];
        dig[r] = (uint32_t)(key[r] >> shift) & mask;
      }
    }
  } else {
#pragma unroll
    for (int r = 0; r < PK_ITEMS; ++r) {
      const uint64_t i = wbase + (uint64_t)r * 64 + lane;
      if (FIRST) {
        const uint64_t k = i < n ? in[i] : ~0ull;
        const uint32_t v = i < n ? (vals_in ? vals_in[i] : (uint32_t)i) : 0u;
        dig[r] = (uint32_t)k & mask;
        key[r] = ((k >> 8) << val_bits) | v;
      } else {
        key[r] = i < n ? in[i] : ~0ull;
        dig[r] = (uint32_t)(key[r] >> shift) & mask;
      }
    }
  }
#pragma unroll
  for (int r = 0; r < PK_ITEMS; r += 2) os_rank_rows(cnt[wave], dig[r], dig[r + 1], &rank[r], &rank[r + 1]);
  __syncthreads();
  {
    // per digit (threads 0..255): tile count, wave offsets, tile-exclusive prefix, look-back
    const bool dthread = tid < RS_RADIX;
    uint32_t c[PK_WAVES], tot = 0;
    if (dthread) {
#pragma unroll
      for (int w = 0; w < PK_WAVES; ++w) {
        c[w] = cnt[w][tid];
        cnt[w][tid] = tot;  // exclusive over waves
        tot += c[w];
      }
    }
    uint32_t block_total;
    const uint32_t ex = block_exclusive_scan<0>(tot, &block_total, lds_wave, lds_prev);  // (threads >= 256 only keep step)
    if (dthread) {
#pragma unroll
      for (int w = 0; w < PK_WAVES; ++w) cnt[w][tid] += ex;  // slot of the wave's first element of this digit in the tile
      const ST excl = os_lookback<ST>(status, tile, tid, tot);
      dst_base[tid] = gb + (uint32_t)excl - ex;
    }
  }
  __syncthreads();
  // reorder in LDS.  Later passes find a slot's digit in the staged word itself; the first pass's digit is no longer part of
  // the word it writes, so the digits are staged too, as bytes over the (then dead) per-wave counters.
#pragma unroll
  for (int r = 0; r < PK_ITEMS; ++r) {
    rank[r] += cnt[wave][dig[r]];  // the element's slot in the tile
    lkeys[rank[r]] = key[r];
  }
  __syncthreads();
  uint8_t* ldig = reinterpret_cast<uint8_t*>(&cnt[0][0]);
  static_assert(sizeof(cnt) >= PK_TILE, "digit bytes alias the wave counters");
  if (FIRST) {
#pragma unroll
    for (int r = 0; r < PK_ITEMS; ++r) ldig[rank[r]] = (uint8_t)dig[r];
    __syncthreads();
  }
  const uint32_t tile_n = (uint32_t)((n - tile_base) < (uint64_t)PK_TILE ? (n - tile_base) : (uint64_t)PK_TILE);
  if (tile_n == (uint32_t)PK_TILE) {
#pragma unroll
    for (int r = 0; r < PK_ITEMS; ++r) {
      const uint32_t p = (uint32_t)r * PK_THREADS + tid;
      const uint64_t k = lkeys[p];
      const uint32_t d = FIRST ? (uint32_t)ldig[p] : (uint32_t)(k >> shift) & mask;
      out[dst_base[d] + p] = k;
    }
  } else {
#pragma unroll
    for (int r = 0; r < PK_ITEMS; ++r) {
      const uint32_t p = (uint32_t)r * PK_THREADS + tid;
      if (p < tile_n) {
        const uint64_t k = lkeys[p];
        const uint32_t d = FIRST ? (uint32_t)ldig[p] : (uint32_t)(k >> shift) & mask;
        out[dst_base[d] + p] = k;
      }
    }
  }
}

// A packed pass over a 9-bit digit (512 bins): one pass fewer where the key bits left after the first pass divide into
// fewer 9-bit than 8-bit digits (34 bits: 9 + 9 + 8 + 8).  Same structure as the 8-bit pass; what differs:
//   * the per-wave running counters are 16-bit halves of 256 words (digit d: word d & 255, half d >> 8; a wave holds at most
//     1024 elements of a digit, a tile 8192) so that the LDS footprint stays at two work-groups per CU; the add of a row's
//     peers goes to the word, shifted into its half;
//   * threads 0..255 turn the counters of BOTH digits of their word into offsets -- the two tile-exclusive prefixes come out of
//     one 256-wide scan over (low count | high count << 16), the high digits' prefix starts at the total of the low ones -- and
//     hand the high digit's count to thread 256 + t, so that all 512 digits look back at once.
constexpr int P9_BINS = 512;
__device__ __forceinline__ void os_rank_rows9(uint32_t* wave_cnt, uint32_t d0, uint32_t d1, uint32_t* r0, uint32_t* r1) {
  os_match a{~0u, ~0u}, b{~0u, ~0u};
  const uint32_t w0 = __hip_atomic_load(wave_cnt + (d0 & 255u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
#pragma unroll
  for (int bit = 0; bit < 9; ++bit) {
    os_match_step(a, d0, bit);
    os_match_step(b, d1, bit);
  }
  const uint32_t below0 = a.below(), below1 = b.below();
  if (below0 == 0)
    __hip_atomic_fetch_add(wave_cnt + (d0 & 255u), a.peers() << ((d0 >> 8) * 16), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  const uint32_t w1 = __hip_atomic_load(wave_cnt + (d1 & 255u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  if (below1 == 0)
    __hip_atomic_fetch_add(wave_cnt + (d1 & 255u), b.peers() << ((d1 >> 8) * 16), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  *r0 = ((w0 >> ((d0 >> 8) * 16)) & 0xffffu) + below0;
  *r1 = ((w1 >> ((d1 >> 8) * 16)) & 0xffffu) + below1;
}

template <typename ST>
__global__ __launch_bounds__(PK_THREADS) void os_pass_packed9_kernel(const uint64_t* __restrict__ in, uint64_t* __restrict__ out,
                                                                      uint64_t n, int shift, const uint32_t* __restrict__ gbase,
                                                                      ST* status, uint32_t* ticket) {
  __shared__ uint64_t lkeys[PK_TILE];
  __shared__ uint32_t cnt[PK_WAVES][256];  // two 16-bit counters per word
  __shared__ uint32_t dst_base[P9_BINS];
  __shared__ uint32_t hi_tot[256], hi_ex[256];
  __shared__ uint32_t lds_wave[PK_THREADS / 64];
  __shared__ uint32_t lds_prev[PK_THREADS];
  __shared__ uint32_t s_tile;
  static_assert(PK_THREADS == P9_BINS, "one thread per digit in the look-back");
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid == 0) s_tile = atomicAdd(ticket, 1u);
  const uint32_t gb = gbase[tid];
  for (int i = tid; i < PK_WAVES * 256; i += PK_THREADS) (&cnt[0][0])[i] = 0;
  __syncthreads();
  const uint32_t tile = s_tile;
  const uint64_t tile_base = (uint64_t)tile * PK_TILE;
  const uint64_t wbase = tile_base + (uint64_t)wave * (64 * PK_ITEMS);
  uint64_t key[PK_ITEMS];
  uint32_t dig[PK_ITEMS];
  uint32_t rank[PK_ITEMS];
  if (tile_base + PK_TILE <= n) {  // every tile but the last: no bounds checks
#pragma unroll
    for (int r = 0; r < PK_ITEMS; ++r) {
      key[r] = in[wbase + (uint64_t)r * 64 + lane];
      dig[r] = (uint32_t)(key[r] >> shift) & (P9_BINS - 1);
    }
  } else {
#pragma unroll
    for (int r = 0; r < PK_ITEMS; ++r) {
      const uint64_t i = wbase + (uint64_t)r * 64 + lane;
      key[r] = i < n ? in[i] : ~0ull;
      dig[r] = (uint32_t)(key[r] >> shift) & (P9_BINS - 1);
    }
  }
#pragma unroll
  for (int r = 0; r < PK_ITEMS; r += 2) os_rank_rows9(cnt[wave], dig[r], dig[r + 1], &rank[r], &rank[r + 1]);
  __syncthreads();
  {
    uint32_t tot = 0;  // low count | high count << 16 of this thread's word (threads 0..255)
    if (tid < 256) {
#pragma unroll
      for (int w = 0; w < PK_WAVES; ++w) {
        const uint32_t c = cnt[w][tid];
        cnt[w][tid] = tot;  // exclusive over waves, both halves at once (a tile holds 8192 elements: no carry)
        tot += c;
      }
    }
    uint32_t block_total;
    const uint32_t ex = block_exclusive_scan<0>(tot, &block_total, lds_wave, lds_prev);  // (threads >= 256 only keep step)
    uint32_t my_tot, my_ex;
    if (tid < 256) {
      const uint32_t ex_lo = ex & 0xffffu, ex_hi = (block_total & 0xffffu) + (ex >> 16);
#pragma unroll
      for (int w = 0; w < PK_WAVES; ++w) cnt[w][tid] += ex_lo | (ex_hi << 16);  // slots of the waves' first elements
      hi_tot[tid] = tot >> 16;
      hi_ex[tid] = ex_hi;
      my_tot = tot & 0xffffu;
      my_ex = ex_lo;
    }
    __syncthreads();
    if (tid >= 256) {
      my_tot = hi_tot[tid - 256];
      my_ex = hi_ex[tid - 256];
    }
    const ST excl = os_lookback<ST>(status, tile, tid, my_tot, P9_BINS);
    dst_base[tid] = gb + (uint32_t)excl - my_ex;
  }
  __syncthreads();
#pragma unroll
  for (int r = 0; r < PK_ITEMS; ++r) {
    rank[r] += (cnt[wave][dig[r] & 255u] >> ((dig[r] >> 8) * 16)) & 0xffffu;  // the element's slot in the tile
    lkeys[rank[r]] = key[r];
  }
  __syncthreads();
  const uint32_t tile_n = (uint32_t)((n - tile_base) < (uint64_t)PK_TILE ? (n - tile_base) : (uint64_t)PK_TILE);
  if (tile_n == (uint32_t)PK_TILE) {
#pragma unroll
    for (int r = 0; r < PK_ITEMS; ++r) {
      const uint32_t p = (uint32_t)r * PK_THREADS + tid;
      const uint64_t k = lkeys[p];
      out[dst_base[(uint32_t)(k >> shift) & (P9_BINS - 1)] + p] = k;
    }
  } else {
#pragma unroll
    for (int r = 0; r < PK_ITEMS; ++r) {
      const uint32_t p = (uint32_t)r * PK_THREADS + tid;
      if (p < tile_n) {
        const uint64_t k = lkeys[p];
        out[dst_base[(uint32_t)(k >> shift) & (P9_BINS - 1)] + p] = k;
      }
    }
  }
}

// Fallback (n >= 2^30): histogram / scan / scatter per pass.
int radix_sort_three_kernel(swg_ctx* ctx, uint64_t** keys, uint32_t** vals, uint64_t** keys_alt, uint32_t** vals_alt,
                            uint64_t n, int begin_bit, int end_bit) {
  const uint32_t ntiles = (uint32_t)((n + RS_TILE - 1) / RS_TILE);
  swg_arena_mark mark = swg_arena_save(ctx);
  uint32_t* hist = swg_alloc<uint32_t>(ctx, (size_t)RS_RADIX * ntiles);
  SWG_CHECK_ARENA(ctx);
  for (int shift = begin_bit; shift < end_bit; shift += 8) {
    const int bits = end_bit - shift < 8 ? end_bit - shift : 8;
    const uint32_t mask = (1u << bits) - 1u;
    SWG_LAUNCH(ctx, "rs_hist", rs_hist_kernel<<<ntiles, RS_THREADS, 0, ctx->stream>>>(*keys, n, shift, mask, hist, ntiles));
    SWG_KERNEL_CHECK(ctx);
    SWG_TRY(swg_exclusive_scan_u32(ctx, hist, hist, (uint64_t)RS_RADIX * ntiles, nullptr));
    SWG_LAUNCH(ctx, "rs_scatter", rs_scatter_kernel<<<ntiles, RS_THREADS, 0, ctx->stream>>>(*keys, *vals, *keys_alt, *vals_alt, n, shift,
                                                                                 mask, hist, ntiles));
    SWG_KERNEL_CHECK(ctx);
    uint64_t* tk = *keys;
    *keys = *keys_alt;
    *keys_alt = tk;
    uint32_t* tv = *vals;
    *vals = *vals_alt;
    *vals_alt = tv;
  }
  swg_arena_restore(ctx, mark);
  return SWG_OK;
}

}  // namespace

static_assert(SWG_RADIX_BINS == 2 * RS_RADIX && SWG_RADIX_MAX_PASSES == OS_MAX_PASSES, "prehist layout");

swg_radix_plan swg_radix_plan_packed(int key_bits) {
  static const bool bits8 = getenv("SWG_SORT_BITS8") != nullptr;
  swg_radix_plan pl{};
  pl.shift[0] = 0;
  pl.bits[0] = (uint8_t)(key_bits < 8 ? key_bits : 8);
  pl.npasses = 1;
  const int rest = key_bits - 8;
  if (rest <= 0) return pl;
  const int q8 = (rest + 7) / 8, q9 = (rest + 8) / 9;
  if (bits8 || q9 == q8) {
    for (int p = 0; p < q8; ++p) {
      pl.shift[1 + p] = (uint8_t)(8 + 8 * p);
      pl.bits[1 + p] = (uint8_t)(rest - 8 * p < 8 ? rest - 8 * p : 8);
    }
    pl.npasses = 1 + q8;
  } else {  // q9 passes, as even as possible: the last `extra` of them one bit wider (at most 9).  The wide digits go on top:
            // the high key bits are segment bits, the same for a whole wavefront of a pair-major input, and a digit made of
            // them alone costs the histogram kernels one LDS add per wavefront; a digit that mixes two or three coordinate
            // bits into them takes four or eight values per wavefront and serialises its adds
    const int base = rest / q9, extra = rest % q9;
    int at = 8;
    for (int p = 0; p < q9; ++p) {
      const int b = base + (p >= q9 - extra ? 1 : 0);
      pl.shift[1 + p] = (uint8_t)at;
      pl.bits[1 + p] = (uint8_t)b;
      at += b;
    }
    pl.npasses = 1 + q9;
  }
  return pl;
}
// ---- sort on a truncated key (round 4) --------------------------------------------------------------------------------
// Words w = (k << val_bits) | value, sorted on ALL `sorted_bits` bits of k by plain 8-byte passes: no pass that reads wider
// elements, no key bits dropped from the word.  The callers get a pass fewer than swg_radix_sort_packed by leaving the LOW
// bits of their real key out of k altogether (k = key >> drop) and ordering the short runs of equal k afterwards, where they
// gather the records anyway (begin_gather_words_kernel, gather_all_words_kernel).
swg_radix_plan swg_radix_plan_words(int sorted_bits) {
  static const bool bits8 = getenv("SWG_SORT_BITS8") != nullptr;
  swg_radix_plan pl{};
  if (sorted_bits <= 0) return pl;
  const int q8 = (sorted_bits + 7) / 8, q9 = (sorted_bits + 8) / 9;
  const int q = (bits8 || q9 == q8) ? q8 : q9;
  if (q > SWG_RADIX_MAX_PASSES) return pl;
  // as even as possible, the wider digits on top (see swg_radix_plan_packed)
  const int base = sorted_bits / q, extra = sorted_bits % q;
  int at = 0;
  for (int p = 0; p < q; ++p) {
    const int b = base + (p >= q - extra ? 1 : 0);
    pl.shift[p] = (uint8_t)at;
    pl.bits[p] = (uint8_t)b;
    at += b;
  }
  pl.npasses = q;
  return pl;
}
// rough cost of a plan in ms per 10^8 elements (measured: 0.41 per 8-bit pass, 0.50 per 9-bit pass, 0.43 for the packed
// sort's first pass): what the choice of `drop` below compares
static double plan_cost(const swg_radix_plan& pl, bool packed_first) {
  double c = 0.0;
  for (int p = 0; p < pl.npasses; ++p) c += (p == 0 && packed_first) ? 0.43 : (pl.bits[p] > 8 ? 0.50 : 0.41);
  return c;
}
// How many low key bits to leave out of the sort (0: none, take the packed sort).  `level`: 0 = up to 10 bits, 1 = up to 7,
// 2+ = never -- the caller raises it when a run of equal truncated keys turned out longer than its gather can order.  A caller
// that knows its keys are sparse passes a larger first cap (`dmax0` > 10: one more level in front of the two).
int swg_radix_drop_bits(uint64_t n, int key_bits, int low_bits, int val_bits, int level, int dmax0) {
  static const char* knob = getenv("SWG_SORT_DROP");  // "0": never (test / A-B knob); "n": at most n bits
  static const bool force_fallback = getenv("SWG_SORT_FALLBACK") != nullptr, force_wide = getenv("SWG_SORT_WIDE") != nullptr,
                    no_packed = getenv("SWG_SORT_PAIRS") != nullptr;
  if (force_fallback || force_wide || no_packed || n < 2 || n >= (uint64_t(1) << 30)) return 0;
  int caps[3], nc = 0;
  if (dmax0 > 10) caps[nc++] = dmax0;
  caps[nc++] = 10;
  caps[nc++] = 7;
  if (level >= nc) return 0;
  int dmax = caps[level];
  if (knob) dmax = std::min(dmax, atoi(knob));
  if (dmax > low_bits) dmax = low_bits;
  const double now = plan_cost(swg_radix_plan_packed(key_bits), true);
  int best = 0;
  double best_cost = now - 0.2;  // a pass fewer, or nothing
  for (int d = 1; d <= dmax; ++d) {
    if (key_bits - d < 1 || key_bits - d + val_bits > 64) continue;
    const swg_radix_plan pl = swg_radix_plan_words(key_bits - d);
    if (pl.npasses == 0) continue;
    const double c = plan_cost(pl, false) + 0.004 * d;
    if (c < best_cost) {
      best_cost = c;
      best = d;
    }
  }
  return best;
}
int swg_radix_sort_words(swg_ctx* ctx, uint64_t* words, uint64_t* scratch, uint64_t n, int sorted_bits, int val_bits,
                         uint32_t* prehist, uint64_t** out) {
  const swg_radix_plan plan = swg_radix_plan_words(sorted_bits);
  const int npasses = plan.npasses;
  if (npasses == 0 || sorted_bits + val_bits > 64 || n < 2 || n >= (uint64_t(1) << 30)) return SWG_ERR_UNSUPPORTED;
  const uint32_t ntiles = (uint32_t)((n + PK_TILE - 1) / PK_TILE);
  const uint32_t htiles = (uint32_t)((n + OS_TILE - 1) / OS_TILE);
  swg_arena_mark mark = swg_arena_save(ctx);
  uint32_t* ghist = prehist ? prehist : swg_alloc<uint32_t>(ctx, (size_t)OS_MAX_PASSES * SWG_RADIX_BINS);
  uint32_t* status = swg_alloc<uint32_t>(ctx, (size_t)ntiles * SWG_RADIX_BINS);
  uint32_t* tickets = swg_alloc<uint32_t>(ctx, OS_MAX_PASSES);
  SWG_CHECK_ARENA(ctx);
  SWG_HIP(ctx, hipMemsetAsync(tickets, 0, sizeof(uint32_t) * OS_MAX_PASSES, ctx->stream));
  if (!prehist) {  // digits straight from the words
    swg_radix_plan wp = plan;
    for (int p = 0; p < npasses; ++p) wp.shift[p] = (uint8_t)(plan.shift[p] + val_bits);
    SWG_HIP(ctx, hipMemsetAsync(ghist, 0, sizeof(uint32_t) * OS_MAX_PASSES * SWG_RADIX_BINS, ctx->stream));
    uint32_t hb = htiles < (uint32_t)ctx->num_cu * 8 ? htiles : (uint32_t)ctx->num_cu * 8;
    SWG_LAUNCH(ctx, "os_hist", os_hist_kernel<<<hb, OS_THREADS, 0, ctx->stream>>>(words, n, wp, ghist));
    SWG_KERNEL_CHECK(ctx);
  }
  SWG_LAUNCH(ctx, "os_scan_hist", os_scan_hist_kernel<<<npasses, SWG_RADIX_BINS, 0, ctx->stream>>>(ghist));
  SWG_KERNEL_CHECK(ctx);
  uint64_t* src = words;
  uint64_t* dst = scratch;
  for (int p = 0; p < npasses; ++p) {
    const int bits = plan.bits[p];
    const uint32_t mask = (1u << bits) - 1u;
    const int wshift = val_bits + plan.shift[p];
    const uint32_t* gb = ghist + (size_t)p * SWG_RADIX_BINS;
    SWG_HIP(ctx, hipMemsetAsync(status, 0, sizeof(uint32_t) * (size_t)ntiles * (bits == 9 ? P9_BINS : RS_RADIX), ctx->stream));
    if (bits == 9)
      SWG_LAUNCH_N(ctx, "os_pass_packed9", n, os_pass_packed9_kernel<uint32_t><<<ntiles, PK_THREADS, 0, ctx->stream>>>(
                                              src, dst, n, wshift, gb, status, tickets + p));
    else
      SWG_LAUNCH_N(ctx, "os_pass_packed", n, os_pass_packed_kernel<uint32_t, false><<<ntiles, PK_THREADS, 0, ctx->stream>>>(
                                              src, nullptr, dst, n, wshift, mask, val_bits, gb, status, tickets + p));
    SWG_KERNEL_CHECK(ctx);
    uint64_t* t = src;
    src = dst;
    dst = t;
  }
  *out = src;
  swg_arena_restore(ctx, mark);
  return SWG_OK;
}

int swg_radix_sort_pairs(swg_ctx* ctx, uint64_t** keys, uint32_t** vals, uint64_t** keys_alt, uint32_t** vals_alt,
                         uint64_t n, int begin_bit, int end_bit, uint32_t* prehist) {
  if (n <= 1 || end_bit <= begin_bit) return SWG_OK;
  if (n >= (uint64_t(1) << 32)) return swg_set_error(ctx, SWG_ERR_RANGE, "radix sort: n >= 2^32");
  const int npasses = (end_bit - begin_bit + 7) / 8;
  const swg_radix_plan plan = swg_radix_plan_pairs(begin_bit, end_bit);
  // SWG_SORT_FALLBACK=1 forces the histogram/scan/scatter path (otherwise only reached for n >= 2^30) so that the
  // tests can exercise it at small sizes
  // SWG_SORT_WIDE=1 forces the 64-bit look-back words (otherwise only used for n >= 2^30)
  static const bool force_fallback = getenv("SWG_SORT_FALLBACK") != nullptr;
  static const bool force_wide = getenv("SWG_SORT_WIDE") != nullptr;
  if (force_fallback || npasses > OS_MAX_PASSES)
    return radix_sort_three_kernel(ctx, keys, vals, keys_alt, vals_alt, n, begin_bit, end_bit);
  const bool wide = force_wide || n >= (uint64_t(1) << 30);
  const size_t word = wide ? sizeof(uint64_t) : sizeof(uint32_t);
  const uint32_t ntiles = (uint32_t)((n + PK_TILE - 1) / PK_TILE);
  const uint32_t htiles = (uint32_t)((n + OS_TILE - 1) / OS_TILE);
  swg_arena_mark mark = swg_arena_save(ctx);
  uint32_t* ghist = prehist ? prehist : swg_alloc<uint32_t>(ctx, (size_t)OS_MAX_PASSES * SWG_RADIX_BINS);
  void* status = swg_arena_alloc(ctx, (size_t)ntiles * RS_RADIX * word);
  uint32_t* tickets = swg_alloc<uint32_t>(ctx, OS_MAX_PASSES);
  SWG_CHECK_ARENA(ctx);
  SWG_HIP(ctx, hipMemsetAsync(tickets, 0, sizeof(uint32_t) * OS_MAX_PASSES, ctx->stream));
  {
    if (!prehist) {
      SWG_HIP(ctx, hipMemsetAsync(ghist, 0, sizeof(uint32_t) * OS_MAX_PASSES * SWG_RADIX_BINS, ctx->stream));
      uint32_t hb = htiles < (uint32_t)ctx->num_cu * 8 ? htiles : (uint32_t)ctx->num_cu * 8;
      SWG_LAUNCH(ctx, "os_hist", os_hist_kernel<<<hb, OS_THREADS, 0, ctx->stream>>>(*keys, n, plan, ghist));
      SWG_KERNEL_CHECK(ctx);
    }
    SWG_LAUNCH(ctx, "os_scan_hist", os_scan_hist_kernel<<<npasses, SWG_RADIX_BINS, 0, ctx->stream>>>(ghist));
    SWG_KERNEL_CHECK(ctx);
  }
  for (int p = 0; p < npasses; ++p) {
    const int shift = begin_bit + 8 * p;
    const int bits = end_bit - shift < 8 ? end_bit - shift : 8;
    const uint32_t mask = (1u << bits) - 1u;
    SWG_HIP(ctx, hipMemsetAsync(status, 0, word * (size_t)ntiles * RS_RADIX, ctx->stream));
    if (wide)
      SWG_LAUNCH_N(ctx, "os_pass", n, os_pass_kernel<uint64_t><<<ntiles, PK_THREADS, 0, ctx->stream>>>(
                                     *keys, *vals, *keys_alt, *vals_alt, n, shift, mask, ghist + (size_t)p * SWG_RADIX_BINS,
                                     static_cast<uint64_t*>(status), tickets + p));
    else
      SWG_LAUNCH_N(ctx, "os_pass", n, os_pass_kernel<uint32_t><<<ntiles, PK_THREADS, 0, ctx->stream>>>(
                                     *keys, *vals, *keys_alt, *vals_alt, n, shift, mask, ghist + (size_t)p * SWG_RADIX_BINS,
                                     static_cast<uint32_t*>(status), tickets + p));
    SWG_KERNEL_CHECK(ctx);
    uint64_t* tk = *keys;
    *keys = *keys_alt;
    *keys_alt = tk;
    uint32_t* tv = *vals;
    *vals = *vals_alt;
    *vals_alt = tv;
  }
  swg_arena_restore(ctx, mark);
  return SWG_OK;
}

bool swg_radix_sort_packed_applies(uint64_t n, int key_bits, int val_bits) {
  static const bool force_fallback = getenv("SWG_SORT_FALLBACK") != nullptr;
  static const bool force_wide = getenv("SWG_SORT_WIDE") != nullptr;
  static const bool no_packed = getenv("SWG_SORT_PAIRS") != nullptr;  // test knob: the 12-byte passes everywhere
  const int npasses = (key_bits + 7) / 8;
  return !(force_fallback || force_wide || no_packed || n < 2 || n >= (uint64_t(1) << 30) || npasses < 2 || npasses > OS_MAX_PASSES ||
           key_bits - 8 + val_bits > 64 || val_bits < 1 || val_bits > 32);
}

int swg_radix_sort_packed(swg_ctx* ctx, uint64_t* keys, const uint32_t* vals, uint64_t* scratch, uint64_t n, int key_bits,
                          int val_bits, uint32_t* prehist, uint64_t** packed_out) {
  if (!swg_radix_sort_packed_applies(n, key_bits, val_bits)) return SWG_ERR_UNSUPPORTED;
  const swg_radix_plan plan = swg_radix_plan_packed(key_bits);
  const int npasses = plan.npasses;
  const uint32_t ntiles = (uint32_t)((n + PK_TILE - 1) / PK_TILE);
  const uint32_t htiles = (uint32_t)((n + OS_TILE - 1) / OS_TILE);
  swg_arena_mark mark = swg_arena_save(ctx);
  uint32_t* ghist = prehist ? prehist : swg_alloc<uint32_t>(ctx, (size_t)OS_MAX_PASSES * SWG_RADIX_BINS);
  uint32_t* status = swg_alloc<uint32_t>(ctx, (size_t)ntiles * SWG_RADIX_BINS);
  uint32_t* tickets = swg_alloc<uint32_t>(ctx, OS_MAX_PASSES);
  SWG_CHECK_ARENA(ctx);
  SWG_HIP(ctx, hipMemsetAsync(tickets, 0, sizeof(uint32_t) * OS_MAX_PASSES, ctx->stream));
  if (!prehist) {
    SWG_HIP(ctx, hipMemsetAsync(ghist, 0, sizeof(uint32_t) * OS_MAX_PASSES * SWG_RADIX_BINS, ctx->stream));
    uint32_t hb = htiles < (uint32_t)ctx->num_cu * 8 ? htiles : (uint32_t)ctx->num_cu * 8;
    SWG_LAUNCH(ctx, "os_hist", os_hist_kernel<<<hb, OS_THREADS, 0, ctx->stream>>>(keys, n, plan, ghist));
    SWG_KERNEL_CHECK(ctx);
  }
  SWG_LAUNCH(ctx, "os_scan_hist", os_scan_hist_kernel<<<npasses, SWG_RADIX_BINS, 0, ctx->stream>>>(ghist));
  SWG_KERNEL_CHECK(ctx);
  uint64_t* src = keys;
  uint64_t* dst = scratch;
  for (int p = 0; p < npasses; ++p) {
    const int bits = plan.bits[p];
    const uint32_t mask = (1u << bits) - 1u;
    const int wshift = val_bits + plan.shift[p] - 8;  // where the digit sits in the packed word (p >= 1)
    const uint32_t* gb = ghist + (size_t)p * SWG_RADIX_BINS;
    SWG_HIP(ctx, hipMemsetAsync(status, 0, sizeof(uint32_t) * (size_t)ntiles * (bits == 9 ? P9_BINS : RS_RADIX), ctx->stream));
    if (p == 0)
      SWG_LAUNCH_N(ctx, "os_pass_packed_first", n, os_pass_packed_kernel<uint32_t, true><<<ntiles, PK_THREADS, 0, ctx->stream>>>(
                                             src, vals, dst, n, 0, mask, val_bits, gb, status, tickets));
    else if (bits == 9)
      SWG_LAUNCH_N(ctx, "os_pass_packed9", n, os_pass_packed9_kernel<uint32_t><<<ntiles, PK_THREADS, 0, ctx->stream>>>(
                                              src, dst, n, wshift, gb, status, tickets + p));
    else
      SWG_LAUNCH_N(ctx, "os_pass_packed", n, os_pass_packed_kernel<uint32_t, false><<<ntiles, PK_THREADS, 0, ctx->stream>>>(
                                              src, nullptr, dst, n, wshift, mask, val_bits, gb, status, tickets + p));
    SWG_KERNEL_CHECK(ctx);
    uint64_t* t = src;
    src = dst;
    dst = t;
  }
  *packed_out = src;
  swg_arena_restore(ctx, mark);
  return SWG_OK;
}
