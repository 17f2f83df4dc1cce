// Times and checks the library's radix sorts on their own (a GPU box tool, not part of the pytest suites):
//   hipcc --offload-arch=gfx950 -O2 -std=c++17 tests/native/sort_bench.cpp -o gpurun_out/sort_bench
//         -Lsweepga_amd -lsweepga_gpu -Wl,-rpath,$PWD/sweepga_amd
//   gpurun_out/sort_bench [n=100000000] [key_bits=42] [reps=5] [packed=1]     (packed=2: no value array, identity taken as read)
// Keys are uniform random key_bits-bit words, values the identity.  Every repetition is verified on the host: the output
// is ordered by (key, value) -- the stable order -- and the values are a permutation (checksum + strict order).
// Prints the per-repetition time of the whole sort (HIP events on the context's stream).
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../../sweepga_amd/csrc/swg_internal.h"

#define CK(x)                                                                   \
  do {                                                                          \
    hipError_t e_ = (x);                                                        \
    if (e_ != hipSuccess) {                                                     \
      fprintf(stderr, "%s: %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); \
      return 2;                                                                 \
    }                                                                           \
  } while (0)

__global__ void fill_kernel(uint64_t n, int key_bits, uint64_t seed, uint64_t* keys, uint32_t* vals) {
  const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint64_t x = (i + seed) * 0x9e3779b97f4a7c15ull;
  x ^= x >> 32;
  x *= 0xd6e8feb86659fd93ull;
  x ^= x >> 32;
  x *= 0xd6e8feb86659fd93ull;
  x ^= x >> 32;
  keys[i] = key_bits >= 64 ? x : x & ((uint64_t(1) << key_bits) - 1);
  vals[i] = (uint32_t)i;
}

int main(int argc, char** argv) {
  const uint64_t n = argc > 1 ? strtoull(argv[1], nullptr, 10) : 100000000ull;
  const int key_bits = argc > 2 ? atoi(argv[2]) : 42;
  const int reps = argc > 3 ? atoi(argv[3]) : 5;
  const int packed_arg = argc > 4 ? atoi(argv[4]) : 1;
  const bool packed = packed_arg != 0;
  const bool verify = getenv("SORT_BENCH_NO_VERIFY") == nullptr;
  swg_ctx* ctx = nullptr;
  if (swg_create(0, &ctx) != SWG_OK) {
    fprintf(stderr, "swg_create: %s\n", swg_last_error(nullptr));
    return 2;
  }
  if (swg_arena_reserve(ctx, (size_t(256) << 20) + n / 8) != SWG_OK) return 2;
  uint64_t *keys = nullptr, *keys_alt = nullptr;
  uint32_t *vals = nullptr, *vals_alt = nullptr;
  CK(hipMalloc(&keys, n * 8));
  CK(hipMalloc(&keys_alt, n * 8));
  CK(hipMalloc(&vals, n * 4));
  CK(hipMalloc(&vals_alt, n * 4));
  int idx_bits = 1;
  while ((uint64_t(1) << idx_bits) < n) ++idx_bits;
  hipEvent_t a, b;
  CK(hipEventCreate(&a));
  CK(hipEventCreate(&b));
  std::vector<uint64_t> h(verify ? n : 0);
  std::vector<uint32_t> hv(verify && !packed ? n : 0);
  int bad = 0;
  for (int rep = 0; rep < reps; ++rep) {
    fill_kernel<<<(unsigned)((n + 255) / 256), 256, 0, ctx->stream>>>(n, key_bits, 1000003ull * rep, keys, vals);
    CK(hipGetLastError());
    swg_arena_reset(ctx);
    uint64_t *k = keys, *ka = keys_alt, *out = nullptr;
    uint32_t *v = vals, *va = vals_alt;
    CK(hipEventRecord(a, ctx->stream));
    int rc;
    if (packed)
      rc = swg_radix_sort_packed(ctx, k, packed_arg == 2 ? nullptr : v, ka, n, key_bits, idx_bits, nullptr, &out);
    else
      rc = swg_radix_sort_pairs(ctx, &k, &v, &ka, &va, n, 0, key_bits);
    CK(hipEventRecord(b, ctx->stream));
    CK(hipStreamSynchronize(ctx->stream));
    if (rc != SWG_OK) {
      fprintf(stderr, "sort returned %d: %s\n", rc, swg_last_error(ctx));
      return 2;
    }
    float ms = 0;
    CK(hipEventElapsedTime(&ms, a, b));
    const int passes = (key_bits + 7) / 8;
    printf("rep %d: %.3f ms  (%d passes, %.1f G elements/s per pass incl. histogram)\n", rep, ms, passes,
           (double)n * passes / ms / 1e6);
    if (!verify) continue;
    uint64_t sum = 0;
    bool ok = true;
    if (packed) {
      CK(hipMemcpy(h.data(), out, n * 8, hipMemcpyDeviceToHost));
      // packed word = ((key >> 8) << idx_bits) | index: ascending (key >> 8, ...) is not the whole order -- the low digit went
      // first -- so recompute each element's key from its index and compare full (key, index) pairs
      const uint64_t imask = (uint64_t(1) << idx_bits) - 1;
      uint64_t pk = 0, pi = 0;
      for (uint64_t i = 0; i < n && ok; ++i) {
        const uint64_t idx = h[i] & imask;
        uint64_t x = (idx + 1000003ull * rep) * 0x9e3779b97f4a7c15ull;
        x ^= x >> 32; x *= 0xd6e8feb86659fd93ull; x ^= x >> 32; x *= 0xd6e8feb86659fd93ull; x ^= x >> 32;
        const uint64_t key = key_bits >= 64 ? x : x & ((uint64_t(1) << key_bits) - 1);
        if ((key >> 8) != (h[i] >> idx_bits)) ok = false;
        if (i && (key < pk || (key == pk && idx <= pi))) ok = false;
        pk = key; pi = idx; sum += idx;
      }
    } else {
      CK(hipMemcpy(h.data(), k, n * 8, hipMemcpyDeviceToHost));
      CK(hipMemcpy(hv.data(), v, n * 4, hipMemcpyDeviceToHost));
      for (uint64_t i = 0; i < n && ok; ++i) {
        const uint64_t idx = hv[i];
        uint64_t x = (idx + 1000003ull * rep) * 0x9e3779b97f4a7c15ull;
        x ^= x >> 32; x *= 0xd6e8feb86659fd93ull; x ^= x >> 32; x *= 0xd6e8feb86659fd93ull; x ^= x >> 32;
        const uint64_t key = key_bits >= 64 ? x : x & ((uint64_t(1) << key_bits) - 1);
        if (key != h[i]) ok = false;
        if (i && (h[i] < h[i - 1] || (h[i] == h[i - 1] && hv[i] <= hv[i - 1]))) ok = false;
        sum += idx;
      }
    }
    if (ok && sum != n * (n - 1) / 2) ok = false;
    printf("rep %d: %s\n", rep, ok ? "sorted, stable, a permutation" : "WRONG");
    if (!ok) ++bad;
  }
  swg_destroy(ctx);
  return bad ? 1 : 0;
}
