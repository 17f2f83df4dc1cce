// Context, error text and the scratch arena of libsweepga_gpu.so.
#include <cstdarg>
#include <cstdio>

#include <algorithm>
#include <vector>

#include <cstdlib>

#include "swg_internal.h"

thread_local std::string swg_create_error;

int swg_set_error(swg_ctx* ctx, int code, const char* fmt, ...) {
  char buf[1024];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  if (ctx)
    ctx->err = buf;
  else
    swg_create_error = buf;
  return code;
}

void* swg_arena_alloc(swg_ctx* ctx, size_t bytes) {
  size_t aligned = (bytes + 255) & ~size_t(255);
  size_t off = ctx->arena_off;
  ctx->arena_off = off + aligned;
  if (ctx->arena_off > ctx->arena_peak) ctx->arena_peak = ctx->arena_off;
  if (ctx->arena_off > ctx->arena_cap) {
    ctx->arena_overflow = true;
    return nullptr;
  }
  return ctx->arena + off;
}

void swg_arena_reset(swg_ctx* ctx) {
  ctx->arena_off = 0;
  ctx->arena_peak = 0;
  ctx->arena_overflow = false;
}

int swg_arena_reserve(swg_ctx* ctx, size_t bytes) {
  if (bytes <= ctx->arena_cap) return SWG_OK;
  SWG_HIP(ctx, hipStreamSynchronize(ctx->stream));
  if (ctx->arena) {
    SWG_HIP(ctx, hipFree(ctx->arena));
    ctx->arena = nullptr;
    ctx->arena_cap = 0;
  }
  void* p = nullptr;
  hipError_t e = hipMalloc(&p, bytes);
  if (e != hipSuccess)
    return swg_set_error(ctx, SWG_ERR_OOM, "hipMalloc of %zu-byte scratch arena failed: %s", bytes,
                         hipGetErrorString(e));
  ctx->arena = static_cast<char*>(p);
  ctx->arena_cap = bytes;
  return SWG_OK;
}

// u64 interval coordinates of one axis -> u32.  Values below 2^32 pass through.  Otherwise (the reference's own test
// puts a mapping at u64::MAX, plane_sweep_exact.rs:804-826) every stretch that no interval covers is shrunk to 1: a
// plane sweep only depends on the order of the event coordinates and on lengths inside covered stretches, so this
// is exact for the sweep seams.  Chaining measures uncovered gaps and never goes through here.
int swg_narrow_coords(swg_ctx* ctx, uint64_t n, const uint64_t* s0, const uint64_t* e0, uint32_t* out_s, uint32_t* out_e,
                      const char* axis) {
  bool wide = false;
  for (uint64_t i = 0; i < n && !wide; ++i) wide = s0[i] > 0xffffffffull || e0[i] > 0xffffffffull;
  if (!wide) {
    for (uint64_t i = 0; i < n; ++i) {
      out_s[i] = (uint32_t)s0[i];
      out_e[i] = (uint32_t)e0[i];
    }
    return SWG_OK;
  }
  std::vector<uint64_t> c;
  c.reserve(2 * n);
  for (uint64_t i = 0; i < n; ++i) {
    c.push_back(s0[i]);
    c.push_back(e0[i]);
  }
  std::sort(c.begin(), c.end());
  c.erase(std::unique(c.begin(), c.end()), c.end());
  auto rank = [&](uint64_t v) { return (size_t)(std::lower_bound(c.begin(), c.end(), v) - c.begin()); };
  std::vector<int64_t> cover(c.size() + 1, 0);
  for (uint64_t i = 0; i < n; ++i)
    if (e0[i] > s0[i]) {
      ++cover[rank(s0[i])];
      --cover[rank(e0[i])];
    }
  std::vector<uint64_t> x(c.size(), 0);
  int64_t depth = 0;
  for (size_t k = 0; k + 1 < c.size(); ++k) {
    depth += cover[k];
    const uint64_t step = depth > 0 ? c[k + 1] - c[k] : 1;
    x[k + 1] = x[k] + step;
    if (x[k + 1] > 0xffffffffull || x[k + 1] < x[k])
      return swg_set_error(ctx, SWG_ERR_RANGE, "covered span of the %s axis does not fit 32 bits", axis);
  }
  for (uint64_t i = 0; i < n; ++i) {
    out_s[i] = (uint32_t)x[rank(s0[i])];
    out_e[i] = (uint32_t)x[rank(e0[i])];
  }
  return SWG_OK;
}

int swg_read_scalars(swg_ctx* ctx, const uint64_t* d_src, uint64_t* h_dst, int count) {
  if (count > 64) return swg_set_error(ctx, SWG_ERR_INVALID, "swg_read_scalars: count > 64");
  ++ctx->n_readbacks;
  SWG_HIP(ctx, hipMemcpyAsync(ctx->h_scalars, d_src, sizeof(uint64_t) * count, hipMemcpyDeviceToHost,
                              ctx->stream));
  SWG_HIP(ctx, hipStreamSynchronize(ctx->stream));
  for (int i = 0; i < count; ++i) h_dst[i] = ctx->h_scalars[i];
  return SWG_OK;
}

swg_prof_scope::swg_prof_scope(swg_ctx* c, const char* kernel_name, uint64_t units) : ctx(c) {
  if (!ctx || !ctx->prof_on) return;
  if (!ctx->prof_only.empty() && ctx->prof_only != kernel_name) return;  // swg_profile_select: events around one kernel only
  for (size_t i = 0; i < ctx->prof_entries.size(); ++i)
    if (ctx->prof_entries[i].name == kernel_name) name = (int)i;
  if (name < 0) {
    ctx->prof_entries.push_back({kernel_name, 0, 0.0, 0});
    name = (int)ctx->prof_entries.size() - 1;
  }
  ctx->prof_entries[name].units += units;
  auto get = [&]() -> hipEvent_t {
    if (!ctx->prof_free_events.empty()) {
      hipEvent_t e = ctx->prof_free_events.back();
      ctx->prof_free_events.pop_back();
      return e;
    }
    hipEvent_t e = nullptr;
    if (hipEventCreate(&e) != hipSuccess) return nullptr;
    return e;
  };
  a = get();
  b = get();
  if (a) (void)hipEventRecord(a, ctx->stream);
}

swg_prof_scope::~swg_prof_scope() {
  if (!ctx || name < 0) return;
  if (a && b) {
    (void)hipEventRecord(b, ctx->stream);
    ctx->prof_pending_list.push_back({name, a, b});
  }
}

int swg_prof_collect(swg_ctx* ctx) {
  if (ctx->prof_pending_list.empty()) return SWG_OK;
  SWG_HIP(ctx, hipStreamSynchronize(ctx->stream));
  for (auto& p : ctx->prof_pending_list) {
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, p.a, p.b) == hipSuccess) {
      ctx->prof_entries[p.name].launches += 1;
      ctx->prof_entries[p.name].ms += ms;
    }
    ctx->prof_free_events.push_back(p.a);
    ctx->prof_free_events.push_back(p.b);
  }
  ctx->prof_pending_list.clear();
  return SWG_OK;
}

extern "C" {

int swg_profile_enable(swg_ctx* ctx, int on) {
  if (!ctx) return SWG_ERR_INVALID;
  if (!on) SWG_TRY(swg_prof_collect(ctx));
  ctx->prof_on = on != 0;
  return SWG_OK;
}
int swg_profile_select(swg_ctx* ctx, const char* kernel_name) {
  if (!ctx) return SWG_ERR_INVALID;
  SWG_TRY(swg_prof_collect(ctx));
  ctx->prof_only = kernel_name ? kernel_name : "";
  return SWG_OK;
}
int swg_profile_reset(swg_ctx* ctx) {
  if (!ctx) return SWG_ERR_INVALID;
  SWG_TRY(swg_prof_collect(ctx));
  ctx->prof_entries.clear();
  return SWG_OK;
}
int swg_profile_count(swg_ctx* ctx) {
  if (!ctx) return SWG_ERR_INVALID;
  if (swg_prof_collect(ctx) != SWG_OK) return SWG_ERR_HIP;
  return (int)ctx->prof_entries.size();
}
int swg_profile_get(swg_ctx* ctx, int i, const char** name, uint64_t* launches, double* total_ms) {
  if (!ctx || i < 0 || i >= (int)ctx->prof_entries.size()) return SWG_ERR_INVALID;
  if (name) *name = ctx->prof_entries[i].name.c_str();
  if (launches) *launches = ctx->prof_entries[i].launches;
  if (total_ms) *total_ms = ctx->prof_entries[i].ms;
  return SWG_OK;
}

int swg_profile_units(swg_ctx* ctx, int i, uint64_t* units) {
  if (!ctx || i < 0 || i >= (int)ctx->prof_entries.size() || !units) return SWG_ERR_INVALID;
  *units = ctx->prof_entries[i].units;
  return SWG_OK;
}

int swg_abi_version(void) { return SWG_ABI_VERSION; }

int swg_create(int device, swg_ctx** out) {
  if (!out) return swg_set_error(nullptr, SWG_ERR_INVALID, "swg_create: out is NULL");
  *out = nullptr;
  int count = 0;
  hipError_t e = hipGetDeviceCount(&count);
  if (e != hipSuccess || count <= 0)
    return swg_set_error(nullptr, SWG_ERR_NO_DEVICE,
                         "no HIP device available (%s); libsweepga_gpu has no CPU fallback",
                         e != hipSuccess ? hipGetErrorString(e) : "device count is 0");
  if (device < 0 || device >= count)
    return swg_set_error(nullptr, SWG_ERR_INVALID, "device %d out of range (0..%d)", device, count - 1);
  swg_ctx* ctx = new (std::nothrow) swg_ctx();
  if (!ctx) return swg_set_error(nullptr, SWG_ERR_OOM, "host allocation failed");
  ctx->device = device;
  auto fail = [&](const char* what, hipError_t err) {
    swg_set_error(nullptr, SWG_ERR_NO_DEVICE, "%s failed: %s", what, hipGetErrorString(err));
    swg_destroy(ctx);
    return SWG_ERR_NO_DEVICE;
  };
  if ((e = hipSetDevice(device)) != hipSuccess) return fail("hipSetDevice", e);
  hipDeviceProp_t prop;
  if ((e = hipGetDeviceProperties(&prop, device)) != hipSuccess) return fail("hipGetDeviceProperties", e);
  ctx->num_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  if ((e = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking)) != hipSuccess)
    return fail("hipStreamCreate", e);
  if ((e = hipEventCreate(&ctx->ev0)) != hipSuccess) return fail("hipEventCreate", e);
  if ((e = hipEventCreate(&ctx->ev1)) != hipSuccess) return fail("hipEventCreate", e);
  void* hp = nullptr;
  if ((e = hipHostMalloc(&hp, 64 * sizeof(uint64_t), hipHostMallocDefault)) != hipSuccess)
    return fail("hipHostMalloc", e);
  ctx->h_scalars = static_cast<uint64_t*>(hp);
  *out = ctx;
  return SWG_OK;
}

void swg_destroy(swg_ctx* ctx) {
  if (!ctx) return;
  if (ctx->device >= 0) (void)hipSetDevice(ctx->device);
  if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
  if (ctx->arena) (void)hipFree(ctx->arena);
  if (ctx->io_block) (void)hipFree(ctx->io_block);
  std::free(ctx->narrow_host);
  if (ctx->h_scalars) (void)hipHostFree(ctx->h_scalars);
  if (ctx->ring) (void)hipHostFree(ctx->ring);
  for (auto& e : ctx->ring_ev)
    if (e) (void)hipEventDestroy(e);
  if (ctx->ev0) (void)hipEventDestroy(ctx->ev0);
  if (ctx->ev1) (void)hipEventDestroy(ctx->ev1);
  for (auto& p : ctx->prof_pending_list) {
    (void)hipEventDestroy(p.a);
    (void)hipEventDestroy(p.b);
  }
  for (auto e : ctx->prof_free_events) (void)hipEventDestroy(e);
  if (ctx->copy_stream) {
    (void)hipStreamSynchronize(ctx->copy_stream);
    (void)hipStreamDestroy(ctx->copy_stream);
  }
  if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
  delete ctx;
}

int swg_memory_info(const swg_ctx* ctx, uint64_t* arena_capacity, uint64_t* arena_peak_last_call) {
  if (!ctx) return SWG_ERR_INVALID;
  if (arena_capacity) *arena_capacity = ctx->arena_cap;
  if (arena_peak_last_call) *arena_peak_last_call = ctx->arena_peak;
  return SWG_OK;
}
int swg_reserve(swg_ctx* ctx, uint64_t arena_bytes) {
  if (!ctx) return SWG_ERR_INVALID;
  SWG_HIP(ctx, hipSetDevice(ctx->device));
  return swg_arena_reserve(ctx, (size_t)arena_bytes);
}

const char* swg_last_error(const swg_ctx* ctx) { return ctx ? ctx->err.c_str() : swg_create_error.c_str(); }

void* swg_stream(swg_ctx* ctx) { return ctx ? static_cast<void*>(ctx->stream) : nullptr; }

int swg_synchronize(swg_ctx* ctx) {
  if (!ctx) return SWG_ERR_INVALID;
  SWG_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return SWG_OK;
}

}  // extern "C"
