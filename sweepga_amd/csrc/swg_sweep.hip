// Plane sweep (src/plane_sweep_exact.rs:268-433) for gfx950, one axis, all segments at once.
//
// Reference semantics (closed form): inside a segment, with priority
// (score desc, axis start asc, index asc), for every event coordinate x let A(x) be the
// intervals with s <= x < e and T(x) the first min(k,|A|) of A(x) by priority.  Then
//     keep_i = (exists x in [s_i,e_i): i in T(x))                               "ever top-k"
//              and not (thr < 1 and exists x: i in A(x)\T(x), some t in T(x), ovl(i,t) > thr)
// and a segment of size <= 1 is returned whole.  The `overlapped` flag is sticky
// (plane_sweep_exact.rs:251-254).  The sequential BTreeSet sweep is replaced by an evaluation
// that is independent per event coordinate:
//
//   1. every interval contributes a Begin and an End event keyed by the composite coordinate
//      X = ((segment+1) << pos_bits) | position; one radix sort orders all events of all
//      segments (dead intervals get X = 0 and fall out in front);
//   2. the sorted event stream is cut into tiles of TE events.  An interval whose Begin lies in
//      an earlier tile and whose End lies in or after tile b is a *carry-in* of tile b; carry-in
//      lists are built once (count, scan, fill) with the interval's keys inlined so the tile
//      kernel streams them;
//   3. one workgroup per tile, one thread per event: thread p evaluates coordinate x_p against
//      the candidates that can be active there -- the tile's own Begin events (scanned
//      backwards from p, cut short by a prefix maximum of interval ends held in LDS) and the
//      carry-ins (streamed through LDS in chunks).  Pass 1 finds T(x_p), pass 2 marks
//      `ever-top` for its members and `overlapped` for every other active interval whose
//      overlap fraction with a member exceeds thr (f64 division, as the reference).
//      k == 1 keeps T(x) in registers; 2 <= k < inf walks the priority order k times
//      (successive minima) so any k works without per-thread storage; k == inf needs no sweep.
//   4. flags are combined: keep = single-in-segment | (ever_top & !overlapped).
//
// Equal coordinates: the reference applies all events at one position before marking
// (plane_sweep_exact.rs:306-334), so a coordinate is evaluated once, by the last event of its
// run, and a run that continues into the next tile is left to that tile (whose carry-ins then
// contain every interval that began at that coordinate earlier).
#include "swg_internal.h"
#include "swg_log.h"

namespace {

constexpr int TE = 256;       // events per tile == threads per workgroup
constexpr int CC = 256;       // carry-in entries staged per LDS chunk
constexpr int EW_THREADS = 256;

struct Prio {  // priority: smaller = better
  uint64_t key;  // sortable -score
  uint64_t s;    // composite axis start
  uint32_t id;   // interval index (input order)
};
__device__ __forceinline__ bool prio_less(uint64_t ak, uint64_t as, uint32_t ai, uint64_t bk, uint64_t bs,
                                          uint32_t bi) {
  if (ak != bk) return ak < bk;
  if (as != bs) return as < bs;
  return ai < bi;
}

// ---- score keys ---------------------------------------------------------------------------
// src/plane_sweep_exact.rs:29-86: length = q_end - q_start (query span on BOTH axes);
// -inf when length <= 0 or (for the identity-using scores) identity <= 0.
__device__ __forceinline__ uint64_t sortable_desc(double score) {
  if (score == 0.0) score = 0.0;  // -0 -> +0 (partial_cmp treats them equal)
  uint64_t b = (uint64_t)__double_as_longlong(score);
  if (score != score) return ~0ull;  // NaN: worst (the reference's order is not total here)
  uint64_t asc = (b >> 63) ? ~b : (b | 0x8000000000000000ull);
  return ~asc;
}

__global__ __launch_bounds__(EW_THREADS) void score_key_kernel(uint64_t n, const uint32_t* __restrict__ qs,
                                                               const uint32_t* __restrict__ qe,
                                                               const double* __restrict__ identity, int scoring,
                                                               uint64_t* __restrict__ key) {
  uint64_t i = (uint64_t)blockIdx.x * EW_THREADS + threadIdx.x;
  if (i >= n) return;
  const double NEG_INF = -__longlong_as_double(0x7ff0000000000000ll);
  const uint64_t len_u = (uint64_t)qe[i] - (uint64_t)qs[i];  // u64 wrapping subtraction as in the reference
  const double length = (double)len_u;
  const double id = identity[i];
  double score;
  switch (scoring) {
    case SWG_SCORE_IDENTITY: score = id <= 0.0 ? NEG_INF : id; break;
    case SWG_SCORE_LENGTH: score = length <= 0.0 ? NEG_INF : length; break;
    case SWG_SCORE_LENGTH_IDENTITY:
    case SWG_SCORE_MATCHES: score = (length <= 0.0 || id <= 0.0) ? NEG_INF : __dmul_rn(length, id); break;
    default: score = (length <= 0.0 || id <= 0.0) ? NEG_INF : __dmul_rn(id, swg_log_glibc(length)); break;
  }
  key[i] = sortable_desc(score);
}

// ---- events ---------------------------------------------------------------------------------
__global__ __launch_bounds__(EW_THREADS) void event_build_kernel(uint64_t n, const uint64_t* __restrict__ seg,
                                                                 const uint32_t* __restrict__ start,
                                                                 const uint32_t* __restrict__ end,
                                                                 const uint8_t* __restrict__ alive, int pos_bits,
                                                                 uint64_t* __restrict__ ev_x,
                                                                 uint32_t* __restrict__ ev_val) {
  uint64_t i = (uint64_t)blockIdx.x * EW_THREADS + threadIdx.x;
  if (i >= n) return;
  const bool live = alive ? alive[i] != 0 : true;
  uint64_t xb = 0, xe = 0;
  uint32_t vb = ((uint32_t)i << 1) | 1u, ve = ((uint32_t)i << 1) | 1u;  // dead: two inert End events at X = 0
  if (live) {
    const uint64_t hi = (seg[i] + 1) << pos_bits;
    xb = hi | start[i];
    xe = hi | end[i];
    vb = ((uint32_t)i << 1);
  }
  // interleaved so that a wave writes 2 x 64 consecutive elements
  ev_x[2 * i] = xb;
  ev_x[2 * i + 1] = xe;
  ev_val[2 * i] = vb;
  ev_val[2 * i + 1] = ve;
}

// After the sort: remember where each interval's Begin/End landed and pull the Begin's
// interval data next to the event so the tile kernel reads only coalesced streams.
__global__ __launch_bounds__(EW_THREADS) void event_gather_kernel(uint64_t n_ev, const uint64_t* __restrict__ ev_x,
                                                                  const uint32_t* __restrict__ ev_val,
                                                                  const uint32_t* __restrict__ end,
                                                                  const uint64_t* __restrict__ score_key,
                                                                  uint32_t* __restrict__ pos_begin,
                                                                  uint32_t* __restrict__ pos_end,
                                                                  uint32_t* __restrict__ ev_end,
                                                                  uint64_t* __restrict__ ev_key) {
  uint64_t p = (uint64_t)blockIdx.x * EW_THREADS + threadIdx.x;
  if (p >= n_ev) return;
  const uint64_t x = ev_x[p];
  const uint32_t v = ev_val[p];
  const uint32_t id = v >> 1;
  uint32_t e = 0;
  uint64_t k = 0;
  if (x != 0) {
    if (v & 1u) {
      pos_end[id] = (uint32_t)p;
    } else {
      pos_begin[id] = (uint32_t)p;
      e = end[id];
      k = score_key[id];
    }
  }
  ev_end[p] = e;
  ev_key[p] = k;
}

// Segments with exactly one live interval (two events) are returned whole by the reference
// (plane_sweep_exact.rs:274-276), zero-length or not.
__global__ __launch_bounds__(EW_THREADS) void single_segment_kernel(uint64_t n_ev, const uint64_t* __restrict__ ev_x,
                                                                    const uint32_t* __restrict__ ev_val, int pos_bits,
                                                                    uint8_t* __restrict__ single) {
  uint64_t p = (uint64_t)blockIdx.x * EW_THREADS + threadIdx.x;
  if (p >= n_ev) return;
  const uint64_t x = ev_x[p];
  if (x == 0) return;
  const uint64_t sg = x >> pos_bits;
  const bool first = p == 0 || (ev_x[p - 1] >> pos_bits) != sg;
  if (!first) return;
  // first event of its segment: the segment has exactly two events iff p+1 is in it and p+2 is not
  const bool second_in = p + 1 < n_ev && (ev_x[p + 1] >> pos_bits) == sg;
  const bool third_in = p + 2 < n_ev && (ev_x[p + 2] >> pos_bits) == sg;
  if (second_in && !third_in) single[ev_val[p] >> 1] = 1;
}

// ---- carry-in lists ---------------------------------------------------------------------------
__global__ __launch_bounds__(EW_THREADS) void carry_count_kernel(uint64_t n, const uint8_t* __restrict__ alive,
                                                                 const uint32_t* __restrict__ pos_begin,
                                                                 const uint32_t* __restrict__ pos_end,
                                                                 uint32_t* __restrict__ tile_count) {
  uint64_t i = (uint64_t)blockIdx.x * EW_THREADS + threadIdx.x;
  if (i >= n) return;
  if (alive && !alive[i]) return;
  const uint32_t tb = pos_begin[i] / TE, te = pos_end[i] / TE;
  for (uint32_t b = tb + 1; b <= te; ++b) atomicAdd(&tile_count[b], 1u);
}

__global__ __launch_bounds__(EW_THREADS) void carry_fill_kernel(
    uint64_t n, const uint8_t* __restrict__ alive, const uint32_t* __restrict__ pos_begin,
    const uint32_t* __restrict__ pos_end, const uint64_t* __restrict__ seg, const uint32_t* __restrict__ start,
    const uint32_t* __restrict__ end, const uint64_t* __restrict__ score_key, int pos_bits,
    const uint32_t* __restrict__ tile_off, uint32_t* __restrict__ tile_cursor, uint64_t* __restrict__ c_s,
    uint64_t* __restrict__ c_e, uint64_t* __restrict__ c_key, uint32_t* __restrict__ c_id) {
  uint64_t i = (uint64_t)blockIdx.x * EW_THREADS + threadIdx.x;
  if (i >= n) return;
  if (alive && !alive[i]) return;
  const uint32_t tb = pos_begin[i] / TE, te = pos_end[i] / TE;
  if (te <= tb) return;
  const uint64_t hi = (seg[i] + 1) << pos_bits;
  const uint64_t s = hi | start[i], e = hi | end[i], k = score_key[i];
  for (uint32_t b = tb + 1; b <= te; ++b) {
    const uint32_t slot = tile_off[b] + atomicAdd(&tile_cursor[b], 1u);
    c_s[slot] = s;
    c_e[slot] = e;
    c_key[slot] = k;
    c_id[slot] = (uint32_t)i;
  }
}

// ---- the tile kernel ----------------------------------------------------------------------------
struct TileArgs {
  uint64_t n_ev;
  const uint64_t* ev_x;
  const uint32_t* ev_val;
  const uint32_t* ev_end;
  const uint64_t* ev_key;
  int pos_bits;
  const uint32_t* tile_off;  // [ntiles + 1]
  const uint64_t* c_s;
  const uint64_t* c_e;
  const uint64_t* c_key;
  const uint32_t* c_id;
  uint64_t k;
  double thr;
  uint8_t* top;
  uint8_t* ovl;
};

__device__ __forceinline__ bool overlap_exceeds(uint64_t as, uint64_t ae, uint64_t bs, uint64_t be, double thr) {
  // query_overlap / target_overlap, plane_sweep_exact.rs:113-144 (composite coords share the segment part)
  const uint64_t os = as > bs ? as : bs;
  const uint64_t oe = ae < be ? ae : be;
  const double ol = oe > os ? (double)(oe - os) : 0.0;
  const uint64_t la = ae - as, lb = be - bs;
  const double ml = (double)(la < lb ? la : lb);
  if (!(ml > 0.0)) return false;
  return __ddiv_rn(ol, ml) > thr;
}

template <bool K1>
__global__ __launch_bounds__(TE) void sweep_tile_kernel(TileArgs a) {
  __shared__ uint64_t sx[TE];    // composite coordinate of event p
  __shared__ uint64_t se[TE];    // composite end of the interval that begins at p (0 if p is an End)
  __shared__ uint64_t skey[TE];  // its score key
  __shared__ uint64_t spm[TE];   // prefix maximum of se
  __shared__ uint32_t sid[TE];   // its interval index
  __shared__ uint64_t wmax[TE / 64];
  __shared__ uint64_t cs[CC], ce[CC], ckey[CC];
  __shared__ uint32_t cid[CC];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const uint64_t p0 = (uint64_t)blockIdx.x * TE;
  const uint64_t p = p0 + tid;
  const bool valid = p < a.n_ev;
  const uint64_t posmask = (uint64_t(1) << a.pos_bits) - 1;
  uint64_t X = ~0ull;
  uint64_t E = 0, KEY = 0;
  uint32_t ID = 0;
  if (valid) {
    X = a.ev_x[p];
    const uint32_t v = a.ev_val[p];
    ID = v >> 1;
    if (X != 0 && !(v & 1u)) {
      E = (X & ~posmask) | a.ev_end[p];
      KEY = a.ev_key[p];
    }
  }
  sx[tid] = X;
  se[tid] = E;
  skey[tid] = KEY;
  sid[tid] = ID;
  // prefix maximum of E over the tile
  uint64_t m = E;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    uint64_t t = __shfl_up(m, d, 64);
    if (lane >= d && t > m) m = t;
  }
  if (lane == 63) wmax[wave] = m;
  __syncthreads();
#pragma unroll
  for (int w = 0; w < TE / 64; ++w)
    if (w < wave && wmax[w] > m) m = wmax[w];
  spm[tid] = m;
  const uint64_t x_next = (p0 + TE < a.n_ev) ? a.ev_x[p0 + TE] : ~0ull;
  __syncthreads();

  // does this thread evaluate its coordinate?
  const bool eval = valid && X != 0 && (tid == TE - 1 || sx[tid + 1] != X) && X != x_next;
  const uint32_t c_begin = a.tile_off[blockIdx.x], c_end = a.tile_off[blockIdx.x + 1];

  // Enumerates every interval active at X: f(s, e, key, id).
  // Own-tile Begins: scan backwards, stop once no earlier interval can reach X.
  auto own_tile = [&](auto&& f) {
    for (int q = tid; q >= 0; --q) {
      if (spm[q] <= X) break;
      const uint64_t ee = se[q];
      if (ee > X) f(sx[q], ee, skey[q], sid[q]);
    }
  };

  if (K1) {
    // ---- pass 1: the best active interval --------------------------------------------------
    uint64_t bk = ~0ull, bs = ~0ull, be = 0;
    uint32_t bi = ~0u;
    bool have = false;
    auto take = [&](uint64_t s, uint64_t e, uint64_t key, uint32_t id) {
      if (!have || prio_less(key, s, id, bk, bs, bi)) {
        bk = key;
        bs = s;
        be = e;
        bi = id;
        have = true;
      }
    };
    if (eval) own_tile(take);
    for (uint32_t c0 = c_begin; c0 < c_end; c0 += CC) {
      __syncthreads();
      const uint32_t cnt = c_end - c0 < CC ? c_end - c0 : CC;
      if ((uint32_t)tid < cnt) {
        cs[tid] = a.c_s[c0 + tid];
        ce[tid] = a.c_e[c0 + tid];
        ckey[tid] = a.c_key[c0 + tid];
        cid[tid] = a.c_id[c0 + tid];
      }
      __syncthreads();
      if (eval)
        for (uint32_t c = 0; c < cnt; ++c)
          if (ce[c] > X) take(cs[c], ce[c], ckey[c], cid[c]);
    }
    if (eval && have) a.top[bi] = 1;
    // ---- pass 2: everything else that is active and overlaps the best too much ----------------
    if (a.thr < 1.0) {
      auto mark = [&](uint64_t s, uint64_t e, uint64_t key, uint32_t id) {
        (void)key;
        if (id != bi && overlap_exceeds(s, e, bs, be, a.thr)) a.ovl[id] = 1;
      };
      if (eval && have) own_tile(mark);
      for (uint32_t c0 = c_begin; c0 < c_end; c0 += CC) {
        __syncthreads();
        const uint32_t cnt = c_end - c0 < CC ? c_end - c0 : CC;
        if ((uint32_t)tid < cnt) {
          cs[tid] = a.c_s[c0 + tid];
          ce[tid] = a.c_e[c0 + tid];
          cid[tid] = a.c_id[c0 + tid];
        }
        __syncthreads();
        if (eval && have)
          for (uint32_t c = 0; c < cnt; ++c)
            if (ce[c] > X) mark(cs[c], ce[c], 0, cid[c]);
      }
    }
  } else {
    // ---- general k: walk the priority order by successive minima; no per-thread storage.
    // Carry-ins are read straight from global memory here (every lane reads the same entry, so
    // the loads coalesce to one request); this path is for 2 <= k < inf.
    if (!eval) return;
    auto all_active = [&](auto&& f) {
      own_tile(f);
      for (uint32_t c = c_begin; c < c_end; ++c) {
        const uint64_t ee = a.c_e[c];
        if (ee > X) f(a.c_s[c], ee, a.c_key[c], a.c_id[c]);
      }
    };
    // next_after(prev): smallest priority strictly greater than prev among the active set
    auto next_after = [&](bool have_prev, uint64_t pk, uint64_t ps, uint32_t pi, uint64_t* nk, uint64_t* ns,
                          uint64_t* ne, uint32_t* ni) -> bool {
      bool found = false;
      uint64_t fk = 0, fs = 0, fe = 0;
      uint32_t fi = 0;
      all_active([&](uint64_t s, uint64_t e, uint64_t key, uint32_t id) {
        if (have_prev && !prio_less(pk, ps, pi, key, s, id)) return;  // not after prev
        if (!found || prio_less(key, s, id, fk, fs, fi)) {
          fk = key;
          fs = s;
          fe = e;
          fi = id;
          found = true;
        }
      });
      *nk = fk;
      *ns = fs;
      *ne = fe;
      *ni = fi;
      return found;
    };
    // phase 1: the k-th best priority (tau); fewer than k actives -> everyone is in T(x)
    uint64_t tk = 0, ts = 0, te_ = 0;
    uint32_t ti = 0;
    bool have_prev = false, exhausted = false;
    for (uint64_t r = 0; r < a.k; ++r) {
      uint64_t nk, ns, ne;
      uint32_t ni;
      if (!next_after(have_prev, tk, ts, ti, &nk, &ns, &ne, &ni)) {
        exhausted = true;
        break;
      }
      tk = nk;
      ts = ns;
      te_ = ne;
      ti = ni;
      have_prev = true;
    }
    (void)te_;
    if (!have_prev) return;  // nothing active
    // phase 2: members of T(x) are `ever top`
    all_active([&](uint64_t s, uint64_t e, uint64_t key, uint32_t id) {
      (void)e;
      if (exhausted || !prio_less(tk, ts, ti, key, s, id)) a.top[id] = 1;
    });
    if (exhausted || !(a.thr < 1.0)) return;  // no non-members, or overlap pass disabled
    // phase 3: every non-member against every member
    uint64_t mk = 0, ms = 0, me = 0;
    uint32_t mi = 0;
    bool mprev = false;
    for (uint64_t r = 0; r < a.k; ++r) {
      uint64_t nk, ns, ne;
      uint32_t ni;
      if (!next_after(mprev, mk, ms, mi, &nk, &ns, &ne, &ni)) break;
      mk = nk;
      ms = ns;
      me = ne;
      mi = ni;
      mprev = true;
      all_active([&](uint64_t s, uint64_t e, uint64_t key, uint32_t id) {
        if (prio_less(tk, ts, ti, key, s, id) && overlap_exceeds(s, e, ms, me, a.thr)) a.ovl[id] = 1;
      });
    }
  }
}

// k == inf: every active interval is always in T(x) (plane_sweep_exact.rs:219-228 with
// usize::MAX), so keep = single-in-segment | (start < end): no sweep.  Only zero-length
// intervals need the segment size; when any exist the caller sorts the events once to get the
// `single` flags (single_segment_kernel), otherwise nothing else is needed.
__global__ __launch_bounds__(EW_THREADS) void kinf_mark_kernel(uint64_t n, const uint32_t* __restrict__ start,
                                                               const uint32_t* __restrict__ end,
                                                               const uint8_t* __restrict__ alive,
                                                               uint8_t* __restrict__ keep,
                                                               uint32_t* __restrict__ n_zero) {
  uint64_t i = (uint64_t)blockIdx.x * EW_THREADS + threadIdx.x;
  if (i >= n) return;
  const bool live = alive ? alive[i] != 0 : true;
  uint8_t kf = 0;
  if (live) {
    if (start[i] < end[i])
      kf = 1;
    else
      atomicAdd(n_zero, 1u);
  }
  keep[i] = kf;
}
__global__ __launch_bounds__(EW_THREADS) void kinf_single_kernel(uint64_t n, const uint8_t* __restrict__ single,
                                                                 uint8_t* __restrict__ keep) {
  uint64_t i = (uint64_t)blockIdx.x * EW_THREADS + threadIdx.x;
  if (i < n && single[i]) keep[i] = 1;
}

__global__ __launch_bounds__(EW_THREADS) void combine_kernel(uint64_t n, const uint8_t* __restrict__ alive,
                                                             const uint8_t* __restrict__ single,
                                                             const uint8_t* __restrict__ top,
                                                             const uint8_t* __restrict__ ovl,
                                                             uint8_t* __restrict__ keep) {
  uint64_t i = (uint64_t)blockIdx.x * EW_THREADS + threadIdx.x;
  if (i >= n) return;
  const bool live = alive ? alive[i] != 0 : true;
  keep[i] = (live && (single[i] || (top[i] && !ovl[i]))) ? 1 : 0;
}

inline unsigned blocks_for(uint64_t n, int threads) { return (unsigned)((n + threads - 1) / threads); }

}  // namespace

int swg_score_keys(swg_ctx* ctx, uint64_t n, const uint32_t* q_start, const uint32_t* q_end,
                   const double* identity, int scoring, uint64_t* key_out) {
  if (n == 0) return SWG_OK;
  SWG_LAUNCH(ctx, "score_key", score_key_kernel<<<blocks_for(n, EW_THREADS), EW_THREADS, 0, ctx->stream>>>(n, q_start, q_end, identity,
                                                                              scoring, key_out));
  SWG_KERNEL_CHECK(ctx);
  return SWG_OK;
}

int swg_sweep_axis(swg_ctx* ctx, const swg_axis_input& in, uint64_t k, double thr, uint8_t* keep) {
  const uint64_t n = in.n;
  if (n == 0) return SWG_OK;
  if (n >= (uint64_t(1) << 31)) return swg_set_error(ctx, SWG_ERR_RANGE, "sweep: n >= 2^31 intervals");
  hipStream_t st = ctx->stream;
  swg_arena_mark mark = swg_arena_save(ctx);

  const int key_bits = in.seg_bits + in.pos_bits;  // seg_bits must cover (max segment id + 1)
  if (key_bits > 64)
    return swg_set_error(ctx, SWG_ERR_RANGE, "sweep: segment id (%d bits) + coordinate (%d bits) exceed 64 bits",
                         in.seg_bits, in.pos_bits);
  const uint64_t n_ev = 2 * n;
  const uint32_t ntiles = (uint32_t)((n_ev + TE - 1) / TE);

  if (k == SWG_K_INF) {
    uint32_t* n_zero = swg_alloc<uint32_t>(ctx, 2);
    SWG_CHECK_ARENA(ctx);
    SWG_HIP(ctx, hipMemsetAsync(n_zero, 0, 8, st));
    SWG_LAUNCH(ctx, "kinf_mark", kinf_mark_kernel<<<blocks_for(n, EW_THREADS), EW_THREADS, 0, st>>>(n, in.start, in.end, in.alive, keep, n_zero));
    SWG_KERNEL_CHECK(ctx);
    uint64_t h = 0;
    SWG_TRY(swg_read_scalars(ctx, reinterpret_cast<uint64_t*>(n_zero), &h, 1));
    if ((uint32_t)h != 0) {  // zero-length intervals exist: need segment sizes -> sort the events once
      uint64_t* ev_x = swg_alloc<uint64_t>(ctx, n_ev);
      uint32_t* ev_val = swg_alloc<uint32_t>(ctx, n_ev);
      uint64_t* ev_x2 = swg_alloc<uint64_t>(ctx, n_ev);
      uint32_t* ev_val2 = swg_alloc<uint32_t>(ctx, n_ev);
      uint8_t* single = swg_alloc<uint8_t>(ctx, n);
      SWG_CHECK_ARENA(ctx);
      SWG_LAUNCH(ctx, "event_build", event_build_kernel<<<blocks_for(n, EW_THREADS), EW_THREADS, 0, st>>>(n, in.seg, in.start, in.end, in.alive,
                                                                            in.pos_bits, ev_x, ev_val));
      SWG_KERNEL_CHECK(ctx);
      SWG_TRY(swg_radix_sort_pairs(ctx, &ev_x, &ev_val, &ev_x2, &ev_val2, n_ev, 0, key_bits));
      SWG_HIP(ctx, hipMemsetAsync(single, 0, n, st));
      SWG_LAUNCH(ctx, "single_segment", single_segment_kernel<<<blocks_for(n_ev, EW_THREADS), EW_THREADS, 0, st>>>(n_ev, ev_x, ev_val, in.pos_bits,
                                                                                  single));
      SWG_KERNEL_CHECK(ctx);
      SWG_LAUNCH(ctx, "kinf_single", kinf_single_kernel<<<blocks_for(n, EW_THREADS), EW_THREADS, 0, st>>>(n, single, keep));
      SWG_KERNEL_CHECK(ctx);
    }
    swg_arena_restore(ctx, mark);
    return SWG_OK;
  }


  uint64_t* ev_x = swg_alloc<uint64_t>(ctx, n_ev);
  uint32_t* ev_val = swg_alloc<uint32_t>(ctx, n_ev);
  uint64_t* ev_x2 = swg_alloc<uint64_t>(ctx, n_ev);  // sort scratch, then ev_key
  uint32_t* ev_val2 = swg_alloc<uint32_t>(ctx, n_ev);  // sort scratch, then ev_end
  uint32_t* pos_begin = swg_alloc<uint32_t>(ctx, n);
  uint32_t* pos_end = swg_alloc<uint32_t>(ctx, n);
  uint8_t* flags = swg_alloc<uint8_t>(ctx, 3 * n);  // single | top | ovl
  uint32_t* tile_cnt = swg_alloc<uint32_t>(ctx, (size_t)ntiles + 1);
  uint32_t* tile_cur = swg_alloc<uint32_t>(ctx, (size_t)ntiles + 1);
  uint64_t* d_total = swg_alloc<uint64_t>(ctx, 1);
  SWG_CHECK_ARENA(ctx);
  uint8_t* single = flags;
  uint8_t* top = flags + n;
  uint8_t* ovl = flags + 2 * n;

  SWG_LAUNCH(ctx, "event_build", event_build_kernel<<<blocks_for(n, EW_THREADS), EW_THREADS, 0, st>>>(n, in.seg, in.start, in.end, in.alive,
                                                                        in.pos_bits, ev_x, ev_val));
  SWG_KERNEL_CHECK(ctx);
  SWG_TRY(swg_radix_sort_pairs(ctx, &ev_x, &ev_val, &ev_x2, &ev_val2, n_ev, 0, key_bits));
  uint64_t* ev_key = ev_x2;
  uint32_t* ev_end = ev_val2;
  SWG_HIP(ctx, hipMemsetAsync(flags, 0, 3 * n, st));
  SWG_HIP(ctx, hipMemsetAsync(tile_cnt, 0, sizeof(uint32_t) * ((size_t)ntiles + 1), st));
  SWG_HIP(ctx, hipMemsetAsync(tile_cur, 0, sizeof(uint32_t) * ((size_t)ntiles + 1), st));
  SWG_LAUNCH(ctx, "event_gather", event_gather_kernel<<<blocks_for(n_ev, EW_THREADS), EW_THREADS, 0, st>>>(n_ev, ev_x, ev_val, in.end,
                                                                            in.score_key, pos_begin, pos_end,
                                                                            ev_end, ev_key));
  SWG_KERNEL_CHECK(ctx);
  SWG_LAUNCH(ctx, "single_segment", single_segment_kernel<<<blocks_for(n_ev, EW_THREADS), EW_THREADS, 0, st>>>(n_ev, ev_x, ev_val, in.pos_bits,
                                                                              single));
  SWG_KERNEL_CHECK(ctx);
  SWG_LAUNCH(ctx, "carry_count", carry_count_kernel<<<blocks_for(n, EW_THREADS), EW_THREADS, 0, st>>>(n, in.alive, pos_begin, pos_end, tile_cnt));
  SWG_KERNEL_CHECK(ctx);
  SWG_TRY(swg_exclusive_scan_u32(ctx, tile_cnt, tile_cnt, (uint64_t)ntiles + 1, d_total));
  uint64_t n_carry = 0;
  SWG_TRY(swg_read_scalars(ctx, d_total, &n_carry, 1));
  uint64_t* c_s = swg_alloc<uint64_t>(ctx, n_carry + 1);
  uint64_t* c_e = swg_alloc<uint64_t>(ctx, n_carry + 1);
  uint64_t* c_key = swg_alloc<uint64_t>(ctx, n_carry + 1);
  uint32_t* c_id = swg_alloc<uint32_t>(ctx, n_carry + 1);
  SWG_CHECK_ARENA(ctx);
  if (n_carry) {
    SWG_LAUNCH(ctx, "carry_fill", carry_fill_kernel<<<blocks_for(n, EW_THREADS), EW_THREADS, 0, st>>>(
        n, in.alive, pos_begin, pos_end, in.seg, in.start, in.end, in.score_key, in.pos_bits, tile_cnt, tile_cur,
        c_s, c_e, c_key, c_id));
    SWG_KERNEL_CHECK(ctx);
  }
  TileArgs ta;
  ta.n_ev = n_ev;
  ta.ev_x = ev_x;
  ta.ev_val = ev_val;
  ta.ev_end = ev_end;
  ta.ev_key = ev_key;
  ta.pos_bits = in.pos_bits;
  ta.tile_off = tile_cnt;
  ta.c_s = c_s;
  ta.c_e = c_e;
  ta.c_key = c_key;
  ta.c_id = c_id;
  ta.k = k;
  ta.thr = thr;
  ta.top = top;
  ta.ovl = ovl;
  if (k == 1)
    SWG_LAUNCH(ctx, "sweep_tile_k1", sweep_tile_kernel<true><<<ntiles, TE, 0, st>>>(ta));
  else
    SWG_LAUNCH(ctx, "sweep_tile_kn", sweep_tile_kernel<false><<<ntiles, TE, 0, st>>>(ta));
  SWG_KERNEL_CHECK(ctx);
  SWG_LAUNCH(ctx, "combine", combine_kernel<<<blocks_for(n, EW_THREADS), EW_THREADS, 0, st>>>(n, in.alive, single, top, ovl, keep));
  SWG_KERNEL_CHECK(ctx);
  swg_arena_restore(ctx, mark);
  return SWG_OK;
}
