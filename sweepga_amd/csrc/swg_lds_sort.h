// The LDS bucket sort of one work-group, shared by pair_sort (swg_pair.hip), seg_sort and the segment-resident sweep
// (swg_segsort.hip).
//
// The scheme: a thread OWNS the records tid, tid + NT, ... of its problem's list; their 32-bit keys are dropped into buckets by a
// monotone map (count -> exclusive offsets -> scatter, unordered inside a bucket), then every slot learns its final place --
// the bucket's begin plus the bucket's elements that order before it by (key, index) -- and the owner reads back where its own
// records ended up, so that every other column reaches its sorted place through LDS by the thread that loaded it (coalesced
// loads, coalesced stores, no gather).  A problem larger than one LDS batch is cut into key ranges first: coarse bins over the
// key range, glued greedily into batches of at most `cap` elements (plan_batches).
//
// Everything here is a piece of that scheme with the work-group's LDS arrays passed in; the bodies keep their own loops over
// the owner's records (what they load differs).  All functions are work-group collectives unless noted: every thread calls them.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

namespace swg_lds {

// orders LDS accesses only: s_waitcnt lgkmcnt(0) + s_barrier (a __syncthreads also waits for the outstanding global loads and
// stores -- vmcnt(0) -- which the phases between these barriers want to keep in flight)
__device__ __forceinline__ void lds_barrier() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_s_waitcnt(0xc07f);
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// exclusive prefix sum of one value per thread in thread order; *total = the sum (ws: NT / 64 + 1 words of LDS)
template <int NT>
__device__ __forceinline__ uint32_t block_excl_sum(uint32_t v, uint32_t* ws, uint32_t* total) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  uint32_t inc = v;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const uint32_t t = __shfl_up(inc, d, 64);
    if (lane >= d) inc += t;
  }
  if (NT == 64) {
    *total = __shfl(inc, 63, 64);
    return inc - v;
  }
  lds_barrier();
  if (lane == 63) ws[w] = inc;
  lds_barrier();
  uint32_t off = 0, tot = 0;
#pragma unroll
  for (int k = 0; k < NT / 64; ++k) {
    const uint32_t x = ws[k];
    off += k < w ? x : 0u;
    tot += x;
  }
  *total = tot;
  return off + inc - v;
}

// exclusive running maximum of one u32 per thread in thread order (0 in front of thread 0)
template <int NT>
__device__ __forceinline__ uint32_t block_excl_max_u32(uint32_t v, uint32_t* ws) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  uint32_t inc = v;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const uint32_t t = __shfl_up(inc, d, 64);
    if (lane >= d) inc = t > inc ? t : inc;
  }
  uint32_t ex = __shfl_up(inc, 1, 64);
  if (lane == 0) ex = 0;
  if (NT == 64) return ex;
  lds_barrier();
  if (lane == 63) ws[w] = inc;
  lds_barrier();
  uint32_t off = 0;
#pragma unroll
  for (int k = 0; k < NT / 64; ++k) {
    const uint32_t x = ws[k];
    off = (k < w && x > off) ? x : off;
  }
  return ex > off ? ex : off;
}

// bins[0 .. NBIN) (counts) -> their exclusive prefix sums, in place
template <int NT, int NBIN>
__device__ __forceinline__ void bins_to_offsets(uint32_t* bins, uint32_t* ws) {
  constexpr int PERB = (NBIN + NT - 1) / NT;
  const int tid = threadIdx.x;
  uint32_t c[PERB], sum = 0, tot;
#pragma unroll
  for (int j = 0; j < PERB; ++j) {
    c[j] = tid * PERB + j < NBIN ? bins[tid * PERB + j] : 0u;
    sum += c[j];
  }
  uint32_t off = block_excl_sum<NT>(sum, ws, &tot);
  lds_barrier();
#pragma unroll
  for (int j = 0; j < PERB; ++j)
    if (tid * PERB + j < NBIN) {
      bins[tid * PERB + j] = off;
      off += c[j];
    }
}

// ONE thread: the batches over the bins' exclusive prefix sums (`total` elements in all) -- a batch takes as many bins as fit
// `cap` (a binary search per batch); b_lo[0 .. nb] = the batches' first bins.  Returns 0 when a single bin holds more than
// `cap` elements or more than maxb batches would be needed: not a case for the LDS sort.
template <int NBIN>
__device__ __forceinline__ uint32_t plan_batches(const uint32_t* bins, uint32_t total, uint32_t cap, uint32_t maxb, uint32_t* b_lo) {
  uint32_t nb = 0, lo = 0;
  b_lo[0] = 0;
  while (lo < (uint32_t)NBIN) {
    const uint32_t start = bins[lo];
    uint32_t l = lo + 1, r = NBIN;
    while (l < r) {
      const uint32_t mid = l + ((r - l + 1) >> 1);
      const uint32_t pm = mid < (uint32_t)NBIN ? bins[mid] : total;
      if (pm - start <= cap) l = mid; else r = mid - 1;
    }
    const uint32_t p1 = l < (uint32_t)NBIN ? bins[l] : total;
    if (p1 - start > cap || nb + 1 >= maxb) return 0u;
    b_lo[++nb] = l;
    lo = l;
  }
  return nb;
}

// cnt[0 .. NBK) (bucket counts) -> exclusive offsets in place; returns the number of elements (in a scalar register)
template <int NT, int NBK>
__device__ __forceinline__ uint32_t bucket_offsets(uint32_t* cnt, uint32_t* ws) {
  constexpr int PER = NBK / NT;
  static_assert(NBK % NT == 0 && PER >= 1, "bucket counters per thread");
  const int tid = threadIdx.x;
  uint32_t c[PER], sum = 0, mb;
#pragma unroll
  for (int j = 0; j < PER; ++j) {
    c[j] = cnt[tid * PER + j];
    sum += c[j];
  }
  uint32_t off = block_excl_sum<NT>(sum, ws, &mb);
  mb = (uint32_t)__builtin_amdgcn_readfirstlane((int)mb);
#pragma unroll
  for (int j = 0; j < PER; ++j) {
    cnt[tid * PER + j] = off;
    off += c[j];
  }
  return mb;
}

// The order inside the buckets.  Before: K[pos] / I[pos] = key and index of the element scattered to slot pos < mb, cnt[b] = END
// of bucket b (what the scatter's atomics leave).  After: K[r] = the keys in order, I[r] = their indices when KEEP_I, and
// RR[pos] = r for every slot -- the owner of the element that was scattered to pos finds its final place there.
// bucket_of(pos, key) -> the bucket slot pos was scattered into.  IT = uint16_t: index and rank share a register (index << 16 |
// rank, rank 0xffff = an empty slot); uint32_t: the index has one of its own.  The thread's ES slots advance together (one
// round trip of LDS reads per step, not one per slot and step), in two halves (registers).  Ends with its barrier passed.
template <int NT, int ES, typename IT, bool KEEP_I, class BucketOf>
__device__ __forceinline__ void rank_buckets(uint32_t* K, IT* I, uint16_t* RR, const uint32_t* cnt, uint32_t mb, uint32_t t_rk, BucketOf&& bucket_of) {
  constexpr bool WIDE = sizeof(IT) == 4;
  static_assert(ES % 2 == 0, "two halves");
  uint32_t rk[ES], rp[ES], ri[WIDE ? ES : 1];
#pragma unroll
  for (int e = 0; e < ES; ++e) {
    const uint32_t pos = t_rk + (uint32_t)e * NT;
    rk[e] = pos < mb ? K[pos] : 0u;
    const uint32_t ix = pos < mb ? (uint32_t)I[pos] : 0u;
    if constexpr (WIDE) {
      ri[e] = ix;
      rp[e] = 0xffffu;
    } else {
      rp[e] = (ix << 16) | 0xffffu;
    }
  }
  auto count_half = [&](auto off_c) {
    constexpr int OFF = decltype(off_c)::value, HS = ES / 2;
    uint32_t lo[HS], hi[HS], longest = 0;
#pragma unroll
    for (int e = 0; e < HS; ++e) {
      const uint32_t pos = t_rk + (uint32_t)(OFF + e) * NT;
      lo[e] = hi[e] = 0;
      if (pos < mb) {
        const uint32_t b = bucket_of(pos, rk[OFF + e]);
        hi[e] = cnt[b];
        lo[e] = b ? cnt[b - 1] : 0u;
        rp[OFF + e] = (rp[OFF + e] & 0xffff0000u) | lo[e];
        longest = hi[e] - lo[e] > longest ? hi[e] - lo[e] : longest;
      }
    }
    for (uint32_t it = 0; it < longest; ++it) {
#pragma unroll
      for (int e = 0; e < HS; ++e) {
        const uint32_t x = lo[e] + it;
        if (x < hi[e]) {
          const uint32_t kx = K[x];
          uint32_t before = kx < rk[OFF + e] ? 1u : 0u;
          if (kx == rk[OFF + e]) {  // (a tie on the key: only then is the other element's index read)
            uint32_t mine;
            if constexpr (WIDE) mine = ri[OFF + e]; else mine = rp[OFF + e] >> 16;
            before = (uint32_t)I[x] < mine ? 1u : 0u;
          }
          rp[OFF + e] += before;
        }
      }
    }
  };
  count_half(std::integral_constant<int, 0>{});
  count_half(std::integral_constant<int, ES / 2>{});
  lds_barrier();
#pragma unroll
  for (int e = 0; e < ES; ++e)
    if ((rp[e] & 0xffffu) != 0xffffu) {
      const uint32_t r = rp[e] & 0xffffu;
      K[r] = rk[e];
      if constexpr (KEEP_I) {
        if constexpr (WIDE) I[r] = ri[e]; else I[r] = (IT)(rp[e] >> 16);
      }
      RR[t_rk + (uint32_t)e * NT] = (uint16_t)r;
    }
  lds_barrier();
}

// the owner's packed slots (two 16-bit slots per word, record e in half e & 1 of word e / 2) -> the records' final places, read
// from RR; records outside `mask` get 0.  The words stay packed (the compiler would otherwise carry ER registers).
template <int ER, typename MASK>
__device__ __forceinline__ void slots_to_ranks(uint32_t (&slotw)[ER / 2], MASK mask, const uint16_t* RR) {
#pragma unroll
  for (int j = 0; j < ER / 2; ++j) {
    const uint32_t w = slotw[j];
    const uint32_t r0 = (mask >> (2 * j)) & 1u ? RR[w & 0xffffu] : 0u, r1 = (mask >> (2 * j + 1)) & 1u ? RR[w >> 16] : 0u;
    slotw[j] = r0 | (r1 << 16);
    asm volatile("" : "+v"(slotw[j]));
  }
}

}  // namespace swg_lds
