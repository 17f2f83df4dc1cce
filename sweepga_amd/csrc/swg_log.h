// Bit-exact restatement of glibc's log() for the arguments the filter feeds it.
//
// The reference computes scores with Rust's f64::ln (plane_sweep_exact.rs:74) and chain
// identities with ln(gap) (paf_filter.rs:903); on Linux that is glibc's `log`.  Scores order
// the plane sweep, so a last-bit difference can flip a tie: the device therefore reproduces
// glibc's algorithm (sysdeps/ieee754/dbl-64/e_log.c, the FMA build `__log_fma` that the ifunc
// selects on every FMA-capable x86-64) operation for operation:
//
//   tmp = ix - OFF; i = (tmp >> 45) & 127; k = (int64)tmp >> 52; z = ix - (tmp & 0xfff<<52)
//   r  = fma(z, invc[i], -1)            w  = fma(k, Ln2hi, logc[i])
//   hi = w + r                          lo = fma(k, Ln2lo, (w - hi) + r)
//   r2 = r*r
//   y  = fma(r*r2, fma(fma(r,A4,A3), r2, fma(r,A2,A1)), fma(r2, A0, lo)) + hi
//
// The fused/unfused split above is the one in the shipped `__log_fma` object code (checked by
// disassembly), so every operation here is written with explicit fma/mul/add primitives and
// the function is compiled with contraction off.  Arguments are integer lengths >= 1
// (as f64), i.e. normal positive doubles; x == 1 returns +0 like glibc; other inputs near 1
// (|x-1| < 1/16, non-integers) take a different glibc branch and are outside this domain.
//
// tests/test_log_exact.py checks this header against the host libm on the CPU, and the
// device build against the host libm on the GPU.
#pragma once
#include <stdint.h>

#include "glibc_log_table.h"

#if defined(__HIPCC__)
#define SWG_HD __host__ __device__
#else
#define SWG_HD
#endif

#if defined(__HIP_DEVICE_COMPILE__)
__device__ __constant__ static const double swg_log_tab[256] = SWG_LOG_TABLE_INIT;
#define SWG_FMA(a, b, c) __fma_rn((a), (b), (c))
#define SWG_MUL(a, b) __dmul_rn((a), (b))
#define SWG_ADD(a, b) __dadd_rn((a), (b))
#define SWG_SUB(a, b) __dsub_rn((a), (b))
#else
static const double swg_log_tab[256] = SWG_LOG_TABLE_INIT;
#define SWG_FMA(a, b, c) __builtin_fma((a), (b), (c))
static inline double swg_mul_(double a, double b) { volatile double r = a * b; return r; }
static inline double swg_add_(double a, double b) { volatile double r = a + b; return r; }
static inline double swg_sub_(double a, double b) { volatile double r = a - b; return r; }
#define SWG_MUL(a, b) swg_mul_((a), (b))
#define SWG_ADD(a, b) swg_add_((a), (b))
#define SWG_SUB(a, b) swg_sub_((a), (b))
#endif

SWG_HD static inline double swg_log_glibc(double x) {
  union { double d; uint64_t u; } cv;
  cv.d = x;
  const uint64_t ix = cv.u;
  if (ix == 0x3ff0000000000000ULL) return 0.0;  // x == 1
  const uint64_t OFF = 0x3fe6000000000000ULL;
  const uint64_t tmp = ix - OFF;
  const int i = (int)((tmp >> 45) & 127);
  const int64_t k = (int64_t)tmp >> 52;
  cv.u = ix - (tmp & (0xfffULL << 52));
  const double z = cv.d;
  const double invc = swg_log_tab[2 * i], logc = swg_log_tab[2 * i + 1];
  const double kd = (double)k;
  const double r = SWG_FMA(z, invc, -1.0);
  const double w = SWG_FMA(kd, SWG_LOG_LN2HI, logc);
  const double hi = SWG_ADD(w, r);
  const double lo = SWG_FMA(kd, SWG_LOG_LN2LO, SWG_ADD(SWG_SUB(w, hi), r));
  const double r2 = SWG_MUL(r, r);
  const double r3 = SWG_MUL(r, r2);
  const double p1 = SWG_FMA(r, SWG_LOG_A2, SWG_LOG_A1);
  const double p2 = SWG_FMA(r, SWG_LOG_A4, SWG_LOG_A3);
  const double q = SWG_FMA(p2, r2, p1);
  const double t1 = SWG_FMA(r2, SWG_LOG_A0, lo);
  const double t2 = SWG_FMA(r3, q, t1);
  return SWG_ADD(t2, hi);
}
