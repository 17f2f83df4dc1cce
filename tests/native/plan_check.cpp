// The digit plans of the radix sorts (swg_radix_plan_pairs / swg_radix_plan_packed, sweepga_amd/csrc/swg_internal.h and
// swg_sort.hip) for every key width: the passes tile the key bits exactly, in ascending order, no digit wider than 9 bits,
// the packed plan's first digit is the key's low 8 bits (the packed word drops exactly those), at most 8 passes, and 9-bit
// digits appear only where they save a pass.  No GPU: the plan functions are host code of libsweepga_gpu.so.
//   hipcc -O1 -std=c++17 tests/native/plan_check.cpp -o plan_check -Lsweepga_amd -lsweepga_gpu -Wl,-rpath,$PWD/sweepga_amd
#include <cstdio>
#include <cstdlib>

#include "../../sweepga_amd/csrc/swg_internal.h"

static int check(const swg_radix_plan& pl, int begin, int end, int max_bits, const char* what) {
  int at = begin, bad = 0;
  if (pl.npasses < 1 || pl.npasses > SWG_RADIX_MAX_PASSES) bad = 1;
  for (int p = 0; p < pl.npasses && !bad; ++p) {
    if (pl.shift[p] != at || pl.bits[p] < 1 || pl.bits[p] > max_bits) bad = 1;
    at += pl.bits[p];
  }
  if (at != end) bad = 1;
  if (bad) printf("bad %s plan for bits [%d, %d): %d passes\n", what, begin, end, pl.npasses);
  return bad;
}

int main() {
  const bool bits8 = getenv("SWG_SORT_BITS8") != nullptr;
  int bad = 0, nine = 0;
  for (int kb = 9; kb <= 64; ++kb) {
    const swg_radix_plan pl = swg_radix_plan_packed(kb);
    bad += check(pl, 0, kb, bits8 ? 8 : 9, "packed");
    if (pl.bits[0] != 8) {
      printf("packed plan for %d bits: first digit %d bits\n", kb, pl.bits[0]);
      ++bad;
    }
    const int eight = 1 + (kb - 8 + 7) / 8;  // passes with 8-bit digits
    bool has9 = false;
    for (int p = 0; p < pl.npasses; ++p) has9 |= pl.bits[p] == 9;
    if (has9) ++nine;
    if (has9 && pl.npasses >= eight) {
      printf("packed plan for %d bits uses 9-bit digits without saving a pass (%d vs %d)\n", kb, pl.npasses, eight);
      ++bad;
    }
    if (!has9 && pl.npasses != eight) {
      printf("packed plan for %d bits: %d passes, expected %d\n", kb, pl.npasses, eight);
      ++bad;
    }
  }
  for (int b = 0; b < 64; b += 7)
    for (int e = b + 1; e <= 64; ++e) {
      const swg_radix_plan pl = swg_radix_plan_pairs(b, e);
      if ((e - b + 7) / 8 > SWG_RADIX_MAX_PASSES) continue;
      bad += check(pl, b, e, 8, "pairs");
    }
  // word sorts (swg_radix_sort_words): the passes tile the sorted bits; and how many low bits a caller may leave out
  // (swg_radix_drop_bits): never more than the level's cap or the coordinate's width, a larger first cap only when asked for
  for (int sb = 1; sb <= 56; ++sb) {
    const swg_radix_plan pl = swg_radix_plan_words(sb);
    if (pl.npasses > 0) bad += check(pl, 0, sb, bits8 ? 8 : 9, "words");
  }
  if (!getenv("SWG_SORT_DROP")) {
    const int caps10[4] = {10, 7, 0, 0}, caps16[4] = {16, 10, 7, 0};
    for (int kb = 20; kb <= 48; ++kb)
      for (int low = 8; low <= 32; low += 4)
        for (int level = 0; level < 4; ++level) {
          const int d10 = swg_radix_drop_bits(100000000ull, kb, low, 27, level), d16 = swg_radix_drop_bits(100000000ull, kb, low, 27, level, 16);
          if (d10 < 0 || d10 > caps10[level] || d10 > low || d16 < 0 || d16 > caps16[level] || d16 > low || kb - d16 < 1) {
            printf("drop bits for a %d-bit key (%d low bits), level %d: %d / %d\n", kb, low, level, d10, d16);
            ++bad;
          }
        }
    if (!bits8 && swg_radix_drop_bits(100000000ull, 43, 28, 27, 0, 16) != 16) {  // S-pan's sort A: 27 bits = three 9-bit passes
      printf("a 43-bit key with sparse low bits is not cut to 27\n");
      ++bad;
    }
  }
  printf("%d bad, %d widths with 9-bit digits\n", bad, nine);
  return bad ? 1 : 0;
}
