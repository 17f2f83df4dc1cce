// Host-side check of sweepga_amd/csrc/swg_log.h against the platform libm log().
// usage: log_check <first> <count> [stride]   -> prints number of mismatching bit patterns.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cstdint>
#include "../../sweepga_amd/csrc/swg_log.h"

int main(int argc, char** argv) {
  uint64_t first = argc > 1 ? strtoull(argv[1], 0, 10) : 1;
  uint64_t count = argc > 2 ? strtoull(argv[2], 0, 10) : 1000000;
  uint64_t stride = argc > 3 ? strtoull(argv[3], 0, 10) : 1;
  uint64_t bad = 0;
#pragma omp parallel for reduction(+ : bad) schedule(static)
  for (uint64_t n = 0; n < count; ++n) {
    double x = (double)(first + n * stride);
    double a = swg_log_glibc(x), b = std::log(x);
    if (std::memcmp(&a, &b, 8) != 0) ++bad;
  }
  printf("%llu\n", (unsigned long long)bad);
  return bad ? 1 : 0;
}
