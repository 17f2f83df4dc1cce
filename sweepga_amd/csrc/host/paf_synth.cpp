// paf-synth: deterministic S-pan-shaped PAF text for end-to-end timing (100 single-chromosome
// genomes g000#1#chr1 ... , ordered non-self pairs, 70 % syntenic / 30 % repeat mappings, lognormal lengths,
// 10 % '-' strand).  Generator only; it has no reference counterpart.
//   paf-synth <n_lines> [n_genomes=100] [seed=2025] [chr_len=150000000] [order=random|query] > out.paf
// order=query: the lines of a query genome form one run (query genomes ascending), the way an aligner writes its output
// query by query; order=random (default): every line draws its query genome at random.
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

static uint64_t s_state;
static inline uint64_t splitmix() {
  uint64_t z = (s_state += 0x9e3779b97f4a7c15ull);
  z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
  z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
  return z ^ (z >> 31);
}
static inline double uni() { return (double)(splitmix() >> 11) * (1.0 / 9007199254740992.0); }
static inline double gauss() {
  const double u = uni() + 1e-300, v = uni();
  return std::sqrt(-2.0 * std::log(u)) * std::cos(6.283185307179586 * v);
}
static inline char* put_u(char* o, uint64_t v) {
  char t[24];
  int k = 0;
  do t[k++] = (char)('0' + v % 10); while (v /= 10);
  while (k) *o++ = t[--k];
  return o;
}

int main(int argc, char** argv) {
  if (argc < 2) {
    std::fprintf(stderr, "usage: paf-synth <n_lines> [n_genomes=100] [seed=2025] [chr_len=150000000]\n");
    return 2;
  }
  const uint64_t n = std::strtoull(argv[1], nullptr, 10);
  const uint64_t g = argc > 2 ? std::strtoull(argv[2], nullptr, 10) : 100;
  s_state = argc > 3 ? std::strtoull(argv[3], nullptr, 10) : 2025;
  const uint64_t L = argc > 4 ? std::strtoull(argv[4], nullptr, 10) : 150000000ull;
  const bool by_query = argc > 5 && std::string(argv[5]) == "query";
  if (g < 2) return 2;
  std::vector<std::string> names(g);
  for (uint64_t i = 0; i < g; ++i) {
    char b[32];
    std::snprintf(b, sizeof b, "g%03llu#1#chr1", (unsigned long long)i);
    names[i] = b;
  }
  std::vector<char> buf(1 << 22);
  char* o = buf.data();
  const std::string len_s = std::to_string(L);
  for (uint64_t i = 0; i < n; ++i) {
    uint64_t a = splitmix() % g;
    if (by_query) a = (uint64_t)((__uint128_t)i * g / n);  // (the draw above is kept: the other columns do not depend on the order)
    uint64_t b = splitmix() % (g - 1);
    if (b >= a) ++b;
    double len = std::exp(std::log(2000.0) + 1.2 * gauss());
    if (len < 100) len = 100;
    if (len > 500000) len = 500000;
    const uint64_t ql = (uint64_t)len;
    const uint64_t qs = (uint64_t)(uni() * (double)(L - ql));
    const bool syn = uni() < 0.7;
    double tsd = syn ? (double)qs + 50000.0 * gauss() : uni() * (double)(L - ql);
    if (tsd < 0) tsd = 0;
    if (tsd > (double)(L - ql - 64)) tsd = (double)(L - ql - 64);
    const uint64_t ts = (uint64_t)tsd;
    const uint64_t tl = ql + splitmix() % 41 - 20;  // ql >= 100
    const uint64_t block = ql > tl ? ql : tl;
    const double u1 = uni(), u2 = uni();
    const double id = 0.70 + 0.30 * std::pow(u1, 1.0 / 5.0) * (1.0 - 0.3 * u2 * u2);  // skewed towards 1
    const uint64_t matches = (uint64_t)(id * (double)block);
    const bool minus = uni() < 0.1;
    for (char c : names[a]) *o++ = c;
    *o++ = '\t';
    for (char c : len_s) *o++ = c;
    *o++ = '\t';
    o = put_u(o, qs);
    *o++ = '\t';
    o = put_u(o, qs + ql);
    *o++ = '\t';
    *o++ = minus ? '-' : '+';
    *o++ = '\t';
    for (char c : names[b]) *o++ = c;
    *o++ = '\t';
    for (char c : len_s) *o++ = c;
    *o++ = '\t';
    o = put_u(o, ts);
    *o++ = '\t';
    o = put_u(o, ts + tl);
    *o++ = '\t';
    o = put_u(o, matches);
    *o++ = '\t';
    o = put_u(o, block);
    *o++ = '\t';
    *o++ = '6';
    *o++ = '0';
    if (i % 3 == 0) {  // a third of the lines carry an extended CIGAR (overrides column 10), as wfmash output does
      for (const char* t = "\tcg:Z:"; *t; ++t) *o++ = *t;
      o = put_u(o, matches);
      *o++ = '=';
      o = put_u(o, block - matches);
      *o++ = 'X';
    }
    *o++ = '\n';
    if ((size_t)(o - buf.data()) > buf.size() - 512) {
      std::fwrite(buf.data(), 1, (size_t)(o - buf.data()), stdout);
      o = buf.data();
    }
  }
  std::fwrite(buf.data(), 1, (size_t)(o - buf.data()), stdout);
  return 0;
}
