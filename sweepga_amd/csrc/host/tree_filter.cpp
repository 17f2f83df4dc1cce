// Tree sparsification of a PAF before the filter: `--sparsify tree:<near>[:<far>[:<random>]]` (or `knn:`), which the
// reference applies to the input file before PafFilter::filter_paf (src/main.rs:3640-3688, src/tree_filter.rs:205-285).
//
//   1. every line that is neither empty nor '#'-led and has >= 11 fields is an alignment: genome = first two '#' parts of
//      the name + '#' (the whole name without '#'), matches / block length = columns 10 / 11 (0 / 1 when unparsable);
//   2. per unordered pair of different genomes: identity = sum(matches) / sum(block length)   (src/tree_filter.rs:39-75);
//   3. every genome keeps its k_nearest best and k_farthest worst neighbours by that identity (:79-135), plus every pair
//      whose DefaultHasher (SipHash-1-3, zero keys) value over the two prefixes is <= random_fraction * 2^64 (:137-156);
//   4. the lines of the selected pairs survive, in input order, newline-normalised (:172-200, 277-282).
// Equal identities: the reference sorts a HashMap's iteration order (arbitrary between runs); here ties fall to the
// neighbour's prefix in ascending byte order, i.e. one of the orders the reference can produce.
// Host code only: the sums are exact integer additions in f64 (< 2^53), nothing here is worth a kernel.
#include <algorithm>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <string>
#include <string_view>
#include <unordered_map>
#include <unordered_set>
#include <vector>

#include "../../../include/sweepga_gpu.h"

namespace {

std::string genome_of(std::string_view name) {  // src/tree_filter.rs:15-24
  const size_t a = name.find('#');
  if (a == std::string_view::npos) return std::string(name);
  const size_t b = name.find('#', a + 1);
  std::string g(name.substr(0, b == std::string_view::npos ? name.size() : b));
  g.push_back('#');
  return g;
}

uint64_t number_or(std::string_view f, uint64_t fallback) {  // str::parse::<u64>().unwrap_or(fallback)
  size_t i = (!f.empty() && f[0] == '+') ? 1 : 0;
  if (i >= f.size()) return fallback;
  unsigned __int128 v = 0;
  for (; i < f.size(); ++i) {
    if (f[i] < '0' || f[i] > '9') return fallback;
    v = v * 10 + (unsigned)(f[i] - '0');
    if (v > UINT64_MAX) return fallback;
  }
  return (uint64_t)v;
}

// SipHash-1-3 with zero keys over one byte string (what DefaultHasher::new() computes for the concatenated writes)
uint64_t sip13(const std::string& bytes) {
  uint64_t v[4] = {0x736f6d6570736575ull, 0x646f72616e646f6dull, 0x6c7967656e657261ull, 0x7465646279746573ull};
  auto rot = [](uint64_t x, int k) { return (x << k) | (x >> (64 - k)); };
  auto mix = [&]() {
    v[0] += v[1]; v[2] += v[3];
    v[1] = rot(v[1], 13) ^ v[0]; v[3] = rot(v[3], 16) ^ v[2];
    v[0] = rot(v[0], 32);
    v[2] += v[1]; v[0] += v[3];
    v[1] = rot(v[1], 17) ^ v[2]; v[3] = rot(v[3], 21) ^ v[0];
    v[2] = rot(v[2], 32);
  };
  const size_t n = bytes.size(), full = n / 8 * 8;
  for (size_t i = 0; i < full; i += 8) {
    uint64_t m;
    std::memcpy(&m, bytes.data() + i, 8);  // little-endian host
    v[3] ^= m;
    mix();
    v[0] ^= m;
  }
  uint64_t last = (uint64_t)(n & 0xff) << 56;
  for (size_t i = full; i < n; ++i) last |= (uint64_t)(unsigned char)bytes[i] << (8 * (i - full));
  v[3] ^= last;
  mix();
  v[0] ^= last;
  v[2] ^= 0xff;
  mix();
  mix();
  mix();
  return v[0] ^ v[1] ^ v[2] ^ v[3];
}

}  // namespace

extern "C" int swg_paf_tree_filter(const char* text, uint64_t len, uint64_t k_nearest, uint64_t k_farthest, double random_fraction,
                                   char** out_text, uint64_t* out_len) {
  if (!out_text || !out_len || (len && !text)) return SWG_ERR_INVALID;
  *out_text = nullptr;
  *out_len = 0;
  struct Line {
    uint64_t off;
    uint32_t len;
    uint32_t gq, gt;
  };
  std::vector<Line> alns;
  std::vector<std::string> genomes;  // id -> prefix
  std::unordered_map<std::string, uint32_t> gid;
  std::unordered_map<std::string_view, uint32_t> name_gid;  // sequence name -> genome id (names repeat, prefixes are built once)
  auto genome_id = [&](std::string_view name) {
    auto it = name_gid.find(name);
    if (it != name_gid.end()) return it->second;
    std::string g = genome_of(name);
    auto jt = gid.find(g);
    uint32_t id;
    if (jt == gid.end()) {
      id = (uint32_t)genomes.size();
      gid.emplace(g, id);
      genomes.push_back(std::move(g));
    } else {
      id = jt->second;
    }
    name_gid.emplace(name, id);
    return id;
  };
  struct Sum {
    double matches = 0.0, block = 0.0;
  };
  std::unordered_map<uint64_t, Sum> sums;  // key = smaller-prefix genome id << 32 | the other one
  auto pair_key = [&](uint32_t a, uint32_t b) {
    if (genomes[b] < genomes[a]) std::swap(a, b);
    return ((uint64_t)a << 32) | b;
  };
  try {
    for (uint64_t pos = 0; pos < len;) {
      const char* nl = static_cast<const char*>(std::memchr(text + pos, '\n', len - pos));
      const uint64_t end = nl ? (uint64_t)(nl - text) : len;
      uint64_t ll = end - pos;
      if (nl && ll && text[end - 1] == '\r') --ll;  // BufRead::lines strips "\n" and "\r\n"
      const std::string_view line(text + pos, ll);
      const uint64_t here = pos;
      pos = end + 1;
      if (line.empty() || line[0] == '#') continue;
      std::string_view f[11];
      size_t s0 = 0;
      int nf = 0;
      while (nf < 11) {
        const size_t t = line.find('\t', s0);
        f[nf++] = line.substr(s0, t == std::string_view::npos ? std::string_view::npos : t - s0);
        if (t == std::string_view::npos) break;
        s0 = t + 1;
      }
      if (nf < 11) continue;
      if (ll > 0xffffffffull) return SWG_ERR_RANGE;
      Line a{here, (uint32_t)ll, genome_id(f[0]), genome_id(f[5])};
      alns.push_back(a);
      if (a.gq == a.gt) continue;
      Sum& s = sums[pair_key(a.gq, a.gt)];
      s.matches += (double)number_or(f[9], 0);
      s.block += (double)number_or(f[10], 1);
    }
    // neighbour lists
    const size_t G = genomes.size();
    struct Nb {
      uint32_t other;
      double identity;
    };
    std::vector<std::vector<Nb>> nbs(G);
    for (const auto& kv : sums) {
      const uint32_t a = (uint32_t)(kv.first >> 32), b = (uint32_t)kv.first;
      const double id = kv.second.block > 0.0 ? kv.second.matches / kv.second.block : 0.0;
      nbs[a].push_back({b, id});
      nbs[b].push_back({a, id});
    }
    std::unordered_set<uint64_t> selected;
    for (uint32_t g = 0; g < G; ++g) {
      std::vector<Nb>& v = nbs[g];
      if (v.empty()) continue;
      // identity descending; ties: neighbour prefix ascending (stable order of the reference's arbitrary one)
      std::sort(v.begin(), v.end(), [&](const Nb& x, const Nb& y) {
        if (x.identity != y.identity) return x.identity > y.identity;
        return genomes[x.other] < genomes[y.other];
      });
      for (size_t k = 0; k < v.size() && k < k_nearest; ++k) selected.insert(pair_key(g, v[k].other));
      for (size_t k = 0; k < v.size() && k < k_farthest; ++k) selected.insert(pair_key(g, v[v.size() - 1 - k].other));  // the reversed list
    }
    if (random_fraction > 0.0) {
      const double scaled = random_fraction * 18446744073709551616.0;  // u64::MAX as f64
      const uint64_t threshold = scaled >= 18446744073709551616.0 ? UINT64_MAX : (uint64_t)scaled;  // `as u64` saturates
      std::string buf;
      for (const auto& kv : sums) {
        buf.assign(genomes[kv.first >> 32]);
        buf.push_back((char)0xff);
        buf.append(genomes[(uint32_t)kv.first]);
        buf.push_back((char)0xff);
        if (sip13(buf) <= threshold) selected.insert(kv.first);
      }
    }
    uint64_t total = 0;
    for (const Line& a : alns)
      if (a.gq != a.gt && selected.count(pair_key(a.gq, a.gt))) total += (uint64_t)a.len + 1;
    char* out = static_cast<char*>(std::malloc(total ? total : 1));
    if (!out) return SWG_ERR_OOM;
    uint64_t o = 0;
    for (const Line& a : alns)
      if (a.gq != a.gt && selected.count(pair_key(a.gq, a.gt))) {
        std::memcpy(out + o, text + a.off, a.len);
        o += a.len;
        out[o++] = '\n';
      }
    *out_text = out;
    *out_len = o;
  } catch (const std::bad_alloc&) {
    return SWG_ERR_OOM;
  }
  return SWG_OK;
}

extern "C" void swg_free(void* p) { std::free(p); }
