// PAF ingest / egress for the filter path (host side, no HIP calls in this file).
//
//   open_paf_input ............ src/paf.rs:10-30        (.gz / .bgz by extension -> BGZF reader; else plain file)
//   extract_metadata .......... src/paf_filter.rs:292-376 (rank = physical line index; < 11 fields skipped but counted;
//                                identity = matches / max(block_len,1); dv:f: and cg:Z: overrides, last writer wins)
//   parse_cigar_counts ........ src/paf.rs:32-64        (only the '=' total matters to the filter)
//   SequenceIndex ............. src/sequence_index.rs:7-31 (ids in first-appearance order, query before target)
//   genome prefixes ........... src/paf_filter.rs:1022-1030, src/plane_sweep_scaffold.rs:13-22
//   write_filtered_output ..... src/paf_filter.rs:1689-1726 (input order, original bytes + ch:Z / st:Z tags)
//   filter_paf ................ src/paf_filter.rs:278-289
//
// The reference makes two sequential passes over the text with one String allocation per line and per field
// vector.  Here the file is mapped once, split at line boundaries into one slice per host thread, counted,
// parsed straight into the SoA columns swg_filter() takes, and the writer reuses the (offset,length) of every
// record instead of re-reading the input.  BGZF blocks are inflated in parallel.
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>

#include <algorithm>
#include <atomic>
#include <cerrno>
#include <chrono>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <string_view>
#include <thread>
#include <unordered_map>
#include <vector>

#include "../../../include/sweepga_gpu.h"
#include "host_internal.h"
#include "threads.h"
#include "rebase.h"

namespace {

thread_local std::string g_paf_error;

int paf_error(int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  std::vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  g_paf_error = buf;
  return code;
}

int pick_threads(int threads) {
  if (threads > 0) return threads > 256 ? 256 : threads;
  const unsigned hc = std::thread::hardware_concurrency();
  return hc ? (int)(hc > 64 ? 64 : hc) : 1;
}

template <class F>
void parallel_for(int threads, F&& body) {  // body(thread_index); nothing escapes a worker (host/threads.h)
  swg_host::run(threads, body);
}

// ---- Rust-compatible scalar parsers (str::parse::<u64>, str::parse::<f64>) ---------------------------
inline bool parse_u64(const char* s, size_t n, uint64_t* out) {
  size_t i = 0;
  if (n == 0) return false;
  if (s[0] == '+') i = 1;
  if (i >= n) return false;
  uint64_t v = 0;
  for (; i < n; ++i) {
    const unsigned d = (unsigned)(s[i] - '0');
    if (d > 9) return false;
    if (v > (UINT64_MAX - d) / 10) return false;
    v = v * 10 + d;
  }
  *out = v;
  return true;
}
inline bool parse_f64(const char* s, size_t n, double* out) {  // no whitespace, no hex floats, no "nan(...)"
  if (n == 0 || n > 400) return false;
  char buf[401];
  for (size_t i = 0; i < n; ++i) {
    const char c = s[i];
    if (c == 'x' || c == 'X' || c == ' ' || c == '\t' || c == '\n' || c == '(' || c == '\0') return false;
    buf[i] = c;
  }
  buf[n] = '\0';
  char* e = nullptr;
  const double v = std::strtod(buf, &e);
  if (e == buf || *e != '\0') return false;
  *out = v;
  return true;
}
// src/paf.rs:32-64: total of '=' run lengths; false when a run length does not parse
inline bool cigar_eq_total(const char* s, size_t n, uint64_t* matches) {
  uint64_t m = 0, cur = 0;
  bool have = false, overflow = false;
  for (size_t i = 0; i < n; ++i) {
    const char ch = s[i];
    const unsigned d = (unsigned)(ch - '0');
    if (d <= 9) {
      if (cur > (UINT64_MAX - d) / 10) overflow = true;
      cur = cur * 10 + d;
      have = true;
    } else {
      if (!have || overflow) return false;
      if (ch == '=') m += cur;
      cur = 0;
      have = false;
      overflow = false;
    }
  }
  *matches = m;
  return true;
}

// ---- input bytes ------------------------------------------------------------------------------------------
struct Text {
  const char* data = nullptr;
  size_t len = 0;
  void* map = nullptr;  // munmap(map, map_len) when set
  size_t map_len = 0;
  dev_t src_dev = 0;  // identity of the mapped input file (only meaningful while `map` is set)
  ino_t src_ino = 0;
  std::vector<char> owned;
  ~Text() {
    if (map) munmap(map, map_len);
  }
};

bool has_gz_ext(const char* path) {  // src/paf.rs:15-19
  const char* dot = std::strrchr(path, '.');
  const char* slash = std::strrchr(path, '/');
  if (!dot || (slash && dot < slash)) return false;
  return !std::strcmp(dot + 1, "gz") || !std::strcmp(dot + 1, "bgz");
}

struct BgzfBlock {
  size_t cdata, clen, out, isize;
  uint32_t crc;
};

// BGZF: every member carries BSIZE in a 'B','C' extra subfield -> blocks can be located without inflating.
bool bgzf_index(const unsigned char* p, size_t n, std::vector<BgzfBlock>* blocks, size_t* total) {
  size_t pos = 0, out = 0;
  while (pos < n) {
    if (n - pos < 18 || p[pos] != 0x1f || p[pos + 1] != 0x8b || p[pos + 2] != 8 || !(p[pos + 3] & 4)) return false;
    const size_t xlen = p[pos + 10] | (size_t)p[pos + 11] << 8;
    if (n - pos < 12 + xlen + 8) return false;
    size_t bsize = 0;
    for (size_t x = pos + 12; x + 4 <= pos + 12 + xlen;) {
      const size_t slen = p[x + 2] | (size_t)p[x + 3] << 8;
      if (p[x] == 'B' && p[x + 1] == 'C' && slen == 2 && x + 6 <= pos + 12 + xlen) bsize = (p[x + 4] | (size_t)p[x + 5] << 8) + 1;
      x += 4 + slen;
    }
    if (!bsize || bsize < 12 + xlen + 8 || n - pos < bsize) return false;
    const unsigned char* tail = p + pos + bsize - 4;
    const size_t isize = tail[0] | (size_t)tail[1] << 8 | (size_t)tail[2] << 16 | (size_t)tail[3] << 24;
    const uint32_t crc = tail[-4] | (uint32_t)tail[-3] << 8 | (uint32_t)tail[-2] << 16 | (uint32_t)tail[-1] << 24;
    blocks->push_back({pos + 12 + xlen, bsize - 12 - xlen - 8, out, isize, crc});
    out += isize;
    pos += bsize;
  }
  *total = out;
  return true;
}

int gunzip_all(const unsigned char* p, size_t n, int threads, std::vector<char>* out) {
  std::vector<BgzfBlock> blocks;
  size_t total = 0;
  if (bgzf_index(p, n, &blocks, &total)) {
    out->resize(total);
    std::atomic<size_t> next{0};
    std::atomic<int> bad{0};
    parallel_for(threads, [&](int) {
      z_stream z{};
      if (inflateInit2(&z, -15) != Z_OK) {
        bad = 1;
        return;
      }
      for (;;) {
        const size_t b = next.fetch_add(1);
        if (b >= blocks.size()) break;
        const BgzfBlock& k = blocks[b];
        unsigned char none;
        unsigned char* dst = k.isize ? reinterpret_cast<unsigned char*>(out->data() + k.out) : &none;
        inflateReset(&z);
        z.next_in = const_cast<unsigned char*>(p + k.cdata);
        z.avail_in = (uInt)k.clen;
        z.next_out = dst;
        z.avail_out = (uInt)(k.isize ? k.isize : 1);
        if (inflate(&z, Z_FINISH) != Z_STREAM_END || z.total_out != k.isize ||
            (uint32_t)crc32(crc32(0L, Z_NULL, 0), dst, (uInt)k.isize) != k.crc)
          bad = 1;
      }
      inflateEnd(&z);
    });
    if (bad) return paf_error(SWG_ERR_INVALID, "corrupt BGZF block");
    return SWG_OK;
  }
  // plain (possibly multi-member) gzip: sequential
  z_stream z{};
  if (inflateInit2(&z, 15 + 16) != Z_OK) return paf_error(SWG_ERR_OOM, "inflateInit2 failed");
  out->clear();
  std::vector<unsigned char> buf(1 << 20);
  size_t pos = 0;
  bool complete = true;  // the input must end exactly at a member end
  while (pos < n) {
    z.next_in = const_cast<unsigned char*>(p + pos);
    z.avail_in = (uInt)((n - pos) > (1u << 30) ? (1u << 30) : (n - pos));
    const size_t fed = z.avail_in;
    int rc;
    do {
      z.next_out = buf.data();
      z.avail_out = (uInt)buf.size();
      rc = inflate(&z, Z_NO_FLUSH);
      if (rc != Z_OK && rc != Z_STREAM_END && rc != Z_BUF_ERROR) {
        inflateEnd(&z);
        return paf_error(SWG_ERR_INVALID, "gzip stream is corrupt (zlib error %d)", rc);
      }
      out->insert(out->end(), buf.data(), buf.data() + (buf.size() - z.avail_out));
    } while (rc == Z_OK && (z.avail_in > 0 || z.avail_out == 0));
    pos += fed - z.avail_in;
    complete = rc == Z_STREAM_END;
    if (rc == Z_STREAM_END) {
      if (pos < n) inflateReset(&z);
    } else if (fed - z.avail_in == 0) {
      break;  // no progress: truncated
    }
  }
  inflateEnd(&z);
  if (!complete) return paf_error(SWG_ERR_INVALID, "gzip stream is truncated");
  return SWG_OK;
}

int load_text(const char* path, int threads, Text* t) {
  const bool is_stdin = !std::strcmp(path, "-");
  int fd = is_stdin ? 0 : open(path, O_RDONLY);
  if (fd < 0) return paf_error(SWG_ERR_INVALID, "cannot open %s: %s", path, std::strerror(errno));
  struct stat st {};
  const bool regular = !is_stdin && fstat(fd, &st) == 0 && S_ISREG(st.st_mode);
  const unsigned char* raw = nullptr;
  size_t raw_len = 0;
  std::vector<char> slurp;
  void* map = nullptr;
  if (regular && st.st_size > 0) {
    map = mmap(nullptr, (size_t)st.st_size, PROT_READ, MAP_PRIVATE, fd, 0);
    if (map == MAP_FAILED) map = nullptr;
  }
  if (map) {
    madvise(map, (size_t)st.st_size, MADV_WILLNEED);  // advice values are not flags: one call per hint
    madvise(map, (size_t)st.st_size, MADV_SEQUENTIAL);
    raw = static_cast<const unsigned char*>(map);
    raw_len = (size_t)st.st_size;
  } else {
    char buf[1 << 16];
    ssize_t k;
    while ((k = read(fd, buf, sizeof buf)) > 0) slurp.insert(slurp.end(), buf, buf + k);
    if (k < 0) {
      if (!is_stdin) close(fd);
      return paf_error(SWG_ERR_INVALID, "read error on %s: %s", path, std::strerror(errno));
    }
    raw = reinterpret_cast<const unsigned char*>(slurp.data());
    raw_len = slurp.size();
  }
  if (!is_stdin) close(fd);
  const bool gz = has_gz_ext(path) || (raw_len >= 2 && raw[0] == 0x1f && raw[1] == 0x8b);
  if (gz && raw_len) {
    const int rc = gunzip_all(raw, raw_len, threads, &t->owned);
    if (map) munmap(map, raw_len);
    if (rc != SWG_OK) return rc;
    t->data = t->owned.data();
    t->len = t->owned.size();
    return SWG_OK;
  }
  if (map) {
    t->map = map;
    t->map_len = raw_len;
    t->src_dev = st.st_dev;
    t->src_ino = st.st_ino;
    t->data = static_cast<const char*>(map);
    t->len = raw_len;
  } else {
    t->owned.swap(slurp);
    t->data = t->owned.data();
    t->len = t->owned.size();
  }
  return SWG_OK;
}

// ---- name interning -----------------------------------------------------------------------------------
struct Interner {  // string_view keys point into the (immutable) input text
  std::unordered_map<std::string_view, uint32_t> ids;
  std::vector<std::string_view> names;
  std::string_view last[2];
  uint32_t last_id[2] = {0, 0};
  uint32_t get(std::string_view nm, int slot) {  // slot 0 = query column, 1 = target column (one-entry caches)
    if (!last[slot].empty() && last[slot] == nm) return last_id[slot];
    uint32_t id;
    auto it = ids.find(nm);
    if (it != ids.end()) {
      id = it->second;
    } else {
      id = (uint32_t)names.size();
      names.push_back(nm);
      ids.emplace(nm, id);
    }
    last[slot] = nm;
    last_id[slot] = id;
    return id;
  }
};

std::string prefix_last(std::string_view n) {  // src/paf_filter.rs:1022-1030
  const size_t p = n.rfind('#');
  return std::string(p == std::string_view::npos ? n : n.substr(0, p + 1));
}
std::string prefix_two(std::string_view n) {  // src/plane_sweep_scaffold.rs:13-22
  const size_t p1 = n.find('#');
  if (p1 == std::string_view::npos) return std::string(n);
  const size_t p2 = n.find('#', p1 + 1);
  std::string r(n.substr(0, p1));
  r += '#';
  r += p2 == std::string_view::npos ? n.substr(p1 + 1) : n.substr(p1 + 1, p2 - p1 - 1);
  r += '#';
  return r;
}
uint32_t genome_table(const std::vector<std::string>& names, std::string (*fn)(std::string_view), std::vector<uint32_t>* out) {
  std::unordered_map<std::string, uint32_t> g;
  out->assign(names.empty() ? 1 : names.size(), 0);
  for (size_t i = 0; i < names.size(); ++i) (*out)[i] = g.emplace(fn(names[i]), (uint32_t)g.size()).first->second;
  return g.empty() ? 1u : (uint32_t)g.size();
}

// Column storage without value-initialisation: the first touch of every page happens in the parsing threads
// (a std::vector would zero-fill ~70 B per record on the calling thread before the parallel part starts).
template <class T>
struct Buf {
  T* p = nullptr;
  size_t n = 0;
  Buf() = default;
  Buf(const Buf&) = delete;
  Buf& operator=(const Buf&) = delete;
  ~Buf() { std::free(p); }
  void alloc(size_t k) {
    std::free(p);
    p = nullptr;
    const size_t bytes = (k ? k : 1) * sizeof(T);
    // Big columns live on 2 MB-aligned storage marked MADV_HUGEPAGE: where transparent huge pages are on "madvise" (the GPU
    // box) the parsing threads take one first-touch fault per 2 MB instead of one per 4 KB -- 10^8 lines, 32 threads: pass 2
    // 841 -> 733 ms, and the parse keeps scaling to 64 threads (618 ms) instead of getting slower.  SWG_PAF_THP=0: plain malloc.
    static const bool thp = !(std::getenv("SWG_PAF_THP") && std::getenv("SWG_PAF_THP")[0] == '0');
    if (thp && bytes >= (size_t(8) << 20)) {
      void* q = nullptr;
      if (posix_memalign(&q, size_t(2) << 20, (bytes + (size_t(2) << 20) - 1) & ~((size_t(2) << 20) - 1)) == 0) {
        madvise(q, bytes, MADV_HUGEPAGE);
        p = static_cast<T*>(q);
      }
    }
    if (!p) p = static_cast<T*>(std::malloc(bytes));
    if (!p) throw std::bad_alloc();
    n = k;
  }
  T* data() { return p; }
  const T* data() const { return p; }
  T& operator[](size_t i) { return p[i]; }
  const T& operator[](size_t i) const { return p[i]; }
};

struct Slice {
  size_t begin = 0, end = 0;     // byte range, whole lines
  uint64_t lines = 0, recs = 0;  // counted in pass 1
  uint64_t line_base = 0, rec_base = 0;
  Interner names;
  std::vector<uint32_t> remap;   // local id -> global id
  int err = SWG_OK;
  uint64_t err_line = 0;
  const char* err_what = nullptr;
  bool wide = false;             // met a coordinate / matches / block length >= 2^32
  bool custom_identity = false;  // a record whose identity is not matches / max(block length, 1) (a dv:f: tag had the last word)
};

}  // namespace

struct swg_paf {
  Text text;
  uint64_t n_lines = 0;
  Buf<uint32_t> q_id, t_id, qs, qe, ts, te, matches, block;
  Buf<double> identity;
  Buf<uint8_t> strand;
  Buf<uint64_t> rank, rec_off;
  Buf<uint32_t> rec_len;
  Buf<uint64_t> wide[6];             // q_start, q_end, t_start, t_end, matches, block length of a file with values >= 2^32
  std::vector<uint64_t> seq_offset;  // what rebasing took off each sequence's coordinates (empty: nothing, or per record:)
  std::vector<uint64_t> rec_off_q, rec_off_t;  // ... off each RECORD's query / target coordinates (rebased per sweep segment)
  std::vector<std::string> names;
  std::vector<uint32_t> g_last, g_two;
  swg_records rec{};
  bool identity_derived = false;     // every record's identity is matches / max(block length, 1)
  double load_ms = 0, parse_ms = 0;
  // ANI view (filled by swg_paf_ani_input)
  bool have_ani = false;
  Buf<uint8_t> ani_eligible;
  Buf<uint32_t> ani_pair;
  Buf<double> ani_matches, ani_block;
  swg_ani_input ani{};
};

namespace {

using clk = std::chrono::steady_clock;
double ms_since(clk::time_point a) { return std::chrono::duration<double, std::milli>(clk::now() - a).count(); }

// One line [p, e) without its "\n"; returns false when it has fewer than 11 tab-separated fields.  The first
// eleven fields are a few bytes each: a byte loop beats eleven memchr calls.
inline bool split11(const char* p, const char* e, const char* f[12]) {
  f[0] = p;
  int k = 1;
  const char* q = p;
  for (; q < e; ++q)
    if (*q == '\t') {
      f[k] = q + 1;
      if (++k == 12) break;
    }
  if (k < 11) return false;
  if (k == 11) f[11] = e + 1;  // ten separators: no tag area (one past the end)
  return true;
}

int parse_text(swg_paf* p, int threads) {
  const char* text = p->text.data;
  const size_t len = p->text.len;
  if (threads > 1 && len < (size_t)threads * 65536) threads = (int)(len / 65536) ? (int)(len / 65536) : 1;
  std::vector<Slice> sl(threads);
  for (int t = 0; t < threads; ++t) {
    size_t b = len / threads * t;
    if (t > 0 && b > 0) {
      const void* nl = std::memchr(text + b - 1, '\n', len - (b - 1));
      b = nl ? (size_t)(static_cast<const char*>(nl) - text) + 1 : len;
    }
    sl[t].begin = b;
    if (t > 0) sl[t - 1].end = b;
  }
  sl[threads - 1].end = len;

  const bool dbg = std::getenv("SWG_PAF_DEBUG") != nullptr;
  if (std::getenv("SWG_PAF_POPULATE") && p->text.map) {  // experiment knob: page tables of the mapped text filled in bulk, slice by slice
#ifdef MADV_POPULATE_READ
    const auto t0 = clk::now();
    parallel_for(threads, [&](int t) {
      const size_t a = sl[t].begin & ~size_t(4095), b = sl[t].end;
      if (b > a) madvise(const_cast<char*>(text) + a, b - a, MADV_POPULATE_READ);
    });
    if (dbg) std::fprintf(stderr, "[swg paf] populate %.1f ms\n", ms_since(t0));
#endif
  }
  auto tp = clk::now();
  auto lap = [&](const char* what) {
    if (dbg) std::fprintf(stderr, "[swg paf] %s %.1f ms\n", what, ms_since(tp));
    tp = clk::now();
  };
  // pass 1: lines per slice (a vectorisable byte count).  A record is a line with >= 11 fields -- nearly every line
  // of a PAF -- so the columns are sized by the line count and every slice parses into the slots starting at its first
  // line; the rare input with skipped lines is closed up afterwards.
  parallel_for(threads, [&](int t) {
    Slice& s = sl[t];
    uint64_t nl = 0;
    size_t i = s.begin;
    for (; i + 8 <= s.end; i += 8) {  // eight bytes at a time: exact zero-byte mask of (word ^ "\n\n...")
      uint64_t w;
      std::memcpy(&w, text + i, 8);
      w ^= 0x0a0a0a0a0a0a0a0aull;
      const uint64_t t = (w & 0x7f7f7f7f7f7f7f7full) + 0x7f7f7f7f7f7f7f7full;
      nl += (uint64_t)__builtin_popcountll(~(t | w | 0x7f7f7f7f7f7f7f7full));
    }
    for (; i < s.end; ++i) nl += text[i] == '\n';
    s.lines = nl + ((s.end > s.begin && text[s.end - 1] != '\n') ? 1 : 0);  // a last line without its newline
  });
  lap("pass 1 (count lines)");
  uint64_t n_lines = 0;
  for (auto& s : sl) {
    s.line_base = n_lines;
    n_lines += s.lines;
  }
  p->n_lines = n_lines;
  const size_t cap = n_lines ? n_lines : 1;
  for (auto* v : {&p->q_id, &p->t_id, &p->qs, &p->qe, &p->ts, &p->te, &p->matches, &p->block, &p->rec_len}) v->alloc(cap);
  p->identity.alloc(cap);
  p->strand.alloc(cap);
  p->rank.alloc(cap);
  p->rec_off.alloc(cap);

  // pass 2: parse into the columns.  Values >= 2^32 (RecordMeta is u64, src/paf_filter.rs:58-62) are rare: the 32-bit
  // columns are filled first, and only a file that has such a value is parsed once more into 64-bit columns, which are
  // then rebased sequence by sequence (host/rebase.h).
  auto pass2 = [&](const bool wide) { parallel_for(threads, [&](int t) {
    Slice& s = sl[t];
    uint64_t line = s.line_base, k = s.line_base;
    const char* f[12];
    auto fail = [&](const char* what) {
      if (s.err == SWG_OK) {
        s.err = SWG_ERR_RANGE;
        s.err_line = line;
        s.err_what = what;
      }
      return 0u;
    };
    for (size_t pos = s.begin; pos < s.end; ++line) {
      const void* nl = std::memchr(text + pos, '\n', s.end - pos);
      const size_t end = nl ? (size_t)(static_cast<const char*>(nl) - text) : s.end;
      size_t ll = end - pos;
      if (nl && ll && text[pos + ll - 1] == '\r') --ll;  // BufRead::lines strips "\n" and "\r\n" (a '\r' only with its '\n')
      const char* b = text + pos;
      const char* e = b + ll;
      pos = end + 1;
      if (!split11(b, e, f)) continue;
      auto fld = [&](int i) { return std::string_view(f[i], (size_t)(f[i + 1] - 1 - f[i])); };
      auto u64_or = [&](int i, uint64_t d) {
        uint64_t v;
        return parse_u64(f[i], (size_t)(f[i + 1] - 1 - f[i]), &v) ? v : d;
      };
      auto narrow = [&](uint64_t v) {
        if (v > 0xffffffffull) s.wide = true;
        return (uint32_t)v;
      };
      uint64_t matches = u64_or(9, 0);
      const uint64_t block = u64_or(10, 1);
      const double denom = (double)(block > 1 ? block : 1);
      double identity = (double)matches / denom;
      for (const char* tg = f[11]; tg <= e;) {  // tags, in order; last writer wins
        const char* te = static_cast<const char*>(std::memchr(tg, '\t', (size_t)(e - tg)));
        if (!te) te = e;
        const size_t tl = (size_t)(te - tg);
        if (tl >= 5 && tg[2] == ':' && tg[4] == ':') {
          if (tg[0] == 'd' && tg[1] == 'v' && tg[3] == 'f') {
            double dv;
            if (parse_f64(tg + 5, tl - 5, &dv)) identity = 1.0 - dv;
          } else if (tg[0] == 'c' && tg[1] == 'g' && tg[3] == 'Z') {
            uint64_t cm;
            if (cigar_eq_total(tg + 5, tl - 5, &cm) && cm > 0) {
              matches = cm;
              identity = (double)cm / denom;
            }
          }
        }
        tg = te + 1;
      }
      p->q_id[k] = s.names.get(fld(0), 0);
      p->t_id[k] = s.names.get(fld(5), 1);
      if (wide) {
        p->wide[0][k] = u64_or(2, 0);
        p->wide[1][k] = u64_or(3, 0);
        p->wide[2][k] = u64_or(7, 0);
        p->wide[3][k] = u64_or(8, 0);
        p->wide[4][k] = matches;
        p->wide[5][k] = block;
      } else {
        p->qs[k] = narrow(u64_or(2, 0));
        p->qe[k] = narrow(u64_or(3, 0));
        p->ts[k] = narrow(u64_or(7, 0));
        p->te[k] = narrow(u64_or(8, 0));
        p->matches[k] = narrow(matches);
        p->block[k] = narrow(block);
      }
      p->identity[k] = identity;
      // (what the device would compute from the 32-bit columns it is given; a value >= 2^32 sends the file to the wide path
      // anyway, whose rebased matches / block lengths are the same numbers)
      if (!(identity == (double)matches / denom)) s.custom_identity = true;
      p->strand[k] = (f[5] - 1 - f[4] == 1 && *f[4] == '+') ? 0 : 1;
      p->rank[k] = line;
      p->rec_off[k] = (uint64_t)(b - text);
      p->rec_len[k] = (uint32_t)ll;
      if (ll > 0xffffffffull) fail("line length");
      ++k;
    }
    s.recs = k - s.line_base;
  }); };
  pass2(false);
  lap("alloc + pass 2 (parse)");
  bool wide = false;
  p->identity_derived = true;
  for (auto& s : sl) {
    wide = wide || s.wide;
    if (s.custom_identity) p->identity_derived = false;
  }
  if (wide) {
    for (auto& w : p->wide) w.alloc(cap);
    pass2(true);
    lap("pass 2 again (64-bit columns)");
  }
  for (auto& s : sl)
    if (s.err != SWG_OK)
      return paf_error(s.err, "%s >= 2^32 on line %llu is not supported", s.err_what, (unsigned long long)(s.err_line + 1));
  // close the gaps left by skipped lines (slices in file order: a slice only ever moves towards the front)
  uint64_t n = 0;
  for (auto& s : sl) {
    s.rec_base = n;
    n += s.recs;
  }
  if (n >= (uint64_t(1) << 31)) return paf_error(SWG_ERR_RANGE, "more than 2^31-1 records");
  if (n != n_lines) {
    auto close_up = [&](auto& col) {
      for (auto& s : sl)
        if (s.recs && s.rec_base != s.line_base) std::memmove(col.data() + s.rec_base, col.data() + s.line_base, s.recs * sizeof(col[0]));
    };
    close_up(p->q_id);
    close_up(p->t_id);
    close_up(p->qs);
    close_up(p->qe);
    close_up(p->ts);
    close_up(p->te);
    close_up(p->matches);
    close_up(p->block);
    close_up(p->rec_len);
    close_up(p->identity);
    close_up(p->strand);
    close_up(p->rank);
    close_up(p->rec_off);
    if (wide)
      for (auto& w : p->wide) close_up(w);
    lap("close gaps");
  }

  // global ids: slices in file order, each slice's names in its own first-appearance order
  {
    std::unordered_map<std::string_view, uint32_t> ids;
    for (auto& s : sl) {
      s.remap.resize(s.names.names.size());
      for (size_t i = 0; i < s.names.names.size(); ++i) {
        auto it = ids.find(s.names.names[i]);
        if (it == ids.end()) {
          it = ids.emplace(s.names.names[i], (uint32_t)p->names.size()).first;
          p->names.emplace_back(s.names.names[i]);
        }
        s.remap[i] = it->second;
      }
    }
  }
  parallel_for(threads, [&](int t) {
    Slice& s = sl[t];
    if (t == 0) return;  // slice 0's local ids are already global
    for (uint64_t k = s.rec_base; k < s.rec_base + s.recs; ++k) {
      p->q_id[k] = s.remap[p->q_id[k]];
      p->t_id[k] = s.remap[p->t_id[k]];
    }
  });
  lap("name merge + remap");
  const uint32_t n_last = genome_table(p->names, prefix_last, &p->g_last);
  const uint32_t n_two = genome_table(p->names, prefix_two, &p->g_two);
  if (wide) {
    const uint32_t n_seq = (uint32_t)(p->names.empty() ? 1 : p->names.size());
    p->seq_offset.assign(n_seq, 0);
    const uint64_t* const c64[6] = {p->wide[0].data(), p->wide[1].data(), p->wide[2].data(),
                                    p->wide[3].data(), p->wide[4].data(), p->wide[5].data()};
    uint32_t* const c32[6] = {p->qs.data(), p->qe.data(), p->ts.data(), p->te.data(), p->matches.data(), p->block.data()};
    swg_rebase::Result rr = swg_rebase::columns(n, p->q_id.data(), p->t_id.data(), c64, n_seq, threads, c32, p->seq_offset.data());
    if (!rr.ok && rr.bad_field < 4 && swg_rebase::axis_tables_fit(n_seq, n_last)) {
      // a sequence touched over 2^32 bases or more: one constant per sweep segment and axis instead (host/rebase.h, columns_by_axis);
      // the columns are then relative to constants the handle does not publish (swg_paf_seq_offsets: NULL)
      p->rec_off_q.assign(n, 0);
      p->rec_off_t.assign(n, 0);
      const swg_rebase::Result r2 = swg_rebase::columns_by_axis(n, p->q_id.data(), p->t_id.data(), c64, n_seq, p->g_last.data(), n_last, threads, c32,
                                                                p->rec_off_q.data(), p->rec_off_t.data());
      if (r2.ok) {
        rr = r2;
        p->seq_offset.clear();
      } else {
        p->rec_off_q.clear();
        p->rec_off_t.clear();
      }
    }
    if (!rr.ok) {
      const uint64_t k = rr.bad_record;
      if (rr.bad_field >= 4)
        return paf_error(SWG_ERR_RANGE, "%s >= 2^32 on line %llu is not supported", swg_rebase::field_name(rr.bad_field),
                         (unsigned long long)(p->rank[k] + 1));
      const uint32_t sid = rr.bad_field < 2 ? p->q_id[k] : p->t_id[k];
      return paf_error(SWG_ERR_RANGE, "the stretch of sequence %s that the mappings against one genome touch spans 2^32 bases or more (%s on "
                       "line %llu, first mapped base %llu): not supported by the 32-bit device layout", p->names[sid].c_str(),
                       swg_rebase::field_name(rr.bad_field), (unsigned long long)(p->rank[k] + 1), (unsigned long long)p->seq_offset[sid]);
    }
    for (auto& w : p->wide) w.alloc(0);
    lap("rebase");
  }
  swg_records& r = p->rec;
  r.n = n;
  r.q_id = p->q_id.data();
  r.t_id = p->t_id.data();
  r.q_start = p->qs.data();
  r.q_end = p->qe.data();
  r.t_start = p->ts.data();
  r.t_end = p->te.data();
  r.identity = p->identity.data();
  r.matches = p->matches.data();
  r.block_len = p->block.data();
  r.strand = p->strand.data();
  r.n_seq = (uint32_t)(p->names.empty() ? 1 : p->names.size());
  r.seq_genome_last = p->g_last.data();
  r.n_genome_last = n_last;
  r.seq_genome_two = p->g_two.data();
  r.n_genome_two = n_two;
  return SWG_OK;
}

const char* const STATUS_TAG[4] = {"", "scaffold", "rescued", "unassigned"};  // src/paf_filter.rs:1708-1718
const size_t STATUS_TAG_LEN[4] = {0, 8, 7, 10};

inline size_t dec_digits(uint32_t v) {
  size_t d = 1;
  while (v >= 10) {
    v /= 10;
    ++d;
  }
  return d;
}
inline size_t out_len(const swg_paf* p, uint64_t k, uint8_t st, uint32_t ch) {
  return (size_t)p->rec_len[k] + (ch ? 12 + dec_digits(ch) : 0) + 6 + STATUS_TAG_LEN[st & 3] + 1;
}
inline char* emit(const swg_paf* p, uint64_t k, uint8_t st, uint32_t ch, char* o) {
  std::memcpy(o, p->text.data + p->rec_off[k], p->rec_len[k]);
  o += p->rec_len[k];
  if (ch) {
    std::memcpy(o, "\tch:Z:chain_", 12);
    o += 12;
    const size_t d = dec_digits(ch);
    for (size_t i = d; i-- > 0; ch /= 10) o[i] = (char)('0' + ch % 10);
    o += d;
  }
  std::memcpy(o, "\tst:Z:", 6);
  o += 6;
  std::memcpy(o, STATUS_TAG[st & 3], STATUS_TAG_LEN[st & 3]);
  o += STATUS_TAG_LEN[st & 3];
  *o++ = '\n';
  return o;
}

bool write_all(int fd, const char* b, size_t n) {
  while (n) {
    const ssize_t w = write(fd, b, n);
    if (w < 0) {
      if (errno == EINTR) continue;
      return false;
    }
    b += w;
    n -= (size_t)w;
  }
  return true;
}
bool pwrite_all(int fd, const char* b, size_t n, off_t at) {
  while (n) {
    const ssize_t w = pwrite(fd, b, n, at);
    if (w < 0) {
      if (errno == EINTR) continue;
      return false;
    }
    b += w;
    n -= (size_t)w;
    at += w;
  }
  return true;
}

}  // namespace

int swg_host_text_load(const char* path, int threads, const char** data, size_t* len, void** handle) {
  Text* t = nullptr;
  try {
    t = new Text;
    const int rc = load_text(path, pick_threads(threads), t);
    if (rc != SWG_OK) {
      delete t;
      return rc;
    }
  } catch (const std::bad_alloc&) {
    delete t;
    return paf_error(SWG_ERR_OOM, "out of host memory reading %s", path);
  }
  *data = t->data ? t->data : "";
  *len = t->len;
  *handle = t;
  return SWG_OK;
}
void swg_host_text_release(void* handle) { delete static_cast<Text*>(handle); }

extern "C" {

const char* swg_paf_last_error(void) { return g_paf_error.c_str(); }

int swg_paf_open(const char* path, int threads, swg_paf** out) {
  if (!path || !out) return paf_error(SWG_ERR_INVALID, "swg_paf_open: NULL argument");
  *out = nullptr;
  threads = pick_threads(threads);
  swg_paf* p = new (std::nothrow) swg_paf;
  if (!p) return paf_error(SWG_ERR_OOM, "host allocation failed");
  int rc;
  try {
    auto t0 = clk::now();
    rc = load_text(path, threads, &p->text);
    p->load_ms = ms_since(t0);
    if (rc == SWG_OK) {
      t0 = clk::now();
      rc = parse_text(p, threads);
      p->parse_ms = ms_since(t0);
    }
  } catch (const std::bad_alloc&) {
    rc = paf_error(SWG_ERR_OOM, "out of host memory while reading %s", path);
  }
  if (rc != SWG_OK) {
    delete p;
    return rc;
  }
  *out = p;
  return SWG_OK;
}

int swg_paf_open_buffer(const char* text, uint64_t len, int threads, swg_paf** out) {
  if ((!text && len) || !out) return paf_error(SWG_ERR_INVALID, "swg_paf_open_buffer: NULL argument");
  *out = nullptr;
  threads = pick_threads(threads);
  swg_paf* p = new (std::nothrow) swg_paf;
  if (!p) return paf_error(SWG_ERR_OOM, "host allocation failed");
  int rc;
  try {
    p->text.owned.assign(text, text + len);  // the handle outlives the caller's buffer
    p->text.data = p->text.owned.data();
    p->text.len = (size_t)len;
    const auto t0 = clk::now();
    rc = parse_text(p, threads);
    p->parse_ms = ms_since(t0);
  } catch (const std::bad_alloc&) {
    rc = paf_error(SWG_ERR_OOM, "out of host memory");
  }
  if (rc != SWG_OK) {
    delete p;
    return rc;
  }
  *out = p;
  return SWG_OK;
}

void swg_paf_close(swg_paf* p) { delete p; }
const swg_records* swg_paf_records(const swg_paf* p) { return p ? &p->rec : nullptr; }
int swg_paf_identity_is_derived(const swg_paf* p) { return p && p->identity_derived ? 1 : 0; }
const uint64_t* swg_paf_seq_offsets(const swg_paf* p) { return (p && !p->seq_offset.empty()) ? p->seq_offset.data() : nullptr; }
const uint64_t* swg_paf_record_offsets(const swg_paf* p, int axis) {
  if (!p || p->rec_off_q.empty()) return nullptr;
  return axis ? p->rec_off_t.data() : p->rec_off_q.data();
}
uint64_t swg_paf_num_lines(const swg_paf* p) { return p ? p->n_lines : 0; }
const uint64_t* swg_paf_ranks(const swg_paf* p) { return p ? p->rank.data() : nullptr; }
uint32_t swg_paf_num_sequences(const swg_paf* p) { return p ? (uint32_t)p->names.size() : 0; }
const char* swg_paf_sequence_name(const swg_paf* p, uint32_t id) {
  return (p && id < p->names.size()) ? p->names[id].c_str() : nullptr;
}
void swg_paf_timing(const swg_paf* p, double* load_ms, double* parse_ms) {
  if (load_ms) *load_ms = p ? p->load_ms : 0;
  if (parse_ms) *parse_ms = p ? p->parse_ms : 0;
}
int swg_paf_text(const swg_paf* p, const char** text, uint64_t* len) {
  if (!p || !text || !len) return paf_error(SWG_ERR_INVALID, "swg_paf_text: NULL argument");
  *text = p->text.data;
  *len = p->text.len;
  return SWG_OK;
}

int swg_paf_write(const swg_paf* p, const char* out_path, const uint8_t* status, const uint32_t* chain, int threads,
                  uint64_t* n_written) {
  if (!p || !out_path) return paf_error(SWG_ERR_INVALID, "swg_paf_write: NULL argument");
  const uint64_t n = p->rec.n;
  if (n && !status) return paf_error(SWG_ERR_INVALID, "swg_paf_write: status is NULL");
  threads = pick_threads(threads);
  if (threads > 16) threads = 16;  // concurrent pwrite()s into one file stop scaling (and then degrade) beyond this
  if ((uint64_t)threads > n / 4096 + 1) threads = (int)(n / 4096 + 1);
  const bool to_stdout = !std::strcmp(out_path, "-");
  // Output path == the mapped input file: truncating it would pull the pages from under the mapping (SIGBUS).  The
  // reference filters into a temporary file first (src/main.rs:3630-3636), so writing in place works there; here the
  // result goes to a sibling temporary file that replaces the input when it is complete.
  // The temporary file sits beside the RESOLVED path (a symlink named on the command line keeps pointing at the file that
  // is replaced), is created exclusively and takes over the replaced file's permission bits.
  std::string tmp_path, final_path;
  mode_t keep_mode = 0644;
  if (!to_stdout && p->text.map) {
    struct stat so {};
    if (stat(out_path, &so) == 0 && so.st_dev == p->text.src_dev && so.st_ino == p->text.src_ino) {
      char* rp = realpath(out_path, nullptr);
      final_path = rp ? rp : out_path;
      std::free(rp);
      tmp_path = final_path + ".swg_tmp." + std::to_string((long)getpid());
      keep_mode = so.st_mode & 07777;
    }
  }
  const char* open_path = tmp_path.empty() ? out_path : tmp_path.c_str();
  int fd = 1;
  if (!to_stdout) {
    if (tmp_path.empty()) {
      fd = open(open_path, O_WRONLY | O_CREAT | O_TRUNC, 0644);
    } else {
      fd = open(open_path, O_WRONLY | O_CREAT | O_EXCL, 0600);
      if (fd < 0 && errno == EEXIST && unlink(open_path) == 0) fd = open(open_path, O_WRONLY | O_CREAT | O_EXCL, 0600);  // a stale one of ours
      if (fd >= 0) (void)fchmod(fd, keep_mode);
    }
  }
  if (fd < 0) return paf_error(SWG_ERR_INVALID, "cannot create %s: %s", open_path, std::strerror(errno));
  std::atomic<int> bad{0};
  bool oom = false;
  // A failed allocation -- also what a worker body that fails turns into (host/threads.h) -- must not unwind through this
  // extern "C" function: it is caught, the descriptor is closed and a temporary file removed below, and the call returns
  // SWG_ERR_OOM.
  try {
  // sizes per thread range
  std::vector<uint64_t> bytes(threads + 1, 0), kept(threads, 0);
  auto lo = [&](int t) { return n / threads * t + (uint64_t)std::min<uint64_t>(t, n % threads); };
  parallel_for(threads, [&](int t) {
    uint64_t b = 0, c = 0;
    for (uint64_t k = lo(t); k < lo(t + 1); ++k)
      if (status[k]) {
        b += out_len(p, k, status[k], chain ? chain[k] : 0);
        ++c;
      }
    bytes[t + 1] = b;
    kept[t] = c;
  });
  for (int t = 0; t < threads; ++t) bytes[t + 1] += bytes[t];
  uint64_t total_kept = 0;
  for (auto c : kept) total_kept += c;
  if (n_written) *n_written = total_kept;
  const bool seekable = !to_stdout && lseek(fd, 0, SEEK_CUR) != (off_t)-1;
  constexpr size_t BUF = size_t(8) << 20;
  // (Tried in round 4: fallocate + a shared mapping of the output, 64 threads formatting straight into it instead of
  // pwrite()ing private buffers -- 1.65-2.1 s against 1.85 s per 5.3 GB on the GPU box: what bounds the writer is the kernel
  // inserting 1.3 M fresh pages into ONE file's page cache, whichever way the bytes arrive.)
  if (seekable) {
    parallel_for(threads, [&](int t) {
      std::vector<char> buf(BUF + (size_t(1) << 16));
      off_t at = (off_t)bytes[t];
      char* o = buf.data();
      for (uint64_t k = lo(t); k < lo(t + 1); ++k) {
        if (!status[k]) continue;
        const size_t need = out_len(p, k, status[k], chain ? chain[k] : 0);
        if ((size_t)(o - buf.data()) + need > buf.size()) {
          if (!pwrite_all(fd, buf.data(), (size_t)(o - buf.data()), at)) bad = errno ? errno : EIO;
          at += o - buf.data();
          o = buf.data();
          if (need > buf.size()) buf.resize(need);
          o = buf.data();
        }
        o = emit(p, k, status[k], chain ? chain[k] : 0, o);
      }
      if (o != buf.data() && !pwrite_all(fd, buf.data(), (size_t)(o - buf.data()), at)) bad = errno ? errno : EIO;
    });
  } else {
    std::vector<char> buf(BUF + (size_t(1) << 16));
    char* o = buf.data();
    for (uint64_t k = 0; k < n && !bad; ++k) {
      if (!status[k]) continue;
      const size_t need = out_len(p, k, status[k], chain ? chain[k] : 0);
      if ((size_t)(o - buf.data()) + need > buf.size()) {
        if (!write_all(fd, buf.data(), (size_t)(o - buf.data()))) bad = errno ? errno : EIO;
        if (need > buf.size()) buf.resize(need);
        o = buf.data();
      }
      o = emit(p, k, status[k], chain ? chain[k] : 0, o);
    }
    if (!bad && o != buf.data() && !write_all(fd, buf.data(), (size_t)(o - buf.data()))) bad = errno ? errno : EIO;
  }
  } catch (const std::bad_alloc&) {
    oom = true;
  } catch (const std::system_error&) {  // a thread could not be started
    oom = true;
  }
  if (!to_stdout && close(fd) != 0 && !bad) bad = errno ? errno : EIO;
  if (!tmp_path.empty()) {
    if (!bad && !oom && rename(tmp_path.c_str(), final_path.c_str()) != 0) bad = errno ? errno : EIO;
    if (bad || oom) unlink(tmp_path.c_str());
  }
  if (oom) return paf_error(SWG_ERR_OOM, "out of host memory (or host threads) while writing %s", out_path);
  if (bad) return paf_error(SWG_ERR_INVALID, "write to %s failed: %s", out_path, std::strerror(bad));
  return SWG_OK;
}

// ---- ANI pre-pass, host part (src/main.rs:405-446, 531-586) ----------------------------------------------
int swg_paf_ani_input(swg_paf* p, int threads, swg_ani_input* out) {
  if (!p || !out) return paf_error(SWG_ERR_INVALID, "swg_paf_ani_input: NULL argument");
  if (p->have_ani) {
    *out = p->ani;
    return SWG_OK;
  }
  const uint64_t n = p->rec.n;
  const uint64_t G = p->rec.n_genome_last;
  if (G * G > 0xffffffffull) return paf_error(SWG_ERR_UNSUPPORTED, "more than 65535 genomes in the ANI pass");
  threads = pick_threads(threads);
  if ((uint64_t)threads > n / 4096 + 1) threads = (int)(n / 4096 + 1);
  try {  // allocations and worker bodies (host/threads.h): nothing unwinds through the C ABI
  const size_t cap = n ? n : 1;
  p->ani_eligible.alloc(cap);
  p->ani_pair.alloc(cap);
  p->ani_matches.alloc(cap);
  p->ani_block.alloc(cap);
  const size_t n_seq = p->names.size();
  // first-seen sequence length per thread range: (record index of first sighting, length)
  struct Seen {
    std::vector<uint8_t> has;
    std::vector<uint64_t> len;
  };
  std::vector<Seen> seen(threads);
  const char* text = p->text.data;
  auto lo = [&](int t) { return n / threads * t + (uint64_t)std::min<uint64_t>(t, n % threads); };
  parallel_for(threads, [&](int t) {
    Seen& sn = seen[t];
    sn.has.assign(n_seq ? n_seq : 1, 0);
    sn.len.assign(n_seq ? n_seq : 1, 0);
    const char* f[12];
    for (uint64_t k = lo(t); k < lo(t + 1); ++k) {
      const char* b = text + p->rec_off[k];
      const char* e = b + p->rec_len[k];
      const uint32_t q = p->q_id[k], tg = p->t_id[k];
      p->ani_eligible[k] = 0;
      p->ani_pair[k] = 0;
      p->ani_matches[k] = 0.0;
      p->ani_block[k] = 0.0;
      if (b == e || *b == '#') continue;                       // main.rs:406-408
      if (p->g_last[q] == p->g_last[tg]) continue;             // main.rs:429-432
      if (!split11(b, e, f)) continue;                         // cannot happen for a record
      auto len = [&](int i) { return (size_t)(f[i + 1] - 1 - f[i]); };
      uint64_t ql = 0, tl = 0;
      if (!parse_u64(f[1], len(1), &ql)) ql = 0;
      if (!parse_u64(f[6], len(6), &tl)) tl = 0;
      double matches, block;
      if (!parse_f64(f[9], len(9), &matches)) matches = 0.0;
      if (!parse_f64(f[10], len(10), &block)) block = 1.0;
      double final_matches = matches;
      for (const char* tgp = f[11]; tgp <= e;) {               // first dv:f: that parses wins (break)
        const char* te = static_cast<const char*>(std::memchr(tgp, '\t', (size_t)(e - tgp)));
        if (!te) te = e;
        const size_t tl2 = (size_t)(te - tgp);
        if (tl2 >= 5 && !std::memcmp(tgp, "dv:f:", 5)) {
          double dv;
          if (parse_f64(tgp + 5, tl2 - 5, &dv)) {
            final_matches = (1.0 - dv) * block;
            break;
          }
        }
        tgp = te + 1;
      }
      p->ani_eligible[k] = 1;
      const uint32_t ga = p->g_last[q], gb = p->g_last[tg];
      p->ani_pair[k] = (uint32_t)((uint64_t)std::min(ga, gb) * G + std::max(ga, gb));
      p->ani_matches[k] = final_matches;
      p->ani_block[k] = block;
      if (!sn.has[q]) {  // genome_sizes.entry(key).or_insert(len): query first, then target
        sn.has[q] = 1;
        sn.len[q] = ql;
      }
      if (!sn.has[tg]) {
        sn.has[tg] = 1;
        sn.len[tg] = tl;
      }
    }
  });
  double total = 0.0;
  for (size_t sq = 0; sq < n_seq; ++sq)
    for (int t = 0; t < threads; ++t)
      if (seen[t].has[sq]) {
        total += (double)seen[t].len[sq];
        break;
      }
  p->ani.n = n;
  p->ani.eligible = p->ani_eligible.data();
  p->ani.pair = p->ani_pair.data();
  p->ani.n_pairs = G * G;
  p->ani.matches = p->ani_matches.data();
  p->ani.block_len = p->ani_block.data();
  p->ani.total_genome_size = total;
  p->have_ani = true;
  *out = p->ani;
  } catch (const std::bad_alloc&) {
    return paf_error(SWG_ERR_OOM, "out of host memory in the ANI pass over %llu records", (unsigned long long)n);
  } catch (const std::system_error& e) {
    return paf_error(SWG_ERR_OOM, "cannot start host threads: %s", e.what());
  }
  return SWG_OK;
}

int swg_paf_ani_stats(swg_ctx* ctx, swg_paf* p, int kind, double percentile, int sort, int threads, double* ani50) {
  if (!ctx || !p || !ani50) return paf_error(SWG_ERR_INVALID, "swg_paf_ani_stats: NULL argument");
  swg_ani_input in;
  int rc = swg_paf_ani_input(p, threads, &in);
  if (rc != SWG_OK) return rc;
  std::vector<uint8_t> status;
  const uint8_t* select = nullptr;
  if (kind == SWG_ANI_ORTHOGONAL) {  // main.rs:343-382: the fixed 1:1 filter, then the ALL pass over its survivors
    swg_config c{};
    c.min_block_length = 1000;
    c.mapping_filter_mode = SWG_MODE_ONE_TO_ONE;
    c.mapping_max_per_query = 1;
    c.mapping_max_per_target = 1;
    c.scaffold_filter_mode = SWG_MODE_ONE_TO_ONE;
    c.scaffold_max_per_query = 1;
    c.scaffold_max_per_target = 1;
    c.overlap_threshold = 0.95;
    c.scaffold_gap = 10000;
    c.min_scaffold_length = 0;
    c.scaffold_overlap_threshold = 0.95;
    c.scaffold_max_deviation = 0;
    c.scoring_function = SWG_SCORE_MATCHES;
    const uint64_t n = p->rec.n;
    status.assign(n ? n : 1, 0);
    std::vector<uint32_t> chain(n ? n : 1, 0);
    if (n) rc = swg_filter(ctx, &p->rec, &c, status.data(), chain.data(), nullptr);
    if (rc != SWG_OK) return paf_error(rc, "%s", swg_last_error(ctx));
    select = status.data();
  }
  rc = swg_ani_median(ctx, &in, select, kind, percentile, sort, ani50);
  if (rc != SWG_OK) return paf_error(rc, "%s", swg_last_error(ctx));
  return SWG_OK;
}

// parse_ani_method, src/main.rs:296-330
int swg_parse_ani_method(const char* s, int* kind, double* percentile, int* sort) {
  if (!s || !kind || !percentile || !sort) return 0;
  std::string lower(s);
  for (auto& c : lower)
    if (c >= 'A' && c <= 'Z') c = (char)(c - 'A' + 'a');
  if (lower == "all") {
    *kind = SWG_ANI_ALL;
    return 1;
  }
  if (lower == "orthogonal" || lower == "1:1") {
    *kind = SWG_ANI_ORTHOGONAL;
    return 1;
  }
  if (lower.empty() || lower[0] != 'n') return 0;
  const std::string rest = lower.substr(1);
  const size_t d1 = rest.find('-');
  const std::string num = rest.substr(0, d1);
  double pct;
  if (!parse_f64(num.data(), num.size(), &pct) || !(pct > 0.0 && pct <= 100.0)) return 0;
  int so = SWG_NSORT_IDENTITY;
  if (d1 != std::string::npos) {
    const size_t d2 = rest.find('-', d1 + 1);
    const std::string part = rest.substr(d1 + 1, d2 == std::string::npos ? std::string::npos : d2 - d1 - 1);
    if (part == "length") so = SWG_NSORT_LENGTH;
    else if (part == "identity") so = SWG_NSORT_IDENTITY;
    else if (part == "score") so = SWG_NSORT_SCORE;
    else return 0;
  }
  *kind = SWG_ANI_NPERCENTILE;
  *percentile = pct;
  *sort = so;
  return 1;
}

// parse_identity_value, src/cli.rs:76-130
int swg_parse_identity_value(const char* s, double ani_percentile, double* out) {
  if (!s || !out) return paf_error(SWG_ERR_INVALID, "swg_parse_identity_value: NULL argument");
  const std::string value(s);
  std::string lower = value;
  for (auto& c : lower)
    if (c >= 'A' && c <= 'Z') c = (char)(c - 'A' + 'a');
  if (lower.rfind("ani", 0) == 0) {
    if (ani_percentile < 0.0) return paf_error(SWG_ERR_INVALID, "Cannot use ANI-based threshold without input alignments");
    const std::string rem = lower.substr(3);
    const size_t plus = rem.find('+');
    const size_t cut = plus != std::string::npos ? plus : rem.find('-');
    if (rem.empty() || cut == std::string::npos) {  // the percentile number itself is ignored: only the median is honoured
      *out = ani_percentile;
      return SWG_OK;
    }
    const std::string off = rem.substr(cut + 1);
    double offset;
    if (!parse_f64(off.data(), off.size(), &offset)) return paf_error(SWG_ERR_INVALID, "Invalid ANI offset: %s", off.c_str());
    if (rem[cut] == '+') {
      const double v = ani_percentile + offset / 100.0;
      *out = v < 1.0 ? v : 1.0;
    } else {
      const double v = ani_percentile - offset / 100.0;
      *out = v > 0.0 ? v : 0.0;
    }
    return SWG_OK;
  }
  double v;
  if (!parse_f64(value.data(), value.size(), &v)) return paf_error(SWG_ERR_INVALID, "Invalid identity value: %s", s);
  *out = v > 1.0 ? v / 100.0 : v;
  return SWG_OK;
}

int swg_filter_paf(swg_ctx* ctx, const char* in_path, const char* out_path, const swg_config* cfg, int threads,
                   swg_stats* stats, double timing_ms[4]) {
  if (!ctx) return paf_error(SWG_ERR_INVALID, "swg_filter_paf: ctx is NULL");
  swg_paf* p = nullptr;
  int rc = swg_paf_open(in_path, threads, &p);
  if (rc != SWG_OK) return rc;
  const uint64_t n = p->rec.n;
  std::vector<uint8_t> status;
  std::vector<uint32_t> chain;
  try {
    status.assign(n ? n : 1, 0);
    chain.assign(n ? n : 1, 0);
  } catch (const std::bad_alloc&) {
    swg_paf_close(p);
    return paf_error(SWG_ERR_OOM, "out of host memory for the results of %llu records", (unsigned long long)n);
  }
  auto t0 = clk::now();
  if (n) rc = swg_filter(ctx, &p->rec, cfg, status.data(), chain.data(), stats);
  const double filter_ms = ms_since(t0);
  if (rc != SWG_OK) {
    paf_error(rc, "%s", swg_last_error(ctx));
    swg_paf_close(p);
    return rc;
  }
  t0 = clk::now();
  rc = swg_paf_write(p, out_path, status.data(), chain.data(), threads, nullptr);
  if (timing_ms) {
    timing_ms[0] = p->load_ms;
    timing_ms[1] = p->parse_ms;
    timing_ms[2] = filter_ms;
    timing_ms[3] = ms_since(t0);
  }
  swg_paf_close(p);
  return rc;
}

}  // extern "C"

// ---- .1aln front end: record derivation (src/unified_filter.rs:83-142) -------------------------------------------------
struct swg_aln {
  std::vector<uint32_t> q_id, t_id, qs, qe, ts, te, matches, block;
  std::vector<double> identity;
  std::vector<uint8_t> strand;
  std::vector<std::string> names;
  std::vector<uint32_t> g_last, g_two;
  std::vector<uint64_t> seq_offset;  // what rebasing took off each sequence's coordinates (empty: nothing, or per record:)
  std::vector<uint64_t> rec_off_q, rec_off_t;  // ... off each RECORD's query / target coordinates (rebased per sweep segment)
  swg_records rec{};
};

namespace {
// char::is_whitespace (Unicode White_Space): byte length of the white-space character at s[i], or 0
size_t ws_len(std::string_view s, size_t i) {
  const unsigned char c = (unsigned char)s[i];
  if ((c >= 0x09 && c <= 0x0d) || c == 0x20) return 1;
  if (c == 0xc2 && i + 1 < s.size()) {
    const unsigned char d = (unsigned char)s[i + 1];
    return (d == 0x85 || d == 0xa0) ? 2 : 0;
  }
  if (i + 2 < s.size()) {
    const unsigned char d = (unsigned char)s[i + 1], e = (unsigned char)s[i + 2];
    if (c == 0xe1 && d == 0x9a && e == 0x80) return 3;
    if (c == 0xe2 && d == 0x80 && ((e >= 0x80 && e <= 0x8a) || e == 0xa8 || e == 0xa9 || e == 0xaf)) return 3;
    if (c == 0xe2 && d == 0x81 && e == 0x9f) return 3;
    if (c == 0xe3 && d == 0x80 && e == 0x80) return 3;
  }
  return 0;
}
// split_whitespace().next().unwrap_or(full), src/unified_filter.rs:83-92
std::string_view first_word_or_all(std::string_view s) {
  size_t i = 0;
  while (i < s.size()) {
    const size_t w = ws_len(s, i);
    if (!w) break;
    i += w;
  }
  if (i >= s.size()) return s;
  size_t j = i;
  while (j < s.size() && !ws_len(s, j)) ++j;
  return s.substr(i, j - i);
}
}  // namespace

extern "C" {

int swg_aln_open(const swg_aln_input* in, swg_aln** out) {
  if (out) *out = nullptr;
  if (!in || !out) return paf_error(SWG_ERR_INVALID, "swg_aln_open: NULL argument");
  const uint64_t n = in->n;
  if (n && (!in->query_name || !in->target_name || !in->query_start || !in->query_end || !in->target_start ||
            !in->target_end || !in->matches || !in->strand))
    return paf_error(SWG_ERR_INVALID, "swg_aln_open: a column is NULL");
  if (n >= (uint64_t(1) << 31)) return paf_error(SWG_ERR_RANGE, "swg_aln_open: more than 2^31-1 alignments");
  swg_aln* a = nullptr;
  try {
    a = new swg_aln;
    const size_t cap = n ? n : 1;
    a->q_id.resize(cap);
    a->t_id.resize(cap);
    a->qs.resize(cap);
    a->qe.resize(cap);
    a->ts.resize(cap);
    a->te.resize(cap);
    a->matches.resize(cap);
    a->block.resize(cap);
    a->identity.resize(cap);
    a->strand.resize(cap);
    std::unordered_map<std::string, uint32_t> ids;  // SequenceIndex: first appearance, query before target
    auto intern = [&](std::string_view nm) {
      auto it = ids.find(std::string(nm));
      if (it != ids.end()) return it->second;
      const uint32_t id = (uint32_t)a->names.size();
      a->names.emplace_back(nm);
      ids.emplace(std::string(nm), id);
      return id;
    };
    bool wide = false;
    for (uint64_t k = 0; k < n; ++k) {
      if (!in->query_name[k] || !in->target_name[k]) {
        delete a;
        return paf_error(SWG_ERR_INVALID, "swg_aln_open: name %llu is NULL", (unsigned long long)k);
      }
      a->q_id[k] = intern(first_word_or_all(in->query_name[k]));
      a->t_id[k] = intern(first_word_or_all(in->target_name[k]));
      const uint64_t q0 = in->query_start[k], q1 = in->query_end[k], t0 = in->target_start[k], t1 = in->target_end[k];
      const uint64_t query_span = q1 - q0, target_span = t1 - t0;  // :107-108 (wrapping, as release Rust)
      const uint64_t block = query_span + target_span;             // :112
      const uint64_t m = in->matches[k];                           // :115
      if ((q0 | q1 | t0 | t1 | block | m) >> 32) wide = true;  // rebased below
      a->qs[k] = (uint32_t)q0;
      a->qe[k] = (uint32_t)q1;
      a->ts[k] = (uint32_t)t0;
      a->te[k] = (uint32_t)t1;
      a->matches[k] = (uint32_t)m;
      a->block[k] = (uint32_t)block;
      a->identity[k] = query_span > 0 ? (double)m / (double)query_span : 0.0;  // :119-123
      a->strand[k] = in->strand[k] == '+' ? 0 : 1;
    }
    if (wide) {  // values >= 2^32: coordinates rebased sequence by sequence (host/rebase.h)
      std::vector<uint64_t> block64(n);
      for (uint64_t k = 0; k < n; ++k)
        block64[k] = (in->query_end[k] - in->query_start[k]) + (in->target_end[k] - in->target_start[k]);
      const uint32_t n_seq = (uint32_t)a->names.size();
      a->seq_offset.assign(n_seq, 0);
      const uint64_t* const c64[6] = {in->query_start, in->query_end, in->target_start, in->target_end, in->matches, block64.data()};
      uint32_t* const c32[6] = {a->qs.data(), a->qe.data(), a->ts.data(), a->te.data(), a->matches.data(), a->block.data()};
      swg_rebase::Result rr = swg_rebase::columns(n, a->q_id.data(), a->t_id.data(), c64, n_seq, pick_threads(0), c32, a->seq_offset.data());
      if (!rr.ok && rr.bad_field < 4) {  // (see swg_paf_open)
        std::vector<uint32_t> g_last;
        const uint32_t n_last = genome_table(a->names, prefix_last, &g_last);
        if (swg_rebase::axis_tables_fit(n_seq, n_last)) {
          a->rec_off_q.assign(n, 0);
          a->rec_off_t.assign(n, 0);
          const swg_rebase::Result r2 = swg_rebase::columns_by_axis(n, a->q_id.data(), a->t_id.data(), c64, n_seq, g_last.data(), n_last, pick_threads(0), c32,
                                                                    a->rec_off_q.data(), a->rec_off_t.data());
          if (r2.ok) {
            rr = r2;
            a->seq_offset.clear();
          } else {
            a->rec_off_q.clear();
            a->rec_off_t.clear();
          }
        }
      }
      if (!rr.ok) {
        const uint64_t k = rr.bad_record;
        const int f = rr.bad_field;
        std::string what = f >= 4 ? std::string(swg_rebase::field_name(f)) + " >= 2^32"
                                  : "the mapped stretch of sequence " + a->names[f < 2 ? a->q_id[k] : a->t_id[k]] +
                                        " spans 2^32 bases or more";
        delete a;
        return paf_error(SWG_ERR_RANGE, "alignment %llu: %s: not supported by the 32-bit device layout", (unsigned long long)k,
                         what.c_str());
      }
    }
    const uint32_t n_last = genome_table(a->names, prefix_last, &a->g_last);
    const uint32_t n_two = genome_table(a->names, prefix_two, &a->g_two);
    swg_records& r = a->rec;
    r.n = n;
    r.q_id = a->q_id.data();
    r.t_id = a->t_id.data();
    r.q_start = a->qs.data();
    r.q_end = a->qe.data();
    r.t_start = a->ts.data();
    r.t_end = a->te.data();
    r.identity = a->identity.data();
    r.matches = a->matches.data();
    r.block_len = a->block.data();
    r.strand = a->strand.data();
    r.n_seq = (uint32_t)(a->names.empty() ? 1 : a->names.size());
    r.seq_genome_last = a->g_last.data();
    r.n_genome_last = n_last;
    r.seq_genome_two = a->g_two.data();
    r.n_genome_two = n_two;
  } catch (const std::bad_alloc&) {
    delete a;
    return paf_error(SWG_ERR_OOM, "swg_aln_open: out of host memory");
  }
  *out = a;
  return SWG_OK;
}
void swg_aln_close(swg_aln* a) { delete a; }
const swg_records* swg_aln_records(const swg_aln* a) { return a ? &a->rec : nullptr; }
const uint64_t* swg_aln_seq_offsets(const swg_aln* a) { return (a && !a->seq_offset.empty()) ? a->seq_offset.data() : nullptr; }
const uint64_t* swg_aln_record_offsets(const swg_aln* a, int axis) {
  if (!a || a->rec_off_q.empty()) return nullptr;
  return axis ? a->rec_off_t.data() : a->rec_off_q.data();
}
uint32_t swg_aln_num_sequences(const swg_aln* a) { return a ? (uint32_t)a->names.size() : 0; }
const char* swg_aln_sequence_name(const swg_aln* a, uint32_t id) {
  return (a && id < a->names.size()) ? a->names[id].c_str() : nullptr;
}

}  // extern "C"
