// Host-side sanitizer harness for sweepga_amd/csrc/host/paf_io.cpp (ASan + UBSan run on the CPU build only: the
// GPU pool has no sanitizer support).  Opens every file given on the command line with several thread counts,
// touches every column, builds the ANI view, writes the records back with all status/chain variants and checks
// a few invariants.  The device entry points paf_io.cpp refers to are stubbed: nothing here needs a GPU.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include "../../include/sweepga_gpu.h"

extern "C" {
int swg_filter(swg_ctx*, const swg_records*, const swg_config*, uint8_t*, uint32_t*, swg_stats*) { return SWG_ERR_NO_DEVICE; }
const char* swg_last_error(const swg_ctx*) { return "stub"; }
int swg_ani_median(swg_ctx*, const swg_ani_input*, const uint8_t*, int, double, int, double*) { return SWG_ERR_NO_DEVICE; }
}

int main(int argc, char** argv) {
  unsigned long long checksum = 0;
  for (int a = 1; a < argc; ++a) {
    for (int threads : {1, 2, 7}) {
      swg_paf* p = nullptr;
      const int rc = swg_paf_open(argv[a], threads, &p);
      if (rc != SWG_OK) {
        std::printf("%s: open failed (%d): %s\n", argv[a], rc, swg_paf_last_error());
        continue;
      }
      const swg_records* r = swg_paf_records(p);
      const uint64_t n = r->n;
      for (uint64_t i = 0; i < n; ++i) {
        checksum += r->q_id[i] + r->t_id[i] + r->q_start[i] + r->q_end[i] + r->t_start[i] + r->t_end[i] + r->matches[i] + r->block_len[i] +
                    r->strand[i] + (unsigned long long)(r->identity[i] * 1000.0) + swg_paf_ranks(p)[i];
        if (r->q_id[i] >= r->n_seq || r->t_id[i] >= r->n_seq) return 10;
      }
      for (uint32_t s = 0; s < swg_paf_num_sequences(p); ++s) {
        checksum += std::string(swg_paf_sequence_name(p, s)).size();
        if (r->seq_genome_last[s] >= r->n_genome_last || r->seq_genome_two[s] >= r->n_genome_two) return 11;
      }
      swg_ani_input in;
      if (swg_paf_ani_input(p, threads, &in) != SWG_OK) return 12;
      for (uint64_t i = 0; i < n; ++i) {
        checksum += in.eligible[i] + in.pair[i] + (unsigned long long)in.matches[i] + (unsigned long long)in.block_len[i];
        if (in.n_pairs && in.pair[i] >= in.n_pairs) return 13;
      }
      std::vector<uint8_t> status(n ? n : 1);
      std::vector<uint32_t> chain(n ? n : 1);
      for (uint64_t i = 0; i < n; ++i) {
        status[i] = (uint8_t)(i % 4);
        chain[i] = (i % 3) ? (uint32_t)(i * 2654435761u) : 0u;
      }
      uint64_t kept = 0;
      const std::string out = std::string(argv[a]) + ".san.out";
      if (swg_paf_write(p, out.c_str(), status.data(), chain.data(), threads, &kept) != SWG_OK) return 14;
      if (swg_paf_write(p, out.c_str(), status.data(), nullptr, threads, &kept) != SWG_OK) return 15;
      std::remove(out.c_str());
      swg_paf_close(p);
    }
  }
  std::printf("ok %llu\n", checksum);
  return 0;
}
