// alnstats -- the reference's src/bin/alnstats.rs command line over libsweepga_gpu.so's host code (swg_alnstats_*).
//   alnstats <file1> [file2] [-d|--detailed]
// One file: print_stats; two files: compare_stats (then -d has no effect, as in the reference).
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../../include/sweepga_gpu.h"

namespace {
const char* USAGE =
    "Statistics for alignment files (PAF, 1aln)\n\n"
    "Usage: alnstats [OPTIONS] <FILE1> [FILE2]\n\n"
    "Arguments:\n"
    "  <FILE1>  First alignment file (PAF format)\n"
    "  [FILE2]  Optional second file for comparison\n\n"
    "Options:\n"
    "  -d, --detailed  Show detailed per-genome-pair statistics\n"
    "  -h, --help      Print help\n";

int fail(const char* msg) {  // anyhow's report of main's Err: "Error: ..." on stderr, status 1
  std::fprintf(stderr, "Error: %s\n", msg);
  return 1;
}
int emit(char* text, uint64_t len) {
  std::fwrite(text, 1, len, stdout);
  swg_free(text);
  return 0;
}
}  // namespace

int main(int argc, char** argv) {
  std::vector<const char*> files;
  bool detailed = false;
  for (int i = 1; i < argc; ++i) {
    const char* a = argv[i];
    if (!std::strcmp(a, "-d") || !std::strcmp(a, "--detailed")) {
      detailed = true;
    } else if (!std::strcmp(a, "-h") || !std::strcmp(a, "--help")) {
      std::fputs(USAGE, stdout);
      return 0;
    } else if (a[0] == '-' && a[1] != 0) {
      std::fprintf(stderr, "error: unexpected argument '%s' found\n\n%s", a, USAGE);
      return 2;
    } else {
      files.push_back(a);
    }
  }
  if (files.empty() || files.size() > 2) {
    std::fprintf(stderr, "error: %s\n\n%s", files.empty() ? "the following required arguments were not provided:\n  <FILE1>" : "unexpected argument found", USAGE);
    return 2;
  }
  swg_alnstats* s1 = nullptr;
  swg_alnstats* s2 = nullptr;
  if (swg_alnstats_open(files[0], 0, &s1) != SWG_OK) return fail(swg_alnstats_last_error());
  char* text = nullptr;
  uint64_t len = 0;
  int rc = 0;
  if (files.size() == 2) {
    if (swg_alnstats_open(files[1], 0, &s2) != SWG_OK) {
      swg_alnstats_close(s1);
      return fail(swg_alnstats_last_error());
    }
    rc = swg_alnstats_compare(s1, s2, files[0], files[1], &text, &len) == SWG_OK ? emit(text, len) : fail(swg_alnstats_last_error());
  } else if (swg_alnstats_report(s1, files[0], detailed, &text, &len) == SWG_OK) {
    rc = emit(text, len);
  } else if (detailed && swg_alnstats_report(s1, files[0], 0, &text, &len) == SWG_OK) {
    // a NaN coverage: the reference has printed the summary and the table header when its sort panics (status 101)
    emit(text, len);
    std::printf("\nPer-genome-pair statistics:\n%s\n", std::string(60, '-').c_str());
    std::fflush(stdout);
    std::fprintf(stderr, "%s\n", swg_alnstats_last_error());
    rc = 101;
  } else {
    rc = fail(swg_alnstats_last_error());
  }
  swg_alnstats_close(s1);
  swg_alnstats_close(s2);
  return rc;
}
