// Shared by the host translation units of libsweepga_gpu.so; not part of the C ABI.
#ifndef SWG_HOST_INTERNAL_H
#define SWG_HOST_INTERNAL_H
#include <cstddef>

// open_paf_input (src/paf.rs:10-30): the whole input as text -- mmap for plain files, parallel BGZF / serial gzip
// inflate for .gz/.bgz (or the gzip magic), "-" = stdin.  *handle owns the bytes until swg_host_text_release.
// Errors: negative code, text in swg_paf_last_error().
int swg_host_text_load(const char* path, int threads, const char** data, size_t* len, void** handle);
void swg_host_text_release(void* handle);
#endif
