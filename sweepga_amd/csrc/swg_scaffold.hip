// Scaffold stage -- under construction.
#include "swg_pipeline.h"

int swg_scaffold_stage(swg_ctx* ctx, const swg_records*, const swg_config*, const uint8_t*, const uint8_t*,
                       const uint64_t*, int, uint8_t*, uint32_t*, swg_stats*) {
  return swg_set_error(ctx, SWG_ERR_UNSUPPORTED, "scaffold stage not built yet");
}

extern "C" int swg_plane_sweep_scaffolds(swg_ctx* ctx, uint64_t, const uint32_t*, const uint32_t*, uint32_t,
                                         const uint32_t*, uint32_t, const uint64_t*, const uint64_t*,
                                         const uint64_t*, const uint64_t*, const double*, int, uint64_t, uint64_t,
                                         double, int, uint64_t*, uint64_t*) {
  return swg_set_error(ctx, SWG_ERR_UNSUPPORTED, "not built yet");
}
extern "C" int swg_merge_chains(swg_ctx* ctx, const swg_records*, uint64_t, uint32_t*, uint32_t*, uint32_t*,
                                uint32_t*, uint32_t*, double*, uint64_t*) {
  return swg_set_error(ctx, SWG_ERR_UNSUPPORTED, "not built yet");
}
extern "C" int swg_union_find_sets(swg_ctx* ctx, uint64_t, uint64_t, const uint32_t*, const uint32_t*, uint32_t*,
                                   uint64_t*) {
  return swg_set_error(ctx, SWG_ERR_UNSUPPORTED, "not built yet");
}
