// Stage entry points of the filter pipeline (internal).
#pragma once
#include "swg_internal.h"

// Mapping-level plane sweep (src/paf_filter.rs:972-1123): query-axis sweep per (query sequence,
// target genome), target-axis sweep per (target sequence, query genome), intersection.
int swg_mapping_sweep(swg_ctx* ctx, const swg_records* r, const swg_config* cfg, const uint8_t* alive,
                      const swg_key_ends* key_ends, int pos_bits, uint8_t* keep);

// Scaffold stage (src/paf_filter.rs:436-747): chaining, span/identity filter, scaffold sweep,
// anchors, inversion capture, rescue.  alive = step-1 survivors, keep1 = mapping-sweep survivors.
int swg_scaffold_stage(swg_ctx* ctx, const swg_records* r, const swg_config* cfg, const uint8_t* alive,
                       const uint8_t* keep1, int pos_bits, uint8_t* status_out,
                       uint32_t* chain_out, swg_stats* stats);
