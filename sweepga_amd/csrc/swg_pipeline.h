// Stage entry points of the filter pipeline (internal).
#pragma once
#include "swg_internal.h"

// Mapping-level plane sweep (src/paf_filter.rs:972-1123): query-axis sweep per (query sequence,
// target genome), target-axis sweep per (target sequence, query genome), intersection.
// q_order (optional, [n]): receives the records in the query axis' sorted order (segment = (query sequence, target genome),
// then q_start, then index; dead records first); *q_order_valid says whether it was produced.
// pair_runs (optional): the input is grouped by (query, target) pair and these are its runs (swg_scaf::PairPlan): with
// score_key (8 bytes per record, written by swg_prepare), key_ends == nullptr and no q_order the axes sort their begins segment
// by segment in LDS (swg_segsort.hip); n_alive = the retained records.
int swg_mapping_sweep(swg_ctx* ctx, const swg_records* r, const swg_config* cfg, const uint8_t* alive,
                      const swg_key_ends* key_ends, int pos_bits, uint8_t* keep, uint32_t* q_order = nullptr,
                      int* q_order_valid = nullptr, const void* pair_runs = nullptr, uint32_t n_pair_runs = 0,
                      const uint64_t* score_key = nullptr, uint64_t n_alive = 0);

// which value columns (identity / matches + block_len) a flag set reads: csrc/swg_filter.hip
void swg_value_columns_needed(const swg_config* cfg, bool* identity_value, bool* weighted_identity);
// Streamed host path (csrc/swg_stream.hip): ranges of whole query genomes, uploads overlapped with the filter.  *taken = 0:
// not applicable (input not grouped by query genome, too small, SWG_STREAM=0), nothing was done; the caller runs its own path.
int swg_stream_try(swg_ctx* const* ctxs, int n_ctx, const swg_records* r, const swg_config* cfg, uint8_t* status_out,
                   uint32_t* chain_out, swg_stats* stats, int* taken);
// Device memory a host-path call needs, reserved ahead (csrc/swg_filter.hip): the staging block for n records, the scratch
// arena for a filter call over n records.
int swg_io_block_reserve(swg_ctx* ctx, uint64_t n, uint32_t n_seq);
int swg_filter_reserve_arena(swg_ctx* ctx, uint64_t n, const swg_records* rec, const swg_config* cfg, bool wide);

// Scaffold stage (src/paf_filter.rs:436-747): chaining, span/identity filter, scaffold sweep,
// anchors, inversion capture, rescue.  alive = step-1 survivors, keep1 = mapping-sweep survivors.
// q_order (optional): the alive records already ordered by q_start inside every (query, target, strand) group (the query
// axis' order of the mapping sweep): sort A then only needs its passes over the group bits.
int swg_scaffold_stage(swg_ctx* ctx, const swg_records* r, const swg_config* cfg, const uint8_t* alive,
                       const uint8_t* keep1, int pos_bits, uint8_t* status_out,
                       uint32_t* chain_out, swg_stats* stats, const uint32_t* q_order = nullptr, uint64_t n_alive = ~0ull,
                       const swg_key_ends* slots = nullptr);
