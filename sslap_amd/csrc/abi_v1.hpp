// abi_v1.hpp -- the version-1 layouts of the two public structs that changed in ABI 2 (include/misslap.h, "ABI
// history"): a caller built against the round-2 header passes a misslap_options of 88 bytes and expects a misslap_meta
// of 376 bytes without a size field.  The library recognises such a caller by the options' struct_size at create and
// serves it with these layouts; nothing else of the C ABI changed shape.
#pragma once
#include <stdint.h>

extern "C" {
typedef struct misslap_options_v1 {
    int32_t struct_size;
    int32_t device;
    int32_t maximize;
    float eps_start;
    int64_t max_iter;
    int32_t input_on_device;
    int32_t tail_threshold;
    int32_t force_f64_values;
    int32_t profile;
    int32_t shard_rank;
    int32_t shard_world;
    int32_t rounds_per_sync;
    int32_t reserved[8];  // [0] tiled_min_K, [1] tiled_shape, [2] tiled_force, [3] shard_min_K, [4] cand_mode,
                          // [5] partial_in_list_order, [6] nnz_limit, [7] cand_build_max_K | (cand_refresh_min << 24)
} misslap_options_v1;

typedef struct misslap_meta_v1 {
    float start_eps, final_eps, target_eps;
    int32_t eCE, soln_found, nreductions;
    int64_t its, n_assigned, n_rows, n_cols, nnz;
    float obj_f32;
    double obj_f64, setup_ms, solve_ms;
    uint64_t edges_scanned, bids_made;
    int64_t grid_rounds, tail_rounds;
    int32_t bytes_per_edge, profiled;
    int64_t bid_launches;
    double bid_ms;
    uint64_t bid_edges;
    int64_t fullscan_launches;
    double fullscan_ms;
    uint64_t fullscan_edges;
    int64_t tail_launches;
    double tail_ms;
    uint64_t tail_edges;
    int64_t tiled_launches;
    double tiled_ms;
    uint64_t tiled_edges;
    int32_t tiled_active, tiled_min_K;
    int64_t merge_launches;
    double merge_ms;
    uint64_t shard_edges, cand_hits, cand_edges;
    double tail_stats[12];
} misslap_meta_v1;
}
static_assert(sizeof(misslap_options_v1) == 88, "version-1 options are 88 bytes");
static_assert(sizeof(misslap_meta_v1) == 376, "version-1 meta is 376 bytes");
