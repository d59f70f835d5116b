// misslap.hip -- host driver + C ABI of libmisslap.so (MI355X / gfx950 only).
//
// Host side of the reference's AuctionSolver (sslap/auction_.pyx:164-523): construction (CSR build on
// the GPU), the eps-scaling outer loop of solve() (:268-306) and the result / meta extraction.  All
// per-edge and per-person work is in the kernels_*.hpp headers; this file only sequences launches on
// one HIP stream and reads back a 100-byte control block when the loop needs a decision.
//
// Build: hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -fno-honor-nans -shared -fPIC misslap.hip -o libmisslap.so
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <thread>
#include <string>
#include <utility>
#include <vector>

#include "../../include/misslap.h"
#include "abi_v1.hpp"
#include "device_common.hpp"
#include "kernels_check.hpp"
#ifdef MISSLAP_DIAG
#include "../../include/misslap_diag.h"
#include "kernels_debug.hpp"
#endif
#include "kernels_ingest.hpp"
#include "kernels_round.hpp"
#include "kernels_tail.hpp"
#include "host_matching.hpp"
#include "host_comm.hpp"
#include "kernels_matching.hpp"
#include "kernels_tiled.hpp"

using namespace misslap;

#define MISSLAP_API extern "C" __attribute__((visibility("default")))
#include "host_base.hpp"
#include "host_batch.hpp"
#include "host_cache.hpp"
#include "host_rounds.hpp"
#include "host_create.hpp"

// ------------------------------------------------------------------------------------------------
MISSLAP_API int misslap_abi_version(void) { return MISSLAP_ABI_VERSION; }
#include "abi_matching.hpp"
#include "abi_util.hpp"

MISSLAP_API int misslap_create(misslap_solver **out, int64_t nnz, const int32_t *loc, const double *val,
                               const misslap_options *opt_in) {
    const double t0 = now_ms();
    misslap_options o2;
    int abi = 0;
    int rc = normalise_options(opt_in, &o2, &abi);
    if (rc) return rc;
    const misslap_options *opt = &o2;
    if (!loc || !val) return fail(MISSLAP_ERR_INVALID, "null loc / val");
    if (nnz <= 0) return fail(MISSLAP_ERR_INVALID, "empty problem (nnz = %lld)", (long long)nnz);
    if (nnz >= nnz_limit(opt))
        return fail(MISSLAP_ERR_INVALID, "nnz must be < %lld (int32 row pointers)", (long long)nnz_limit(opt));
    misslap_solver *h = nullptr;
    CreateTrace trace(nullptr);
    rc = new_handle(out, opt, abi, &h);
    if (rc) return rc;
    trace.st = h->stream;
    trace.stage("new handle");
    h->nnz = nnz;
    if ((rc = sync_device_inputs(opt, h->stream))) {
        free_all(h);
        return rc;
    }
    trace.stage("inputs ordered");
    const int *d_loc = nullptr;
    const double *d_val = nullptr;
    int *own_loc = nullptr;
    double *own_val = nullptr;
    int last_row = -1;
    auto cleanup = [&](int code) {
        if (own_loc) (void)hipFree(own_loc);
        if (own_val) (void)hipFree(own_val);
        if (code) free_all(h);
        return code;
    };
    if (opt->input_on_device) {
        d_loc = loc;
        d_val = val;
        if (hipMemcpy(&last_row, loc + 2 * (nnz - 1), sizeof(int), hipMemcpyDeviceToHost) != hipSuccess)
            return cleanup(fail(MISSLAP_ERR_HIP, "cannot read loc from the device"));
    } else {
        if ((rc = dev_alloc(&own_loc, (size_t)nnz * 2))) return cleanup(rc);
        if ((rc = dev_alloc(&own_val, (size_t)nnz))) return cleanup(rc);
        if (hipMemcpyAsync(own_loc, loc, sizeof(int) * 2 * (size_t)nnz, hipMemcpyHostToDevice, h->stream) != hipSuccess ||
            hipMemcpyAsync(own_val, val, sizeof(double) * (size_t)nnz, hipMemcpyHostToDevice, h->stream) != hipSuccess)
            return cleanup(fail(MISSLAP_ERR_HIP, "host-to-device copy of the COO input failed"));
        d_loc = own_loc;
        d_val = own_val;
        last_row = loc[2 * (nnz - 1)];
    }
    trace.stage("last row read");
    rc = build_from_device_coo(h, d_loc, d_val, last_row, opt);
    if (rc) return cleanup(rc);
    cleanup(0);
    h->setup_ms = now_ms() - t0;
    trace.stage("(build, see above) + cleanup");
    *out = h;
    return MISSLAP_OK;
}

MISSLAP_API int misslap_create_dense(misslap_solver **out, int64_t n_rows, int64_t n_cols, const double *mat,
                                     const misslap_options *opt_in, int64_t *nnz_out) {
    const double t0 = now_ms();
    misslap_options o2;
    int abi = 0;
    int rc = normalise_options(opt_in, &o2, &abi);
    if (rc) return rc;
    const misslap_options *opt = &o2;
    if (!mat) return fail(MISSLAP_ERR_INVALID, "null mat");
    if (n_rows <= 0 || n_cols <= 0 || n_rows > 0x7ffffffe || n_cols > 0x7ffffffe)
        return fail(MISSLAP_ERR_INVALID, "bad dense shape");
    misslap_solver *h = nullptr;
    rc = new_handle(out, opt, abi, &h);
    if (rc) return rc;
    double *d_mat = nullptr, *d_val = nullptr;
    int *d_cnt = nullptr, *d_ptr = nullptr, *d_loc = nullptr;
    IngestStats *d_st = nullptr;
    auto cleanup = [&](int code) {
        for (void *p : {(void *)d_mat, (void *)d_val, (void *)d_cnt, (void *)d_ptr, (void *)d_loc, (void *)d_st})
            if (p) (void)hipFree(p);
        if (code) free_all(h);
        return code;
    };
    const size_t cells = (size_t)n_rows * (size_t)n_cols;
    if ((rc = sync_device_inputs(opt, h->stream))) return cleanup(rc);
    if ((rc = dev_alloc(&d_mat, cells))) return cleanup(rc);
    if ((rc = dev_alloc(&d_cnt, (size_t)n_rows))) return cleanup(rc);
    if ((rc = dev_alloc(&d_ptr, (size_t)n_rows + 1))) return cleanup(rc);
    if ((rc = dev_alloc(&d_st, 1))) return cleanup(rc);
    const void *src = mat;
    if (hipMemsetAsync(d_st, 0, sizeof(IngestStats), h->stream) != hipSuccess ||
        hipMemcpyAsync(d_mat, src, sizeof(double) * cells,
                       opt->input_on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, h->stream) != hipSuccess)
        return cleanup(fail(MISSLAP_ERR_HIP, "copy of the dense input failed"));
    const int g4 = blocks_for(n_rows, 4);
    hipLaunchKernelGGL(k_dense_count, dim3(g4), dim3(256), 0, h->stream, d_mat, (int)n_rows, (int)n_cols, d_cnt);
    hipLaunchKernelGGL(k_dense_scan, dim3(1), dim3(1024), 0, h->stream, d_cnt, (int)n_rows, d_ptr, d_st);
    IngestStats st;
    if (hipMemcpyAsync(&st, d_st, sizeof(st), hipMemcpyDeviceToHost, h->stream) != hipSuccess ||
        hipStreamSynchronize(h->stream) != hipSuccess)
        return cleanup(fail(MISSLAP_ERR_HIP, "dense ingest failed: %s", hipGetErrorString(hipGetLastError())));
    // the valid entries are counted in 64 bits on the device: a 70 000 x 70 000 matrix fits the GPU but not an int32
    if (nnz_out) *nnz_out = (int64_t)st.dense_total;
    if ((st.err & kErrTooMany) || st.dense_total >= nnz_limit(opt))
        return cleanup(fail(MISSLAP_ERR_INVALID, "the matrix holds %lld valid entries; a solver handle takes fewer than %lld "
                            "(int32 row pointers)", (long long)st.dense_total, (long long)nnz_limit(opt)));
    const int total = (int)st.dense_total;
    if (total < n_rows)  // the caller raises the reference's ValueError (auction_.pyx:559-560)
        return cleanup(fail(MISSLAP_ERR_INVALID, "Fewer than %lld valid values provided for %lld rows.",
                            (long long)n_rows, (long long)n_rows));
    if (st.err & kErrRowGap)
        return cleanup(fail(MISSLAP_ERR_INVALID, "every row must have at least one valid (>= 0) entry"));
    h->nnz = total;
    if ((rc = dev_alloc(&d_loc, (size_t)total * 2))) return cleanup(rc);
    if ((rc = dev_alloc(&d_val, (size_t)total))) return cleanup(rc);
    hipLaunchKernelGGL(k_dense_fill, dim3(g4), dim3(256), 0, h->stream, d_mat, (int)n_rows, (int)n_cols, d_ptr,
                       d_loc, d_val);
    rc = build_from_device_coo(h, d_loc, d_val, (int)n_rows - 1, opt);
    if (rc) return cleanup(rc);
    cleanup(0);
    h->setup_ms = now_ms() - t0;
    *out = h;
    return MISSLAP_OK;
}

MISSLAP_API int misslap_destroy(misslap_solver *h) {
    free_all(h);
    return MISSLAP_OK;
}

MISSLAP_API int misslap_dims(const misslap_solver *h, int64_t *n_rows, int64_t *n_cols, int64_t *nnz) {
    if (!h) return fail(MISSLAP_ERR_INVALID, "null handle");
    if (n_rows) *n_rows = h->n_rows;
    if (n_cols) *n_cols = h->n_cols;
    if (nnz) *nnz = h->nnz;
    return MISSLAP_OK;
}

MISSLAP_API int misslap_set_stream(misslap_solver *h, void *hip_stream) {
    if (!h) return fail(MISSLAP_ERR_INVALID, "null handle");
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(hipStreamSynchronize(h->stream));
    if (h->own_stream) (void)hipStreamDestroy(h->stream);
    h->stream = (hipStream_t)hip_stream;
    h->own_stream = false;
    return MISSLAP_OK;
}

MISSLAP_API int misslap_exchange_buffers(misslap_solver *h, void **best_key, void **best_pos, int64_t *n_objects) {
    if (!h) return fail(MISSLAP_ERR_INVALID, "null handle");
    if (best_key) *best_key = h->best_key;
    if (best_pos) *best_pos = h->best_pos;
    if (n_objects) *n_objects = h->n_cols;
    return MISSLAP_OK;
}

MISSLAP_API int misslap_round_bid(misslap_solver *h) {
    if (!h) return fail(MISSLAP_ERR_INVALID, "null handle");
    HIP_TRY(hipSetDevice(h->device));
    return launch_bid(h);
}
MISSLAP_API int misslap_round_tiebreak(misslap_solver *h) {
    if (!h) return fail(MISSLAP_ERR_INVALID, "null handle");
    HIP_TRY(hipSetDevice(h->device));
    return launch_tiebreak(h);
}
MISSLAP_API int misslap_round_apply(misslap_solver *h) {
    if (!h) return fail(MISSLAP_ERR_INVALID, "null handle");
    HIP_TRY(hipSetDevice(h->device));
    return launch_apply(h);
}
MISSLAP_API int misslap_run_tail(misslap_solver *h) {
    if (!h) return fail(MISSLAP_ERR_INVALID, "null handle");
    HIP_TRY(hipSetDevice(h->device));
    return launch_tail(h);
}

MISSLAP_API int misslap_get_status(misslap_solver *h, misslap_status *st) {
    if (!h || !st) return fail(MISSLAP_ERR_INVALID, "null argument");
    HIP_TRY(hipSetDevice(h->device));
    int rc = read_ctl(h);
    st->error_bits = h->h_ctl->err;
    st->its = h->h_ctl->nits;
    st->K = h->h_ctl->K;
    st->nreductions = h->nreductions;
    st->eps = h->eps;
    st->target_eps = h->target_eps;
    st->finished = h->finished ? 1 : 0;
    st->tail_threshold = h->thr;
    st->rounds_per_sync = h->rounds_per_sync;
    st->shard_min_K = h->shard_min_K;
    return rc;
}

MISSLAP_API int misslap_check_ece(misslap_solver *h, float eps, int32_t *satisfied) {
    if (!h || !satisfied) return fail(MISSLAP_ERR_INVALID, "null argument");
    HIP_TRY(hipSetDevice(h->device));
    int ok = 0;
    int rc = run_ece(h, eps, &ok);
    *satisfied = ok;
    return rc;
}

// Loop control after the rounds of a phase, auction_.pyx:275-292.
MISSLAP_API int misslap_phase_end(misslap_solver *h, int32_t *finished) {
    if (!h) return fail(MISSLAP_ERR_INVALID, "null handle");
    HIP_TRY(hipSetDevice(h->device));
    int rc = read_status(h);  // (K, nits and the error bits are all this needs)
    if (rc) return rc;
    const Ctl &c = *h->h_ctl;
    if (c.nits >= h->max_iter) {  // terminate(), first clause (:308-309)
        h->finished = true;
    } else if (c.K == 0) {
        int ok = 0;
        if ((rc = run_ece(h, h->target_eps, &ok))) return rc;  // is_optimal() (:433-439)
        if (ok) {
            h->finished = true;
        } else if (h->eps < h->target_eps) {  // :280
            h->finished = true;
        } else {
            h->eps = h->eps * h->theta;  // :283 (fp32 product)
            MISSLAP_LAUNCH(h, k_reset_phase, (F_k_reset_phase), 256, dim3(blocks_for(h->n_rows > h->n_cols ? h->n_rows : h->n_cols, 256)),
                           dim3(256), h->ctl, h->p2o, h->o2p, h->rec, h->U, h->n_rows, h->n_cols);
            HIP_TRY(hipGetLastError());
            h->live_valid = false;  // (K was changed by a launch without a ticket; the mirror below is current)
            h->nreductions += 1;  // :292
            h->K_ub = h->n_rows;
            h->K_exact = true;  // k_reset_phase sets K = n_rows ...
            h->h_ctl->K = h->n_rows;  // ... and nothing else the mirror holds: it stays current (ctl_fresh)
            h->h_ctl->nholes = 0;
            h->h_ctl->nleft = 0;
            h->phase_fresh = true;
            h->ece_flag_clear = true;
            begin_phase(h);
        }
    }
    if (finished) *finished = h->finished ? 1 : 0;
    return MISSLAP_OK;
}

MISSLAP_API int misslap_finish(misslap_solver *h, int32_t *person_to_object_out, misslap_meta *meta_out) {
    if (!h) return fail(MISSLAP_ERR_INVALID, "null handle");
    HIP_TRY(hipSetDevice(h->device));
    int rc;
    if ((rc = read_ctl(h))) return rc;
    const bool complete = h->h_ctl->K == 0;  // eCE_satisfied is False while anybody is unassigned (auction_.pyx:446-447)
    h->ctl_fresh = false;
    // ONE pass over the rows for meta['eCE'] / soln_found (:297, :300), the objective (:302, :489-523) and the validity
    // flags of the assignment (benchmarking.py:56-64): all three look for the stored entry (i, sol[i])
    MISSLAP_LAUNCH_PLAIN(h, k_final_reset, dim3(1), dim3(1), 0, h->ctl);
    const FinalOut fo{1, h->maximize, h->o2p, h->contrib, h->nmatch, h->n_rows, h->n_cols, h->fin_slots};
    int n_slots = 0;
    // (a workgroup without rows returns before it writes its slot)
    HIP_TRY(stream_memset(h, h->fin_slots, 0, sizeof(FinSlot) * (size_t)h->fin_slots_n));
    if ((rc = launch_rows_all(h, h->target_eps, fo, &n_slots))) return rc;
    h->ece_flag_clear = false;  // the final pass leaves its verdict in Ctl::ece_fail: the next test must clear it
    if (h->f32) {
        EdgesF32 ed{h->edges32};
        MISSLAP_LAUNCH_PLAIN(h, k_obj_sum<EdgesF32>, dim3(1), dim3(1024), 0, h->ctl, ed, h->row_ptr, h->p2o,
                             h->n_rows, h->maximize, h->contrib, h->nmatch, h->fin_slots, n_slots);
    } else {
        EdgesF64 ed{h->col, h->val64};
        MISSLAP_LAUNCH_PLAIN(h, k_obj_sum<EdgesF64>, dim3(1), dim3(1024), 0, h->ctl, ed, h->row_ptr, h->p2o,
                             h->n_rows, h->maximize, h->contrib, h->nmatch, h->fin_slots, n_slots);
    }
    HIP_TRY(hipGetLastError());
    if (person_to_object_out)
        HIP_TRY(stream_memcpy(h, person_to_object_out, h->p2o, sizeof(int) * (size_t)h->n_rows, hipMemcpyDeviceToHost));
    // the bid kernels' statistics (a slot per workgroup, RoundArgs::wg_stats) -> the control block
    MISSLAP_LAUNCH_PLAIN(h, k_collect_stats, dim3(1), dim3(1024), 0, h->ctl, h->wg_stats, h->wg_stats_slots);
    HIP_TRY(hipGetLastError());
    h->ctl_fresh = false;
    if ((rc = read_ctl(h))) return rc;
    const int ece = complete && !h->h_ctl->ece_fail ? 1 : 0;
    if (!meta_out) return MISSLAP_OK;
    // how much the caller's struct holds: a version-1 caller (88-byte options at create) gets the version-1 layout
    size_t out_bytes = sizeof(misslap_meta_v1);
    if (h->abi >= 2) {
        const int32_t sz = meta_out->struct_size;
        if (sz < (int32_t)offsetof(misslap_meta, edges_scanned) || sz > 65536)
            return fail(MISSLAP_ERR_INVALID, "misslap_meta.struct_size = %d: set it to sizeof(misslap_meta) before the call", sz);
        out_bytes = std::min<size_t>((size_t)sz, sizeof(misslap_meta));
    }
    const Ctl &c = *h->h_ctl;
    misslap_meta full;
    misslap_meta *meta = &full;
    std::memset(meta, 0, sizeof(*meta));
    meta->struct_size = (int32_t)out_bytes;
    meta->abi_version = MISSLAP_ABI_VERSION;
    {
        const unsigned long long distinct = c.val_cnt[0], n_neg = c.val_cnt[1], n_big = c.val_cnt[2], n_invalid = c.val_cnt[3];
        const unsigned long long uniq = distinct + (n_neg ? 1ull : 0ull);  // np.unique counts -1 as one value
        meta->complete_assignment = (uniq == (unsigned long long)h->n_rows ? 1 : 0) | (n_neg == 0 ? 2 : 0) | (n_big == 0 ? 4 : 0);
        meta->valid_assignment = n_invalid == 0 ? 1 : 0;
        meta->lines_active = (h->cand != nullptr && !h->lines_dropped) ? 1 : 0;  // (at the END of the solve: see phases_with_lines)
    }
    meta->start_eps = h->start_eps;
    meta->final_eps = h->eps;
    meta->target_eps = h->target_eps;
    meta->eCE = ece;
    meta->soln_found = (c.K == 0) ? ece : 0;
    meta->nreductions = h->nreductions;
    meta->its = c.nits;
    meta->n_assigned = (int64_t)h->n_rows - c.K;
    meta->n_rows = h->n_rows;
    meta->n_cols = h->n_cols;
    meta->nnz = h->nnz;
    meta->obj_f64 = c.obj;
    meta->obj_f32 = (float)c.obj;  // get_obj returns a C float (:489)
    meta->setup_ms = h->setup_ms;
    meta->solve_ms = h->solve_ms;
    meta->edges_scanned = c.edges;
    meta->bids_made = c.bids;
    meta->grid_rounds = c.grid_rounds;
    meta->tail_rounds = c.tail_rounds;
    meta->tail_edges = c.tail_edges;
    meta->bytes_per_edge = h->f32 ? 8 : 12;
    meta->profiled = h->profile ? 1 : 0;
    meta->tiled_active = h->tiled_ok ? 1 : 0;
    meta->tiled_min_K = h->tiled_min_K;
    meta->shard_edges = c.shard_edges;
    for (int k = 0; k < 6; ++k) meta->tail_stats[k] = (double)c.dbg[k];  // tail: rounds / 10-ns ticks per mode
    for (int k = 0; k < 4; ++k) meta->tail_stats[6 + k] = (double)c.dbg[12 + k];  // bids, line hits, builds, hit edges
#if defined(MISSLAP_TAIL_STAMP) || defined(MISSLAP_TAIL_STAMP_SOLO) || defined(MISSLAP_TAIL_STAMP_TEAM) || \
    defined(MISSLAP_TAIL_STAMP_BLOCK) || defined(MISSLAP_TILED_STAMP) || defined(MISSLAP_TAIL_STAMP_DUO)
    for (int k = 0; k < 6; ++k) meta->tail_stats[k] = (double)c.dbg[6 + k];  // diagnostic build: solo-round segments (cycles)
#endif
    meta->cand_hits = c.cand_hits;
    meta->cand_edges = c.cand_edges;
    meta->sharded_rounds = h->sharded_rounds;
    meta->tiled_format = h->tiled_ok ? h->tiled_fmt : 0;
    meta->phases_with_lines = h->phases_with_lines;
    meta->eps_phases = h->phases_run;
    meta->filter_undecided = -1;  // (reserved: the slot of round 5's fp32-tile filter scans, removed in round 6)
    if (h->profile && h->prof_used) {
        std::vector<unsigned long long> le(2 * (size_t)h->launch_idx);
        if (h->launch_idx)
            HIP_TRY(hipMemcpy(le.data(), h->launch_edges, sizeof(unsigned long long) * le.size(),
                              hipMemcpyDeviceToHost));
        for (size_t k = 0; k < h->prof_used; ++k) {
            const ProfRec &r = h->prof[k];
            float ms = 0.f;
            if (hipEventElapsedTime(&ms, r.start, r.stop) != hipSuccess) continue;
            if (r.kind == 0 || r.kind == 2) {
                const unsigned long long e = le[2 * (size_t)r.launch_idx], eh = le[2 * (size_t)r.launch_idx + 1];
                if (r.kind == 0) {
                    meta->bid_launches += 1;  // self-skipped (no-op) launches included, like a kernel trace
                    meta->bid_ms += ms;
                    meta->bid_edges += e;
                    meta->bid_edges_read += e - eh;
                } else {
                    meta->tiled_launches += 1;
                    meta->tiled_ms += ms;
                    meta->tiled_edges += e;
                }
                if (r.fullscan) {
                    meta->fullscan_launches += 1;
                    meta->fullscan_ms += ms;
                    meta->fullscan_edges += e;
                    meta->fullscan_edges_read += e - (r.kind == 0 ? eh : 0ull);
                }
            } else {
                meta->tail_launches += 1;
                meta->tail_ms += ms;
            }
        }
    }
    if (h->abi >= 2) {
        std::memcpy(meta_out, &full, out_bytes);
    } else {  // version-1 layout: the same fields without the two leading words and the appended ones
        misslap_meta_v1 v1;
        static_assert(offsetof(misslap_meta, complete_assignment) - offsetof(misslap_meta, start_eps) == sizeof(misslap_meta_v1),
                      "misslap_meta = {struct_size, abi_version} + the version-1 fields + appended fields");
        std::memcpy(&v1, &full.start_eps, sizeof(v1));
        v1.merge_launches = 0;  // (version 2 re-uses these two slots: bid_edges_read / fullscan_edges_read)
        v1.merge_ms = 0.0;
        std::memcpy(meta_out, &v1, sizeof(v1));
    }
    return MISSLAP_OK;
}

// The handle's own round operations for the solve loop (host_comm.hpp: drive_sharded).
misslap_round_ops handle_round_ops(misslap_solver *h) {
    misslap_round_ops o;
    std::memset(&o, 0, sizeof(o));
    o.struct_size = (int32_t)sizeof(o);
    o.tail_threshold = h->thr;
    o.shard_min_K = h->shard_min_K;
    o.rounds_per_sync = h->rounds_per_sync_auto && !h->live_off ? kRoundsPerSyncLive : h->rounds_per_sync;
    o.large_round_K = kRoundSmallMax;
    o.rounds_per_sync_large = kRoundsPerSyncLargeK;
    o.max_iter = h->max_iter;
    o.ctx = h;
    o.status = [](void *x, int64_t *K, int64_t *its) {
        misslap_solver *s = static_cast<misslap_solver *>(x);
        const int rc = read_status(s);
        *K = s->h_ctl->K;
        *its = s->h_ctl->nits;
        return rc;
    };
    o.status_post = [](void *x, int32_t slot) { return status_enqueue(static_cast<misslap_solver *>(x), slot & 1); };
    o.status_take = [](void *x, int32_t slot, int64_t *K, int64_t *its) {
        misslap_solver *s = static_cast<misslap_solver *>(x);
        const int rc = status_wait(s, slot & 1);
        *K = s->h_stat[slot & 1].K;
        *its = s->h_stat[slot & 1].nits;
        return rc;
    };
    o.round_bid = [](void *x) { return launch_bid(static_cast<misslap_solver *>(x)); };
    o.round_tiebreak = [](void *x) { return launch_tiebreak(static_cast<misslap_solver *>(x)); };
    o.round_apply = [](void *x) { return launch_apply(static_cast<misslap_solver *>(x)); };
    o.run_tail = [](void *x) { return launch_tail(static_cast<misslap_solver *>(x)); };
    o.phase_end = [](void *x, int32_t *fin) { return misslap_phase_end(static_cast<misslap_solver *>(x), fin); };
    o.best_key = h->best_key;
    o.best_pos = h->best_pos;
    o.n_objects = h->n_cols;
    o.stream = h->stream;
    return o;
}

// AuctionSolver.solve(), auction_.pyx:268-306: the loop of host_comm.hpp with no communicator (the rounds at or above
// the full-scan threshold are issued one at a time, K known exactly -- neither bid kernel is then launched for a round
// the other one takes; the others in batches).
MISSLAP_API int misslap_solve(misslap_solver *h, int32_t *person_to_object_out, misslap_meta *meta) {
    if (!h) return fail(MISSLAP_ERR_INVALID, "null handle");
    if (h->world != 1) return fail(MISSLAP_ERR_STATE, "misslap_solve drives one GPU; sharded handles use misslap_solve_sharded");
    return misslap_solve_sharded(h, nullptr, person_to_object_out, meta);
}
// (the sharded solve is defined below: misslap_solve calls it)
#include "abi_comm.hpp"
#include "abi_batch.hpp"

MISSLAP_API int misslap_get_state(misslap_solver *h, double *prices, int32_t *unassigned, int32_t *person_to_object,
                                  int32_t *object_to_person) {
    if (!h) return fail(MISSLAP_ERR_INVALID, "null handle");
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(hipStreamSynchronize(h->stream));
    if (prices) HIP_TRY(hipMemcpy(prices, h->price, sizeof(double) * (size_t)h->n_cols, hipMemcpyDeviceToHost));
    if (unassigned) HIP_TRY(hipMemcpy(unassigned, h->U, sizeof(int) * (size_t)h->n_rows, hipMemcpyDeviceToHost));
    if (person_to_object)
        HIP_TRY(hipMemcpy(person_to_object, h->p2o, sizeof(int) * (size_t)h->n_rows, hipMemcpyDeviceToHost));
    if (object_to_person)
        HIP_TRY(hipMemcpy(object_to_person, h->o2p, sizeof(int) * (size_t)h->n_cols, hipMemcpyDeviceToHost));
    return MISSLAP_OK;
}
#include "abi_diag.hpp"
