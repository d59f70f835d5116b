// abi_matching.hpp -- C ABI: the feasibility guard (host Hopcroft-Karp and the GPU matcher).
// (part of the single translation unit misslap.hip; included in the order given there)
#pragma once

// Feasibility guard (host side by design, like the reference's): see host_matching.hpp.
MISSLAP_API int misslap_hopcroft_karp(const int32_t *loc, int64_t nnz, int32_t n_rows, int32_t n_cols,
                                      int32_t *size, int32_t *left_pairings, int32_t *right_pairings) {
    if (!size || nnz < 0 || n_rows < 0 || n_cols < 0 || (nnz > 0 && !loc))
        return fail(MISSLAP_ERR_INVALID, "bad argument");
    for (int64_t k = 0; k < nnz; ++k) {
        const int32_t i = loc[2 * k], j = loc[2 * k + 1];
        if (i < 0 || i >= n_rows || j < 0 || j >= n_cols)
            return fail(MISSLAP_ERR_INVALID, "loc entry %lld = (%d, %d) outside %d x %d", (long long)k, i, j, n_rows, n_cols);
        if (k && i < loc[2 * (k - 1)]) return fail(MISSLAP_ERR_INVALID, "loc rows must be sorted in ascending order");
    }
    try {
        HopcroftKarp hk(loc, nnz, n_rows, n_cols);
        *size = hk.solve();
        if (left_pairings) std::copy(hk.pair_u.begin(), hk.pair_u.end(), left_pairings);
        if (right_pairings) std::copy(hk.pair_v.begin(), hk.pair_v.end(), right_pairings);
    } catch (const std::bad_alloc &) {
        return fail(MISSLAP_ERR_HIP, "out of host memory in misslap_hopcroft_karp");
    }
    return MISSLAP_OK;
}
// The same guard on the GPU (kernels_matching.hpp): BFS-layered maximum matching; the cardinality equals the host
// version's (and the reference's), the pairings are a maximum matching but not necessarily the same one.
// Greedy start + phases of the GPU matcher (kernels_matching.hpp) on a CSR already in device memory; the matched-row
// count is left in a.counters[2].
// The matcher augments ONE path per BFS tree and phase, so its set of augmenting paths is not maximal and the
// O(sqrt n) phase bound of Hopcroft-Karp does not hold; every BFS layer costs a launch and a status read.  Chain-like
// graphs could need O(n) layers times many phases: the phases / layers are budgeted, and when the budget runs out
// *gave_up is set -- the caller finishes with the host matcher seeded by the matching found so far.
static int run_matching_phases(hipStream_t st, const MatchArgs &a, int *nph_out, bool *gave_up) {
    const int n_rows = a.n_rows, n_cols = a.n_cols;
    const long long root_n = (long long)std::sqrt((double)std::max(n_rows, 1)) + 1;
    const long long max_phases = 4 * root_n + 64, max_layers = std::max<long long>(2048, 64 * root_n);
    long long layers = 0;
    *gave_up = false;
    long long layer_budget = max_layers;
    if (const char *e = std::getenv("MISSLAP_MATCHING_MAX_LAYERS")) layer_budget = std::atoll(e);  // (tests of the fallback)
    const int gV = blocks_for(std::max(n_rows, n_cols), 256), gW = blocks_for(n_rows, 4);
    hipLaunchKernelGGL(k_m_init, dim3(gV), dim3(256), 0, st, a);
    hipLaunchKernelGGL(k_m_greedy, dim3(gV), dim3(256), 0, st, a);
    HIP_TRY(hipGetLastError());
    int nph = 0;
    for (;;) {  // phases (:199-211)
        HIP_TRY(hipMemsetAsync(a.counters, 0, 4 * sizeof(int), st));
        hipLaunchKernelGGL(k_m_phase_init, dim3(gV), dim3(256), 0, st, a);
        int cnt[4] = {0, 0, 0, 0};
        bool augmented = false;
        for (int L = 0; L <= n_rows; ++L) {
            if (++layers > layer_budget || nph >= max_phases) {
                *gave_up = true;
                break;
            }
            HIP_TRY(hipMemsetAsync(a.counters + 1, 0, sizeof(int), st));
            hipLaunchKernelGGL(k_m_bfs_layer, dim3(gW), dim3(256), 0, st, a, L);
            HIP_TRY(hipMemcpyAsync(cnt, a.counters, sizeof(cnt), hipMemcpyDeviceToHost, st));
            HIP_TRY(hipStreamSynchronize(st));
            if (cnt[0] > 0) {  // this layer reached free columns: flip one shortest path per tree
                hipLaunchKernelGGL(k_m_augment, dim3(blocks_for(n_rows, 256)), dim3(256), 0, st, a);
                augmented = true;
                break;
            }
            if (cnt[1] == 0) break;  // the layering is exhausted: no augmenting path is left
        }
        HIP_TRY(hipGetLastError());
        if (!augmented || *gave_up) break;
        ++nph;
    }
    HIP_TRY(hipMemsetAsync(a.counters + 2, 0, sizeof(int), st));
    hipLaunchKernelGGL(k_m_count, dim3(blocks_for(n_rows, 256)), dim3(256), 0, st, a);
    *nph_out = nph;
    return MISSLAP_OK;
}

MISSLAP_API int misslap_matching_gpu(const int32_t *loc, int64_t nnz, int32_t n_rows, int32_t n_cols, int32_t device,
                                     int32_t *size, int32_t *left_pairings, int32_t *right_pairings, int32_t *phases) {
    if (!size || nnz < 0 || n_rows < 0 || n_cols < 0 || (nnz > 0 && !loc)) return fail(MISSLAP_ERR_INVALID, "bad argument");
    if (nnz >= (int64_t)0x7fffffff) return fail(MISSLAP_ERR_INVALID, "nnz must be < 2^31 (int32 row pointers)");
    for (int64_t k = 0; k < nnz; ++k) {
        const int32_t i = loc[2 * k], j = loc[2 * k + 1];
        if (i < 0 || i >= n_rows || j < 0 || j >= n_cols)
            return fail(MISSLAP_ERR_INVALID, "loc entry %lld = (%d, %d) outside %d x %d", (long long)k, i, j, n_rows, n_cols);
        if (k && i < loc[2 * (k - 1)]) return fail(MISSLAP_ERR_INVALID, "loc rows must be sorted in ascending order");
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(MISSLAP_ERR_NO_DEVICE, "no HIP device available: misslap_matching_gpu has no CPU fallback "
                    "(misslap_hopcroft_karp is the host matcher)");
    if (device < 0 || device >= ndev) return fail(MISSLAP_ERR_INVALID, "device %d out of range", device);
    HIP_TRY(hipSetDevice(device));
    *size = 0;
    if (phases) *phases = 0;
    if (left_pairings) std::fill(left_pairings, left_pairings + n_rows, -1);
    if (right_pairings) std::fill(right_pairings, right_pairings + n_cols, -1);
    if (nnz == 0 || n_rows == 0) return MISSLAP_OK;
    DevScratch tmp;
    int rc;
    int *d_loc = nullptr, *d_err = nullptr;
    MatchArgs a{};
    a.n_rows = n_rows;
    a.n_cols = n_cols;
    int *row_ptr = nullptr, *col = nullptr;
    if ((rc = tmp.alloc(&d_loc, (size_t)nnz * 2))) return rc;
    if ((rc = tmp.alloc(&row_ptr, (size_t)n_rows + 1))) return rc;
    if ((rc = tmp.alloc(&col, (size_t)nnz))) return rc;
    if ((rc = tmp.alloc(&a.match_row, (size_t)n_rows))) return rc;
    if ((rc = tmp.alloc(&a.match_col, (size_t)n_cols))) return rc;
    if ((rc = tmp.alloc(&a.level, (size_t)n_rows))) return rc;
    if ((rc = tmp.alloc(&a.root, (size_t)n_rows))) return rc;
    if ((rc = tmp.alloc(&a.pred_col, (size_t)n_cols))) return rc;
    if ((rc = tmp.alloc(&a.end_of_root, (size_t)n_rows))) return rc;
    if ((rc = tmp.alloc(&a.counters, 4))) return rc;
    if ((rc = tmp.alloc(&d_err, 1))) return rc;
    a.row_ptr = row_ptr;
    a.col = col;
    a.col_stride = 1;
    hipStream_t st = nullptr;  // the default stream: this entry point is synchronous
    HIP_TRY(hipMemcpyAsync(d_loc, loc, sizeof(int) * 2 * (size_t)nnz, hipMemcpyHostToDevice, st));
    HIP_TRY(hipMemsetAsync(d_err, 0, sizeof(int), st));
    const int gE = blocks_for(nnz, 256 * 4);
    hipLaunchKernelGGL(k_m_row_ptr, dim3(gE), dim3(256), 0, st, d_loc, (long long)nnz, n_rows, row_ptr, col, d_err);
    int nph = 0;
    bool gave_up = false;
    if ((rc = run_matching_phases(st, a, &nph, &gave_up))) return rc;
    int out[4] = {0, 0, 0, 0}, err = 0;
    std::vector<int> mr, mc;
    if (gave_up) {
        mr.resize((size_t)n_rows);
        mc.resize((size_t)n_cols);
    }
    int *lp = gave_up ? mr.data() : left_pairings, *rp = gave_up ? mc.data() : right_pairings;
    HIP_TRY(hipMemcpyAsync(out, a.counters, sizeof(out), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipMemcpyAsync(&err, d_err, sizeof(int), hipMemcpyDeviceToHost, st));
    if (lp) HIP_TRY(hipMemcpyAsync(lp, a.match_row, sizeof(int) * (size_t)n_rows, hipMemcpyDeviceToHost, st));
    if (rp) HIP_TRY(hipMemcpyAsync(rp, a.match_col, sizeof(int) * (size_t)n_cols, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    tmp.drained = true;
    if (err) return fail(MISSLAP_ERR_INVALID, "loc rows must be sorted in ascending order");
    *size = out[2];
    if (phases) *phases = nph;
    if (gave_up) {  // finish on the host from the matching found so far (same cardinality: both are maximum)
        try {
            HopcroftKarp hk(loc, nnz, n_rows, n_cols);
            hk.seed(mr.data(), mc.data());
            *size = hk.solve();
            if (left_pairings) std::copy(hk.pair_u.begin(), hk.pair_u.end(), left_pairings);
            if (right_pairings) std::copy(hk.pair_v.begin(), hk.pair_v.end(), right_pairings);
        } catch (const std::bad_alloc &) {
            return fail(MISSLAP_ERR_HIP, "out of host memory in misslap_matching_gpu");
        }
    }
    return MISSLAP_OK;
}

// The same matcher on the graph a solver handle already holds in device memory (its CSR): no host copy of the
// entries, no second upload -- what the front-end's feasibility guard uses after it has created the handle.
MISSLAP_API int misslap_matching_of(misslap_solver *h, int32_t *size, int32_t *phases) {
    if (!h || !size) return fail(MISSLAP_ERR_INVALID, "bad argument");
    HIP_TRY(hipSetDevice(h->device));
    DevScratch tmp;
    int rc;
    MatchArgs a{};
    a.n_rows = h->n_rows;
    a.n_cols = h->n_cols;
    a.row_ptr = h->row_ptr;
    a.col = h->f32 ? reinterpret_cast<const int *>(h->edges32) : h->col;
    a.col_stride = h->f32 ? 2 : 1;
    {
        DevBlock blk;
        blk.want(&a.match_row, (size_t)h->n_rows);
        blk.want(&a.match_col, (size_t)h->n_cols);
        blk.want(&a.level, (size_t)h->n_rows);
        blk.want(&a.root, (size_t)h->n_rows);
        blk.want(&a.pred_col, (size_t)h->n_cols);
        blk.want(&a.end_of_root, (size_t)h->n_rows);
        blk.want(&a.counters, 4);
        tmp.blks.emplace_back();
        if ((rc = blk.commit(&tmp.blks.back()))) return rc;
    }
    hipStream_t st = h->stream;
    int nph = 0;
    bool gave_up = false;
    if ((rc = run_matching_phases(st, a, &nph, &gave_up))) return rc;
    int out[4] = {0, 0, 0, 0};
    HIP_TRY(hipMemcpyAsync(out, a.counters, sizeof(out), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    *size = out[2];
    if (phases) *phases = nph;
    if (gave_up) {  // the budget ran out: the handle's CSR and the matching so far go to the host matcher
        try {
            const size_t stride = h->f32 ? 2 : 1;
            std::vector<int> rp((size_t)h->n_rows + 1), cols((size_t)h->nnz * stride), mr((size_t)h->n_rows), mc((size_t)h->n_cols);
            HIP_TRY(hipMemcpy(rp.data(), h->row_ptr, sizeof(int) * rp.size(), hipMemcpyDeviceToHost));
            HIP_TRY(hipMemcpy(cols.data(), a.col, sizeof(int) * cols.size(), hipMemcpyDeviceToHost));
            HIP_TRY(hipMemcpy(mr.data(), a.match_row, sizeof(int) * mr.size(), hipMemcpyDeviceToHost));
            HIP_TRY(hipMemcpy(mc.data(), a.match_col, sizeof(int) * mc.size(), hipMemcpyDeviceToHost));
            rp[(size_t)h->n_rows] = (int)h->nnz;  // (the device array's last entry is written by the ingest as well)
            HopcroftKarp hk(rp.data(), cols.data(), (int)stride, h->n_rows, h->n_cols);
            hk.seed(mr.data(), mc.data());
            *size = hk.solve();
        } catch (const std::bad_alloc &) {
            return fail(MISSLAP_ERR_HIP, "out of host memory in misslap_matching_of");
        }
    }
    tmp.drained = true;
    return MISSLAP_OK;
}
