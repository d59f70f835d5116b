// abi_diag.hpp -- C ABI of libmisslap_diag.so only (-DMISSLAP_DIAG): timed launches of ablated bid kernels.
// (part of the single translation unit misslap.hip; included in the order given there)
#pragma once

#ifdef MISSLAP_DIAG  // built into libmisslap_diag.so only (python -m sslap_amd.build diag), for tools/
// Diagnostics: average duration (ms) of `reps` launches of an ablated full-scan bid kernel over the
// current unassigned list (K == n_rows right after create).  mode: 0 complete, 1 no price gather,
// 2 no cross-lane reduction, 3 edge stream only.  Results are discarded; solver state is untouched.
namespace {
__global__ __launch_bounds__(256) void k_flush_read(const uint4 *src, size_t n16, unsigned *sink) {  // plain loads: the lines stay cached, clean
    unsigned acc = 0;
    for (size_t k = (size_t)blockIdx.x * 256 + threadIdx.x; k < n16; k += (size_t)gridDim.x * 256) {
        const uint4 v = src[k];
        acc ^= v.x ^ v.y ^ v.z ^ v.w;
    }
    if (acc == 0x9e3779b9u) *sink = acc;
}
}  // namespace

MISSLAP_API int misslap_debug_time_bid(misslap_solver *h, int32_t mode, int32_t reps, float *ms_avg) {
    if (!h || !ms_avg || reps <= 0) return fail(MISSLAP_ERR_INVALID, "bad argument");
    if (!h->f32) return fail(MISSLAP_ERR_STATE, "ablation kernels are instantiated for the 8 B/edge layout only");
    HIP_TRY(hipSetDevice(h->device));
    // mode | 0x100: COLD -- 1 GiB of other data is written between the launches and every launch is timed on its own (a
    // full scan's 260 MB of edges + tables fit the 256 MB of MALL: back to back they never come from HBM)
    const bool cold = (mode & 0x100) != 0;
    mode &= 0xff;
    if (mode >= 10) {  // LDS-tiled kernel: 10 complete, 11 no fill, 12 no arithmetic, 13 no edge loads ... (shape 3: one
                       // loader); 20 / 21 / 22 / 27: complete / no fill / no arithmetic / half fill in the PRODUCTION shape 0
        if (!h->tiled_ok || (mode < 20 && h->tiled_shape != 3 && mode != 10) || (mode >= 20 && h->tiled_shape != 0))
            return fail(MISSLAP_ERR_STATE, "tiled ablations 11-17 need tiled_shape 3, 20-27 tiled_shape 0");
        const hipFuncAttribute at = hipFuncAttributeMaxDynamicSharedMemorySize;
        const int ldsb = (int)tiled_lds_bytes(kTiledShapes[h->tiled_shape][4]);
        HIP_TRY(hipFuncSetAttribute((const void *)k_bid_tiled<1024, 4, 2, 2, kTileColsHalf, 1, 1>, at, ldsb));
        HIP_TRY(hipFuncSetAttribute((const void *)k_bid_tiled<1024, 4, 2, 2, kTileColsHalf, 1, 2>, at, ldsb));
        HIP_TRY(hipFuncSetAttribute((const void *)k_bid_tiled<1024, 4, 2, 2, kTileColsHalf, 1, 3>, at, ldsb));
        HIP_TRY(hipFuncSetAttribute((const void *)k_bid_tiled<1024, 4, 2, 2, kTileColsHalf, 1, 4>, at, ldsb));
        HIP_TRY(hipFuncSetAttribute((const void *)k_bid_tiled<1024, 4, 2, 2, kTileColsHalf, 1, 5>, at, ldsb));
        HIP_TRY(hipFuncSetAttribute((const void *)k_bid_tiled<1024, 4, 2, 2, kTileColsHalf, 1, 6>, at, ldsb));
        HIP_TRY(hipFuncSetAttribute((const void *)k_bid_tiled<1024, 4, 2, 2, kTileColsHalf, 1, 7>, at, ldsb));
        HIP_TRY(hipFuncSetAttribute((const void *)k_bid_tiled<1024, 4, 2, 2, kTileColsHalf, 3, 1>, at, ldsb));
        HIP_TRY(hipFuncSetAttribute((const void *)k_bid_tiled<1024, 4, 2, 2, kTileColsHalf, 3, 2>, at, ldsb));
        HIP_TRY(hipFuncSetAttribute((const void *)k_bid_tiled<1024, 4, 2, 2, kTileColsHalf, 3, 7>, at, ldsb));
        RoundArgs a = round_args(h);
        a.launch_edges = nullptr;
        TiledArgs ta{h->tiled, h->tcol, h->seg4, h->T, 1, h->n_tiled,
                     nullptr, nullptr, h->ovf_ptr, h->ovf_q, h->ovf_cap, h->part_vw, h->part_g, h->n_rows, h->split_cnt, FinalOut{}};
        hipEvent_t t0, t1;
        HIP_TRY(hipEventCreate(&t0));
        HIP_TRY(hipEventCreate(&t1));
        auto launch_t = [&]() {
            const dim3 g(256), b(1024);
            switch (mode) {
                case 10:  // the product kernel in the handle's launch shape
                    switch (h->tiled_shape) {
#define X(I, TH, R, B, D, TC, LD, GL, CS) \
    case I: hipLaunchKernelGGL((k_bid_tiled<TH, R, B, D, TC, LD, 0, GL, CS>), g, dim3(TH), ldsb, h->stream, a, ta); break;
                        MISSLAP_FOR_TILED_SHAPES(X)
#undef X
                    }
                    break;
                case 11: hipLaunchKernelGGL((k_bid_tiled<1024, 4, 2, 2, kTileColsHalf, 1, 1>), g, b, ldsb, h->stream, a, ta); break;
                case 12: hipLaunchKernelGGL((k_bid_tiled<1024, 4, 2, 2, kTileColsHalf, 1, 2>), g, b, ldsb, h->stream, a, ta); break;
                case 14: hipLaunchKernelGGL((k_bid_tiled<1024, 4, 2, 2, kTileColsHalf, 1, 4>), g, b, ldsb, h->stream, a, ta); break;
                case 15: hipLaunchKernelGGL((k_bid_tiled<1024, 4, 2, 2, kTileColsHalf, 1, 5>), g, b, ldsb, h->stream, a, ta); break;
                case 16: hipLaunchKernelGGL((k_bid_tiled<1024, 4, 2, 2, kTileColsHalf, 1, 6>), g, b, ldsb, h->stream, a, ta); break;
                case 17: hipLaunchKernelGGL((k_bid_tiled<1024, 4, 2, 2, kTileColsHalf, 1, 7>), g, b, ldsb, h->stream, a, ta); break;
                case 20: hipLaunchKernelGGL((k_bid_tiled<1024, 4, 2, 2, kTileColsHalf, 3, 0>), g, b, ldsb, h->stream, a, ta); break;
                case 21: hipLaunchKernelGGL((k_bid_tiled<1024, 4, 2, 2, kTileColsHalf, 3, 1>), g, b, ldsb, h->stream, a, ta); break;
                case 22: hipLaunchKernelGGL((k_bid_tiled<1024, 4, 2, 2, kTileColsHalf, 3, 2>), g, b, ldsb, h->stream, a, ta); break;
                case 27: hipLaunchKernelGGL((k_bid_tiled<1024, 4, 2, 2, kTileColsHalf, 3, 7>), g, b, ldsb, h->stream, a, ta); break;
                default: hipLaunchKernelGGL((k_bid_tiled<1024, 4, 2, 2, kTileColsHalf, 1, 3>), g, b, ldsb, h->stream, a, ta); break;
            }
        };
        launch_t();
        HIP_TRY(hipGetLastError());
        float ms = 0.f;
        if (cold) {
            // (evicted by READS of other data: a memset would leave 256 MB of dirty lines whose write-back the timed scan
            // would pay for)
            void *flush = nullptr;
            const size_t flush_bytes = (size_t)1 << 30;
            HIP_TRY(hipMalloc(&flush, flush_bytes + 64));
            HIP_TRY(hipMemsetAsync(flush, 0, flush_bytes + 64, h->stream));
            HIP_TRY(hipStreamSynchronize(h->stream));
            for (int r = 0; r < reps; ++r) {
                hipLaunchKernelGGL(k_flush_read, dim3(2048), dim3(256), 0, h->stream, (const uint4 *)flush, flush_bytes / 16,
                                   (unsigned *)((char *)flush + flush_bytes));
                HIP_TRY(hipEventRecord(t0, h->stream));
                launch_t();
                HIP_TRY(hipEventRecord(t1, h->stream));
                HIP_TRY(hipEventSynchronize(t1));
                float one = 0.f;
                HIP_TRY(hipEventElapsedTime(&one, t0, t1));
                ms += one;
            }
            (void)hipFree(flush);
        } else {
            HIP_TRY(hipEventRecord(t0, h->stream));
            for (int r = 0; r < reps; ++r) launch_t();
            HIP_TRY(hipGetLastError());
            HIP_TRY(hipEventRecord(t1, h->stream));
            HIP_TRY(hipEventSynchronize(t1));
            HIP_TRY(hipEventElapsedTime(&ms, t0, t1));
        }
        *ms_avg = ms / (float)reps;
        (void)hipEventDestroy(t0);
        (void)hipEventDestroy(t1);
        // the launches polluted the per-object maxima: restore the "no bid" state
        HIP_TRY(hipMemsetAsync(h->best_key, 0, sizeof(unsigned long long) * (size_t)h->n_cols, h->stream));
        HIP_TRY(hipStreamSynchronize(h->stream));
        return MISSLAP_OK;
    }
    hipEvent_t e0, e1;
    HIP_TRY(hipEventCreate(&e0));
    HIP_TRY(hipEventCreate(&e1));
    EdgesF32 ed{h->edges32};
    const int grid = blocks_for(h->n_rows, 4);
    unsigned long long *sink = h->bid_key;
    auto launch = [&]() {
        switch (mode) {
            case 0: hipLaunchKernelGGL((k_bid_ablate<EdgesF32, 0>), dim3(grid), dim3(256), 0, h->stream, h->U, h->row_ptr, h->price, ed, h->n_rows, (double)h->eps, sink); break;
            case 1: hipLaunchKernelGGL((k_bid_ablate<EdgesF32, 1>), dim3(grid), dim3(256), 0, h->stream, h->U, h->row_ptr, h->price, ed, h->n_rows, (double)h->eps, sink); break;
            case 2: hipLaunchKernelGGL((k_bid_ablate<EdgesF32, 2>), dim3(grid), dim3(256), 0, h->stream, h->U, h->row_ptr, h->price, ed, h->n_rows, (double)h->eps, sink); break;
            case 4: hipLaunchKernelGGL((k_bid_ablate<EdgesF32, 4>), dim3(grid), dim3(256), 0, h->stream, h->U, h->row_ptr, h->price, ed, h->n_rows, (double)h->eps, sink); break;
            default: hipLaunchKernelGGL((k_bid_ablate<EdgesF32, 3>), dim3(grid), dim3(256), 0, h->stream, h->U, h->row_ptr, h->price, ed, h->n_rows, (double)h->eps, sink); break;
        }
    };
    launch();  // warm-up
    HIP_TRY(hipEventRecord(e0, h->stream));
    for (int r = 0; r < reps; ++r) launch();
    HIP_TRY(hipEventRecord(e1, h->stream));
    HIP_TRY(hipEventSynchronize(e1));
    float ms = 0.f;
    HIP_TRY(hipEventElapsedTime(&ms, e0, e1));
    *ms_avg = ms / (float)reps;
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    return MISSLAP_OK;
}
#endif  // MISSLAP_DIAG
