// host_create.hpp -- handle construction and destruction: option normalisation, CSR build and both edge layouts on the device, the tile-major copy, state blocks, the eps schedule (auction_.pyx:202-265).
// (part of the single translation unit misslap.hip; included in the order given there)
#pragma once

namespace {
void free_all(misslap_solver *h) {
    if (!h) return;
    (void)hipSetDevice(h->device);
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    for (const Blk &b : h->blocks) block_free(h->device, b.p, b.bytes);
    for (auto &r : h->prof) {
        (void)hipEventDestroy(r.start);
        (void)hipEventDestroy(r.stop);
    }
    HostRes res;
    res.stream = h->own_stream ? h->stream : nullptr;
    res.h_ctl = h->h_ctl;
    res.ev[0] = h->stat_ev[0];
    res.ev[1] = h->stat_ev[1];
    const bool whole = res.stream && res.h_ctl && res.ev[0] && res.ev[1];
    if (!whole || !host_pool().park(h->device, res)) {
        if (res.h_ctl) (void)hipHostFree(res.h_ctl);
        for (hipEvent_t e : res.ev)
            if (e) (void)hipEventDestroy(e);
        if (res.stream) (void)hipStreamDestroy(res.stream);
    }
    delete h;
}

// Shared tail of the two constructors: d_loc / d_val are device-resident COO arrays.
int build_from_device_coo(misslap_solver *h, const int *d_loc, const double *d_val, int last_row,
                          const misslap_options *opt) {
    const int64_t nnz = h->nnz;
    if (last_row < 0) return fail(MISSLAP_ERR_INVALID, "negative row index");
    h->n_rows = last_row + 1;  // auction_.pyx:209 (rows are ascending, so the last one is the maximum)
    int rc;
    DevScratch tmp;  // every temporary below: released on every return path
    CreateTrace trace(h->stream);
    IngestStats *d_st = nullptr;
    if ((rc = tmp.alloc(&d_st, 1))) return rc;
    HIP_TRY(hipMemsetAsync(d_st, 0, sizeof(IngestStats), h->stream));
    {
        const int init = -1;
        HIP_TRY(hipMemcpyAsync(&d_st->max_col, &init, sizeof(int), hipMemcpyHostToDevice, h->stream));
        HIP_TRY(hipStreamSynchronize(h->stream));  // `init` lives on this stack frame
    }
    {
        DevBlock blk;
        blk.want(&h->row_ptr, (size_t)h->n_rows + 1);
        h->blocks.emplace_back();
        if ((rc = blk.commit(&h->blocks.back()))) return rc;
    }
    const int grid = blocks_for(nnz, 256 * 8);
    hipLaunchKernelGGL(k_ingest_rows, dim3(grid), dim3(256), 0, h->stream, d_loc, (long long)nnz, h->n_rows,
                       h->row_ptr, d_st);
    hipLaunchKernelGGL(k_ingest_vals, dim3(grid), dim3(256), 0, h->stream, d_val, (long long)nnz, d_st);
    hipLaunchKernelGGL(k_max_row_len, dim3(blocks_for(h->n_rows, 256)), dim3(256), 0, h->stream, h->row_ptr, h->n_rows, d_st);
    IngestStats st;
    HIP_TRY(hipMemcpyAsync(&st, d_st, sizeof(st), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    trace.stage("ingest rows / values");
    if (st.err & kErrColNegative) return fail(MISSLAP_ERR_INVALID, "loc holds a negative row or column index");
    if (st.err & kErrRowsUnsorted)
        return fail(MISSLAP_ERR_INVALID, "loc rows must be sorted in ascending order (auction_.pyx:33-48 contract)");
    if (st.err & kErrRowGap)
        return fail(MISSLAP_ERR_INVALID,
                    "every row 0..N-1 must have at least one entry (auction_.pyx:33-48 contract)");
    if (st.err & kErrNonFinite) return fail(MISSLAP_ERR_INVALID, "val holds a NaN or an infinity");
    if (st.max_col >= 0x7ffffffe) return fail(MISSLAP_ERR_INVALID, "column index too large (max + 1 must fit an int32)");
    h->n_cols = st.max_col + 1;  // auction_.pyx:210
    h->f32 = !st.not_f32 && !opt->force_f64_values;
    // (candidate lines and eps: see misslap_solver::lines_safe_eps; the decision is taken per eps-phase, begin_phase)
    int cand_mode = opt->cand_mode;
    {
        double max_abs_d;
        const long long b = (long long)st.max_abs_bits;
        std::memcpy(&max_abs_d, &b, sizeof(double));
        h->lines_safe_eps = max_abs_d * 0x1p-44;  // < 2^9 ulps of the largest cost
    }
    // Lines for long rows (k_refresh_long) pay where a row scan is long: dense 8000^2 1.79 -> 0.60 s.  At a few
    // hundred edges per row the pass costs more than the scans it saves (C4, 300 edges per row, 176 rounds: 13.4 ->
    // 18.3 ms), so it runs from kLongRowsFrom edges per row on average.
    const long long avg_row = nnz / h->n_rows;
    h->avg_row_len = avg_row;
    h->max_row_len = st.max_row_len;
    h->long_rows = st.max_row_len > kCandRowMax && avg_row >= (long long)kLongRowsFrom && avg_row <= kCandLongMax;
    // ... below that (C4's 300 edges per row, a dense 600^2) only once the solve has shown that its tail is long:
    // launch_tail switches the builder on after max(100, n_rows / 64) tail rounds
    // (... and so do the long rows of a handle whose AVERAGE row keeps a line, where they are many: 40 000 rows of 256
    // edges on average, half of them longer: 208 -> 119 ms per solve; a few stragglers -- C3 has rows of 260 edges -- are left
    // to their scans, a pass over all rows every few hundred tail rounds costs more than they do)
    const bool many_long = (long long)st.long_rows * 100 >= (long long)kLongRowsMixedPercent * h->n_rows;
    h->long_rows_later = !h->long_rows && (avg_row > kCandRowMax || (st.max_row_len > kCandRowMax && many_long)) &&
                         avg_row <= kCandLongMax && cand_mode != 1;
    if (h->thr < 0) {  // library default: by whether the persons will have candidate lines (rows of <= 256 edges)
        const bool lines = cand_mode != 1 && (avg_row <= kCandRowMax || h->long_rows);
        h->thr = lines ? kDefaultTailThreshold : kDefaultTailThresholdNoLines;
    }
    const int flip = h->maximize ? 0 : 1;
    if (h->f32) {
        DevBlock blk;
        blk.want(&h->edges32, (size_t)nnz + 4 * kWave);  // the tail kernel reads up to 256 entries past a row start
        h->blocks.emplace_back();
        if ((rc = blk.commit(&h->blocks.back()))) return rc;
        HIP_TRY(hipMemsetAsync(h->edges32 + nnz, 0, sizeof(int2) * 4 * kWave, h->stream));
        hipLaunchKernelGGL(k_build_edges_f32, dim3(grid), dim3(256), 0, h->stream, d_loc, d_val, (long long)nnz,
                           flip, h->edges32);
    } else {
        DevBlock blk;
        blk.want(&h->col, (size_t)nnz + 4 * kWave);
        blk.want(&h->val64, (size_t)nnz + 4 * kWave);
        h->blocks.emplace_back();
        if ((rc = blk.commit(&h->blocks.back()))) return rc;
        HIP_TRY(hipMemsetAsync(h->col + nnz, 0, sizeof(int) * 4 * kWave, h->stream));
        HIP_TRY(hipMemsetAsync(h->val64 + nnz, 0, sizeof(double) * 4 * kWave, h->stream));
        hipLaunchKernelGGL(k_build_edges_f64, dim3(grid), dim3(256), 0, h->stream, d_loc, d_val, (long long)nnz,
                           flip, h->col, h->val64);
    }
    trace.stage("edge layout");
    const size_t N = (size_t)h->n_rows, M = (size_t)h->n_cols;
    // second, tile-major copy of the edges for the full-scan engine (kernels_tiled.hpp; the big rounds and the eCE pass)
    // launch shape: options.tiled_shape = k + 1 picks shape k (tuning); 0 = by the average (person, tile) segment length
    const bool shape_auto = !(opt->tiled_shape >= 1 && opt->tiled_shape <= kNumTiledShapes);
    h->tiled_shape = shape_auto ? 0 : opt->tiled_shape - 1;

    const int tiled_opt = opt->tiled_min_K;  // 0 default, < 0 never, > 0 minimum K for the full-scan engines
    size_t Mpad = M;
    const bool forced_engine = opt->tiled_force != 0 && tiled_opt > 0;  // tests / tuning: any size
    if (tiled_opt >= 0 && (N >= 4096 || forced_engine)) {
        const bool forced = forced_engine;  // tests / tuning: skip the density heuristics
        const int tcols = kTiledShapes[h->tiled_shape][4];
        const int T = (int)((M + tcols - 1) / tcols);
        if (shape_auto) {
            // lanes per person by the average (person, tile) segment: a step covers 2 edges x 2 loads per lane without
            // entering the leftover loop, whose every pass costs a memory latency (C3: 10 edges per segment -> 4
            // lanes, C4: 20 -> 8 lanes, C2: 50 -> 16 lanes)
            const double seg = (double)nnz / ((double)N * T);
            h->tiled_shape = seg <= 14.0 ? 0 : seg <= 28.0 ? 8 : 9;
        }
        const int rb = kTileRB;
        const long long nblk = ((long long)N + rb - 1) / rb;
        const long long L = nblk * T * rb;
        // the columns of the row-major CSR, whichever layout it has
        const int *cols = h->f32 ? reinterpret_cast<const int *>(h->edges32) : h->col;
        const int cs = h->f32 ? 2 : 1;
        // both tables are addressed with 32-bit byte offsets (8 B per entry): < 2^29 entries each
        if ((forced || (double)nnz / ((double)N * T) >= 4.0) && L < 0x1fffffffLL) {
            h->T = T;
            const int nchunks = (int)((L + kScanChunk - 1) / kScanChunk);
            int *cnt = nullptr, *len = nullptr, *lrel = nullptr, *start = nullptr, *sums = nullptr, *flag = nullptr;
            {
                DevBlock blk;
                blk.want(&cnt, (size_t)L);
                blk.want(&len, (size_t)L);
                blk.want(&lrel, (size_t)L);
                blk.want(&start, (size_t)L + 1);
                blk.want(&sums, (size_t)nchunks + 1);
                blk.want(&flag, 1);
                tmp.blks.emplace_back();
                if ((rc = blk.commit(&tmp.blks.back()))) return rc;
            }
            {
                DevBlock blk;
                blk.want(&h->ovf_ptr, N + 2);
                h->blocks.emplace_back();
                if ((rc = blk.commit(&h->blocks.back()))) return rc;
            }
            int unsorted = 0, total = 0, n_ovf = 0;
            // segment lengths (any = rows whose columns are not ascending: counted per edge, no binary search), their
            // padded scan, the overflow lists' sizes; then the three numbers the host needs
            auto count_and_scan = [&](bool any) -> int {
                HIP_TRY(hipMemsetAsync(cnt, 0, sizeof(int) * (size_t)L, h->stream));
                HIP_TRY(hipMemsetAsync(len, 0, sizeof(int) * (size_t)L, h->stream));
                if (!any) {
                    HIP_TRY(hipMemsetAsync(flag, 0, sizeof(int), h->stream));
                    hipLaunchKernelGGL(k_tile_count, dim3(blocks_for((long long)N, 4)), dim3(256), 0, h->stream, cols, cs,
                                       h->row_ptr, h->n_rows, T, tcols, rb, cnt, len, lrel, flag);
                } else {
                    hipLaunchKernelGGL(k_tile_count_any, dim3(blocks_for((long long)N, 4)), dim3(256), 0, h->stream, cols, cs,
                                       h->row_ptr, h->n_rows, T, tcols, rb, len);
                    hipLaunchKernelGGL(k_tile_even, dim3(blocks_for(L, 256)), dim3(256), 0, h->stream, len, L, cnt);
                }
                hipLaunchKernelGGL(k_scan_sums, dim3(nchunks), dim3(1024), 0, h->stream, cnt, L, sums);
                hipLaunchKernelGGL(k_scan_of_sums, dim3(1), dim3(1024), 0, h->stream, sums, nchunks);
                hipLaunchKernelGGL(k_scan_apply, dim3(nchunks), dim3(1024), 0, h->stream, cnt, L, sums, start);
                // overflow lists (kernels_tiled.hpp, k_ovf_count): the edges of a (person, tile) segment beyond what the
                // launch shape's pipelined loads cover.  `cnt` is free again: per-person counts, then their scan
                const int *shp0 = kTiledShapes[h->tiled_shape];
                h->ovf_cap = 2 * shp0[6] * shp0[3];
                const int nch = (int)(((long long)N + 1 + kScanChunk - 1) / kScanChunk);
                hipLaunchKernelGGL(k_ovf_count, dim3(blocks_for((long long)N, 256)), dim3(256), 0, h->stream, len, h->n_rows, T, rb,
                                   h->ovf_cap, cnt);
                hipLaunchKernelGGL(k_scan_sums, dim3(nch), dim3(1024), 0, h->stream, cnt, (long long)N, sums);
                hipLaunchKernelGGL(k_scan_of_sums, dim3(1), dim3(1024), 0, h->stream, sums, nch);
                hipLaunchKernelGGL(k_scan_apply, dim3(nch), dim3(1024), 0, h->stream, cnt, (long long)N, sums, h->ovf_ptr);
                if (!any) HIP_TRY(hipMemcpyAsync(&unsorted, flag, sizeof(int), hipMemcpyDeviceToHost, h->stream));
                HIP_TRY(hipMemcpyAsync(&total, start + L, sizeof(int), hipMemcpyDeviceToHost, h->stream));
                HIP_TRY(hipMemcpyAsync(&n_ovf, h->ovf_ptr + N, sizeof(int), hipMemcpyDeviceToHost, h->stream));
                HIP_TRY(hipStreamSynchronize(h->stream));
                return MISSLAP_OK;
            };
            if ((rc = count_and_scan(false))) return rc;
            // A row that stores an entry more than once (legal: the last one is the choice, auction_.pyx:467-471) is
            // ascending but not strictly: it takes the stored-index formats too -- formats 0 / 1 order equal values by
            // the COLUMN (k_bid_tiled, kKeyCol), which must then be unique within a row.
            bool carry = (unsorted & 2) != 0;
            bool usable = true;
            if (carry) usable = st.max_row_len <= 65536;
            if (unsorted & 1) {
                // Rows whose columns are not ascending (legal in the reference: cumulative_idxs, auction_.pyx:33-48, only
                // needs the ROWS sorted, and the bid loop takes the stored order, :343-357).  The copy only needs the edges
                // grouped by tile; what the in-row tie rule (:351) needs -- the stored index -- travels with every edge
                // (16 bits: rows of at most 65 536 edges; longer ones keep to the wave-per-row kernel).
                carry = true;
                usable = st.max_row_len <= 65536;
                if (usable && (rc = count_and_scan(true))) return rc;
            }
            trace.stage("tile counts + scans");
            h->tiled_fmt = (h->f32 ? 0 : 1) + (carry ? 2 : 0);
            if (h->tiled_fmt != 0) {  // formats 1..3 exist for the three production shapes (4 / 8 / 16 lanes per person)
                const int gl0 = kTiledShapes[h->tiled_shape][6];
                const int want_shape = gl0 == 4 ? 0 : gl0 == 8 ? 8 : 9;
                if (kTiledShapes[want_shape][3] * 2 * gl0 != h->ovf_cap) usable = false;  // (a tuning shape with another depth)
                h->tiled_shape = want_shape;
            }
            const int rec_bytes = tile_rec_bytes(h->tiled_fmt);
            // The engine pays where segments fit the pipelined loads.  Where more than 1 / 16 of the edges would sit on
            // overflow lists (rows that are dense inside a tile: the `mat=` shapes) the wave-per-row scan is the better
            // full-scan kernel anyway -- a dense row reads the price table in order -- and the second copy is not built.
            const bool fits = forced || (long long)n_ovf * 16 <= (long long)nnz;
            // (records are addressed with 32-bit byte offsets)
            const long long total_max = std::min<long long>(0x1ffffff0LL, (0xfffff000LL / rec_bytes) * 2);
            if (usable && total > 0 && total < total_max && fits) {
                h->n_tiled = total;
                {
                    DevBlock blk;
                    blk.want(&h->ovf_q, (size_t)n_ovf + 1);
                    h->blocks.emplace_back();
                    if ((rc = blk.commit(&h->blocks.back()))) return rc;
                }
                HIP_TRY(hipMemsetAsync(h->ovf_q, 0, sizeof(int4), h->stream));  // (entry 0 is read by idle lanes)
                // + kTilePadRecords ZEROED records behind the copy: the unconditional loads of a lane group reach up to
                // lanes x loads-per-segment records past a segment's start whatever its length (16 x 2 = 32 for the widest
                // shape); what a masked-off lane reads must be a FINITE value (its price is +inf, and NaN - inf would
                // poison the running second-best through fmin), so the bytes behind the last segment are zeros, not
                // whatever the block held before
                const size_t tiled_words = ((size_t)total / 2 + kTilePadRecords) * (size_t)(rec_bytes / 4);
                {
                    DevBlock blk;
                    blk.want(&h->tiled, tiled_words);
                    blk.want(&h->seg4, (size_t)L + 2);
                    blk.want(&h->tcol, (size_t)total + 16);
                    h->blocks.emplace_back();
                    if ((rc = blk.commit(&h->blocks.back()))) return rc;
                }
                HIP_TRY(hipMemsetAsync(h->tiled, 0, sizeof(unsigned) * tiled_words, h->stream));
                HIP_TRY(hipMemsetAsync(h->tcol, 0, sizeof(int) * ((size_t)total + 16), h->stream));
                if (carry) HIP_TRY(hipMemsetAsync(lrel, 0, sizeof(int) * (size_t)L, h->stream));  // the segments' running fill
                // packed edges holding price slots (buffer stride of the double-buffered shapes)
                const int buf_stride = tcols == kTileColsBig ? 0 : tcols + 128;
                const dim3 gs(blocks_for((long long)N, 4)), bs(256);
                const EdgesF32 e32{h->edges32};
                const EdgesF64 e64{h->col, h->val64};
                switch (h->tiled_fmt) {
                    case 0: hipLaunchKernelGGL((k_tile_scatter<EdgesF32, 0>), gs, bs, 0, h->stream, e32, h->row_ptr, h->n_rows, T, tcols, rb, start, lrel, lrel, h->tiled, h->tcol, buf_stride); break;
                    case 1: hipLaunchKernelGGL((k_tile_scatter<EdgesF64, 1>), gs, bs, 0, h->stream, e64, h->row_ptr, h->n_rows, T, tcols, rb, start, lrel, lrel, h->tiled, h->tcol, buf_stride); break;
                    case 2: hipLaunchKernelGGL((k_tile_scatter<EdgesF32, 2>), gs, bs, 0, h->stream, e32, h->row_ptr, h->n_rows, T, tcols, rb, start, lrel, lrel, h->tiled, h->tcol, buf_stride); break;
                    default: hipLaunchKernelGGL((k_tile_scatter<EdgesF64, 3>), gs, bs, 0, h->stream, e64, h->row_ptr, h->n_rows, T, tcols, rb, start, lrel, lrel, h->tiled, h->tcol, buf_stride); break;
                }
                hipLaunchKernelGGL(k_pack_seg4, dim3(blocks_for(L + 1, 256)), dim3(256), 0, h->stream, start, len, L, h->seg4);
                hipLaunchKernelGGL(k_ovf_fill, dim3(blocks_for((long long)N, 256)), dim3(256), 0, h->stream, len, start,
                                   h->n_rows, T, rb, h->ovf_cap, h->ovf_ptr, h->tiled, h->tcol, h->ovf_q, h->tiled_fmt);
                HIP_TRY(hipGetLastError());
                HIP_TRY(hipStreamSynchronize(h->stream));
                trace.stage("tile-major copy");
                h->tiled_ok = true;
                // break-even against k_bid (cost ~ K) measured at C3: the full-scan engines have a fixed cost
                // (price fills, barriers / the merge pass) of about a fifth of a full k_bid scan: 0.3 N.  Where the rows
                // keep candidate lines (<= 256 edges) the engine takes the rounds from 0.7 N on only (round 6): a partial
                // round reads nearly the whole tile-major copy whoever bids (a (person, tile) segment is half a fabric line:
                // C3, 55 % of the persons bidding: 237 of 241 MB), while k_bid answers most of those bids from lines --
                // same box, threshold 0.3 / 0.7 N: C3 solve 359.3 / 357.8 ms, C2 123.4 / 122.2, all launches of the engine
                // 42.8 / 48.4 % and 41.6 / 51.6 % of 8 TB/s; C4 (300 edges per row: no lines) 5.88 / 6.09 ms: stays at 0.3 N
                const bool rows_keep_lines = cand_mode != 1 && avg_row <= (long long)kCandRowMax;
                h->tiled_min_K = tiled_opt > 0 ? tiled_opt : (int)std::max<size_t>((N * (rows_keep_lines ? 7 : 3)) / 10, 8192);
                Mpad = (size_t)T * tcols;  // whole tiles: the LDS fills need no bounds test
                const hipFuncAttribute at = hipFuncAttributeMaxDynamicSharedMemorySize;
                // per create, i.e. per device: the > 64 KB dynamic-LDS opt-in is a property of the function ON
                // the current device, so a process-wide "done" flag would leave a second device without it
                if (h->tiled_fmt == 0) {
                    switch (h->tiled_shape) {
#define X(I, TH, R, B, D, TC, LD, GL, CS)                                                                           \
    case I:                                                                                                          \
        HIP_TRY(hipFuncSetAttribute((const void *)k_bid_tiled<TH, R, B, D, TC, LD, 0, GL, CS>, at,                     \
                                    (int)tiled_lds_bytes(TC)));                                                      \
        break;
                        MISSLAP_FOR_TILED_SHAPES(X)
#undef X
                    }
                    HIP_TRY(hipFuncSetAttribute((const void *)MISSLAP_BID_KERNEL_REV(4, 0), at, (int)tiled_lds_bytes(kTileColsHalf)));
                    HIP_TRY(hipFuncSetAttribute((const void *)MISSLAP_BID_KERNEL_REV(8, 0), at, (int)tiled_lds_bytes(kTileColsHalf)));
                    HIP_TRY(hipFuncSetAttribute((const void *)MISSLAP_BID_KERNEL_REV(16, 0), at, (int)tiled_lds_bytes(kTileColsHalf)));
                    switch (check_lanes(h)) {  // the check pass on the same engine (launch_rows_all)
#define X(GL)                                                                                                        \
    case GL:                                                                                                         \
        HIP_TRY(hipFuncSetAttribute((const void *)MISSLAP_CHECK_KERNEL(GL), at, (int)tiled_lds_bytes(kTileColsHalf)));  \
        break;
                        MISSLAP_FOR_CHECK_LANES(X)
#undef X
                        default: break;
                    }
                } else {
                    switch (h->tiled_fmt * 100 + kTiledShapes[h->tiled_shape][6]) {
#define X(FMT, GL)                                                                                                   \
    case FMT * 100 + GL:                                                                                             \
        HIP_TRY(hipFuncSetAttribute((const void *)MISSLAP_BID_KERNEL_FMT(GL, FMT), at, (int)tiled_lds_bytes(kTileColsHalf)));   \
        if (FMT == 1) HIP_TRY(hipFuncSetAttribute((const void *)MISSLAP_BID_KERNEL_REV(GL, 1), at, (int)tiled_lds_bytes(kTileColsHalf))); \
        HIP_TRY(hipFuncSetAttribute((const void *)MISSLAP_CHECK_KERNEL_FMT(GL, FMT), at, (int)tiled_lds_bytes(kTileColsHalf))); \
        break;
                        MISSLAP_FOR_FMT_LANES(X)
#undef X
                        default: break;
                    }
                }
            }
            HIP_TRY(hipStreamSynchronize(h->stream));  // the temporaries are released at scope exit
        }
    }
    trace.stage("tile engine attributes");
    {
        DevBlock blk;
        blk.want(&h->price, Mpad);
        blk.want(&h->rec, M);
        {
            // the fp32 filter of the wave-per-row kernel's full scans: where that kernel does the full scans (no tile-major
            // copy) and the fp64 price table exceeds an XCD's L2 share (>= 3 MB); costs of ordinary magnitude only (the
            // error bound of the filter is relative: no subnormal fp32 values, no overflow of fl32(price)).
            // MISSLAP_F32_FILTER=0 / 1: never / whatever the table's size (A/B timing, tests)
            const char *fe = std::getenv("MISSLAP_F32_FILTER");  // (read per create: the tests switch it)
            const int env = fe ? std::atoi(fe) : -1;
            double max_abs_d;
            const long long b = (long long)st.max_abs_bits;
            std::memcpy(&max_abs_d, &b, sizeof(double));
            const bool range_ok = max_abs_d > 0x1p-100 && max_abs_d < 0x1p60;
            if (!h->tiled_ok && range_ok && env != 0 && (env == 1 || M * sizeof(double) >= ((size_t)3 << 20))) {
                blk.want(&h->price32, M);
                blk.want(&h->pmax_bits, 1);
                h->cmax32 = (float)max_abs_d;
            }
        }
        h->line_maintenance = cand_mode != 2;
        if (cand_mode != 1) {  // candidate lines (cand_mode 1: off -- A/B timing, parity tests, the precision guard)
            blk.want(&h->cand, N * (size_t)kCandLanes);
            if (!h->f32) blk.want(&h->cand64, N * (size_t)kCandLanes);  // 12 B/edge layout: the costs as fp64
        }
        blk.want(&h->p2o, N);
        blk.want(&h->o2p, M);
        blk.want(&h->U, N);
        blk.want(&h->bid_key, N);
        blk.want(&h->bid_obj, N);
        blk.want(&h->bid_rec, (size_t)kRoundSmallMax);
        blk.want(&h->best_key, M);
        blk.want(&h->best_pos, M);
        blk.want(&h->cnt, 2 * ((N + kChunk - 1) / kChunk) + 2);
        blk.want(&h->hole_list, N);
        blk.want(&h->mover_list, N);
        blk.want(&h->need_list, N);
        blk.want(&h->ctl, 1);
        blk.want(&h->contrib, N);
        blk.want(&h->nmatch, N);
        // >= any grid of the final pass: the gather form launches at most kMaxGridBlocks workgroups, the engine form
        // (launch_rows_all) ceil(N / persons per workgroup) with at least (1024 - 192) / 16 lane groups x 4 persons = 208
        // persons per workgroup (16 lanes per person), or one workgroup per CU
        h->fin_slots_n = (int)std::max<size_t>(kMaxGridBlocks, final_pass_grid_max(N, h->n_cus) + 1);
        blk.want(&h->fin_slots, (size_t)h->fin_slots_n);
        h->wg_stats_slots = (int)std::min<size_t>(N / 64 + 4096, 1u << 22);  // >= kMaxGridBlocks and any scan grid
        blk.want(&h->wg_stats, (size_t)kStatWords * (size_t)h->wg_stats_slots);
        if (h->tiled_ok && kTiledShapes[h->tiled_shape][7] > 1) {
            blk.want(&h->part_vw, (size_t)kTiledShapes[h->tiled_shape][7] * N);
            blk.want(&h->part_g, (size_t)kTiledShapes[h->tiled_shape][7] * N);
            blk.want(&h->split_cnt, (size_t)N / 256 + 1024);  // >= slices of any launch (a slice holds >= 256 bidders or the grid is one CU round)
        }
        if (h->profile) {
            h->launch_edges_cap = 1 << 20;
            blk.want(&h->launch_edges, 2 * (size_t)h->launch_edges_cap);  // {edges, of which answered from lines} per launch
        }
        h->blocks.emplace_back();
        if ((rc = blk.commit(&h->blocks.back()))) return rc;
    }
    HIP_TRY(hipMemsetAsync(h->price, 0, sizeof(double) * Mpad, h->stream));
    HIP_TRY(hipMemsetAsync(h->bid_rec, 0, sizeof(int4) * kRoundSmallMax, h->stream));
    HIP_TRY(hipMemsetAsync(h->wg_stats, 0, sizeof(unsigned long long) * kStatWords * (size_t)h->wg_stats_slots, h->stream));
    if (h->split_cnt) HIP_TRY(hipMemsetAsync(h->split_cnt, 0, sizeof(int) * ((size_t)N / 256 + 1024), h->stream));
    if (h->profile)
        HIP_TRY(hipMemsetAsync(h->launch_edges, 0, sizeof(unsigned long long) * 2 * (size_t)h->launch_edges_cap, h->stream));
    // the mirror, the two trailing status copies and the live status words (kept together: one pooled allocation)
    // (coherent + mapped EXPLICITLY: with HIP_HOST_COHERENT=0 in the environment a default allocation is not coherent,
    // and the kernels' system-scope stores to the live words would become visible at sync points only)
    if (!h->h_ctl) HIP_TRY(hipHostMalloc((void **)&h->h_ctl, 3 * sizeof(Ctl) + 128, hipHostMallocCoherent | hipHostMallocMapped));
    h->h_stat = h->h_ctl + 1;
    {
        char *base = reinterpret_cast<char *>(h->h_ctl + 3);
        base += (64 - (reinterpret_cast<uintptr_t>(base) & 63)) & 63;
        h->live = reinterpret_cast<volatile unsigned long long *>(base);
        for (int k = 0; k < 5; ++k) h->live[k] = 0ull;  // ticket 0 = nothing posted (tickets start at 1); [4]: the eCE verdict
        void *dev = nullptr;
        if (hipHostGetDevicePointer(&dev, base, 0) == hipSuccess) h->live_dev = static_cast<unsigned long long *>(dev);
        const char *e = std::getenv("MISSLAP_LIVE_STATUS");
        h->live_off = h->live_dev == nullptr || (e && e[0] == '0');
        h->live_every_round = e && e[0] == '2';
        const char *f = std::getenv("MISSLAP_ROUND_FUSED");
        h->round_fused = !(f && f[0] == '0');
        h->ticket = 0;
        h->live_valid = false;
    }
    for (hipEvent_t &e : h->stat_ev)
        if (!e) HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    h->shard_min_K = (int)std::max<size_t>((N * 3) / 10, 8192);  // (multi-GPU: the rounds that are sharded and exchanged)
    if (h->tiled_ok && tiled_opt > 0) h->shard_min_K = h->tiled_min_K;  // (a caller's engine threshold moves it along)
    if (opt->shard_min_K > 0) h->shard_min_K = opt->shard_min_K;
    if (opt->shard_min_K < 0) h->shard_min_K = 0;  // every grid round sharded + exchanged
    // candidate lines are used and built below the full-scan regime (0.3 N): C5 with lines built in every round
    // 3.85 s and a 939 us full scan (it writes a 256-byte line per person), with this limit 3.87 s and 588 us
    if (h->cand_build_max_K == 0x7fffffff)
        h->cand_build_max_K = (int)std::max<size_t>((N * 3) / 10, 8192) - 1;
    h->max_iter = opt->max_iter < 1 ? 1 : opt->max_iter;  // the loop body runs before the first test (:271-275)
    hipLaunchKernelGGL(k_init_state, dim3(blocks_for((long long)(N > M ? N : M), 256)), dim3(256), 0, h->stream,
                       h->ctl, h->price, h->rec, h->p2o, h->o2p, h->U, h->best_key, h->best_pos, h->cand, h->n_rows, h->n_cols,
                       (long long)h->max_iter);
    HIP_TRY(hipGetLastError());
    h->ece_flag_clear = true;
    // eps schedule, fp32 exactly as the generated C of the reference
    double max_abs;
    {
        const long long b = (long long)st.max_abs_bits;
        std::memcpy(&max_abs, &b, sizeof(double));
    }
    const float C = (float)max_abs;               // auction_.pyx:242-243
    h->eps = (float)((double)C / 2.0);            // :246
    h->target_eps = (float)(1.0 / (double)h->n_rows);  // :247
    h->theta = (float)0.15;                       // :248
    if (opt->eps_start > 0) h->eps = opt->eps_start;  // :251-252
    h->start_eps = h->eps;
    begin_phase(h);
    h->K_ub = h->n_rows;
    h->K_exact = true;
    h->phase_fresh = true;
    HIP_TRY(hipStreamSynchronize(h->stream));
    tmp.drained = true;
    trace.stage("state blocks + init");
    return MISSLAP_OK;
}

// entries a handle can hold: row pointers are int32 (options.nnz_limit > 0 lowers the limit: guard tests)
int64_t nnz_limit(const misslap_options *opt) {
    return opt->nnz_limit > 0 ? (int64_t)opt->nnz_limit : (int64_t)0x7fffffff;
}

// Device-resident inputs: the library works on a private non-blocking stream, which is not ordered behind the
// stream(s) that produced the caller's buffers.  With options.input_stream the solver's stream waits for an event
// recorded on the producer's stream (nothing else of the caller is held up); without it the whole device is waited for
// once, before anything reads the buffers.  (The few synchronous host reads of the inputs -- the last row index -- go
// through hipMemcpy on the null stream and are therefore made after a wait for that event as well.)
int sync_device_inputs(const misslap_options *opt, hipStream_t solver_stream) {
    if (!opt->input_on_device) return MISSLAP_OK;
    if (!opt->input_stream) {
        HIP_TRY(hipDeviceSynchronize());
        return MISSLAP_OK;
    }
    hipEvent_t ev = nullptr;
    HIP_TRY(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    hipError_t e = hipEventRecord(ev, (hipStream_t)opt->input_stream);
    if (e == hipSuccess) e = hipStreamWaitEvent(solver_stream, ev, 0);
    if (e == hipSuccess) e = hipEventSynchronize(ev);  // host reads of the inputs below
    (void)hipEventDestroy(ev);
    if (e != hipSuccess) return fail(MISSLAP_ERR_HIP, "cannot order the solver behind options.input_stream: %s", hipGetErrorString(e));
    return MISSLAP_OK;
}

// The caller's options in the current layout.  struct_size 88 = a version-1 caller (abi_v1.hpp): its reserved[] knobs
// are mapped onto the named fields and the handle remembers to answer with the version-1 misslap_meta.  A version-2
// struct may be shorter than this library's (built against an older version-2 header: the missing tail is zero =
// defaults) but not longer than it knows how to read.
int normalise_options(const misslap_options *in, misslap_options *out, int *abi) {
    if (!in) return fail(MISSLAP_ERR_INVALID, "null options");
    std::memset(out, 0, sizeof(*out));
    if (in->struct_size == (int32_t)sizeof(misslap_options_v1)) {
        misslap_options_v1 v1;
        std::memcpy(&v1, in, sizeof(v1));
        std::memcpy(out, &v1, offsetof(misslap_options_v1, reserved));  // identical prefix
        out->tiled_min_K = v1.reserved[0];
        out->tiled_shape = v1.reserved[1];
        out->tiled_force = v1.reserved[2];
        out->shard_min_K = v1.reserved[3];
        out->cand_mode = v1.reserved[4];
        out->partial_in_list_order = v1.reserved[5];
        out->nnz_limit = v1.reserved[6];
        out->cand_build_max_K = v1.reserved[7] & 0xffffff;
        out->cand_refresh_min = (v1.reserved[7] >> 24) & 63;
        *abi = 1;
    } else {
        // (the named knobs end where version 1's 88 bytes end: a version-2 struct is told apart by being longer)
        constexpr int32_t kMinV2 = (int32_t)offsetof(misslap_options, reserved) + 4;
        static_assert(offsetof(misslap_options, reserved) == sizeof(misslap_options_v1), "see above");
        if (in->struct_size < kMinV2 || in->struct_size > (int32_t)sizeof(misslap_options))
            return fail(MISSLAP_ERR_INVALID, "misslap_options.struct_size %d: expected %d (ABI %d; %d = ABI 1 is accepted too)",
                        in->struct_size, (int)sizeof(misslap_options), MISSLAP_ABI_VERSION, (int)sizeof(misslap_options_v1));
        std::memcpy(out, in, (size_t)in->struct_size);
        for (int32_t r : out->reserved)
            if (r != 0) return fail(MISSLAP_ERR_INVALID, "misslap_options.reserved must be zero");
        *abi = 2;
    }
    out->struct_size = (int32_t)sizeof(misslap_options);
    if (out->cand_mode < 0 || out->cand_mode > 2) return fail(MISSLAP_ERR_INVALID, "cand_mode %d: 0, 1 or 2", out->cand_mode);
    if (out->cand_refresh_min < 0 || out->cand_refresh_min > 32)
        return fail(MISSLAP_ERR_INVALID, "cand_refresh_min %d: 0 .. 32", out->cand_refresh_min);
    if (out->cand_build_max_K < 0) return fail(MISSLAP_ERR_INVALID, "cand_build_max_K must not be negative");
    if (out->tiled_shape < 0 || out->tiled_shape > kNumTiledShapes)
        return fail(MISSLAP_ERR_INVALID, "tiled_shape %d: 0 (automatic) .. %d", out->tiled_shape, kNumTiledShapes);
    return MISSLAP_OK;
}

int new_handle(misslap_solver **out, const misslap_options *opt, int abi, misslap_solver **hp) {
    if (!out || !opt) return fail(MISSLAP_ERR_INVALID, "null argument");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(MISSLAP_ERR_NO_DEVICE, "no HIP device available: libmisslap has no CPU fallback");
    if (opt->device < 0 || opt->device >= ndev) return fail(MISSLAP_ERR_INVALID, "device %d out of range", opt->device);
    if (opt->tail_threshold > kTailMax)
        return fail(MISSLAP_ERR_INVALID, "tail_threshold %d exceeds %d", opt->tail_threshold, kTailMax);
    if (opt->shard_world < 0 || (opt->shard_world > 0 && (opt->shard_rank < 0 || opt->shard_rank >= opt->shard_world)))
        return fail(MISSLAP_ERR_INVALID, "bad shard rank/world");
    HIP_TRY(hipSetDevice(opt->device));
    misslap_solver *h = new misslap_solver();
    h->abi = abi;
    h->device = opt->device;
    {
        int cus = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, opt->device) == hipSuccess && cus > 0)
            h->n_cus = cus;
    }
    h->maximize = opt->maximize ? 1 : 0;
    h->thr = opt->tail_threshold >= 0 ? opt->tail_threshold : -1;  // -1: resolved in build_from_device_coo
    if (const char *e = std::getenv("MISSLAP_TAIL_LAUNCH_ROUNDS")) h->tail_launch_rounds = std::max(1, std::atoi(e));  // (read per create)
    if (opt->cand_build_max_K > 0) h->cand_build_max_K = opt->cand_build_max_K;
    if (opt->cand_refresh_min > 0) h->cand_refresh_min = opt->cand_refresh_min - 1;
    h->rounds_per_sync = opt->rounds_per_sync > 0 ? opt->rounds_per_sync : kDefaultRoundsPerSync;
    h->rounds_per_sync_auto = opt->rounds_per_sync <= 0;
    h->world = opt->shard_world > 0 ? opt->shard_world : 1;
    h->rank = opt->shard_world > 0 ? opt->shard_rank : 0;
    h->profile = opt->profile != 0;
    h->profile_all = opt->profile >= 2;
    HostRes res;
    if (host_pool().take(h->device, &res)) {
        h->stream = res.stream;
        h->h_ctl = res.h_ctl;
        h->stat_ev[0] = res.ev[0];
        h->stat_ev[1] = res.ev[1];
    }
    if (!h->stream && hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking) != hipSuccess) {
        delete h;
        return fail(MISSLAP_ERR_HIP, "hipStreamCreate failed");
    }
    h->own_stream = true;
    *hp = h;
    return MISSLAP_OK;
}
}  // namespace
