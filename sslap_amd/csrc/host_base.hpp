// host_base.hpp -- error plumbing, tuning constants, launch shapes of the full-scan engine, profiling records and the solver handle.
// (part of the single translation unit misslap.hip; included in the order given there)
#pragma once

namespace {

thread_local std::string g_err;

int fail(int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

#define HIP_TRY(expr)                                                                              \
    do {                                                                                           \
        hipError_t _e = (expr);                                                                    \
        if (_e != hipSuccess)                                                                      \
            return fail(MISSLAP_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, \
                        __LINE__);                                                                 \
    } while (0)

double now_ms() {
    using clk = std::chrono::steady_clock;
    return std::chrono::duration<double, std::milli>(clk::now().time_since_epoch()).count();
}
// Diagnostics build (-DMISSLAP_DIAG), MISSLAP_TRACE_CREATE=1: where a handle's setup time goes (stderr, one line per stage;
// the stream is drained at every stage, so the sum is a little above an untraced create)
struct CreateTrace {
    bool on;
    double t0;
    hipStream_t st;
    explicit CreateTrace(hipStream_t s) : st(s) {
#ifdef MISSLAP_DIAG
        const char *e = std::getenv("MISSLAP_TRACE_CREATE");
        on = e && e[0] == '1';
#else
        on = false;
#endif
        t0 = now_ms();
    }
    void stage(const char *name) {
        if (!on) return;
        (void)hipStreamSynchronize(st);
        const double t = now_ms();
        fprintf(stderr, "[misslap create] %-28s %8.3f ms\n", name, t - t0);
        t0 = t;
    }
};

// Rounds with K <= threshold run in the tail kernels.  Break-even against a grid round (two launches: k_bid +
// k_round_small, ~12 us), measured with tools/sweep_thr.py / tools/tail_stats.py after every change of either side.
// Round 2 with the 16-wavefront block kernel: C3 96: 485 ms, 192: 479, 256: 482, 384: 480, 512: 481; C5 128: 3.797 s,
// 256: 3.790, 448: 3.823; C2 96: 160.9 ms, 192: 159.4, 320: 158.9 -- flat above ~150.
constexpr int kDefaultTailThreshold = 192;
// ... without candidate lines (12 B/edge layout, rows too long for a line) every bid of the tail is a row scan and the
// tail only pays while few of them are in flight: C2 with fp64 values 192: 314 ms, 40: 279; C4 (300 edges per row) 192:
// 13.7 ms, 40: 13.1
constexpr int kDefaultTailThresholdNoLines = 40;
constexpr int kLongRowsFrom = 1024;
constexpr int kLongRowsMixedPercent = 20;
constexpr long long kLongRowsAfterTailRoundsMin = 100;  // ... or n_rows / 64 tail rounds, whichever is more (launch_tail)
// Tail rounds between two maintenance passes of a long-row handle (the tail kernels cannot rebuild the line of a long
// row in place: a missed line stays missed until the next pass, and every miss is a scan of the whole row).  Dense
// 8000^2: no limit 341 ms, 4096: 324, 1024: 220, 256: 121, 128: 120, 64: 144, 32: 200; with a quarter of the budget for
// the block instance (most bidders per round, i.e. most lines spent) 512: 123, 256: 101, 128: 108.
constexpr int kLongRowTailBudget = 192;
// ... and what a pass rebuilds: after a few hundred rounds of a dense problem nearly EVERY line has lost some of its
// thirty candidates (everybody's candidates are the same popular objects), so a pass at the strict threshold of the
// short rows (24 live candidates) re-reads the whole matrix -- 512 MB per pass at dense 8000^2, 141 us.  Long rows are
// rebuilt below 12 live candidates.  Dense 8000^2, threshold x rounds between passes: 24 x 256: 115 ms, 12 x 256: 102,
// 12 x 192: 88.6, 12 x 128: 89.9, 10 x 192: 88.5, 14 x 192: 91.2, 16 x 128: 93.6, 12 x 384: 114, 4 x 256: 125.
constexpr int kLongRowMinAlive = 12;
// Where a line is rebuilt matters more than whether it hits: lines are built in the grid rounds but earn their keep in
// the tail kernels, tens of thousands of rounds later, and a line that still hits but is nearly spent would miss THERE,
// where a row scan is the whole round and not one of hundreds in flight.  Two mechanisms, both on the number of
// candidates still at or above tau ("live"):
//   * k_refresh_lines, the maintenance pass over ALL persons ahead of the tail kernels (once per eps-phase): a line
//     with fewer than kCandMaintenanceMin live candidates is rebuilt.  After it the tail misses nothing at all.
//   * k_bid may answer a hit that leaves fewer than cand_refresh_min live candidates by a full scan + rebuild.  Before the
//     maintenance pass existed this was the big lever (C3 590 ms without, 511 (10), 498 (16), 493 (22), 491 (24), 496
//     (31); C2 187 -> 160 ms, C5 4.61 -> 3.86 s); with the pass it is redundant -- C3 24 / 8 / 0: 428 / 427 / 425 ms, C2
//     144.8 / 142.2 / 143.9, C5 3.518 / 3.514 / 3.506 s -- and off by default (options.reserved[7] turns it on).
// With the pass at 6 instead of 24, C5 loses 4 % (lines spent before the tail ends).
constexpr int kDefaultCandRefresh = 0;
constexpr int kCandMaintenanceMin = 24;
constexpr int kDefaultRoundsPerSync = 16;
constexpr int kRoundsPerSyncLive = 4;  // ... with live status: a status read is a poll of host memory, and a short batch
                                       // wastes fewer launches on rounds that turn out not to be live (same box, 16 / 4:
                                       // C4 6.1 / 5.8 ms per solve, C1 9.7 / 9.2, C2 and C3 unchanged; tools/sweep_rps.sh)
constexpr int kRoundsPerSyncLargeK = 2;  // batch length while K > kRoundSmallMax
constexpr int kMaxGridBlocks = 2048;  // 256 CUs x 8 resident 256-thread blocks
constexpr int kNumTiledShapes = 10;
// (shape 0: three loader wavefronts measured 1-2 % faster than one inside a solve; the round-2 shape 4 -- eight persons
// per 8-lane group, one load per segment -- spilled 34 VGPRs and was retired: the index now names the column-split
// variant of shape 0)
// launch shapes of k_bid_tiled: {threads, persons per lane group, persons in flight, loads per segment,
// prices per LDS tile, loader wavefronts, lanes per person, column split}; see kernels_tiled.hpp
const int kTiledShapes[kNumTiledShapes][8] = {
    {1024, 4, 2, 2, kTileColsHalf, 3, 4, 1}, {1024, 4, 2, 2, kTileColsHalf, 0, 4, 1}, {1024, 4, 2, 3, kTileColsBig, 0, 4, 1},
    {1024, 4, 2, 2, kTileColsHalf, 1, 4, 1}, {1024, 8, 2, 2, kTileColsHalf, 3, 4, 2}, {1024, 4, 1, 2, kTileColsHalf, 1, 4, 1},
    {1024, 4, 2, 3, kTileColsHalf, 1, 4, 1}, {1024, 4, 2, 2, kTileColsHalf, 2, 4, 1},
    // longer (person, tile) segments: 8 / 16 lanes per person, i.e. 32 / 64 edges per step
    {1024, 4, 2, 2, kTileColsHalf, 3, 8, 1}, {1024, 4, 2, 2, kTileColsHalf, 3, 16, 1}};
#define MISSLAP_FOR_TILED_SHAPES(X)                                                                                  \
    X(0, 1024, 4, 2, 2, kTileColsHalf, 3, 4, 1) X(1, 1024, 4, 2, 2, kTileColsHalf, 0, 4, 1)                          \
    X(2, 1024, 4, 2, 3, kTileColsBig, 0, 4, 1) X(3, 1024, 4, 2, 2, kTileColsHalf, 1, 4, 1)                           \
    X(4, 1024, 8, 2, 2, kTileColsHalf, 3, 4, 2) X(5, 1024, 4, 1, 2, kTileColsHalf, 1, 4, 1)                          \
    X(6, 1024, 4, 2, 3, kTileColsHalf, 1, 4, 1) X(7, 1024, 4, 2, 2, kTileColsHalf, 2, 4, 1)                          \
    X(8, 1024, 4, 2, 2, kTileColsHalf, 3, 8, 1) X(9, 1024, 4, 2, 2, kTileColsHalf, 3, 16, 1)
// ... and for the record formats 1..3 of the tile-major copy (fp64 values, rows with unsorted columns): the shapes 0 / 8 /
// 9, i.e. {1024 threads, 4 persons per lane group, 2 in flight, 2 loads per segment, half tiles, 3 loaders} x lanes
#define MISSLAP_BID_KERNEL_FMT(GL, FMT) k_bid_tiled<1024, 4, 2, 2, kTileColsHalf, 3, 0, GL, 1, 0, FMT>
// ... and the backward walk (kRev) of the column-keyed formats 0 / 1 in the same three shapes
#define MISSLAP_BID_KERNEL_REV(GL, FMT) k_bid_tiled<1024, 4, 2, 2, kTileColsHalf, 3, 0, GL, 1, 0, FMT, 1>
#define MISSLAP_FOR_FMT_LANES(X) X(1, 4) X(1, 8) X(1, 16) X(2, 4) X(2, 8) X(2, 16) X(3, 4) X(3, 8) X(3, 16)
inline size_t tiled_lds_bytes(int tile_cols) {  // see the LDS map in k_bid_tiled; + statistics scratch
    const size_t doubles = tile_cols == kTileColsBig ? (size_t)tile_cols + 2 : 2 * (size_t)tile_cols + 128;
    return doubles * sizeof(double) + kTileStatBytes + kTileTouchBytes;  // + statistics scratch (incl. the arrival word of a column-split shape) + the loaders' touch scratch
}

// Profiled launches (options.profile): the two events are handed to the launch itself (hipExtLaunchKernel), so they
// carry the begin / end timestamps of the KERNEL -- what a rocprofv3 kernel trace reports.  Events recorded around a
// launch on the stream bracket the dispatch gap as well (~7 us per launch at C3: 92.4 against 85.6 us in round 2).
#define MISSLAP_LAUNCH_TIMED(PR, KERNEL, GRID, BLOCK, LDS, STREAM, ...)                              \
    do {                                                                                            \
        if (PR) {                                                              \
            hipExtLaunchKernelGGL(KERNEL, GRID, BLOCK, LDS, STREAM, (PR)->start, (PR)->stop, 0, __VA_ARGS__); \
        } else {                                                                                    \
            if (PR) (void)hipEventRecord((PR)->start, STREAM);                                      \
            hipLaunchKernelGGL(KERNEL, GRID, BLOCK, LDS, STREAM, __VA_ARGS__);                      \
            if (PR) (void)hipEventRecord((PR)->stop, STREAM);                                       \
        }                                                                                           \
    } while (0)

struct ProfRec {
    hipEvent_t start, stop;
    int kind;        // 0 = k_bid, 1 = k_tail, 2 = k_bid_tiled
    int fullscan;    // bid launch with K == n_rows
    int launch_idx;  // index into launch_edges (kind 0)
};

}  // namespace

namespace {
struct Blk {  // a device block and its size (the size it is returned to the block cache with)
    void *p = nullptr;
    size_t bytes = 0;
};
int block_alloc(void **p, size_t bytes, size_t *got);
void block_free(int device, void *p, size_t bytes);
}  // namespace

namespace {
struct BatchFiber;  // host_batch.hpp
}
struct misslap_solver {
    BatchFiber *batch = nullptr;  // inside misslap_solve_batch: the fiber this handle's solve loop runs on (its launches are recorded)
    int abi = MISSLAP_ABI_VERSION;  // 1: created with version-1 options (88 bytes) -> version-1 misslap_meta layout
    int n_cus = 256;                // compute units of the device (one k_bid_tiled workgroup per CU)
    // Candidate lines are exact only while prices never fall (device_common.hpp).  A price update is fl(fl(c - w) + eps)
    // with w <= fl(c - p): it can land BELOW p once eps is smaller than the rounding error of those operations, i.e. for
    // huge |cost| in the LAST eps-phases (eps falls to 0.15 / N).  The lines are used while the phase's eps is at or above
    // lines_safe_eps = max|cost| x 2^-44 (2^9 ulps of the largest cost) and dropped for good from the first phase below it
    // (begin_phase; the full scans never depend on the invariant; kErrPriceFell is the run-time backstop).
    double lines_safe_eps = 0.0;
    bool lines_dropped = false;     // ... that phase has been reached: the lines are no longer read or maintained
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    int64_t nnz = 0;
    int n_rows = 0, n_cols = 0;
    int maximize = 0;
    bool f32 = true;
    // device buffers
    int2 *edges32 = nullptr;
    int *col = nullptr;
    double *val64 = nullptr;
    int *row_ptr = nullptr;
    double *price = nullptr;
    float *price32 = nullptr;  // fp32 mirror of the prices for the filter scans of the wave-per-row kernel (wave_bid_filter);
    int *pmax_bits = nullptr;  // nullptr: no filter.  pmax_bits: the largest price at the last rebuild of the mirror
    float cmax32 = 0.f;        // (float) max |cost|
    PriceRec *rec = nullptr;
    int2 *cand = nullptr;  // candidate lines, 256 B per person
    double *cand64 = nullptr;  // ... and 256 B of fp64 costs per person in the 12 B/edge layout
    int *p2o = nullptr, *o2p = nullptr, *U = nullptr;
    unsigned long long *bid_key = nullptr;
    int *bid_obj = nullptr;
    int4 *bid_rec = nullptr;
    unsigned long long *best_key = nullptr;
    int *best_pos = nullptr;
    int *cnt = nullptr, *hole_list = nullptr, *mover_list = nullptr;
    int *need_list = nullptr;  // RoundArgs::need_list
    Ctl *ctl = nullptr;
    double *contrib = nullptr;
    int *nmatch = nullptr;
    FinSlot *fin_slots = nullptr;  // per-workgroup results of the final pass (kernels_check.hpp)
    int fin_slots_n = 0;
    unsigned long long *launch_edges = nullptr;
    int launch_edges_cap = 0;
    // tile-major second copy of the edges for k_bid_tiled (kernels_tiled.hpp)
    unsigned *tiled = nullptr;  // packed records, two edges each, in format tiled_fmt (kernels_tiled.hpp: TileFmt)
    int tiled_fmt = 0;          // 0: 6 B/edge {slot, f32}; 1: 10 B/edge {slot, f64}; 2 / 3: + the stored index (unsorted rows)
    int *seg4 = nullptr;  // k_bid_tiled's 4-byte segment table
    int *tcol = nullptr;  // real columns of the tile-major copy (k_bid_tiled stores LDS offsets in `tiled`)
    int *ovf_ptr = nullptr;  // per-person lists of the edges beyond ovf_cap in a (person, tile) segment
    int4 *ovf_q = nullptr;   // ... their entries {tile-major position, column, value bits, 0}
    int ovf_cap = 0;
    double2 *part_vw = nullptr;  // column-split launch shapes: per-(share of the tiles, bidder slot) partial top-2
    int *part_g = nullptr;
    unsigned long long *wg_stats = nullptr;  // RoundArgs::wg_stats (statistics of the bid kernels, a slot per workgroup)
    int wg_stats_slots = 0;                  // ... allocated
    int take_edges_n = 0;                    // a profiled bid launch whose edge count the round's k_tiebreak still has to
    unsigned long long *take_edges_out = nullptr;  // add up: its grid and where the count goes
    int *split_cnt = nullptr;    // ... and the arrival counter of every slice 
    int n_tiled = 0;  // entries of `tiled` including the padding entries
    int T = 0;
    bool tiled_ok = false;
    int tiled_min_K = 0;
    int tiled_shape = 0;  // index into kShapes of launch_bid_tiled
    std::vector<Blk> blocks;  // device memory of the arrays above (DevBlock), released as a whole
    Ctl *h_ctl = nullptr;  // pinned mirror
    Ctl *h_stat = nullptr;  // pinned [2]: status copies that trail the grid rounds by one batch (status_enqueue)
    // live status (device_common.hpp, post_live_status): four pinned words behind the mirrors, the ticket of the last
    // round-closing launch, and whether everything enqueued since the last full read is covered by that ticket
    volatile unsigned long long *live = nullptr;
    unsigned long long *live_dev = nullptr;  // the same words as the device addresses them
    unsigned ticket = 0;
    bool live_valid = false;
    bool live_off = false;      // MISSLAP_LIVE_STATUS=0, or a poll has timed out: status reads by copy + stream drain
    bool live_every_round = false;  // MISSLAP_LIVE_STATUS=2 (A/B): k_round_small posts as well
    unsigned slot_ticket[2] = {0, 0};
    bool slot_live[2] = {false, false};
    hipEvent_t stat_ev[2] = {nullptr, nullptr};
    // scalar solver state (auction_.pyx:180-187)
    float eps = 0, target_eps = 0, theta = 0, start_eps = 0;
    int nreductions = 0;
    bool finished = false;
    int64_t max_iter = 0;
    int thr = -1;
    bool round_small = false;  // the current round's bids skip the global atomicMax and k_round_small finishes it
    bool round_fused = true;   // ... in the same launch (k_round_fused); MISSLAP_ROUND_FUSED=0: two launches
    bool round_done = false;   // the bid launch of the current round has closed it
    int cand_build_max_K = 0x7fffffff;
    int tail_round_budget = kLongRowTailBudget;
    // No tail kernel instance runs more than so many rounds before it returns to the host (which looks at the status and
    // launches again): a degenerate instance -- integer costs of 1e7 against eps of 1e-5 are price wars of 1e10 rounds --
    // would otherwise sit in ONE launch for minutes (until max_iter).  ~2-4 s of rounds; MISSLAP_TAIL_LAUNCH_ROUNDS (read
    // per create) changes it, tests use a small value.
    int tail_launch_rounds = 1 << 22;
    int max_row_len = 0;
    long long avg_row_len = 0;
    // Lines in USE: the handle has them and its rows can keep one -- rows of at most kCandRowMax edges, or longer ones
    // once the long-row builder runs (k_refresh_long).  Otherwise (C4: 300 edges per row) every line is empty for the
    // whole solve, and loading + evaluating it in front of every bid, and the maintenance pass over all of them, are
    // pure overhead: the kernels then run as for a handle without lines.
    bool lines_live() const { return cand != nullptr && !lines_dropped && (avg_row_len <= kCandRowMax || long_rows); }
    bool long_rows_later = false;  // rows of a few hundred edges: k_refresh_long only if the tail turns out long
    long long tail_rounds_host = 0;  // rounds the tail kernels have run so far, from the round counts of the status reads
    long long tail_nits0 = -1;       // (the control block's own counter reaches the host with a full read only)
                                     // round count in front of the tail launches whose rounds are not yet counted; -1: none
    bool long_rows = false;  // some row is longer than kCandRowMax: k_refresh_long has work
    bool line_maintenance = true;  // k_refresh_lines ahead of the tail kernels (options.reserved[4] = 2: off)
    int cand_refresh_min = kDefaultCandRefresh;
    bool round_ordered = false;  // the current round's bidders were taken in person order (k_order_*, partial tiled rounds)
    int apply_bidders_ratio = 2;  // k_apply_bidders while K * ratio <= M
    bool profile_all = false;  // profile >= 2: events around every k_bid launch, not only the full scans
    int rounds_per_sync = kDefaultRoundsPerSync;
    bool rounds_per_sync_auto = true;  // not set by the caller: kRoundsPerSyncLive while the live status is in use
    int rank = 0, world = 1;
    long long sharded_rounds = 0;  // rounds of the last solve that were sharded and exchanged (misslap_solve_sharded)
    int phases_run = 0, phases_with_lines = 0;  // eps-phases begun so far / of which with candidate lines in use
    int shard_min_K = 0;  // multi-GPU: only rounds with K >= this are sharded and exchanged
    bool profile = false;
    int K_ub = 0;  // host-side upper bound of K (K never grows inside a phase)
    bool K_exact = false;  // K_ub was read from the device and no round has been enqueued since
    bool walk_rev_next = false;  // the next engine launch of this eps-phase walks the column tiles backwards (walk_backwards)
    bool ece_flag_clear = false;  // Ctl::ece_fail is 0 on the device (k_init_state, k_reset_phase) and no test has run since
    int ctl_fresh = 0;  // nothing enqueued since the last read and the pinned mirror h_ctl holds: 2 = the device's whole
                        // control block (read_ctl), 1 = its K / nits / error bits (a live status read), 0 = neither
    bool phase_fresh = true;  // no round of the current eps-phase has been enqueued yet
    std::vector<ProfRec> prof;
    size_t prof_used = 0;
    int launch_idx = 0;
    double setup_ms = 0, solve_ms = 0;
};
