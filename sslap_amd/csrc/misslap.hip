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
// MISSLAP_TRACE_CREATE=1: where a handle's setup time goes (stderr, one line per stage; the stream is drained at every
// stage, so the sum is a little above an untraced create)
struct CreateTrace {
    bool on;
    double t0;
    hipStream_t st;
    explicit CreateTrace(hipStream_t s) : st(s) {
        const char *e = std::getenv("MISSLAP_TRACE_CREATE");
        on = e && e[0] == '1';
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
#define MISSLAP_FOR_FMT_LANES(X) X(1, 4) X(1, 8) X(1, 16) X(2, 4) X(2, 8) X(2, 16) X(3, 4) X(3, 8) X(3, 16)
inline size_t tiled_lds_bytes(int tile_cols) {  // see the LDS map in k_bid_tiled; + statistics scratch
    const size_t doubles = tile_cols == kTileColsBig ? (size_t)tile_cols + 2 : 2 * (size_t)tile_cols + 128;
    return doubles * sizeof(double) + kTileStatBytes + kTileTouchBytes;  // + statistics scratch (incl. the arrival word of a column-split shape) + the loaders' touch scratch
}

// Profiled launches (options.profile): the two events are handed to the launch itself (hipExtLaunchKernel), so they
// carry the begin / end timestamps of the KERNEL -- what a rocprofv3 kernel trace reports.  Events recorded around a
// launch on the stream bracket the dispatch gap as well (~7 us per launch at C3: 92.4 against 85.6 us in round 2).
// MISSLAP_PROFILE_PLAIN_EVENTS=1 selects the bracketing form (A/B of the two clocks).
inline bool plain_events() {
    static const bool v = [] {
        const char *e = std::getenv("MISSLAP_PROFILE_PLAIN_EVENTS");
        return e && e[0] == '1';
    }();
    return v;
}
#define MISSLAP_LAUNCH_TIMED(PR, KERNEL, GRID, BLOCK, LDS, STREAM, ...)                              \
    do {                                                                                            \
        if ((PR) && !plain_events()) {                                                              \
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

struct misslap_solver {
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
    bool order_partial = true;   // ... which options.reserved[5] = 1 turns off (A/B, parity tests)
    int apply_bidders_ratio = 2;  // k_apply_bidders while K * ratio <= M (env MISSLAP_APPLY_BIDDERS_RATIO: A/B; huge = never)
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
    bool ece_flag_clear = false;  // Ctl::ece_fail is 0 on the device (k_init_state, k_reset_phase) and no test has run since
    int ctl_fresh = 0;  // nothing enqueued since the last read and the pinned mirror h_ctl holds: 2 = the device's whole
                        // control block (read_ctl), 1 = its K / nits / error bits (a live status read), 0 = neither
    bool phase_fresh = true;  // no round of the current eps-phase has been enqueued yet
    std::vector<ProfRec> prof;
    size_t prof_used = 0;
    int launch_idx = 0;
    double setup_ms = 0, solve_ms = 0;
};

namespace {

template <class T>
int dev_alloc(T **p, size_t n) {
    HIP_TRY(hipMalloc((void **)p, (n ? n : 1) * sizeof(T)));
    return MISSLAP_OK;
}

// Device temporaries of a constructor: freed when the scope is left, on every path.
struct DevScratch {
    std::vector<Blk> blks;
    int device = 0;
    bool drained = false;  // set by the owner after it has synchronised the stream(s) that used the blocks
    DevScratch() { (void)hipGetDevice(&device); }
    DevScratch(const DevScratch &) = delete;
    DevScratch &operator=(const DevScratch &) = delete;
    ~DevScratch() {
        // The blocks go back to a process-wide cache (not through hipFree, which would synchronise): on an error
        // return kernels may still be running on them, and another thread's handle could be handed that memory.
        if (!drained && !blks.empty()) (void)hipDeviceSynchronize();
        for (const Blk &b : blks) block_free(device, b.p, b.bytes);
    }
    template <class T>
    int alloc(T **p, size_t n) {
        Blk b;
        const int rc = block_alloc(&b.p, (n ? n : 1) * sizeof(T), &b.bytes);
        if (rc == MISSLAP_OK) {
            *p = static_cast<T *>(b.p);
            blks.push_back(b);
        }
        return rc;
    }
};

// Several device arrays carved from ONE hipMalloc (256-byte aligned): hipMalloc / hipFree cost tens of microseconds
// each and a handle holds some thirty arrays -- allocated one by one they are a fifth of the time it takes to set a
// 40 M-edge problem up.  `want` registers an array, `commit` allocates and hands the pointers out; the block is
// released as a whole (by the handle: misslap_solver::blocks, or by a DevScratch).
struct DevBlock {
    struct Item {
        void **target;
        size_t bytes;
    };
    std::vector<Item> items;
    template <class T>
    void want(T **p, size_t n) {
        items.push_back({reinterpret_cast<void **>(p), (n ? n : 1) * sizeof(T)});
    }
    int commit(Blk *out) {
        size_t total = 0;
        for (const Item &it : items) total += (it.bytes + 255) & ~(size_t)255;
        const int rc = block_alloc(&out->p, total ? total : 256, &out->bytes);
        if (rc) return rc;
        char *base = static_cast<char *>(out->p);
        size_t off = 0;
        for (const Item &it : items) {
            *it.target = base + off;
            off += (it.bytes + 255) & ~(size_t)255;
        }
        items.clear();
        return MISSLAP_OK;
    }
};

// Host-side resources are kept across handles: creating a stream (a hardware queue) takes several milliseconds -- more
// than everything else a handle's setup does --, the pinned mirror of the control block and the two status events
// another tenth of a millisecond.  A destroyed handle parks its idle bundle here; the next handle on that device takes it.
struct HostRes {
    hipStream_t stream = nullptr;
    Ctl *h_ctl = nullptr;  // pinned, 3 blocks: the mirror and the two trailing status copies
    hipEvent_t ev[2] = {nullptr, nullptr};
};
struct HostResPool {
    std::mutex m;
    std::vector<std::pair<int, HostRes>> idle;
    static constexpr size_t kMaxIdle = 8;
    bool take(int device, HostRes *out) {
        std::lock_guard<std::mutex> g(m);
        for (size_t k = 0; k < idle.size(); ++k)
            if (idle[k].first == device) {
                *out = idle[k].second;
                idle.erase(idle.begin() + (long)k);
                return true;
            }
        return false;
    }
    bool park(int device, const HostRes &r) {
        std::lock_guard<std::mutex> g(m);
        if (idle.size() >= kMaxIdle) return false;
        idle.emplace_back(device, r);
        return true;
    }
};
HostResPool &host_pool() {
    static HostResPool *pool = new HostResPool();  // never destroyed: the HIP runtime may be gone at static teardown
    return *pool;
}

// ... and so are small device blocks: hipMalloc + hipFree of a handle's four blocks cost a quarter of a millisecond,
// which is what a 20 x 20 problem takes to SOLVE.  Freed blocks of at most kMaxEach bytes wait here (at most
// kMaxEntries, kMaxHeld bytes in total) for a request they fit within a factor of two.
struct BlockCache {
    struct Ent {
        int device;
        size_t bytes;
        void *p;
    };
    std::mutex m;
    std::vector<Ent> idle;
    size_t held = 0;
    // limits (misslap_set_cache_limits; MISSLAP_BLOCK_CACHE_MB in the environment sets the first two at start-up).  The
    // defaults -- 4 GB of a 288 GB device, blocks of up to 1 GB -- hold the blocks of one or two problems of the
    // BASELINE sizes (C3: 0.9 GB per handle): hipMalloc + hipFree of those cost a millisecond per create / destroy pair
    // (C4: setup 4.0 -> 3.0 ms), and hipFree waits for every stream of the device, i.e. for other solves' kernels.  An
    // application that solves many large problems at a time raises them further
    size_t kMaxHeld = (size_t)4 << 30, kMaxEach = (size_t)1 << 30, kMaxEntries = 64;
    bool explicit_limits = false;  // set by the caller (misslap_set_cache_limits / MISSLAP_BLOCK_CACHE_MB)
    bool sized = false;            // the default total has been bounded by the device's memory (first block parked)
    BlockCache() {
        if (const char *e = std::getenv("MISSLAP_BLOCK_CACHE_MB")) {
            const long long mb = std::atoll(e);
            if (mb >= 0) {
                kMaxHeld = (size_t)mb << 20;
                kMaxEach = kMaxHeld;
                kMaxEntries = 4096;
                explicit_limits = true;
            }
        }
    }
    // the DEFAULT total never exceeds 1 / 64 of the device's memory (4 GB of an MI355X's 288 GB; 1 GB of a 64 GB part)
    void size_default(size_t device_bytes) {
        std::lock_guard<std::mutex> g(m);
        if (explicit_limits || sized) return;
        sized = true;
        kMaxHeld = std::min(kMaxHeld, device_bytes / 64);
        kMaxEach = std::min(kMaxEach, kMaxHeld / 4);
    }
    void *take(int device, size_t bytes, size_t *got) {
        std::lock_guard<std::mutex> g(m);
        size_t best = idle.size();
        for (size_t k = 0; k < idle.size(); ++k)
            if (idle[k].device == device && idle[k].bytes >= bytes && idle[k].bytes <= 2 * bytes + 4096 &&
                (best == idle.size() || idle[k].bytes < idle[best].bytes))
                best = k;
        if (best == idle.size()) return nullptr;
        void *p = idle[best].p;
        *got = idle[best].bytes;
        held -= idle[best].bytes;
        idle.erase(idle.begin() + (long)best);
        return p;
    }
    bool give(int device, void *p, size_t bytes) {
        if (bytes == 0 || bytes > kMaxEach) return false;
        std::lock_guard<std::mutex> g(m);
        if (idle.size() >= kMaxEntries || held + bytes > kMaxHeld) return false;
        idle.push_back({device, bytes, p});
        held += bytes;
        return true;
    }
};
BlockCache &block_cache() {
    static BlockCache *c = new BlockCache();
    return *c;
}
// a device block of at least `bytes` on the current device, from the cache if one fits; *got = its real size
int block_alloc(void **p, size_t bytes, size_t *got) {
    int dev = 0;
    HIP_TRY(hipGetDevice(&dev));
    *p = block_cache().take(dev, bytes, got);
    if (*p) return MISSLAP_OK;
    HIP_TRY(hipMalloc(p, bytes));
    *got = bytes;
    return MISSLAP_OK;
}
void block_free(int device, void *p, size_t bytes) {
    BlockCache &bc = block_cache();
    if (p && !bc.sized && !bc.explicit_limits) {
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) == hipSuccess && total_b > 0) bc.size_default(total_b);
    }
    if (p && !bc.give(device, p, bytes)) (void)hipFree(p);
}

// The start of an eps-phase (create: the first; misslap_phase_end: every later one): are the candidate lines still exact
// at this phase's eps?  (fp32 eps promoted to double exactly as the bid does, auction_.pyx:360.)
void begin_phase(misslap_solver *h) {
    if (h->cand != nullptr && !h->lines_dropped && (double)h->eps < h->lines_safe_eps) h->lines_dropped = true;
    h->phases_run += 1;
    h->phases_with_lines += h->lines_live() ? 1 : 0;
}

RoundArgs round_args(misslap_solver *h) {
    RoundArgs a;
    a.ctl = h->ctl;
    a.row_ptr = h->row_ptr;
    a.price = h->price;
    a.rec = h->rec;
    a.p2o = h->p2o;
    a.o2p = h->o2p;
    a.U = h->U;
    a.bid_key = h->bid_key;
    a.bid_obj = h->bid_obj;
    a.bid_rec = h->bid_rec;
    a.best_key = h->best_key;
    a.best_pos = h->best_pos;
    a.cnt = h->cnt;
    a.hole_list = h->hole_list;
    a.mover_list = h->mover_list;
    a.launch_edges = h->profile ? h->launch_edges : nullptr;
    a.n_rows = h->n_rows;
    a.n_cols = h->n_cols;
    a.thr = h->thr;
    a.rank = h->rank;
    a.world = h->world;
    a.shard_min_K = h->world > 1 ? h->shard_min_K : 0;
    a.eps = h->eps;
    a.launch_idx = 0;
    a.wg_stats = h->wg_stats;
    a.need_list = h->need_list;
    a.live = nullptr;
    a.ticket = 0;
    a.gather_max_K = h->tiled_ok ? h->tiled_min_K : 0;
    a.cand = h->lines_live() ? h->cand : nullptr;
    a.cand64 = h->lines_live() ? h->cand64 : nullptr;
    a.cand_build_max_K = h->cand_build_max_K;
    static const int build_min_env = [] {
        const char *e = std::getenv("MISSLAP_BUILD_MIN_K");
        return e ? std::atoi(e) : kRoundSmallMax;  // (same box, 2048 vs 0: C3 400.1 vs 401.4 ms, C2 132.0 vs 132.5, C1 9.69 vs 9.85)
    }();
    a.cand_build_min_K = (h->thr > 0 && h->line_maintenance) ? build_min_env : 0;  // (no maintenance pass: nobody else rebuilds)
    a.cand_refresh_min = h->cand_refresh_min;
    a.price32 = nullptr;  // (set by launch_bid for the launches that scan through the filter)
    a.pmax_bits = h->pmax_bits;
    a.cmax = h->cmax32;
    return a;
}

int blocks_for(long long items, int per_block) {
    long long b = (items + per_block - 1) / per_block;
    if (b < 1) b = 1;
    if (b > kMaxGridBlocks) b = kMaxGridBlocks;
    return (int)b;
}

ProfRec *prof_next(misslap_solver *h, int kind) {
    if (h->prof_used == h->prof.size()) {
        ProfRec r{};
        if (hipEventCreate(&r.start) != hipSuccess || hipEventCreate(&r.stop) != hipSuccess) return nullptr;
        h->prof.push_back(r);
    }
    ProfRec *r = &h->prof[h->prof_used++];
    r->kind = kind;
    r->fullscan = 0;
    r->launch_idx = -1;
    return r;
}

// (A status read is a stream drain: ~20 us.  A solve of a small problem is a few hundred rounds of ~1 us inside one
// tail launch per eps-phase and was spending most of its time in the five reads per phase; with the mirror reused
// while nothing has been enqueued since the last read, two remain.)
// (behind every status read) the rounds of the tail launches enqueued before it
void count_tail_rounds(misslap_solver *h) {
    if (h->tail_nits0 < 0) return;
    h->tail_rounds_host += h->h_ctl->nits - h->tail_nits0;
    h->tail_nits0 = -1;
}

int read_ctl(misslap_solver *h) {
    if (h->ctl_fresh == 2) {
        if (h->h_ctl->err)
            return fail(MISSLAP_ERR_STATE, "device-side invariant violated (error bits 0x%x)", h->h_ctl->err);
        return MISSLAP_OK;
    }
    HIP_TRY(hipMemcpyAsync(h->h_ctl, h->ctl, sizeof(Ctl), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    h->ctl_fresh = 2;
    h->K_ub = h->h_ctl->K;
    h->K_exact = true;
    count_tail_rounds(h);
    if (h->h_ctl->err)
        return fail(MISSLAP_ERR_STATE, "device-side invariant violated (error bits 0x%x)", h->h_ctl->err);
    return MISSLAP_OK;
}

// Wait for the status a round-closing launch posts (post_live_status): exact = the ticket `want` itself, otherwise any
// ticket at or behind it.  Returns false on a timeout (the caller falls back to a copy + drain and stops using the words).
bool live_poll(misslap_solver *h, unsigned want, bool exact, int *K, int *err, long long *nits) {
    volatile unsigned long long *w = h->live;
    const double t_end = now_ms() + 20000.0;
    for (unsigned spins = 0;; ++spins) {
        const unsigned long long a = w[0], b = w[1], c = w[2], d = w[3];
        const unsigned t = (unsigned)(a >> 32);
        if ((unsigned)(b >> 32) == t && (unsigned)(c >> 32) == t && (unsigned)(d >> 32) == t && w[0] == a &&
            (exact ? t == want : (int)(t - want) >= 0)) {
            *K = (int)(unsigned)(a & 0xffffffffull);
            *err = (int)(unsigned)(b & 0xffffffffull);
            *nits = (long long)((c & 0xffffffffull) | ((d & 0xffffffffull) << 32));
            return true;
        }
        if (spins < 4000) {
            __builtin_ia32_pause();
        } else {
            std::this_thread::yield();  // (a tail kernel runs for milliseconds: do not burn a core another solve needs)
            if ((spins & 1023) == 0 && now_ms() > t_end) return false;
        }
    }
}
// K / nits / error bits of everything enqueued so far, into the mirror's fields: from the live words where the last
// thing enqueued that changes them was a ticketed launch, by a full read otherwise.
// the live words cover everything enqueued: if the last launch that changed K / nits carried no ticket, one that only
// posts the status is enqueued behind it
void ensure_posted(misslap_solver *h) {
    if (h->live_valid || h->live_off) return;
    hipLaunchKernelGGL(k_post_status, dim3(1), dim3(1), 0, h->stream, h->ctl, h->live_dev, ++h->ticket);
    h->live_valid = true;
}
int read_status(misslap_solver *h) {
    if (h->ctl_fresh == 2 || h->live_off) return read_ctl(h);
    if (h->ctl_fresh == 1) {  // (K, nits and the error bits of the mirror are current)
        if (h->h_ctl->err) return fail(MISSLAP_ERR_STATE, "device-side invariant violated (error bits 0x%x)", h->h_ctl->err);
        return MISSLAP_OK;
    }
    ensure_posted(h);
    int K = 0, err = 0;
    long long nits = 0;
    if (!live_poll(h, h->ticket, true, &K, &err, &nits)) {
        h->live_off = true;
        return read_ctl(h);
    }
    h->h_ctl->K = K;
    h->h_ctl->nits = nits;
    h->h_ctl->err = err;
    h->K_ub = K;
    h->K_exact = true;
    count_tail_rounds(h);
    if (err) return fail(MISSLAP_ERR_STATE, "device-side invariant violated (error bits 0x%x)", err);
    return MISSLAP_OK;
}

// Status of the round loop WITHOUT draining the stream: a copy of the control block is enqueued behind a batch of
// rounds and read while the next batch runs.  K never grows inside an eps-phase, so a status that is one batch old
// is still an upper bound for the launch grids, and every round kernel is a no-op once the round is not live: a
// batch enqueued on a stale "go on" costs its launches and nothing else.
int status_enqueue(misslap_solver *h, int slot) {
    h->slot_live[slot] = !h->live_off;
    if (h->slot_live[slot]) {  // no copy: the closing kernel of the batch's last round has posted, or k_post_status does
        ensure_posted(h);
        h->slot_ticket[slot] = h->ticket;
        return MISSLAP_OK;
    }
    h->ctl_fresh = false;
    HIP_TRY(hipMemcpyAsync(&h->h_stat[slot], h->ctl, sizeof(Ctl), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipEventRecord(h->stat_ev[slot], h->stream));
    return MISSLAP_OK;
}
int status_wait(misslap_solver *h, int slot) {
    if (h->slot_live[slot]) {
        int K = 0, err = 0;
        long long nits = 0;
        if (live_poll(h, h->slot_ticket[slot], false, &K, &err, &nits)) {
            h->h_stat[slot].K = K;
            h->h_stat[slot].nits = nits;
            h->h_stat[slot].err = err;
            h->K_ub = K;
            h->K_exact = false;
            if (err) return fail(MISSLAP_ERR_STATE, "device-side invariant violated (error bits 0x%x)", err);
            return MISSLAP_OK;
        }
        h->live_off = true;  // timed out: drain the stream and read the control block
        int rc = read_ctl(h);
        h->h_stat[slot] = *h->h_ctl;
        h->K_exact = false;
        return rc;
    }
    HIP_TRY(hipEventSynchronize(h->stat_ev[slot]));
    const Ctl &c = h->h_stat[slot];
    h->K_ub = c.K;
    h->K_exact = false;  // rounds have been enqueued behind this copy
    if (c.err) return fail(MISSLAP_ERR_STATE, "device-side invariant violated (error bits 0x%x)", c.err);
    return MISSLAP_OK;
}

int launch_bid_tiled(misslap_solver *h) {
    h->ctl_fresh = false;
    RoundArgs a = round_args(h);
    // K_ub is only an upper bound unless the host has just read K: the device decides sharded / replicated from
    // the exact K, so the smaller sharded grid is used only when the host knows the same K
    const bool sharded = h->world > 1 && h->K_exact && h->K_ub >= h->shard_min_K;
    const long long share = sharded ? ((long long)h->K_ub + h->world - 1) / h->world : h->K_ub;
    const int *shp = kTiledShapes[h->tiled_shape];
    const int groups = (shp[0] - 64 * shp[5]) / shp[6];  // lane groups; loader wavefronts own no persons
    const int per_wg_max = groups * shp[1];
    const int cs = shp[7];  // column split: `cs` workgroups share a slice of bidders, each with 1 / cs of the tiles
    long long grid = (share + per_wg_max - 1) / per_wg_max;
    const long long resident = h->n_cus / cs;  // one workgroup per CU: its two price tiles take the whole LDS
    const long long spread = std::min<long long>(resident, (share + groups - 1) / groups);
    if (grid < spread) grid = spread;
    grid *= cs;
    TiledArgs ta{h->tiled, h->tcol, h->seg4, h->T, h->tiled_min_K, h->n_tiled,
                 nullptr, nullptr, h->ovf_ptr, h->ovf_q, h->ovf_cap, h->part_vw, h->part_g, h->n_rows, h->split_cnt, FinalOut{}};
    // A partial round whose K the host knows: bidders in person order (kernels_tiled.hpp, k_order_*).  Scratch that
    // is idle during a bid phase: the compaction lists (the tie-break reads order_pos before they are rewritten),
    // the chunk counters, the objective's match counters.
    // (K < N in every round of a phase but the first: every winner of the first round takes an unowned object.  The
    // ordering kernels and the scan take K from the device, so the host need not know it exactly.)
    h->round_ordered = h->order_partial && !h->phase_fresh;
    if (h->round_ordered) {
        int *pos_of = h->nmatch, *order_person = h->hole_list, *order_pos = h->mover_list, *sums = h->cnt;
        const int nchunks = (h->n_rows + kScanChunk - 1) / kScanChunk;
        // (each returns at once when the scan itself will: a round enqueued on a stale upper bound of K)
        hipLaunchKernelGGL(k_order_prepare, dim3(std::max(nchunks, blocks_for(h->K_ub, 1024))), dim3(1024), 0, h->stream, h->ctl, h->U,
                           pos_of, h->p2o, h->n_rows, nchunks, sums, h->thr, h->tiled_min_K);
        hipLaunchKernelGGL(k_order_scatter, dim3(nchunks), dim3(1024), 0, h->stream, h->ctl, h->p2o, h->n_rows, sums, pos_of,
                           order_person, order_pos, h->thr, h->tiled_min_K);
        ta.order_person = order_person;
        ta.order_pos = order_pos;
    }
    if (grid > h->wg_stats_slots) return fail(MISSLAP_ERR_STATE, "scan grid %lld exceeds the statistics slots (%d)", grid, h->wg_stats_slots);
    ProfRec *pr = nullptr;
    if (h->profile) {
        if (h->launch_idx >= h->launch_edges_cap)
            return fail(MISSLAP_ERR_STATE, "profile buffer exhausted (%d bid launches)", h->launch_idx);
        pr = prof_next(h, 2);
        if (!pr) return fail(MISSLAP_ERR_HIP, "hipEventCreate failed");
        pr->fullscan = h->phase_fresh;  // K == N; with several ranks: this rank's share of the full scan
        pr->launch_idx = a.launch_idx = h->launch_idx++;
    }
    const size_t lds = tiled_lds_bytes(shp[4]);
    const dim3 g((unsigned)grid);
    if (h->tiled_fmt == 0) {
        switch (h->tiled_shape) {
#define X(I, TH, R, B, D, TC, LD, GL, CS) \
    case I: MISSLAP_LAUNCH_TIMED(pr, (k_bid_tiled<TH, R, B, D, TC, LD, 0, GL, CS>), g, dim3(TH), (unsigned)lds, h->stream, a, ta); break;
            MISSLAP_FOR_TILED_SHAPES(X)
#undef X
            default: return fail(MISSLAP_ERR_STATE, "bad tiled shape");
        }
    } else {  // formats 1..3 (fp64 values / unsorted rows): the three production shapes, 4 / 8 / 16 lanes per person
        const int key = h->tiled_fmt * 100 + shp[6];
        switch (key) {
#define X(FMT, GL) \
    case FMT * 100 + GL: MISSLAP_LAUNCH_TIMED(pr, (MISSLAP_BID_KERNEL_FMT(GL, FMT)), g, dim3(1024), (unsigned)lds, h->stream, a, ta); break;
            MISSLAP_FOR_FMT_LANES(X)
#undef X
            default: return fail(MISSLAP_ERR_STATE, "no full-scan instance for format %d with %d lanes per person", h->tiled_fmt, shp[6]);
        }
    }
    if (pr) {  // (the round's k_tiebreak adds the workgroups' counts up: no launch of its own inside a timed solve)
        h->take_edges_n = (int)grid;
        h->take_edges_out = h->launch_edges + 2 * (size_t)pr->launch_idx;
    }
    HIP_TRY(hipGetLastError());
    return MISSLAP_OK;
}

// rounds with few bidders that are not sharded over GPUs: tiebreak, apply and compaction in one launch
bool use_round_small(const misslap_solver *h) {
    return h->K_ub <= kRoundSmallMax && (h->world == 1 || h->K_ub < h->shard_min_K);
}

int launch_bid(misslap_solver *h) {
    h->ctl_fresh = false;
    // (not behind a full-scan engine launch: the engines always feed best_key, which k_round_small ignores)
    h->round_small = use_round_small(h) && !(h->tiled_ok && h->K_ub >= h->tiled_min_K);
    if (h->tiled_ok && h->K_ub >= h->tiled_min_K) {
        int rc = launch_bid_tiled(h);  // no-op on the device when K < tiled_min_K
        if (rc) return rc;
        if (h->K_exact) {  // the host has just read K: k_bid would be a no-op
            h->phase_fresh = false;
            h->K_exact = false;
            return MISSLAP_OK;
        }
    }
    h->K_exact = false;
    RoundArgs a = round_args(h);
    const long long share = h->K_ub;  // upper bound: unsharded rounds bid for every list position
    // a round with few bidders is ONE launch (k_round_fused: bids by 16-wavefront workgroups, the rest by the last of them)
    const bool fused = h->round_small && h->round_fused;
    const int grid = blocks_for(share, (fused ? 1024 : kBidBlock) / kWave);
    ProfRec *pr = nullptr;
    // profile 1 times the full scans only (two event records around each of the ~3000 small launches of a solve
    // cost more host time than the launches themselves); profile 2 / 3 time every launch
    const bool fullscan = h->phase_fresh && !(h->tiled_ok && h->K_ub >= h->tiled_min_K);
    if (!(h->profile && (h->profile_all || fullscan))) a.launch_edges = nullptr;
    if (h->profile && (h->profile_all || fullscan)) {
        if (h->launch_idx >= h->launch_edges_cap)
            return fail(MISSLAP_ERR_STATE, "profile buffer exhausted (%d bid launches)", h->launch_idx);
        pr = prof_next(h, 0);
        if (!pr) return fail(MISSLAP_ERR_HIP, "hipEventCreate failed");
        pr->fullscan = fullscan;
        pr->launch_idx = a.launch_idx = h->launch_idx++;
    }
    const EdgesF32 e32{h->edges32};
    const EdgesF64 e64{h->col, h->val64};
    const dim3 g(grid), b(fused ? 1024 : kBidBlock);
    if (fused) {  // the launch closes the round: it carries the round's ticket (launch_apply has nothing left to do)
        a.live = (h->live_off || !h->live_every_round) ? nullptr : h->live_dev;
        a.ticket = ++h->ticket;
        h->live_valid = a.live != nullptr;
    }
    // variant: 2 = lines used and rebuilt; 1 = lines used, lean scan, nothing built (the full-scan regime); 0 = no lines
    const int variant = !h->lines_live() ? 0 : h->K_ub > h->cand_build_max_K ? 1 : 2;
    // big rounds of a handle whose price table does not fit an XCD's L2: the lean scans go through the fp32 filter
    // (wave_bid_filter); the mirror is rebuilt from the prices in front of the launch (12 bytes per object)
    if (h->price32 && !h->round_small && variant != 2 && (long long)h->K_ub * 8 >= h->n_rows) {
        HIP_TRY(hipMemsetAsync(h->pmax_bits, 0, sizeof(int), h->stream));
        hipLaunchKernelGGL(k_price_mirror, dim3(blocks_for(h->n_cols, 256 * 4)), dim3(256), 0, h->stream, h->ctl, h->price,
                           h->price32, h->n_cols, h->pmax_bits, h->thr, a.gather_max_K);
        a.price32 = h->price32;
    }
#define MISSLAP_LAUNCH_BID(E, ED)                                                                                   \
    do {                                                                                                            \
        if (fused) MISSLAP_LAUNCH_TIMED(pr, (k_round_fused<E>), g, b, 0, h->stream, a, ED);                         \
        else if (h->round_small) MISSLAP_LAUNCH_TIMED(pr, (k_bid<E, RecSource, 2>), g, b, 0, h->stream, a, ED);     \
        else if (variant == 0) MISSLAP_LAUNCH_TIMED(pr, (k_bid<E, PriceSource, 0>), g, b, 0, h->stream, a, ED);     \
        else if (variant == 1) MISSLAP_LAUNCH_TIMED(pr, (k_bid<E, PriceSource, 1>), g, b, 0, h->stream, a, ED);     \
        else MISSLAP_LAUNCH_TIMED(pr, (k_bid<E, PriceSource, 2>), g, b, 0, h->stream, a, ED);                       \
    } while (0)
    if (h->f32) MISSLAP_LAUNCH_BID(EdgesF32, e32);  // (rounds that k_round_small finishes: bids with the owners)
    else MISSLAP_LAUNCH_BID(EdgesF64, e64);
#undef MISSLAP_LAUNCH_BID
    if (pr && !h->round_small) {
        h->take_edges_n = std::max(h->take_edges_n, grid);  // (a full-scan engine launch of the same round may be pending too)
        h->take_edges_out = h->launch_edges + 2 * (size_t)pr->launch_idx;
    } else if (pr) {  // (no k_tiebreak in a round that k_round_small finishes; such launches are profiled at level 2 / 3 only)
        hipLaunchKernelGGL(k_take_launch_edges, dim3(1), dim3(1024), 0, h->stream, h->wg_stats, grid, h->launch_edges + 2 * (size_t)pr->launch_idx);
    }
    HIP_TRY(hipGetLastError());
    h->phase_fresh = false;
    h->round_done = fused;
    return MISSLAP_OK;
}

int launch_tiebreak(misslap_solver *h) {
    h->ctl_fresh = false;
    if (h->round_small) return MISSLAP_OK;  // k_round_small (launch_apply) resolves the ties itself
    RoundArgs a = round_args(h);
    const long long share = h->K_ub;
    hipLaunchKernelGGL(k_tiebreak, dim3(blocks_for(share, 256)), dim3(256), 0, h->stream, a,
                       h->round_ordered ? h->mover_list : nullptr, h->tiled_min_K, h->take_edges_n, h->take_edges_out);
    h->take_edges_n = 0;
    h->take_edges_out = nullptr;
    HIP_TRY(hipGetLastError());
    return MISSLAP_OK;
}

int launch_apply(misslap_solver *h) {
    h->ctl_fresh = false;
    if (h->round_small && h->round_done) {  // (k_round_fused has closed the round)
        h->round_small = h->round_done = false;
        h->K_exact = false;
        return MISSLAP_OK;
    }
    RoundArgs a = round_args(h);
    // (a round that k_round_small closes posts nothing: four stores to host memory are 1.5 us on a 3-5 us kernel that
    // runs thousands of times per solve -- a batch of such rounds is followed by k_post_status instead, ensure_posted)
    a.live = (h->live_off || (h->round_small && !h->live_every_round)) ? nullptr : h->live_dev;
    a.ticket = ++h->ticket;
    h->live_valid = a.live != nullptr;
    if (h->round_small) {
        h->round_small = false;
        h->K_exact = false;
        hipLaunchKernelGGL(k_round_small, dim3(1), dim3(1024), 0, h->stream, a);
        HIP_TRY(hipGetLastError());
        return MISSLAP_OK;
    }
    h->K_exact = false;
    h->round_ordered = false;
    // by the bidders where they are few against the objects (every rank holds every bid only in unsharded rounds)
    if ((h->world == 1 || h->K_ub < h->shard_min_K) && (long long)h->K_ub * h->apply_bidders_ratio <= h->n_cols)
        hipLaunchKernelGGL(k_apply_bidders, dim3(blocks_for(h->K_ub, 256)), dim3(256), 0, h->stream, a);
    else
        hipLaunchKernelGGL(k_apply, dim3(blocks_for(h->n_cols, 256)), dim3(256), 0, h->stream, a);
    if (h->K_ub <= kCompactSmallMax) {
        hipLaunchKernelGGL(k_compact_small, dim3(1), dim3(1024), 0, h->stream, a);
    } else {
        const int cb = blocks_for(h->K_ub, kChunk);
        hipLaunchKernelGGL(k_compact_count, dim3(cb), dim3(256), 0, h->stream, a);
        hipLaunchKernelGGL(k_compact_scatter, dim3(cb), dim3(256), 0, h->stream, a);
        hipLaunchKernelGGL(k_compact_fill, dim3(blocks_for(h->K_ub, 256)), dim3(256), 0, h->stream, a);
    }
    HIP_TRY(hipGetLastError());
    return MISSLAP_OK;
}

int launch_tail(misslap_solver *h) {
    if (h->thr <= 0) return MISSLAP_OK;
    h->ctl_fresh = false;
    // Rows of a few hundred edges keep no lines until the solve has shown that its tail is long: that many tail rounds
    // (a tail round without a line is a row scan by one wavefront, 1.5-4 us at 300-1000 edges; the pass that builds the
    // lines of every row costs milliseconds at C4's 100 000 rows and pays for itself within a phase at a dense
    // 1000 x 1000).  The tail kernels of such a handle return after as many rounds, so that a first phase with thousands
    // of tail rounds does not run to its end without lines (dense 1000^2: 13 of 16 ms were its first two tail launches).
    static const long long after_env = [] {
        const char *e = std::getenv("MISSLAP_LONG_AFTER_TAIL_ROUNDS");
        return e ? std::atoll(e) : -1ll;
    }();
    const long long long_after = after_env >= 0 ? after_env : std::max<long long>(kLongRowsAfterTailRoundsMin, h->n_rows / 64);
    if (h->long_rows_later && h->tail_rounds_host >= long_after) {  // (status read just before)
        h->long_rows = true;
        h->long_rows_later = false;
    }
    TailArgs a;
    a.ctl = h->ctl;
    a.row_ptr = h->row_ptr;
    a.price = h->price;
    a.rec = h->rec;
    a.p2o = h->p2o;
    a.o2p = h->o2p;
    a.U = h->U;
    const bool lines = h->lines_live();
    a.cand = lines ? h->cand : nullptr;
    a.cand64 = lines ? h->cand64 : nullptr;
    static const int budget_env = [] {
        const char *e = std::getenv("MISSLAP_LONG_TAIL_BUDGET");
        return e ? std::atoi(e) : 0;
    }();
    a.round_budget = h->long_rows && lines && h->line_maintenance ? (budget_env > 0 ? budget_env : h->tail_round_budget)
                     : h->long_rows_later                            ? (int)std::min<long long>(std::max<long long>(long_after, 1), 1 << 30)
                                                                     : 0;
    a.thr = h->thr;
    a.eps = h->eps;
    ProfRec *pr = nullptr;
    if (h->profile) {
        pr = prof_next(h, 1);
        if (!pr) return fail(MISSLAP_ERR_HIP, "hipEventCreate failed");
        HIP_TRY(hipEventRecord(pr->start, h->stream));
    }
    const EdgesF32 e32{h->edges32};
    const EdgesF64 e64{h->col, h->val64};
    // rows the long-row builder takes (it runs right behind the pass over all lines, on the list that pass leaves)
    const int long_max = !(lines && h->line_maintenance && h->long_rows) ? 0 : (h->max_row_len <= 256 * kLongPer ? 256 : 512) * kLongPer;
    static const int min_alive_long = [] {
        const char *e = std::getenv("MISSLAP_LONG_MIN_ALIVE");
        return e ? std::atoi(e) : kLongRowMinAlive;
    }();
    // every line checked at today's prices (kernels_round.hpp); then the rounds with more than kTeamMax bidders, with
    // sixteen wavefronts (kernels_tail.hpp); then -- lines only -- the rounds with 3..kTeamMax bidders, one list slot
    // per wavefront; then the rest: with lines the two-wavefront duo / chain instance, without them the 512-thread
    // instance that holds every mode
#define MISSLAP_LAUNCH_TAIL(E, ED)                                                                                       \
    do {                                                                                                                 \
        if (lines && h->line_maintenance)                                                                                \
            hipLaunchKernelGGL(k_refresh_lines<E>, dim3(blocks_for((h->n_rows + 1) / 2, kBidBlock / kWave)),             \
                               dim3(kBidBlock), 0, h->stream, round_args(h), ED, kCandMaintenanceMin, long_max, min_alive_long); \
        if (lines && h->line_maintenance && h->long_rows) {                                                              \
            if (h->max_row_len <= 256 * kLongPer)                                                                        \
                hipLaunchKernelGGL((k_refresh_long<E, 256>), dim3(blocks_for(h->n_rows, 1)), dim3(256), 0, h->stream,    \
                                   round_args(h), ED);                                                                   \
            else                                                                                                         \
                hipLaunchKernelGGL((k_refresh_long<E, 512>), dim3(blocks_for(h->n_rows, 1)), dim3(512), 0, h->stream,    \
                                   round_args(h), ED);                                                                   \
        }                                                                                                                \
        if (h->K_ub > kTeamMax)                                                                                          \
            hipLaunchKernelGGL((k_tail<E, 2 * kTailMax>), dim3(1), dim3(2 * kTailMax), 0, h->stream, a, ED);             \
        if (h->K_ub > 2 && lines)                                                                                        \
            hipLaunchKernelGGL((k_tail<E, 2 * kTailMax, true>), dim3(1), dim3(2 * kTailMax), 0, h->stream, a, ED);       \
        if (lines) hipLaunchKernelGGL((k_tail<E, 2 * kWave>), dim3(1), dim3(2 * kWave), 0, h->stream, a, ED);            \
        else hipLaunchKernelGGL((k_tail<E, kTailMax>), dim3(1), dim3(kTailMax), 0, h->stream, a, ED);                    \
    } while (0)
    if (h->f32) MISSLAP_LAUNCH_TAIL(EdgesF32, e32);
    else MISSLAP_LAUNCH_TAIL(EdgesF64, e64);
#undef MISSLAP_LAUNCH_TAIL
    if (pr) HIP_TRY(hipEventRecord(pr->stop, h->stream));
    // the tail keeps only the price records current: rebuild price / o2p / p2o from them
    h->live_valid = !h->live_off && h->live_dev != nullptr;
    hipLaunchKernelGGL(k_sync_from_rec, dim3(blocks_for(h->n_cols, 256)), dim3(256), 0, h->stream, h->ctl, h->rec, h->price,
                       h->o2p, h->p2o, h->U, h->n_cols, (h->cand != nullptr && !h->lines_dropped) ? 1 : 0,
                       h->live_valid ? h->live_dev : nullptr, ++h->ticket);
    HIP_TRY(hipGetLastError());
    h->phase_fresh = false;
    if (h->tail_nits0 < 0) h->tail_nits0 = h->h_ctl->nits;  // (the status read in front of this launch)
    return MISSLAP_OK;
}

// The pass over all rows behind eCE_satisfied / get_obj / the validity flags (kernels_check.hpp) runs on the
// full-scan engine where the handle has the tile-major copy in a shape the check instances cover: lanes per person of
// that shape (the overflow lists are built for 2 x lanes x 2 loads per segment), 0 = the pass on the row-major CSR.
int check_lanes(const misslap_solver *h) {
    if (!h->tiled_ok) return 0;
    const int *shp = kTiledShapes[h->tiled_shape];
    return (shp[3] == 2 && shp[4] == kTileColsHalf) ? shp[6] : 0;
}
#define MISSLAP_FOR_CHECK_LANES(X) X(4) X(8) X(16)
#define MISSLAP_CHECK_KERNEL(GL) k_bid_tiled<1024, 4, 2, 2, kTileColsHalf, 3, 0, GL, 1, 1>
#define MISSLAP_CHECK_KERNEL_FMT(GL, FMT) k_bid_tiled<1024, 4, 2, 2, kTileColsHalf, 3, 0, GL, 1, 1, FMT>

// rows [0, n_rows) on the row-major CSR (the sample of run_ece; every row where there is no tile-major copy)
// the largest grid launch_rows_all can ask for on n_rows persons (over every lanes-per-person shape of the check pass)
size_t final_pass_grid_max(size_t n_rows, int n_cus) {
    const size_t per_wg_min = (size_t)((1024 - 64 * 3) / 16) * 4;  // 16 lanes per person
    return std::max<size_t>((n_rows + per_wg_min - 1) / per_wg_min, (size_t)n_cus);
}
int launch_rows_gather(misslap_solver *h, float eps, const FinalOut &fo, int n_rows, int *n_blocks = nullptr) {
    const int grid = blocks_for(n_rows, 4);
    if (n_blocks) *n_blocks = grid;
    if (h->f32) {
        EdgesF32 ed{h->edges32};
        hipLaunchKernelGGL(k_ece<EdgesF32>, dim3(grid), dim3(256), 0, h->stream, h->ctl, ed, h->row_ptr, h->price,
                           h->p2o, n_rows, eps, fo);
    } else {
        EdgesF64 ed{h->col, h->val64};
        hipLaunchKernelGGL(k_ece<EdgesF64>, dim3(grid), dim3(256), 0, h->stream, h->ctl, ed, h->row_ptr, h->price,
                           h->p2o, n_rows, eps, fo);
    }
    HIP_TRY(hipGetLastError());
    return MISSLAP_OK;
}
// every row, on the engine the handle has
int launch_rows_all(misslap_solver *h, float eps, const FinalOut &fo, int *n_blocks = nullptr) {
    const int gl = check_lanes(h);
    if (!gl) return launch_rows_gather(h, eps, fo, h->n_rows, n_blocks);
    RoundArgs a = round_args(h);
    a.eps = eps;
    a.launch_edges = nullptr;
    TiledArgs ta{h->tiled, h->tcol, h->seg4, h->T, 0, h->n_tiled,
                 nullptr, nullptr, h->ovf_ptr, h->ovf_q, h->ovf_cap, nullptr, nullptr, h->n_rows, nullptr, fo};
    const int groups = (1024 - 64 * 3) / gl, per_wg_max = groups * 4;
    long long grid = ((long long)h->n_rows + per_wg_max - 1) / per_wg_max;
    const long long spread = std::min<long long>(h->n_cus, ((long long)h->n_rows + groups - 1) / groups);
    if (grid < spread) grid = spread;
    if (fo.fin && grid > h->fin_slots_n) return fail(MISSLAP_ERR_STATE, "final pass: grid %lld exceeds its result slots (%d)", grid, h->fin_slots_n);
    if (n_blocks) *n_blocks = (int)grid;
    const size_t lds = tiled_lds_bytes(kTileColsHalf);
    if (h->tiled_fmt == 0) {
        switch (gl) {
#define X(GL) \
    case GL: hipLaunchKernelGGL((MISSLAP_CHECK_KERNEL(GL)), dim3((unsigned)grid), dim3(1024), (unsigned)lds, h->stream, a, ta); break;
            MISSLAP_FOR_CHECK_LANES(X)
#undef X
        }
    } else {
        switch (h->tiled_fmt * 100 + gl) {
#define X(FMT, GL) \
    case FMT * 100 + GL: hipLaunchKernelGGL((MISSLAP_CHECK_KERNEL_FMT(GL, FMT)), dim3((unsigned)grid), dim3(1024), (unsigned)lds, h->stream, a, ta); break;
            MISSLAP_FOR_FMT_LANES(X)
#undef X
        }
    }
    HIP_TRY(hipGetLastError());
    return MISSLAP_OK;
}

// eCE_satisfied(eps), auction_.pyx:443-485.  The sample pass first (kernels_check.hpp: a failing test fails within the
// first few rows), then every row -- a launch that returns at once when the sample has set the flag.
int run_ece(misslap_solver *h, float eps, int *ok) {
    int rc = read_status(h);
    if (rc) return rc;
    if (h->h_ctl->K > 0) {  // auction_.pyx:446-447
        *ok = 0;
        return MISSLAP_OK;
    }
    h->ctl_fresh = false;
    // (the flag is clear behind the state initialisation and behind every k_reset_phase: one runtime fill kernel less
    // per phase; a second test on the same state -- misslap_check_ece -- clears it itself)
    if (!h->ece_flag_clear) HIP_TRY(hipMemsetAsync(&h->ctl->ece_fail, 0, sizeof(int), h->stream));
    h->ece_flag_clear = false;
    const FinalOut fo{0, h->maximize, h->o2p, h->contrib, h->nmatch, h->n_rows, h->n_cols, nullptr};
    const int sample = std::min(h->n_rows, kEceSampleRows);
    if ((rc = launch_rows_gather(h, eps, fo, sample))) return rc;
    if (sample < h->n_rows && (rc = launch_rows_all(h, eps, fo))) return rc;
    if (!h->live_off) {  // the verdict through the live words: no copy of the control block, no stream drain
        hipLaunchKernelGGL(k_post_ece, dim3(1), dim3(1), 0, h->stream, h->ctl, h->live_dev, ++h->ticket);
        h->live_valid = true;
        int K = 0, err = 0;
        long long nits = 0;
        if (live_poll(h, h->ticket, true, &K, &err, &nits)) {
            // (k_post_ece stores the verdict word right behind the four status words; the wait is bounded by time,
            // like live_poll's)
            volatile unsigned long long *w = h->live + 4;
            unsigned long long v = *w;
            const double t_end = now_ms() + 2000.0;
            for (unsigned spins = 0; (unsigned)(v >> 32) != h->ticket; ++spins) {
                __builtin_ia32_pause();
                if ((spins & 4095) == 4095 && now_ms() > t_end) break;
                v = *w;
            }
            if ((unsigned)(v >> 32) == h->ticket) {
                h->h_ctl->K = K;
                h->h_ctl->nits = nits;
                h->h_ctl->err = err;
                h->h_ctl->ece_fail = (int)(unsigned)(v & 0xffffffffull);
                h->K_ub = K;
                h->K_exact = true;
                h->ctl_fresh = 1;
                count_tail_rounds(h);
                if (err) return fail(MISSLAP_ERR_STATE, "device-side invariant violated (error bits 0x%x)", err);
                *ok = h->h_ctl->ece_fail ? 0 : 1;
                return MISSLAP_OK;
            }
        }
        h->live_off = true;  // timed out: from here on by copy + drain
    }
    rc = read_ctl(h);
    if (rc) return rc;
    *ok = h->h_ctl->ece_fail ? 0 : 1;
    return MISSLAP_OK;
}

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
    static const long long long_from_env = [] {
        const char *e = std::getenv("MISSLAP_LONG_ROWS_FROM");
        return e ? std::atoll(e) : (long long)kLongRowsFrom;
    }();
    h->long_rows = st.max_row_len > kCandRowMax && avg_row >= long_from_env && avg_row <= kCandLongMax;
    // ... below that (C4's 300 edges per row, a dense 600^2) only once the solve has shown that its tail is long:
    // launch_tail switches the builder on after max(100, n_rows / 64) tail rounds
    // (... and so do the long rows of a handle whose AVERAGE row keeps a line, where they are many: 40 000 rows of 256
    // edges on average, half of them longer: 208 -> 119 ms per solve; a few stragglers -- C3 has rows of 260 edges -- are left
    // to their scans, a pass over all rows every few hundred tail rounds costs more than they do)
    static const int mixed_pct_env = [] {
        const char *e = std::getenv("MISSLAP_LONG_MIXED_PCT");
        return e ? std::atoi(e) : kLongRowsMixedPercent;
    }();
    const bool many_long = (long long)st.long_rows * 100 >= (long long)mixed_pct_env * h->n_rows;
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
            // MISSLAP_TILED_CARRY_INDEX=1 (A/B timing, tests): the stored-index formats also for column-sorted rows
            static const bool carry_env = [] {
                const char *e = std::getenv("MISSLAP_TILED_CARRY_INDEX");
                return e && e[0] == '1';
            }();
            if ((rc = count_and_scan(false))) return rc;
            bool carry = carry_env;
            bool usable = true;
            if (unsorted) {
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
                const size_t tiled_words = ((size_t)total / 2 + 8) * (size_t)(rec_bytes / 4);  // + 16 entries of padding
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
                // (price fills, barriers / the merge pass) of about a fifth of a full k_bid scan
                h->tiled_min_K = tiled_opt > 0 ? tiled_opt : (int)std::max<size_t>((N * 3) / 10, 8192);
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
    h->shard_min_K = h->tiled_ok ? h->tiled_min_K : (int)std::max<size_t>((N * 3) / 10, 8192);
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
    h->order_partial = opt->partial_in_list_order == 0;
    if (const char *e = std::getenv("MISSLAP_APPLY_BIDDERS_RATIO")) h->apply_bidders_ratio = std::max(1, std::atoi(e));
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

// ------------------------------------------------------------------------------------------------
MISSLAP_API int misslap_abi_version(void) { return MISSLAP_ABI_VERSION; }

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

MISSLAP_API const char *misslap_last_error(void) { return g_err.c_str(); }

MISSLAP_API int misslap_set_cache_limits(int64_t max_total_bytes, int64_t max_block_bytes, int32_t max_blocks) {
    if (max_total_bytes < 0 || max_block_bytes < 0 || max_blocks < 0) return fail(MISSLAP_ERR_INVALID, "negative limit");
    BlockCache &bc = block_cache();
    std::lock_guard<std::mutex> g(bc.m);
    bc.kMaxHeld = (size_t)max_total_bytes;
    bc.kMaxEach = (size_t)max_block_bytes;
    bc.kMaxEntries = (size_t)max_blocks;
    bc.explicit_limits = true;
    return MISSLAP_OK;
}

MISSLAP_API int misslap_trim_caches(int64_t *freed_bytes) {
    int64_t freed = 0;
    int keep_dev = 0;
    const bool have_dev = hipGetDevice(&keep_dev) == hipSuccess;
    {
        BlockCache &bc = block_cache();
        std::vector<BlockCache::Ent> take;
        {
            std::lock_guard<std::mutex> g(bc.m);
            take.swap(bc.idle);
            bc.held = 0;
        }
        for (const BlockCache::Ent &e : take) {
            if (hipSetDevice(e.device) == hipSuccess) {
                (void)hipDeviceSynchronize();  // nothing may still be running on a parked block
                (void)hipFree(e.p);
                freed += (int64_t)e.bytes;
            }
        }
    }
    {
        HostResPool &hp = host_pool();
        std::vector<std::pair<int, HostRes>> take;
        {
            std::lock_guard<std::mutex> g(hp.m);
            take.swap(hp.idle);
        }
        for (auto &pr : take) {
            if (hipSetDevice(pr.first) != hipSuccess) continue;
            if (pr.second.stream) {
                (void)hipStreamSynchronize(pr.second.stream);
                (void)hipStreamDestroy(pr.second.stream);
            }
            if (pr.second.h_ctl) (void)hipHostFree(pr.second.h_ctl);
            for (hipEvent_t e : pr.second.ev)
                if (e) (void)hipEventDestroy(e);
        }
    }
    if (have_dev) (void)hipSetDevice(keep_dev);
    if (freed_bytes) *freed_bytes = freed;
    return MISSLAP_OK;
}

MISSLAP_API int misslap_device_info(int32_t device, char *name, int32_t name_len, int32_t *compute_units,
                                    int64_t *hbm_bytes) {
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(MISSLAP_ERR_NO_DEVICE, "no HIP device available");
    hipDeviceProp_t p;
    HIP_TRY(hipGetDeviceProperties(&p, device));
    if (name && name_len > 0) snprintf(name, (size_t)name_len, "%s (%s)", p.name, p.gcnArchName);
    if (compute_units) *compute_units = p.multiProcessorCount;
    if (hbm_bytes) *hbm_bytes = (int64_t)p.totalGlobalMem;
    return MISSLAP_OK;
}

MISSLAP_API int misslap_device_uuid(int32_t device, char *uuid_hex, int32_t len) {
    if (!uuid_hex || len < 33) return fail(MISSLAP_ERR_INVALID, "uuid buffer of at least 33 bytes expected");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(MISSLAP_ERR_NO_DEVICE, "no HIP device available");
    hipUUID id;
    HIP_TRY(hipDeviceGetUuid(&id, device));
    for (int k = 0; k < 16; ++k) snprintf(uuid_hex + 2 * k, 3, "%02x", (unsigned)(unsigned char)id.bytes[k]);
    return MISSLAP_OK;
}

// Streaming rates of this device (see the header).  The shape is the fastest of tools/micro/stream_bench.hip
// (profiles/r04_micro_stream.txt): every workgroup walks ONE contiguous chunk of the buffer, four 16-byte non-temporal
// loads in flight per lane -- 6.9-7.0 TB/s read-only against 5.3 TB/s for a grid-stride loop over the whole buffer
// with plain loads (6.1 for contiguous chunks with plain loads); a copy reaches 6.0-6.3 TB/s (read + written bytes).
namespace {
typedef unsigned v4u_t __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void k_stream_read(const v4u_t *src, size_t n16, unsigned *sink) {
    const size_t per = (n16 + gridDim.x - 1) / gridDim.x;
    size_t k = (size_t)blockIdx.x * per + threadIdx.x;
    const size_t end = min(n16, (size_t)(blockIdx.x + 1) * per);
    unsigned acc = 0;
    for (; k + 3 * 256 < end; k += 4 * 256) {
        v4u_t v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = __builtin_nontemporal_load(src + k + u * 256);
#pragma unroll
        for (int u = 0; u < 4; ++u) acc ^= v[u].x ^ v[u].y ^ v[u].z ^ v[u].w;
    }
    for (; k < end; k += 256) {
        const v4u_t a = src[k];
        acc ^= a.x ^ a.y ^ a.z ^ a.w;
    }
    if (acc == 0x9e3779b9u) *sink = acc;  // keeps the loads alive; the buffer is zero-filled, so nothing is stored
}
__global__ __launch_bounds__(256) void k_stream_copy(const v4u_t *src, v4u_t *dst, size_t n16) {
    const size_t per = (n16 + gridDim.x - 1) / gridDim.x;
    size_t k = (size_t)blockIdx.x * per + threadIdx.x;
    const size_t end = min(n16, (size_t)(blockIdx.x + 1) * per);
    for (; k + 3 * 256 < end; k += 4 * 256) {
        v4u_t v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = __builtin_nontemporal_load(src + k + u * 256);
#pragma unroll
        for (int u = 0; u < 4; ++u) __builtin_nontemporal_store(v[u], dst + k + u * 256);
    }
    for (; k < end; k += 256) dst[k] = src[k];
}
}  // namespace
MISSLAP_API int misslap_measure_hbm(int32_t device, int64_t bytes, int32_t reps, double *read_GBs, double *copy_GBs) {
    if (bytes < (1 << 20) || reps < 1 || (!read_GBs && !copy_GBs)) return fail(MISSLAP_ERR_INVALID, "bad argument");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(MISSLAP_ERR_NO_DEVICE, "no HIP device available");
    if (device < 0 || device >= ndev) return fail(MISSLAP_ERR_INVALID, "device %d out of range", device);
    HIP_TRY(hipSetDevice(device));
    int cus = 256;
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device);
    const size_t n16 = (size_t)bytes / 16;
    v4u_t *src = nullptr, *dst = nullptr;
    unsigned *sink = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    auto done = [&](int code) {
        if (src) (void)hipFree(src);
        if (dst) (void)hipFree(dst);
        if (sink) (void)hipFree(sink);
        if (e0) (void)hipEventDestroy(e0);
        if (e1) (void)hipEventDestroy(e1);
        return code;
    };
    if (hipMalloc((void **)&src, n16 * 16) != hipSuccess || hipMalloc((void **)&sink, 256) != hipSuccess ||
        (copy_GBs && hipMalloc((void **)&dst, n16 * 16) != hipSuccess) || hipEventCreate(&e0) != hipSuccess ||
        hipEventCreate(&e1) != hipSuccess || hipMemset(src, 0, n16 * 16) != hipSuccess)
        return done(fail(MISSLAP_ERR_HIP, "misslap_measure_hbm: allocation failed: %s", hipGetErrorString(hipGetLastError())));
    const dim3 block(256);
    auto timed = [&](bool copy, double *out, double bytes_moved) {
        // read: 16 workgroups per CU; copy: one workgroup per 16 KB (the two best grids of the microbenchmark)
        const dim3 grid(copy ? (unsigned)std::max<size_t>(1, n16 / 1024) : (unsigned)cus * 16);
        for (int r = -2; r < reps; ++r) {  // two warm-up launches
            if (r == 0 && hipEventRecord(e0, nullptr) != hipSuccess) return false;
            if (copy) hipLaunchKernelGGL(k_stream_copy, grid, block, 0, nullptr, src, dst, n16);
            else hipLaunchKernelGGL(k_stream_read, grid, block, 0, nullptr, src, n16, sink);
        }
        float ms = 0.f;
        if (hipEventRecord(e1, nullptr) != hipSuccess || hipEventSynchronize(e1) != hipSuccess ||
            hipEventElapsedTime(&ms, e0, e1) != hipSuccess || ms <= 0.f)
            return false;
        *out = bytes_moved * reps / (ms * 1e-3) / 1e9;
        return true;
    };
    if (read_GBs && !timed(false, read_GBs, (double)n16 * 16)) return done(fail(MISSLAP_ERR_HIP, "misslap_measure_hbm: timing failed"));
    if (copy_GBs && !timed(true, copy_GBs, 2.0 * (double)n16 * 16)) return done(fail(MISSLAP_ERR_HIP, "misslap_measure_hbm: timing failed"));
    return done(MISSLAP_OK);
}

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
            hipLaunchKernelGGL(k_reset_phase, dim3(blocks_for(h->n_rows > h->n_cols ? h->n_rows : h->n_cols, 256)),
                               dim3(256), 0, h->stream, h->ctl, h->p2o, h->o2p, h->rec, h->U, h->n_rows, h->n_cols);
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
    hipLaunchKernelGGL(k_final_reset, dim3(1), dim3(1), 0, h->stream, h->ctl);
    const FinalOut fo{1, h->maximize, h->o2p, h->contrib, h->nmatch, h->n_rows, h->n_cols, h->fin_slots};
    int n_slots = 0;
    // (a workgroup without rows returns before it writes its slot)
    HIP_TRY(hipMemsetAsync(h->fin_slots, 0, sizeof(FinSlot) * (size_t)h->fin_slots_n, h->stream));
    if ((rc = launch_rows_all(h, h->target_eps, fo, &n_slots))) return rc;
    h->ece_flag_clear = false;  // the final pass leaves its verdict in Ctl::ece_fail: the next test must clear it
    if (h->f32) {
        EdgesF32 ed{h->edges32};
        hipLaunchKernelGGL(k_obj_sum<EdgesF32>, dim3(1), dim3(1024), 0, h->stream, h->ctl, ed, h->row_ptr, h->p2o,
                           h->n_rows, h->maximize, h->contrib, h->nmatch, h->fin_slots, n_slots);
    } else {
        EdgesF64 ed{h->col, h->val64};
        hipLaunchKernelGGL(k_obj_sum<EdgesF64>, dim3(1), dim3(1024), 0, h->stream, h->ctl, ed, h->row_ptr, h->p2o,
                           h->n_rows, h->maximize, h->contrib, h->nmatch, h->fin_slots, n_slots);
    }
    HIP_TRY(hipGetLastError());
    if (person_to_object_out)
        HIP_TRY(hipMemcpyAsync(person_to_object_out, h->p2o, sizeof(int) * (size_t)h->n_rows, hipMemcpyDeviceToHost,
                               h->stream));
    // the bid kernels' statistics (a slot per workgroup, RoundArgs::wg_stats) -> the control block
    hipLaunchKernelGGL(k_collect_stats, dim3(1), dim3(1024), 0, h->stream, h->ctl, h->wg_stats, h->wg_stats_slots);
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
    defined(MISSLAP_TAIL_STAMP_BLOCK) || defined(MISSLAP_TILED_STAMP)
    for (int k = 0; k < 6; ++k) meta->tail_stats[k] = (double)c.dbg[6 + k];  // diagnostic build: solo-round segments (cycles)
#endif
    meta->cand_hits = c.cand_hits;
    meta->cand_edges = c.cand_edges;
    meta->sharded_rounds = h->sharded_rounds;
    meta->tiled_format = h->tiled_ok ? h->tiled_fmt : 0;
    meta->phases_with_lines = h->phases_with_lines;
    meta->eps_phases = h->phases_run;
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
    // (MISSLAP_BIG_ROUNDS_BATCHED=1, host_comm.hpp: the rounds of the full-scan regime through the batched path, one per
    // batch, so that the host's upper bound of K is at most two rounds old when the regime ends)
    if (h->world == 1 && !big_rounds_exact_env()) o.rounds_per_sync_large = 1;
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

// ---- multi-GPU: communicators and the sharded solve (host_comm.hpp) ------------------------------------------------
MISSLAP_API int misslap_rccl_unique_id(void *id_out) {
    if (!id_out) return fail(MISSLAP_ERR_INVALID, "null argument");
    RcclApi &api = rccl_api();
    if (!api.error.empty()) return fail(MISSLAP_ERR_HIP, "%s", api.error.c_str());
    RcclApi::UniqueId id;
    const int rc = api.GetUniqueId(&id);
    if (rc) return fail(MISSLAP_ERR_HIP, "ncclGetUniqueId failed: %s", api.GetErrorString(rc));
    std::memcpy(id_out, &id, sizeof(id));
    return MISSLAP_OK;
}

MISSLAP_API int misslap_rccl_selfcheck(int32_t *n_symbols, int32_t enums[6], char *lib_path, int32_t lib_path_len) {
    RcclApi &api = rccl_api();
    if (!api.handle) return fail(MISSLAP_ERR_HIP, "%s", api.error.c_str());
    int n = 0;
    n += api.GetUniqueId != nullptr;
    n += api.CommInitRank != nullptr;
    n += api.CommDestroy != nullptr;
    n += api.AllReduce != nullptr;
    n += api.GetErrorString != nullptr;
    n += api.CommCount != nullptr;
    if (n_symbols) *n_symbols = n;
    if (enums) {
        enums[0] = kNcclInt32;
        enums[1] = kNcclInt64;
        enums[2] = kNcclMax;
        enums[3] = kNcclMin;
        enums[4] = (int32_t)sizeof(RcclApi::UniqueId);
        enums[5] = 0;
        if (auto ver = reinterpret_cast<int (*)(int *)>(dlsym(api.handle, "ncclGetVersion"))) {
            int v = 0;
            if (ver(&v) == 0) enums[5] = v;
        }
    }
    if (lib_path && lib_path_len > 0) {
        lib_path[0] = 0;
        Dl_info di;
        if (api.AllReduce && dladdr(reinterpret_cast<void *>(api.AllReduce), &di) && di.dli_fname)
            snprintf(lib_path, (size_t)lib_path_len, "%s", di.dli_fname);
    }
    if (n != MISSLAP_RCCL_SYMBOLS) return fail(MISSLAP_ERR_HIP, "%s", api.error.empty() ? "librccl: a symbol is missing" : api.error.c_str());
    return MISSLAP_OK;
}

MISSLAP_API int misslap_comm_init_rccl(misslap_comm **out, const void *unique_id, int32_t rank, int32_t world,
                                       int32_t device) {
    if (!out || !unique_id || world < 1 || rank < 0 || rank >= world) return fail(MISSLAP_ERR_INVALID, "bad argument");
    RcclApi &api = rccl_api();
    if (!api.error.empty()) return fail(MISSLAP_ERR_HIP, "%s", api.error.c_str());
    HIP_TRY(hipSetDevice(device));
    RcclApi::UniqueId id;
    std::memcpy(&id, unique_id, sizeof(id));
    misslap_comm *c = new misslap_comm();
    c->rank = rank;
    c->world = world;
    c->device = device;
    const int rc = api.CommInitRank(&c->nccl_comm, world, id, rank);
    if (rc) {
        delete c;
        return fail(MISSLAP_ERR_HIP, "ncclCommInitRank failed: %s", api.GetErrorString(rc));
    }
    *out = c;
    return MISSLAP_OK;
}

MISSLAP_API int misslap_comm_init_custom(misslap_comm **out, const misslap_comm_ops *ops) {
    if (!out || !ops || ops->struct_size != (int32_t)sizeof(misslap_comm_ops) || !ops->allreduce_max_i64 ||
        !ops->allreduce_min_i32 || ops->world < 1 || ops->rank < 0 || ops->rank >= ops->world)
        return fail(MISSLAP_ERR_INVALID, "bad misslap_comm_ops");
    misslap_comm *c = new misslap_comm();
    c->rank = ops->rank;
    c->world = ops->world;
    c->ops = *ops;
    c->custom = true;
    *out = c;
    return MISSLAP_OK;
}

MISSLAP_API int misslap_comm_info(const misslap_comm *c, int32_t *kind, int32_t *rank, int32_t *world, int32_t *transport_ranks) {
    if (!c) return fail(MISSLAP_ERR_INVALID, "null communicator");
    if (kind) *kind = c->custom ? 0 : 1;
    if (rank) *rank = c->rank;
    if (world) *world = c->world;
    if (transport_ranks) {
        *transport_ranks = c->ops.world;
        if (!c->custom) {
            RcclApi &api = rccl_api();
            int n = 0;
            if (!api.CommCount) return fail(MISSLAP_ERR_HIP, "librccl: ncclCommCount is missing");
            const int rc = api.CommCount(c->nccl_comm, &n);
            if (rc) return fail(MISSLAP_ERR_HIP, "ncclCommCount failed: %s", api.GetErrorString(rc));
            *transport_ranks = n;
        }
    }
    return MISSLAP_OK;
}

MISSLAP_API int misslap_comm_destroy(misslap_comm *c) {
    if (!c) return MISSLAP_OK;
    if (!c->custom && c->nccl_comm) (void)rccl_api().CommDestroy(c->nccl_comm);
    delete c;
    return MISSLAP_OK;
}

MISSLAP_API int misslap_drive_sharded(const misslap_round_ops *ops, misslap_comm *comm) {
    if (!ops || ops->struct_size != (int32_t)sizeof(misslap_round_ops) || !ops->status || !ops->round_bid ||
        !ops->round_tiebreak || !ops->round_apply || !ops->run_tail || !ops->phase_end)
        return fail(MISSLAP_ERR_INVALID, "bad misslap_round_ops");
    return drive_sharded(ops, comm, fail);
}

MISSLAP_API int misslap_solve_sharded(misslap_solver *h, misslap_comm *comm, int32_t *person_to_object_out,
                                      misslap_meta *meta) {
    if (!h) return fail(MISSLAP_ERR_INVALID, "null handle");
    if (comm && (comm->world != h->world || comm->rank != h->rank))
        return fail(MISSLAP_ERR_INVALID, "communicator is rank %d of %d, the handle was created as shard %d of %d",
                    comm->rank, comm->world, h->rank, h->world);
    if (!comm && h->world != 1) return fail(MISSLAP_ERR_INVALID, "a handle of %d shards needs a communicator", h->world);
    if (comm && !comm->custom && comm->device != h->device)  // an all-reduce enqueued on another device's stream fails late or hangs
        return fail(MISSLAP_ERR_INVALID, "the RCCL communicator lives on device %d, the handle on device %d", comm->device,
                    h->device);
    // (before any work: a caller that forgot the size must not pay for a solve to learn it)
    if (meta && h->abi >= 2 && (meta->struct_size < (int32_t)offsetof(misslap_meta, edges_scanned) || meta->struct_size > 65536))
        return fail(MISSLAP_ERR_INVALID, "misslap_meta.struct_size = %d: set it to sizeof(misslap_meta) before the call", meta->struct_size);
    HIP_TRY(hipSetDevice(h->device));
    const double t0 = now_ms();
    const misslap_round_ops o = handle_round_ops(h);
    if (comm) comm->sharded_rounds = 0;
    int rc = drive_sharded(&o, comm, fail);
    h->sharded_rounds = comm ? comm->sharded_rounds : 0;
    if (rc) return rc;
    HIP_TRY(hipStreamSynchronize(h->stream));
    h->solve_ms += now_ms() - t0;
    return misslap_finish(h, person_to_object_out, meta);
}

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

#ifdef MISSLAP_DIAG  // built into libmisslap_diag.so only (python -m sslap_amd.build diag), for tools/
// Diagnostics: average duration (ms) of `reps` launches of an ablated full-scan bid kernel over the
// current unassigned list (K == n_rows right after create).  mode: 0 complete, 1 no price gather,
// 2 no cross-lane reduction, 3 edge stream only.  Results are discarded; solver state is untouched.
MISSLAP_API int misslap_debug_time_bid(misslap_solver *h, int32_t mode, int32_t reps, float *ms_avg) {
    if (!h || !ms_avg || reps <= 0) return fail(MISSLAP_ERR_INVALID, "bad argument");
    if (!h->f32) return fail(MISSLAP_ERR_STATE, "ablation kernels are instantiated for the 8 B/edge layout only");
    HIP_TRY(hipSetDevice(h->device));
    if (mode >= 10) {  // LDS-tiled kernel, shape 0: 10 complete, 11 no fill, 12 no arithmetic, 13 no edge loads
        if (!h->tiled_ok || (h->tiled_shape != 3 && mode != 10))  // the ablations are instantiated for shape 3
            return fail(MISSLAP_ERR_STATE, "tiled ablations need tiled_shape 3");
        const hipFuncAttribute at = hipFuncAttributeMaxDynamicSharedMemorySize;
        const int ldsb = (int)tiled_lds_bytes(kTiledShapes[h->tiled_shape][4]);
        HIP_TRY(hipFuncSetAttribute((const void *)k_bid_tiled<1024, 4, 2, 2, kTileColsHalf, 1, 1>, at, ldsb));
        HIP_TRY(hipFuncSetAttribute((const void *)k_bid_tiled<1024, 4, 2, 2, kTileColsHalf, 1, 2>, at, ldsb));
        HIP_TRY(hipFuncSetAttribute((const void *)k_bid_tiled<1024, 4, 2, 2, kTileColsHalf, 1, 3>, at, ldsb));
        HIP_TRY(hipFuncSetAttribute((const void *)k_bid_tiled<1024, 4, 2, 2, kTileColsHalf, 1, 4>, at, ldsb));
        HIP_TRY(hipFuncSetAttribute((const void *)k_bid_tiled<1024, 4, 2, 2, kTileColsHalf, 1, 5>, at, ldsb));
        HIP_TRY(hipFuncSetAttribute((const void *)k_bid_tiled<1024, 4, 2, 2, kTileColsHalf, 1, 6>, at, ldsb));
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
                default: hipLaunchKernelGGL((k_bid_tiled<1024, 4, 2, 2, kTileColsHalf, 1, 3>), g, b, ldsb, h->stream, a, ta); break;
            }
        };
        launch_t();
        HIP_TRY(hipEventRecord(t0, h->stream));
        for (int r = 0; r < reps; ++r) launch_t();
        HIP_TRY(hipEventRecord(t1, h->stream));
        HIP_TRY(hipEventSynchronize(t1));
        float ms = 0.f;
        HIP_TRY(hipEventElapsedTime(&ms, t0, t1));
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
