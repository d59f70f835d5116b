// kernels_round.hpp -- one Jacobi auction round as grid kernels (rounds with many bidders).
//
// Reference: AuctionSolver.bid_and_assign, sslap/auction_.pyx:313-430.
//   k_bid       BID phase (:339-365) + running per-object maximum (first half of RESOLVE, :375-385)
//   k_tiebreak  "first bidder in list order wins equal bids" (strict '>' at :379)
//   k_apply     ASSIGN phase (:388-427), one thread per object instead of the O(M) sequential walk
//   k_compact_* push_all_left (:137-162, called at :430) as two counting passes + scatter
//   round end   K += evicted - assigned (:429), nits += 1 (:273): by the kernel that closes the round (k_compact_fill,
//               k_compact_small, k_round_fused)
// Every kernel is a no-op unless the round is live (K > tail_threshold, nits < max_iter), so the
// host can enqueue several rounds without reading K back.
#pragma once
#include "device_common.hpp"

namespace misslap {

struct RoundArgs {
    Ctl *ctl;
    const int *row_ptr;
    double *price;
    PriceRec *rec;                // {price, owner, owner's row start} per object (tail kernel's gather)
    int *p2o;
    int *o2p;
    int *U;
    unsigned long long *bid_key;  // [N] bid of U position n (as key)
    int *bid_obj;                 // [N] object bid on by U position n
    int4 *bid_rec;                // [kRoundSmallMax] small rounds: {object, its owner, bidder, bidder's row start} of
                                  // U position n -- everything k_round_small needs besides the bid
    unsigned long long *best_key; // [M]
    int *best_pos;                // [M]
    int *cnt;                     // [2 * nblocks_compact] per-chunk (left holes, movers)
    int *hole_list;               // [N] positions of left holes
    int *mover_list;              // [N] persons to move left
    unsigned long long *launch_edges;  // per-bid-launch edge counters (profiling), may be null
    int n_rows, n_cols;
    int thr;                      // tail threshold
    int rank, world;              // bidder shard
    int shard_min_K;              // shard only rounds with K >= this (multi-GPU), see shard_range
    float eps;
    int launch_idx;
    int gather_max_K;             // k_bid runs only for K < gather_max_K (k_bid_tiled takes the rest); 0 = no limit
    int2 *cand;                   // candidate lines (device_common.hpp); nullptr = none
    double *cand64;               // ... their fp64 costs (12 B/edge layout only, nullptr otherwise)
    int cand_build_max_K;         // k_bid uses and (re)builds lines only in rounds with K <= this
    int cand_build_min_K;         // ... and rebuilds a line that missed only in rounds with K > this (below, a round lasts
                                  // as long as its slowest bidder, and a rebuild doubles a miss: the maintenance pass ahead
                                  // of the tail kernels rebuilds what the small rounds leave spent)
    int cand_refresh_min;         // ... and treats a hit that leaves fewer live candidates than this as a miss
    // Statistics of the bid kernels, one 64-byte slot per WORKGROUP (kStat* below), added to with plain loads / stores
    // by thread 0 of the workgroup -- launches of a stream are ordered, so nobody else touches the slot -- and summed
    // once, when the solve is finished (k_collect_stats).  Atomics on the control block are not an option: they are
    // performed one after the other, ~5.5 ns each, and a launch is not over before the last one has retired -- 2048
    // workgroups x 4-5 counters made a bid launch of a mid round 39 us instead of 12 (C3; 13 ms per solve).
    unsigned long long *wg_stats;
    int *need_list;               // [n_rows] persons with LONG rows whose line k_refresh_lines found spent (Ctl::n_need of
                                  // them): the work list of k_refresh_long
    unsigned long long *live;     // pinned host words the round-closing kernel posts the status to (post_live_status);
    unsigned ticket;              // nullptr: none.  ticket: the host's number of that launch
    // single-precision filter of the full scans (device_common.hpp, wave_bid_filter): the fp32 mirror of the prices built
    // in front of this launch (nullptr: exact scans), the bit pattern of the largest price at that time, max |cost|
    const float *price32;
    const int *pmax_bits;
    float cmax;
};
constexpr int kStatEdges = 0, kStatBids = 1, kStatHits = 2, kStatHitEdges = 3, kStatShardEdges = 4, kStatLaunchEdges = 5,
              kStatLaunchHitEdges = 6,  // of kStatLaunchEdges: rows a candidate line answered (counted, never read)
              kStatWords = 8;

__device__ __forceinline__ bool round_live(const Ctl *c, int thr) {
    return c->K > thr && c->K > 0 && c->nits < c->max_iter;
}
// The small-round kernels request the control block and their first data together.  The compiler sinks a load below
// an early exit that does not need it, so the exit is made to need it: `never` is a condition on the loaded value
// that holds for no valid content.
struct CtlHead {
    int K, err;
    long long nits, max_iter;
    __device__ __forceinline__ explicit CtlHead(const Ctl *c) : K(c->K), err(c->err), nits(c->nits), max_iter(c->max_iter) {
        __builtin_amdgcn_sched_barrier(0);  // the requests above are issued before anything below waits
    }
    __device__ __forceinline__ bool live(int thr, bool never) const {
        return (K > thr) & (K > 0) & (nits < max_iter) & !never;
    }
};
// Bidders of a round are sharded over the ranks only while K >= shard_min_K (the few big rounds where the
// bid phase is bandwidth-bound); below that every rank bids for everybody -- the replicas stay identical
// without any exchange, and a per-round all-reduce would cost more than the round.
__device__ __forceinline__ void shard_range(int K, int rank, int world, int shard_min_K, int &lo, int &hi) {
    if (K < shard_min_K) {
        lo = 0;
        hi = K;
        return;
    }
    lo = (int)(((long long)K * rank) / world);
    hi = (int)(((long long)K * (rank + 1)) / world);
}

constexpr int kBidBlock = 256;  // 4 wavefronts; one wavefront per bidder
// A bid is first tried on the person's candidate line (device_common.hpp) and only on a miss by a full scan of the
// row, which also (re)builds the line: in the rounds this kernel serves (K below the full-scan threshold) about
// 85 % of the bids are answered from 256 bytes and <= 30 price look-ups.
// Src = PriceSource: prices only (8 B per look-up); RecSource (rounds finished by k_round_small): the
// 16-byte price records, whose owner fields give the resolve kernel the person a winning bid evicts.  Such a round
// is a chain of dependent memory latencies and nothing else, so the wavefront's first list entry is requested before
// the control block is read (any position below n_rows is readable; it is used only if the round is live).
template <class Src>
struct SrcOf;
template <>
struct SrcOf<PriceSource> {
    static constexpr bool kOwners = false;
    static __device__ __forceinline__ PriceSource make(const RoundArgs &a) { return PriceSource{a.price}; }
};
template <>
struct SrcOf<RecSource> {
    static constexpr bool kOwners = true;
    static __device__ __forceinline__ RecSource make(const RoundArgs &a) { return RecSource{a.rec}; }
};
// kLines: 2 = lines used, refreshed and rebuilt (rounds with K <= cand_build_max_K); 1 = lines used where they still
// answer, misses by the lean scan, nothing built (the rounds above: a line built there is spent long before the tail
// kernel could use it, and the rounds in between rebuild it anyway); 0 = no lines (12 B/edge layout, cand = off).
// The host picks the variant from its upper bound of K; a stale bound only means a round or two more without builds.
// What a wavefront's bids add to the statistics (tally_flush: per workgroup, to its slot of RoundArgs::wg_stats).
struct BidTally {
    unsigned long long edges = 0, hit_edges = 0;
    int nb = 0, nh = 0, err = 0;
};
// hand-over of bids between the workgroups of ONE launch (k_round_fused): device-scope relaxed atomics are performed
// at the level all XCDs share (the XCDs' L2s are not coherent with each other, and a device-scope fence would write
// back and invalidate a whole L2 per workgroup); the order is made by hand, see k_round_fused.
__device__ __forceinline__ void handover_store(unsigned long long *p, unsigned long long v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ unsigned long long handover_load(const unsigned long long *p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// The bids of the list positions first, first + stride, ... below `hi` (one wavefront per position).
template <class E, class Src, int kLines, bool kHandOver>
__device__ __forceinline__ void bid_positions(const RoundArgs &a, const E &ed, int lo, int hi, int first, int stride,
                                              int i_first, BidTally &tl) {
    const int lane = threadIdx.x & 63;
    const double eps = (double)a.eps;  // float promoted to double, auction_.pyx:360
    const Src src = SrcOf<Src>::make(a);
    double hint = 0.0;  // cand_build's search distance, carried from one build of this wavefront to the next
    const bool lines = kLines > 0 && E::kCand && a.cand != nullptr;
    for (int n = lo + first; n < hi; n += stride) {
        const int i = n == first ? i_first : a.U[n];
        const int s = a.row_ptr[i], e = a.row_ptr[i + 1];
        CandBid b[2];
        b[0].hit = false;
        if (lines) {
            typename E::Slot sl = LineIO<typename E::Slot>::load(a.cand, a.cand64,
                                                                 (size_t)i * kCandLanes + (lane & (kCandLanes - 1)));
            int alive[2];
            cand_eval2(sl, true, false, src, eps, b, tl.err, NoEarly(), NoStamp(), alive);
            // A hit on a line with little life left is answered by a full scan all the same -- the bid is the same
            // bid -- so that the line is rebuilt HERE, where a scan is one of many in flight, and not by a miss in
            // the tail kernel, where a scan is the whole round.
            if (kLines == 2 && alive[0] < a.cand_refresh_min) b[0].hit = false;
        }
        if (!b[0].hit) {  // wave-uniform
            if (kLines == 2) {
                CandBuildArgs ba;
                const typename E::Raw none[4] = {};
                wave_bid_full<E, Src, false, false>(ed, src, s, e, none, eps, b[0], ba, tl.err);
                if (lines && ba.want && hi > a.cand_build_min_K) cand_build(a.cand, a.cand64, i, ba, eps, hint);
            } else if (!SrcOf<Src>::kOwners && a.price32 != nullptr) {  // (wave-uniform) big price table: through the fp32 filter
                // 2 delta = 2^-21 (max |cost| + max price): see wave_bid_filter; an infinite price (one-entry rows bid
                // +inf) makes it infinite, which sends every row with a finite third value to the exact scan
                const float two_delta = 0x1p-21f * (a.cmax + __int_as_float(*a.pmax_bits));
                wave_bid_filter(ed, a.price, a.price32, two_delta, s, e, eps, b[0], tl.err);
            } else {
                wave_bid_lean(ed, src, s, e, eps, b[0], tl.err);
            }
        } else {
            tl.nh += 1;
            tl.hit_edges += (unsigned long long)b[0].len;
        }
        if (lane == 0) {
            if (kHandOver) {  // (the resolve part of the same launch, in another workgroup, reads these)
                handover_store(&a.bid_key[n], b[0].key);
                unsigned long long *r = reinterpret_cast<unsigned long long *>(&a.bid_rec[n]);
                handover_store(r, (unsigned long long)(unsigned)b[0].obj | ((unsigned long long)(unsigned)b[0].prev << 32));
                handover_store(r + 1, (unsigned long long)(unsigned)i | ((unsigned long long)(unsigned)s << 32));
            } else {
                a.bid_key[n] = b[0].key;
                if (SrcOf<Src>::kOwners) {  // k_round_small forms the maxima itself
                    a.bid_rec[n] = make_int4(b[0].obj, b[0].prev, i, s);
                } else {
                    a.bid_obj[n] = b[0].obj;
                    atomicMax(&a.best_key[b[0].obj], b[0].key);
                }
            }
        }
        tl.edges += (unsigned long long)b[0].len;
        tl.nb += 1;
    }
}

template <int kWaves>
__device__ __forceinline__ void tally_flush(const RoundArgs &a, const BidTally &tl, int K) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    __shared__ unsigned long long s_edges[kWaves], s_hedges[kWaves];
    __shared__ int s_nb[kWaves], s_nh[kWaves];
    if (lane == 0) {
        s_edges[wave] = tl.edges;
        s_hedges[wave] = tl.hit_edges;
        s_nb[wave] = tl.nb;
        s_nh[wave] = tl.nh;
        if (tl.err) atomicOr(&a.ctl->err, tl.err);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long te = 0, the = 0;
        int tb = 0, th = 0;
        for (int w = 0; w < kWaves; ++w) {
            te += s_edges[w];
            the += s_hedges[w];
            tb += s_nb[w];
            th += s_nh[w];
        }
        if (tb) {
            unsigned long long *st = a.wg_stats + (size_t)kStatWords * blockIdx.x;
            st[kStatEdges] += te;
            st[kStatBids] += (unsigned long long)tb;
            if (a.world > 1 && K >= a.shard_min_K) st[kStatShardEdges] += te;
            if (a.launch_edges) {  // a profiled launch: claimed by k_take_launch_edges / the round's k_tiebreak
                st[kStatLaunchEdges] += te;
                st[kStatLaunchHitEdges] += the;
            }
            if (th) {
                st[kStatHits] += (unsigned long long)th;
                st[kStatHitEdges] += the;
            }
        }
    }
}

template <class E, class Src, int kLines>
__device__ __forceinline__ void k_bid_body(RoundArgs a, E ed) {
    static_assert(kLines == 2 || !SrcOf<Src>::kOwners, "the lean scan does not carry the owners k_round_small needs");
    const int wave = threadIdx.x >> 6;
    const int wpb = kBidBlock / kWave;
    const int first = blockIdx.x * wpb + wave;
    const Ctl *ctl = a.ctl;
    const CtlHead head(ctl);
    const int i_first = a.U[min(first, a.n_rows - 1)];
    if (!head.live(a.thr, i_first < -1)) return;  // (list entries are persons or -1)
    if (a.gather_max_K > 0 && head.K >= a.gather_max_K) return;
    int lo, hi;
    shard_range(head.K, a.rank, a.world, a.shard_min_K, lo, hi);
    BidTally tl;
    bid_positions<E, Src, kLines, false>(a, ed, lo, hi, first, gridDim.x * wpb, i_first, tl);
    tally_flush<kBidBlock / kWave>(a, tl, head.K);
}
template <class E, class Src, int kLines>
__global__ __launch_bounds__(kBidBlock) void k_bid(RoundArgs a, E ed) { k_bid_body<E, Src, kLines>(a, ed); }
template <class E, class Src, int kLines>
struct F_k_bid {  // (the body as a callable: what a batched launch runs per problem, csrc/host_batch.hpp)
    static __device__ __forceinline__ void run(RoundArgs a, E ed) { k_bid_body<E, Src, kLines>(a, ed); }
};


// The fp32 mirror of the prices for wave_bid_filter, and the largest price (bit pattern: prices are >= 0, so the patterns
// order like integers; *pmax_bits is zeroed by the host in front of the launch).  Only in a live round that k_bid serves.
__device__ __forceinline__ void k_price_mirror_body(const Ctl *ctl, const double *price, float *price32, int n_cols,
                                                       int *pmax_bits, int thr, int gather_max_K) {
    if (!round_live(ctl, thr) || (gather_max_K > 0 && ctl->K >= gather_max_K)) return;
    __shared__ int s_w[16];
    float m = 0.f;
    for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < n_cols; j += gridDim.x * blockDim.x) {
        const float p = (float)price[j];  // round to nearest
        price32[j] = p;
        m = __builtin_fmaxf(m, __builtin_fabsf(p));
    }
    int b = __float_as_int(m);
    for (int off = 32; off >= 1; off >>= 1) b = max(b, __shfl_xor(b, off));
    if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = b;
    __syncthreads();
    if (threadIdx.x == 0) {  // ONE atomic per workgroup: same-address atomics retire one after the other, ~5.5 ns each
        for (int w = 1; w < (int)blockDim.x / kWave; ++w) b = max(b, s_w[w]);
        if (b > 0) atomicMax(pmax_bits, b);
    }
}
__global__ __launch_bounds__(1024) void k_price_mirror(const Ctl *ctl, const double *price, float *price32, int n_cols,
                                                       int *pmax_bits, int thr, int gather_max_K) { k_price_mirror_body(ctl, price, price32, n_cols, pmax_bits, thr, gather_max_K); }
struct F_k_price_mirror {  // (the body as a callable: what a batched launch runs per problem, csrc/host_batch.hpp)
    static __device__ __forceinline__ void run(const Ctl *ctl, const double *price, float *price32, int n_cols, int *pmax_bits, int thr, int gather_max_K) { k_price_mirror_body(ctl, price, price32, n_cols, pmax_bits, thr, gather_max_K); }
};


// Line maintenance ahead of the tail kernels (once per eps-phase, when K has fallen to the tail threshold).  The grid
// rounds refresh the lines of the persons who BID in them; a person who won its object in one of the big rounds (which
// handle no lines) still carries the line of an earlier phase, and if it is evicted in the tail its stale line misses
// there -- where a row scan is the whole round (measured at C3: 7 % of the single-bidder rounds, 36 ms per solve).  So
// every person's line is evaluated once at the current prices (two persons per wavefront), and a line that would not
// answer, or has fewer than `min_alive` live candidates left, is rebuilt from a full scan -- here, where a scan is
// one of thousands in flight.  Nothing is bid: lines only decide which edges a later bid looks at.
// long_max > 0: the long-row builder runs behind this pass and takes rows of up to so many edges; their lines are
// rebuilt below min_alive_long live candidates.
template <class E>
__device__ __forceinline__ void k_refresh_lines_body(RoundArgs a, E ed, int min_alive, int long_max, int min_alive_long) {
    if (!E::kCand || a.cand == nullptr) return;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int wpb = kBidBlock / kWave;
    const double eps = (double)a.eps;
    const PriceSource src{a.price};
    double hint = 0.0;
    int err = 0;
    for (int w = blockIdx.x * wpb + wave; 2 * w < a.n_rows; w += gridDim.x * wpb) {
        const int i0 = 2 * w, i1 = 2 * w + 1;
        const bool act1 = i1 < a.n_rows;
        const int ime = (lane < kCandLanes || !act1) ? i0 : i1;
        typename E::Slot sl = LineIO<typename E::Slot>::load(a.cand, a.cand64,
                                                             (size_t)ime * kCandLanes + (lane & (kCandLanes - 1)));
        CandBid b[2];
        int alive[2];
        cand_eval2(sl, true, act1, src, eps, b, err, NoEarly(), NoStamp(), alive);
#pragma unroll
        for (int X = 0; X < 2; ++X) {
            if (X == 1 && !act1) continue;  // wave-uniform
            if (b[X].hit && alive[X] >= max(min_alive, min_alive_long)) continue;
            const int i = X ? i1 : i0;
            const int s = a.row_ptr[i], e = a.row_ptr[i + 1];
            if (e - s > kCandRowMax) {  // (wave-uniform) too long for a rebuild here: k_refresh_long's, if it can hold the row
                if (long_max > 0 && e - s <= long_max && !(b[X].hit && alive[X] >= min_alive_long) && lane == 0)
                    a.need_list[atomicAdd(&a.ctl->n_need, 1)] = i;
                continue;
            }
            if (b[X].hit && alive[X] >= min_alive) continue;
            CandBid full;
            CandBuildArgs ba;
            const typename E::Raw none[4] = {};
            wave_bid_full<E, PriceSource, false, false>(ed, src, s, e, none, eps, full, ba, err);
            if (ba.want) cand_build(a.cand, a.cand64, i, ba, eps, hint);
        }
    }
    if (lane == 0 && err) atomicOr(&a.ctl->err, err);
}
template <class E>
__global__ __launch_bounds__(kBidBlock) void k_refresh_lines(RoundArgs a, E ed, int min_alive, int long_max, int min_alive_long) { k_refresh_lines_body<E>(a, ed, min_alive, long_max, min_alive_long); }
template <class E>
struct F_k_refresh_lines {  // (the body as a callable: what a batched launch runs per problem, csrc/host_batch.hpp)
    static __device__ __forceinline__ void run(RoundArgs a, E ed, int min_alive, int long_max, int min_alive_long) { k_refresh_lines_body<E>(a, ed, min_alive, long_max, min_alive_long); }
};


// The same maintenance for LONG rows (kCandRowMax < row length <= kCandLongMax: the dense `mat=` entry of the reference,
// rows of thousands of edges).  A full scan of such a row keeps nothing in registers that a line could be built from,
// so their lines are built here and only here: one 256-thread workgroup per person holds the values cost - price of the
// whole row in registers (up to 32 per thread), finds by bisection a threshold t <= W with 24..30 values at or above it
// (a count per probe: compares in registers, one ballot per wavefront, one LDS add), collects the qualifying edges in
// stored order and writes the line.  Everywhere else a missed line of a long row is answered by a scan without a
// rebuild; the pass runs right before the tail kernels, which is where the lines are needed.
constexpr int kLongPer = 32;                   // values per thread
constexpr int kCandLongMax = 512 * kLongPer;  // 16384: the longest row that keeps a line (512-thread instance; 1024
                                              // threads leave 128 registers per thread, which spills the row's values)
template <class E, int kLongThreads>          // 256 threads: rows <= 8192 edges; 512 threads: rows <= 16384
__device__ __forceinline__ void k_refresh_long_body(RoundArgs a, E ed) {
    if (!E::kCand || a.cand == nullptr) return;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    __shared__ int s_cnt[2], s_n, s_g[kCandMax + 2], s_col[kCandMax + 2];
    __shared__ double s_cost[kCandMax + 2], s_red[2][kLongThreads / kWave];
    const double ninf = -__builtin_huge_val();
    const PriceSource src{a.price};
    // (the persons come from k_refresh_lines, which has evaluated every line at today's prices two per wavefront: a
    // workgroup per person re-doing that evaluation for ALL persons was 179 us per pass at dense 8000 x 8000, where a
    // pass finds a few hundred spent lines)
    const int n_need = a.ctl->n_need;
    for (int idx = blockIdx.x; idx < n_need; idx += gridDim.x) {
        const int i = a.need_list[idx];
        const int s = a.row_ptr[i], e = a.row_ptr[i + 1], len = e - s;
        if (len <= kCandRowMax || len > kLongThreads * kLongPer) continue;  // uniform over the workgroup (never: the list's rule)
        if (t == 0) s_n = 0;
        __syncthreads();
        // the row's values, in registers
        double v[kLongPer];
        double m1 = ninf, m2 = ninf;  // my best two (multiplicity counted)
#pragma unroll
        for (int k = 0; k < kLongPer; ++k) {
            const int g = s + k * kLongThreads + t;
            int c;
            double cost;
            ed.load(min(g, e - 1), c, cost);
            const double x = g < e ? cost - src.get(c).price : ninf;
            v[k] = x;
            m2 = __builtin_fmax(m2, __builtin_fmin(x, m1));
            m1 = __builtin_fmax(m1, x);
        }
        // W = the row's second-best value (multiplicity counted), lo = its smallest: wave-wide, then across wavefronts
        double lo = __builtin_huge_val();
#pragma unroll
        for (int k = 0; k < kLongPer; ++k) lo = v[k] == ninf ? lo : __builtin_fmin(lo, v[k]);
        {
            Top2 x;
            x.v = m1;
            x.w = m2;
            x.g = t;
            const Top2 r = top2_wave_reduce(x);
            const double wl = -wave_max_f64(-lo);
            if (lane == 0) {
                s_red[0][wave] = r.v;
                s_red[1][wave] = r.w;
                s_cost[wave] = wl;  // (scratch)
            }
        }
        __syncthreads();
        double V = ninf, W = ninf, LO = __builtin_huge_val();
        for (int w2 = 0; w2 < kLongThreads / kWave; ++w2) {
            const double bv = s_red[0][w2], bw = s_red[1][w2];
            W = __builtin_fmax(W, __builtin_fmax(__builtin_fmin(bv, V), bw));
            V = __builtin_fmax(V, bv);
            LO = __builtin_fmin(LO, s_cost[w2]);
        }
        __syncthreads();
        auto count_ge = [&](double thr, int par) {  // rows' values >= thr, uniform over the workgroup
            int n = 0;
#pragma unroll
            for (int k = 0; k < kLongPer; ++k) n += v[k] >= thr;
            for (int off = 32; off >= 1; off >>= 1) n += __shfl_xor(n, off);
            if (t == 0) s_cnt[par] = 0;
            __syncthreads();
            if (lane == 0) atomicAdd(&s_cnt[par], n);
            __syncthreads();
            return s_cnt[par];
        };
        // threshold: count(v >= W) >= 2; more than kCandMax ties at the top: no line
        double tsel = W;
        int par = 0;
        int nsel = count_ge(W, par);
        par ^= 1;
        bool ok = nsel <= kCandMax && W > ninf;
        if (ok && nsel < kCandMin && LO < W) {
            double hi_t = W, lo_t = LO;  // count(hi_t) = nsel < kCandMin; count(lo_t) = len > kCandMax
            for (int it = 0; it < 48; ++it) {
                const double mid = 0.5 * (hi_t + lo_t);
                if (!(mid < hi_t) || !(mid > lo_t)) break;
                const int n = count_ge(mid, par);
                par ^= 1;
                if (n > kCandMax) lo_t = mid;
                else {
                    hi_t = mid;
                    tsel = mid;
                    nsel = n;
                    if (n >= kCandMin) break;
                }
            }
        }
        // collect the qualifying edges (at most kCandMax) and order them by stored index
        if (ok) {
#pragma unroll
            for (int k = 0; k < kLongPer; ++k) {
                if (v[k] >= tsel) {
                    const int g = s + k * kLongThreads + t;
                    int c;
                    double cost;
                    ed.load(g, c, cost);
                    const int at = atomicAdd(&s_n, 1);
                    if (at < kCandMax) {
                        s_g[at] = g;
                        s_col[at] = c;
                        s_cost[at] = cost;
                    }
                }
            }
        }
        __syncthreads();
        if (wave == 0 && lane < kCandLanes) {
            const int n = ok ? min(s_n, kCandMax) : 0;
            int2 x = make_int2(-1, 0);  // empty slot
            size_t at = (size_t)i * kCandLanes + lane;
            if (lane == 0) {
                const double tau = ok ? tsel : __builtin_huge_val();  // +inf: a line that never answers
                x = make_int2(__double2loint(tau), __double2hiint(tau));
            } else if (lane == kCandLanes - 1) {
                x = make_int2(len, 0);
            }
            a.cand[at] = x;
            __builtin_amdgcn_wave_barrier();
            if (lane < n) {  // entry `lane` goes to slot 1 + (number of entries with a smaller stored index)
                int rank = 0;
                for (int q = 0; q < n; ++q) rank += s_g[q] < s_g[lane];
                at = (size_t)i * kCandLanes + 1 + rank;
                a.cand[at] = make_int2(s_col[lane], a.cand64 ? 0 : __float_as_int((float)s_cost[lane]));
                if (a.cand64) a.cand64[at] = s_cost[lane];
            }
        }
        __syncthreads();
    }
}
template <class E, int kLongThreads>
__global__ __launch_bounds__(kLongThreads) void k_refresh_long(RoundArgs a, E ed) { k_refresh_long_body<E, kLongThreads>(a, ed); }
template <class E, int kLongThreads>
struct F_k_refresh_long {  // (the body as a callable: what a batched launch runs per problem, csrc/host_batch.hpp)
    static __device__ __forceinline__ void run(RoundArgs a, E ed) { k_refresh_long_body<E, kLongThreads>(a, ed); }
};


// order_pos: the bidders of this rank's shard were taken in person order (k_bid_tiled, partial rounds): shard slot
// -> list position; nullptr = the shard is a range of list positions
// take_n / take_out (profiling only): the round's bid launch was a profiled one -- the first wavefront adds up what its
// take_n workgroups scanned (RoundArgs::wg_stats), like k_take_launch_edges, without a launch of its own
// order_min_K: the person-ordered scan (and the kernels that prepare order_pos) ran only if K >= order_min_K -- a round
// enqueued on a stale upper bound of K may have been bid by k_bid instead, in list order
__device__ __forceinline__ void k_tiebreak_body(RoundArgs a, const int *order_pos, int order_min_K, int take_n, unsigned long long *take_out) {
    const Ctl *ctl = a.ctl;
    if (!round_live(ctl, a.thr)) return;
    if (ctl->K < order_min_K) order_pos = nullptr;
    if (take_n > 0 && blockIdx.x == 0 && threadIdx.x < kWave) {
        unsigned long long v = 0, vh = 0;
        for (int k = threadIdx.x; k < take_n; k += kWave) {
            unsigned long long *st = a.wg_stats + (size_t)kStatWords * k;
            v += st[kStatLaunchEdges];
            vh += st[kStatLaunchHitEdges];
            st[kStatLaunchEdges] = 0ull;
            st[kStatLaunchHitEdges] = 0ull;
        }
        for (int off = 32; off >= 1; off >>= 1) {
            v += ((unsigned long long)__shfl_xor((unsigned)(v >> 32), off) << 32) | (unsigned long long)__shfl_xor((unsigned)(v & 0xffffffffull), off);
            vh += ((unsigned long long)__shfl_xor((unsigned)(vh >> 32), off) << 32) | (unsigned long long)__shfl_xor((unsigned)(vh & 0xffffffffull), off);
        }
        if (threadIdx.x == 0) {  // {edges of the bidders' rows, of which answered from lines}
            take_out[0] += v;
            take_out[1] += vh;
        }
    }
    int lo, hi;
    shard_range(ctl->K, a.rank, a.world, a.shard_min_K, lo, hi);
    for (int r = lo + blockIdx.x * blockDim.x + threadIdx.x; r < hi; r += gridDim.x * blockDim.x) {
        const int n = order_pos ? order_pos[r] : r;
        const int j = a.bid_obj[n];
        if (a.bid_key[n] == a.best_key[j]) atomicMin(&a.best_pos[j], n);
    }
}
__global__ __launch_bounds__(256) void k_tiebreak(RoundArgs a, const int *order_pos, int order_min_K, int take_n, unsigned long long *take_out) { k_tiebreak_body(a, order_pos, order_min_K, take_n, take_out); }
struct F_k_tiebreak {  // (the body as a callable: what a batched launch runs per problem, csrc/host_batch.hpp)
    static __device__ __forceinline__ void run(RoundArgs a, const int *order_pos, int order_min_K, int take_n, unsigned long long *take_out) { k_tiebreak_body(a, order_pos, order_min_K, take_n, take_out); }
};


// ASSIGN (auction_.pyx:388-427).  All writes of one round are disjoint: winners are distinct unassigned persons,
// evicted owners are distinct assigned persons, and a winner's slot in U is its own.
// The workgroups' statistics slots (RoundArgs::wg_stats) -> the control block; one 1024-thread workgroup, once per solve.
__device__ __forceinline__ unsigned long long block_sum_u64(unsigned long long v, unsigned long long *s_w) {
    for (int off = 32; off >= 1; off >>= 1)
        v += ((unsigned long long)__shfl_xor((unsigned)(v >> 32), off) << 32) | (unsigned long long)__shfl_xor((unsigned)(v & 0xffffffffull), off);
    __syncthreads();  // s_w of the previous call is no longer read
    if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = v;
    __syncthreads();
    unsigned long long t = 0;
    for (int w = 0; w < (int)blockDim.x / kWave; ++w) t += s_w[w];
    return t;
}
__global__ __launch_bounds__(1024) void k_collect_stats(Ctl *ctl, unsigned long long *wg_stats, int n_slots) {
    __shared__ unsigned long long s_w[16];
    unsigned long long v[5] = {0, 0, 0, 0, 0};
    for (int k = threadIdx.x; k < n_slots; k += 1024) {
        unsigned long long *st = wg_stats + (size_t)kStatWords * k;
#pragma unroll
        for (int q = 0; q < 5; ++q) {
            v[q] += st[q];
            st[q] = 0ull;
        }
    }
    unsigned long long t[5];
    for (int q = 0; q < 5; ++q) t[q] = block_sum_u64(v[q], s_w);
    if (threadIdx.x == 0) {
        ctl->edges += t[kStatEdges];
        ctl->bids += t[kStatBids];
        ctl->cand_hits += t[kStatHits];
        ctl->cand_edges += t[kStatHitEdges];
        ctl->shard_edges += t[kStatShardEdges];
    }
}
// ... and, behind a PROFILED bid launch, the edges that launch scanned (options.profile)
__global__ __launch_bounds__(1024) void k_take_launch_edges(unsigned long long *wg_stats, int n_slots, unsigned long long *out) {
    __shared__ unsigned long long s_w[16];
    unsigned long long v = 0, vh = 0;
    for (int k = threadIdx.x; k < n_slots; k += 1024) {
        unsigned long long *st = wg_stats + (size_t)kStatWords * k;
        v += st[kStatLaunchEdges];
        vh += st[kStatLaunchHitEdges];
        st[kStatLaunchEdges] = 0ull;
        st[kStatLaunchHitEdges] = 0ull;
    }
    const unsigned long long t = block_sum_u64(v, s_w);
    const unsigned long long th = block_sum_u64(vh, s_w);
    if (threadIdx.x == 0) {
        out[0] += t;
        out[1] += th;
    }
}
// list position n has won object j: price, eviction, assignment; returns 1 if the slot becomes a hole
__device__ __forceinline__ int apply_winner_of(const RoundArgs &a, Ctl *ctl, int j, int n) {
    const int i = a.U[n];
    PriceRec r;
    r.price = key_to_bid(a.best_key[j]);     // p[j] = best_bids[j]   (:397)
    r.owner = i;
    r.ostart = a.row_ptr[i];
    // (never with eps above the rounding error of a price update; only the candidate lines rely on it)
    if (a.cand != nullptr && r.price < a.price[j]) atomicOr(&ctl->err, kErrPriceFell);
    a.rec[j] = r;
    a.price[j] = r.price;
    const int prev = a.o2p[j];               // :401
    a.p2o[i] = j;                            // :417
    a.o2p[j] = i;                            // :418
    a.best_key[j] = 0ull;                    // :421-422
    a.best_pos[j] = kPosNone;
    if (prev != -1) {
        a.p2o[prev] = -1;                    // :404
        a.U[n] = prev;                       // :409 evicted owner inherits the slot
        return 0;
    }
    a.U[n] = -1;                             // :412 hole
    return 1;
}
__device__ __forceinline__ void count_holes(Ctl *ctl, int holes) {  // one atomic per workgroup (256 threads)
    __shared__ int s_h[4];
    for (int off = 32; off >= 1; off >>= 1) holes += __shfl_xor(holes, off);
    if ((threadIdx.x & 63) == 0) s_h[threadIdx.x >> 6] = holes;
    __syncthreads();
    if (threadIdx.x == 0) {
        const int t = s_h[0] + s_h[1] + s_h[2] + s_h[3];
        if (t) atomicAdd(&ctl->nholes, t);
    }
}
// One thread per object: instead of the reference's O(M) sequential walk (:394).
__device__ __forceinline__ void k_apply_body(RoundArgs a) {
    Ctl *ctl = a.ctl;
    if (!round_live(ctl, a.thr)) return;
    int holes = 0;
    for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < a.n_cols; j += gridDim.x * blockDim.x) {
        const int n = a.best_pos[j];
        if (n != kPosNone) holes += apply_winner_of(a, ctl, j, n);
    }
    count_holes(ctl, holes);
}
__global__ __launch_bounds__(256) void k_apply(RoundArgs a) { k_apply_body(a); }
struct F_k_apply {  // (the body as a callable: what a batched launch runs per problem, csrc/host_batch.hpp)
    static __device__ __forceinline__ void run(RoundArgs a) { k_apply_body(a); }
};

// One thread per BIDDER, for rounds with far fewer bidders than objects (every list position below K has bid in this
// round, so bid_obj[n] is this round's -- on every rank only in rounds that are not sharded): position n has won the
// object it bid on iff k_tiebreak left n in best_pos.  A loser that reads best_pos after the winner has reset it sees
// "none", which is not its position either.
__device__ __forceinline__ void k_apply_bidders_body(RoundArgs a) {
    Ctl *ctl = a.ctl;
    if (!round_live(ctl, a.thr)) return;
    const int K = ctl->K;
    int holes = 0;
    for (int n = blockIdx.x * blockDim.x + threadIdx.x; n < K; n += gridDim.x * blockDim.x) {
        const int j = a.bid_obj[n];
        if (a.best_pos[j] == n) holes += apply_winner_of(a, ctl, j, n);
    }
    count_holes(ctl, holes);
}
__global__ __launch_bounds__(256) void k_apply_bidders(RoundArgs a) { k_apply_bidders_body(a); }
struct F_k_apply_bidders {  // (the body as a callable: what a batched launch runs per problem, csrc/host_batch.hpp)
    static __device__ __forceinline__ void run(RoundArgs a) { k_apply_bidders_body(a); }
};


// push_all_left: the k-th empty slot in [0, K') receives the k-th person found in [K', K).
constexpr int kChunk = 1024;  // U positions per block (256 threads x 4)
__device__ __forceinline__ void k_compact_count_body(RoundArgs a) {
    const Ctl *ctl = a.ctl;
    if (!round_live(ctl, a.thr)) return;
    const int K = ctl->K, Kn = K - ctl->nholes;
    if (ctl->nholes == 0) return;
    const int nchunks = (K + kChunk - 1) / kChunk;
    __shared__ int sl[4], sm[4];
    for (int b = blockIdx.x; b < nchunks; b += gridDim.x) {
        int l = 0, m = 0;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int n = b * kChunk + q * 256 + threadIdx.x;
            if (n < K) {
                const int u = a.U[n];
                l += (n < Kn && u == -1);
                m += (n >= Kn && u != -1);
            }
        }
        for (int off = 32; off >= 1; off >>= 1) {
            l += __shfl_xor(l, off);
            m += __shfl_xor(m, off);
        }
        if ((threadIdx.x & 63) == 0) {
            sl[threadIdx.x >> 6] = l;
            sm[threadIdx.x >> 6] = m;
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            a.cnt[2 * b] = sl[0] + sl[1] + sl[2] + sl[3];
            a.cnt[2 * b + 1] = sm[0] + sm[1] + sm[2] + sm[3];
        }
        __syncthreads();
    }
}
__global__ __launch_bounds__(256) void k_compact_count(RoundArgs a) { k_compact_count_body(a); }
struct F_k_compact_count {  // (the body as a callable: what a batched launch runs per problem, csrc/host_batch.hpp)
    static __device__ __forceinline__ void run(RoundArgs a) { k_compact_count_body(a); }
};


__device__ __forceinline__ void k_compact_scatter_body(RoundArgs a) {
    Ctl *ctl = a.ctl;
    if (!round_live(ctl, a.thr)) return;
    const int K = ctl->K, Kn = K - ctl->nholes;
    if (ctl->nholes == 0) return;
    const int nchunks = (K + kChunk - 1) / kChunk;
    __shared__ int s_red[2][4];
    __shared__ int s_wl[4], s_wm[4];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int b = blockIdx.x; b < nchunks; b += gridDim.x) {
        // exclusive offsets of this chunk = sum of the counts of all earlier chunks
        int pl = 0, pm = 0;
        for (int c = threadIdx.x; c < b; c += 256) {
            pl += a.cnt[2 * c];
            pm += a.cnt[2 * c + 1];
        }
        for (int off = 32; off >= 1; off >>= 1) {
            pl += __shfl_xor(pl, off);
            pm += __shfl_xor(pm, off);
        }
        if (lane == 0) {
            s_red[0][wave] = pl;
            s_red[1][wave] = pm;
        }
        __syncthreads();
        int offl = s_red[0][0] + s_red[0][1] + s_red[0][2] + s_red[0][3];
        int offm = s_red[1][0] + s_red[1][1] + s_red[1][2] + s_red[1][3];
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int n = b * kChunk + q * 256 + threadIdx.x;
            int u = -1;
            if (n < K) u = a.U[n];
            const bool isl = (n < Kn) && (u == -1);
            const bool ism = (n >= Kn) && (n < K) && (u != -1);
            const unsigned long long bl = __ballot(isl), bm = __ballot(ism);
            if (lane == 0) {
                s_wl[wave] = __popcll(bl);
                s_wm[wave] = __popcll(bm);
            }
            __syncthreads();
            int wl = 0, wm = 0, tl = 0, tm = 0;
            for (int w2 = 0; w2 < 4; ++w2) {
                if (w2 < wave) {
                    wl += s_wl[w2];
                    wm += s_wm[w2];
                }
                tl += s_wl[w2];
                tm += s_wm[w2];
            }
            if (isl) a.hole_list[offl + wl + __popcll(bl & lanemask_lt())] = n;
            if (ism) {
                a.mover_list[offm + wm + __popcll(bm & lanemask_lt())] = u;
                a.U[n] = -1;  // data[right_track] = -1   (:159)
            }
            offl += tl;
            offm += tm;
            __syncthreads();
        }
        if (b == nchunks - 1 && threadIdx.x == 0) ctl->nleft = offl;  // total left holes (== movers)
    }
}
__global__ __launch_bounds__(256) void k_compact_scatter(RoundArgs a) { k_compact_scatter_body(a); }
struct F_k_compact_scatter {  // (the body as a callable: what a batched launch runs per problem, csrc/host_batch.hpp)
    static __device__ __forceinline__ void run(RoundArgs a) { k_compact_scatter_body(a); }
};


// ... and the round's end (K += evicted - assigned :429, nits += 1 :273; a launch of one thread until round 4): every
// workgroup reads the control block when it starts and counts itself in on Ctl::arrive when it is done; the one that
// arrives last knows that nobody reads the old K any more and writes the new one.
__device__ __forceinline__ void k_compact_fill_body(RoundArgs a) {
    Ctl *ctl = a.ctl;
    const CtlHead head(ctl);
    if (!head.live(a.thr, false)) {  // (uniform over the launch)
        if (blockIdx.x == 0 && threadIdx.x == 0) post_live_status(a.live, a.ticket, head.K, head.err, head.nits);
        return;
    }
    const int nholes = ctl->nholes;
    if (nholes != 0) {
        const int nl = ctl->nleft;
        for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < nl; k += gridDim.x * blockDim.x)
            a.U[a.hole_list[k]] = a.mover_list[k];  // data[left_track] = i   (:158)
    }
    __syncthreads();  // (every thread of the workgroup has read what it needs of the control block)
    if (threadIdx.x == 0 &&
        __hip_atomic_fetch_add(&ctl->arrive, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (int)gridDim.x - 1) {
        __hip_atomic_store(&ctl->arrive, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // for the next launch
        const int Kn = head.K - nholes;
        ctl->K = Kn;  // :429
        ctl->nholes = 0;
        ctl->nleft = 0;
        ctl->nits = head.nits + 1;  // :273
        ctl->grid_rounds += 1;
        post_live_status(a.live, a.ticket, Kn, head.err, head.nits + 1);
    }
}
__global__ __launch_bounds__(256) void k_compact_fill(RoundArgs a) { k_compact_fill_body(a); }
struct F_k_compact_fill {  // (the body as a callable: what a batched launch runs per problem, csrc/host_batch.hpp)
    static __device__ __forceinline__ void run(RoundArgs a) { k_compact_fill_body(a); }
};


// push_all_left + round end in ONE launch for moderate K: a single 1024-thread workgroup walks U[0,K) in
// chunks (block scan with a running carry), writes the two lists, fills, and closes the round.  Replaces four
// launches (count, scatter, fill, round_end) -- most grid rounds have K of a few hundred to a few thousand
// and are bound by launch boundaries, not by work.  The host uses it while K_ub <= kCompactSmallMax.
constexpr int kCompactSmallMax = 32768;
// one 1024-thread workgroup, `nholes` uniform.  4096 list positions per pass: the four loads of a thread are in flight
// together, the per-wavefront counts go through LDS once (double-buffered by pass parity: ONE barrier per pass), and
// the running list lengths are the same in every thread's registers (a pass used to be 1024 positions behind one
// exposed memory latency and three barriers: 10 us for the 10 000 - 30 000 positions of a mid round).
__device__ __forceinline__ void compact_small_body(const RoundArgs &a, Ctl *ctl, int K, int nholes, int err_seen) {
    const int Kn = K - nholes;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    constexpr int kQ = 4;
    __shared__ int s_w[2][2][kQ][16];  // [pass parity][left holes / movers][quarter][wavefront]
    if (nholes > 0) {
        int run_l = 0, run_m = 0, par = 0;  // (uniform)
        // The list entries of pass p + 1 are requested while pass p is worked on, and the barrier of a pass is an
        // LDS-only one (the per-wavefront counts are all that crosses it; the list stores are read behind the full
        // barrier at the end), so the request stays in flight across it: one exposed memory latency per CALL instead
        // of one per pass (K = 20 000: 5 passes; the kernel 10 -> 6 us at C4's mid rounds).
        int un[kQ];
#pragma unroll
        for (int q = 0; q < kQ; ++q) un[q] = a.U[min(q * 1024 + t, K - 1)];
        for (int base = 0; base < K; base += kQ * 1024, par ^= 1) {
            int u[kQ];
            unsigned long long bl[kQ], bm[kQ];
            bool isl[kQ], ism[kQ];
#pragma unroll
            for (int q = 0; q < kQ; ++q) u[q] = un[q];
#pragma unroll
            for (int q = 0; q < kQ; ++q) {  // unconditional (a load inside a branch is waited for at once), masked below
                const int n = base + kQ * 1024 + q * 1024 + t;
                un[q] = a.U[min(n, K - 1)];
            }
#pragma unroll
            for (int q = 0; q < kQ; ++q) {
                const int n = base + q * 1024 + t;
                isl[q] = (n < Kn) & (u[q] == -1);
                ism[q] = (n >= Kn) & (n < K) & (u[q] != -1);
                bl[q] = __ballot(isl[q]);
                bm[q] = __ballot(ism[q]);
                if (lane == 0) {
                    s_w[par][0][q][wave] = __popcll(bl[q]);
                    s_w[par][1][q][wave] = __popcll(bm[q]);
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");  // (LDS only: the next pass's loads stay in flight)
            const int wave_s = __builtin_amdgcn_readfirstlane(wave);
#pragma unroll
            for (int q = 0; q < kQ; ++q) {
                const int n = base + q * 1024 + t;
                // the 16 wavefronts' counts in lanes 0..15 (one DPP row): inclusive scan by row shifts, my wavefront's
                // exclusive prefix and the total by readlane -- 2 LDS reads per quarter instead of 32
                const int cx = lane < 16 ? s_w[par][0][q][lane & 15] : 0, cy = lane < 16 ? s_w[par][1][q][lane & 15] : 0;
                int x = cx, y = cy;
                x += __builtin_amdgcn_update_dpp(0, x, 0x111, 0xF, 0xF, true);  // row_shr:1 (lanes shifted in from outside read 0)
                y += __builtin_amdgcn_update_dpp(0, y, 0x111, 0xF, 0xF, true);
                x += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xF, 0xF, true);
                y += __builtin_amdgcn_update_dpp(0, y, 0x112, 0xF, 0xF, true);
                x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xF, 0xF, true);
                y += __builtin_amdgcn_update_dpp(0, y, 0x114, 0xF, 0xF, true);
                x += __builtin_amdgcn_update_dpp(0, x, 0x118, 0xF, 0xF, true);
                y += __builtin_amdgcn_update_dpp(0, y, 0x118, 0xF, 0xF, true);
                const int wl = __builtin_amdgcn_readlane(x - cx, wave_s), wm = __builtin_amdgcn_readlane(y - cy, wave_s);
                const int tl = __builtin_amdgcn_readlane(x, 15), tm = __builtin_amdgcn_readlane(y, 15);
                if (isl[q]) a.hole_list[run_l + wl + __popcll(bl[q] & lanemask_lt())] = n;
                if (ism[q]) {
                    a.mover_list[run_m + wm + __popcll(bm[q] & lanemask_lt())] = u[q];
                    a.U[n] = -1;  // data[right_track] = -1   (:159)
                }
                run_l += tl;
                run_m += tm;
            }
        }
        __syncthreads();  // the lists are complete (written by other threads of this workgroup)
        const int nl = run_l;  // left holes == movers
        for (int k = t; k < nl; k += 1024) a.U[a.hole_list[k]] = a.mover_list[k];  // data[left_track] = i   (:158)
    }
    if (t == 0) {
        const long long nits = ctl->nits + 1;  // :273
        ctl->K = Kn;  // :429
        ctl->nholes = 0;
        ctl->nleft = 0;
        ctl->nits = nits;
        ctl->grid_rounds += 1;
        post_live_status(a.live, a.ticket, Kn, err_seen, nits);
    }
}

__device__ __forceinline__ void k_compact_small_body(RoundArgs a) {
    Ctl *ctl = a.ctl;
    const CtlHead head(ctl);
    if (!head.live(a.thr, false)) {  // (the round was not live: its ticket is posted all the same -- the host waits for it)
        if (threadIdx.x == 0) post_live_status(a.live, a.ticket, head.K, head.err, head.nits);
        return;
    }
    compact_small_body(a, ctl, head.K, ctl->nholes, head.err);
}
__global__ __launch_bounds__(1024) void k_compact_small(RoundArgs a) { k_compact_small_body(a); }
struct F_k_compact_small {  // (the body as a callable: what a batched launch runs per problem, csrc/host_batch.hpp)
    static __device__ __forceinline__ void run(RoundArgs a) { k_compact_small_body(a); }
};


// RESOLVE + ASSIGN + push_all_left + round end of a round with few bidders in ONE launch (a single 1024-thread
// workgroup; the host uses it while K_ub <= kRoundSmallMax, the round is not sharded over GPUs and the bids were
// made by k_bid<E, RecSource>, i.e. WITHOUT the global atomicMax).  The per-object arg-max of
// :375-385 -- highest bid, earliest list position among equal bids -- is formed in an LDS hash table keyed by
// object (64-bit ds_max on the bid's bit pattern, then ds_min on the position among the holders of the maximum),
// so the resolve phase costs LDS latencies instead of two global atomic round trips; best_key / best_pos are not
// touched.  Such rounds are bound by latency and launch boundaries, not by work.
constexpr int kRoundSmallMax = 2048;
constexpr int kRoundSmallSlots = kRoundSmallMax / 1024;  // list positions per thread, kept in registers
constexpr int kRoundSmallHash = 2 * kRoundSmallMax;      // load factor <= 0.5
// (the part behind the bids' arrival: k_round_small gets them from the launch before, k_round_fused from the other
// workgroups of its own launch)
__device__ __forceinline__ void round_small_body(const RoundArgs &a, Ctl *ctl, const CtlHead &head,
                                                 const int4 (&br)[kRoundSmallSlots],
                                                 const unsigned long long (&key)[kRoundSmallSlots]) {
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int K = head.K;
    __shared__ int s_cnt[16], s_wl[16], s_wm[16];
    __shared__ int s_hole[kRoundSmallMax], s_mover[kRoundSmallMax];  // push_all_left lists
    __shared__ int hObj[kRoundSmallHash], hPos[kRoundSmallHash];
    __shared__ unsigned long long hKey[kRoundSmallHash];
    for (int h = t; h < kRoundSmallHash; h += 1024) {
        hObj[h] = -1;
        hKey[h] = 0ull;
        hPos[h] = kPosNone;
    }
    int obj[kRoundSmallSlots], who[kRoundSmallSlots], prev[kRoundSmallSlots], rstart[kRoundSmallSlots];
    int hs[kRoundSmallSlots];
#pragma unroll
    for (int q = 0; q < kRoundSmallSlots; ++q) {
        obj[q] = br[q].x;
        prev[q] = br[q].y;   // :401 (nothing was assigned since the bid was made)
        who[q] = br[q].z;
        rstart[q] = br[q].w;
    }
    __syncthreads();  // table cleared
#pragma unroll
    for (int q = 0; q < kRoundSmallSlots; ++q) {
        hs[q] = 0;
        if (q * 1024 + t < K) {
            int h = (int)(((unsigned)obj[q] * 2654435761u) >> 16) & (kRoundSmallHash - 1);
            for (;;) {  // open addressing; K <= half the table
                const int old = atomicCAS(&hObj[h], -1, obj[q]);
                if (old == -1 || old == obj[q]) break;
                h = (h + 1) & (kRoundSmallHash - 1);
            }
            hs[q] = h;
            atomicMax(&hKey[h], key[q]);  // best bid of the object
        }
    }
    __syncthreads();
    // first bidder in list order among those holding the best bid of an object (:379, strict ">")
#pragma unroll
    for (int q = 0; q < kRoundSmallSlots; ++q)
        if (q * 1024 + t < K && hKey[hs[q]] == key[q]) atomicMin(&hPos[hs[q]], q * 1024 + t);
    __syncthreads();
    int bpos[kRoundSmallSlots];
#pragma unroll
    for (int q = 0; q < kRoundSmallSlots; ++q) bpos[q] = hPos[hs[q]];
    int holes = 0;
    int u[kRoundSmallSlots];  // U[n] after the assignment phase
#pragma unroll
    for (int q = 0; q < kRoundSmallSlots; ++q) {
        const int n = q * 1024 + t, j = obj[q];
        u[q] = who[q];
        if (n < K && bpos[q] == n) {  // winner of object j; losers see the winner's position
            const int i = who[q];
            PriceRec r;
            r.price = key_to_bid(key[q]);            // p[j] = best_bids[j]   (:397)
            r.owner = i;
            r.ostart = rstart[q];
            a.rec[j] = r;
            a.price[j] = r.price;
            if (prev[q] != -1) a.p2o[prev[q]] = -1;  // :404
            u[q] = prev[q];                          // :409 evicted owner inherits the slot / :412 hole (-1)
            holes += prev[q] == -1;
            a.p2o[i] = j;                            // :417
            a.o2p[j] = i;                            // :418 (:421-422: best_key / best_pos were never written)
        }
    }
    for (int off = 32; off >= 1; off >>= 1) holes += __shfl_xor(holes, off);
    if (lane == 0) s_cnt[wave] = holes;
    __syncthreads();
    int nholes = 0;
    for (int w = 0; w < 16; ++w) nholes += s_cnt[w];
    // push_all_left (:137-162) on the register copy of U: k-th hole in [0, K') <- k-th person in [K', K)
    const int Kn = K - nholes;
    int cl = 0, cm = 0;  // running list lengths (uniform)
    unsigned filled = 0;  // my slots that are left holes: their final value comes from the mover list
#pragma unroll
    for (int q = 0; q < kRoundSmallSlots; ++q) {
        const int n = q * 1024 + t;
        if (q * 1024 >= K) break;  // uniform
        const bool isl = (n < Kn) && (u[q] == -1);
        const bool ism = (n >= Kn) && (n < K) && (u[q] != -1);
        const unsigned long long bl = __ballot(isl), bm = __ballot(ism);
        __syncthreads();  // s_wl / s_wm of the previous slot are no longer read
        if (lane == 0) {
            s_wl[wave] = __popcll(bl);
            s_wm[wave] = __popcll(bm);
        }
        __syncthreads();
        // (the 16 wavefronts' counts in one DPP row, scanned by row shifts: see compact_small_body)
        const int cx = lane < 16 ? s_wl[lane & 15] : 0, cy = lane < 16 ? s_wm[lane & 15] : 0;
        int x = cx, y = cy;
        x += __builtin_amdgcn_update_dpp(0, x, 0x111, 0xF, 0xF, true);
        y += __builtin_amdgcn_update_dpp(0, y, 0x111, 0xF, 0xF, true);
        x += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xF, 0xF, true);
        y += __builtin_amdgcn_update_dpp(0, y, 0x112, 0xF, 0xF, true);
        x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xF, 0xF, true);
        y += __builtin_amdgcn_update_dpp(0, y, 0x114, 0xF, 0xF, true);
        x += __builtin_amdgcn_update_dpp(0, x, 0x118, 0xF, 0xF, true);
        y += __builtin_amdgcn_update_dpp(0, y, 0x118, 0xF, 0xF, true);
        const int wave_s = __builtin_amdgcn_readfirstlane(wave);
        const int wl = __builtin_amdgcn_readlane(x - cx, wave_s), wm = __builtin_amdgcn_readlane(y - cy, wave_s);
        const int tl = __builtin_amdgcn_readlane(x, 15), tm = __builtin_amdgcn_readlane(y, 15);
        if (ism) {
            s_mover[cm + wm + __popcll(bm & lanemask_lt())] = u[q];
            u[q] = -1;  // data[right_track] = -1   (:159)
        }
        if (isl) {
            s_hole[cl + wl + __popcll(bl & lanemask_lt())] = n;
            filled |= 1u << q;
        }
        cl += tl;
        cm += tm;
    }
    __syncthreads();
    // every position is written exactly once: a left hole takes its mover from the LDS lists (as many movers
    // as left holes), every other position its value after the assignment phase
#pragma unroll
    for (int q = 0; q < kRoundSmallSlots; ++q) {
        const int n = q * 1024 + t;
        if (n < K && !((filled >> q) & 1u)) a.U[n] = u[q];
    }
    for (int k = t; k < cl; k += 1024) a.U[s_hole[k]] = s_mover[k];  // data[left_track] = i   (:158)
    if (t == 0) {
        ctl->K = Kn;  // :429
        ctl->nholes = 0;
        ctl->nleft = 0;
        ctl->nits = head.nits + 1;  // :273
        ctl->grid_rounds += 1;
        post_live_status(a.live, a.ticket, Kn, head.err, head.nits + 1);
    }
}

__device__ __forceinline__ void k_round_small_body(RoundArgs a) {
    const int t = threadIdx.x;
    // Everything a position needs comes from its bidder (k_bid<E, RecSource>) in two loads, requested before the
    // control block is read (positions at or beyond K hold stale bids, masked below) and kept in registers.
    Ctl *ctl = a.ctl;
    const CtlHead head(ctl);
    int4 br[kRoundSmallSlots];
    unsigned long long key[kRoundSmallSlots];
    bool never = false;  // (objects are >= 0, a key is the bit pattern of a finite bid + 1)
#pragma unroll
    for (int q = 0; q < kRoundSmallSlots; ++q) {
        br[q] = a.bid_rec[q * 1024 + t];
        key[q] = a.bid_key[min(q * 1024 + t, a.n_rows - 1)];
        never |= (br[q].x < 0) | (key[q] == ~0ull);
    }
    if (!head.live(a.thr, never)) {
        if (t == 0) post_live_status(a.live, a.ticket, head.K, head.err, head.nits);
        return;
    }
    round_small_body(a, ctl, head, br, key);
}
__global__ __launch_bounds__(1024) void k_round_small(RoundArgs a) { k_round_small_body(a); }
struct F_k_round_small {  // (the body as a callable: what a batched launch runs per problem, csrc/host_batch.hpp)
    static __device__ __forceinline__ void run(RoundArgs a) { k_round_small_body(a); }
};


// A whole round with few bidders in ONE launch: the bids of k_bid<E, RecSource, 2> by 1024-thread workgroups (one
// wavefront per list position), and the rest of the round (round_small_body) by whichever workgroup finishes its bids
// LAST.  Every workgroup hands its bids over and counts itself in on Ctl::arrive; nobody waits for anybody, so the
// launch cannot hang whatever else occupies the CUs.  The hand-over does without fences: the bids are device-scope
// atomic stores, complete (vmcnt) before the barrier behind which thread 0 counts the workgroup in with a device-scope
// atomic; the last arriver reads them with device-scope atomic loads issued behind the barrier that hands it the
// count.  Everything else the resolve part touches (U, records, prices, p2o / o2p) nobody wrote in this launch.
// Saves one launch boundary per round (~5 us of ~10.5; 1 400 - 2 100 such rounds per C3 solve), and one launch in four
// of a solve is what bounds several solves in flight on one GPU.
template <class E>
__device__ __forceinline__ void k_round_fused_body(RoundArgs a, E ed) {
    const int t = threadIdx.x, wave = t >> 6;
    constexpr int wpb = 1024 / kWave;
    const int first = blockIdx.x * wpb + wave;
    Ctl *ctl = a.ctl;
    const CtlHead head(ctl);
    const int i_first = a.U[min(first, a.n_rows - 1)];
    // (the exit below depends on the control block ALONE -- it must be the same in every workgroup of the launch, or
    // the count-in would never complete -- and the list entry is requested ahead of it all the same: the empty asm
    // makes it a value the code in front of the exit needs)
    asm volatile("" ::"v"(i_first));
    if (!head.live(a.thr, false)) {  // (uniform over the launch: nobody counts in)
        if (blockIdx.x == 0 && t == 0) post_live_status(a.live, a.ticket, head.K, head.err, head.nits);
        return;
    }
    BidTally tl;
    bid_positions<E, RecSource, 2, true>(a, ed, 0, head.K, first, gridDim.x * wpb, i_first, tl);
    tally_flush<wpb>(a, tl, head.K);
    __shared__ int s_arrived;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (t == 0) s_arrived = __hip_atomic_fetch_add(&ctl->arrive, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    if (s_arrived != (int)gridDim.x - 1) return;  // uniform over the workgroup
    if (t == 0) __hip_atomic_store(&ctl->arrive, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // for the next launch
    int4 br[kRoundSmallSlots];
    unsigned long long key[kRoundSmallSlots];
#pragma unroll
    for (int q = 0; q < kRoundSmallSlots; ++q) {
        const int n = min(q * 1024 + t, a.n_rows - 1);
        const unsigned long long *r = reinterpret_cast<const unsigned long long *>(&a.bid_rec[q * 1024 + t]);
        const unsigned long long r0 = handover_load(r), r1 = handover_load(r + 1);
        br[q] = make_int4((int)(unsigned)(r0 & 0xffffffffull), (int)(unsigned)(r0 >> 32), (int)(unsigned)(r1 & 0xffffffffull),
                          (int)(unsigned)(r1 >> 32));
        key[q] = handover_load(&a.bid_key[n]);
    }
    round_small_body(a, ctl, head, br, key);
}
template <class E>
__global__ __launch_bounds__(1024) void k_round_fused(RoundArgs a, E ed) { k_round_fused_body<E>(a, ed); }
template <class E>
struct F_k_round_fused {  // (the body as a callable: what a batched launch runs per problem, csrc/host_batch.hpp)
    static __device__ __forceinline__ void run(RoundArgs a, E ed) { k_round_fused_body<E>(a, ed); }
};


// The status of everything enqueued so far, posted by a launch of its own: behind a batch of small rounds, whose
// closing kernel (k_round_small, a few microseconds each, thousands per solve) does not pay for four stores to host
// memory every round.
__device__ __forceinline__ void k_post_status_body(const Ctl *ctl, unsigned long long *live, unsigned ticket) {
    post_live_status(live, ticket, ctl->K, ctl->err, ctl->nits);
}
__global__ void k_post_status(const Ctl *ctl, unsigned long long *live, unsigned ticket) { k_post_status_body(ctl, live, ticket); }
struct F_k_post_status {  // (the body as a callable: what a batched launch runs per problem, csrc/host_batch.hpp)
    static __device__ __forceinline__ void run(const Ctl *ctl, unsigned long long *live, unsigned ticket) { k_post_status_body(ctl, live, ticket); }
};

// ... and, behind the eCE pass of a phase end, its verdict in a fifth word
__device__ __forceinline__ void k_post_ece_body(const Ctl *ctl, unsigned long long *live, unsigned ticket) {
    post_live_status(live, ticket, ctl->K, ctl->err, ctl->nits);
    __hip_atomic_store(&live[4], ((unsigned long long)ticket << 32) | (unsigned)ctl->ece_fail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__global__ void k_post_ece(const Ctl *ctl, unsigned long long *live, unsigned ticket) { k_post_ece_body(ctl, live, ticket); }
struct F_k_post_ece {  // (the body as a callable: what a batched launch runs per problem, csrc/host_batch.hpp)
    static __device__ __forceinline__ void run(const Ctl *ctl, unsigned long long *live, unsigned ticket) { k_post_ece_body(ctl, live, ticket); }
};


// eps-phase restart (auction_.pyx:286-290): forget assignments, keep prices.
__device__ __forceinline__ void k_reset_phase_body(Ctl *ctl, int *p2o, int *o2p, PriceRec *rec, int *U,
                                                     int n_rows, int n_cols) {
    const int stride = gridDim.x * blockDim.x;
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    for (int i = t; i < n_rows; i += stride) {
        p2o[i] = -1;
        U[i] = i;
    }
    for (int j = t; j < n_cols; j += stride) {
        o2p[j] = -1;
        rec[j].owner = -1;  // the price stays
    }
    if (t == 0) {
        ctl->K = n_rows;
        ctl->nholes = 0;
        ctl->nleft = 0;
        ctl->arrive = 0;  // (k_round_fused leaves it at 0 itself; a phase starts from a known count whatever came before)
        ctl->ece_fail = 0;  // (the eCE test at the end of this phase finds its flag clear: no fill launch in front of it)
    }
}
__global__ __launch_bounds__(256) void k_reset_phase(Ctl *ctl, int *p2o, int *o2p, PriceRec *rec, int *U,
                                                     int n_rows, int n_cols) { k_reset_phase_body(ctl, p2o, o2p, rec, U, n_rows, n_cols); }
struct F_k_reset_phase {  // (the body as a callable: what a batched launch runs per problem, csrc/host_batch.hpp)
    static __device__ __forceinline__ void run(Ctl *ctl, int *p2o, int *o2p, PriceRec *rec, int *U, int n_rows, int n_cols) { k_reset_phase_body(ctl, p2o, o2p, rec, U, n_rows, n_cols); }
};


}  // namespace misslap
