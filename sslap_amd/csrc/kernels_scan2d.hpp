// kernels_scan2d.hpp -- full-scan bid phase as a 2-D decomposition: (row block) x (column slice).
//
// k_bid_tiled walks ALL column tiles in every workgroup: every CU re-reads the whole price table per scan
// (1.6 MB of LDS fills next to 1.25 MB of edges at C3) and synchronises once per tile.  Here workgroup
// (r, c) keeps ONE column slice c of the prices resident in LDS for the whole launch (one LDS-DMA fill, one
// barrier) and streams the edges of row block r that fall into that slice -- one contiguous range of the
// tile-major copy built with (persons per block = rows of a row block, tile = slice).  There is no per-row
// state: every (person, slice) segment yields a partial top-2 (V, W, G), written to HBM (20 B) and merged
// over the C slices by k_merge2d (one thread per bidder), which also forms the bid (auction_.pyx:360) and
// the running per-object maximum (:375-385).  Extra traffic: 2 x 20 B x C per bidder (C3, C = 16: 128 MB
// next to 321 MB of edges) -- the price of having no tile loop, no refills and no barriers.
//
// Status: correct (parity tests run it), but at C3 it measures 175 us per full scan against 135 us for
// k_bid_tiled -- the per-segment cross-lane reduction and address arithmetic (~460 wave instructions per
// 400 edges) make it issue-bound -- so it is opt-in (options.reserved[2] = 2) and not the default.
//
// Merging partial top-2 sets is exact: values are compared, never recomputed; "later stored index wins
// equal values" (:351) holds across slices because positions in the tile-major copy grow with the slice.
#pragma once
#include "device_common.hpp"
#include "kernels_round.hpp"
#include "kernels_tiled.hpp"

namespace misslap {

struct Scan2dArgs {
    const int2 *tiled;  // tile-major edges, blocks of `rb` persons x C slices
    const int2 *seg;    // {start, real length} per segment
    int C;              // column slices
    int rb;             // persons per row block
    int slice_cols;     // prices per slice (multiple of 128)
    int min_K;          // runs only for K >= min_K
    int nnz;
    double *part_v;     // [C][N] best value of (slice, person)
    double *part_w;     // [C][N] second best
    int *part_g;        // [C][N] position of the best, -1 = no edge in the slice
};

template <int kThreads, int kDepth, int kBatch>
__global__ __launch_bounds__(kThreads) void k_scan2d(RoundArgs a, Scan2dArgs sa) {
    extern __shared__ __attribute__((aligned(16))) double s_price[];  // slice_cols + 2 (+inf slot)
    const Ctl *ctl = a.ctl;
    if (!round_live(ctl, a.thr) || ctl->K < sa.min_K) return;
    const int C = sa.C;
    const int r = (int)blockIdx.x / C, c = (int)blockIdx.x % C;
    const int t = threadIdx.x, lane = t & 63, gl = lane & 7, group = t >> 3;
    constexpr int kGroups = kThreads / 8;
    constexpr int kWaves = kThreads / kWave;
    const int wave_u = __builtin_amdgcn_readfirstlane(t >> 6);
    const int c0 = c * sa.slice_cols;
    {  // the slice of prices, once
        const int pieces = sa.slice_cols / 128;
        const double *gsrc = a.price + c0 + 2 * lane;
        for (int piece = wave_u; piece < pieces; piece += kWaves)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(gsrc + piece * 128),
                                             (__attribute__((address_space(3))) void *)(s_price + piece * 128), 16, 0,
                                             0);
        if (t == 0) s_price[sa.slice_cols] = __builtin_huge_val();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
    const int i0 = r * sa.rb, i1 = min(a.n_rows, i0 + sa.rb);
    const int last = sa.nnz - 1;
    const double ninf = -__builtin_huge_val();
    const size_t pbase = (size_t)c * a.n_rows;

    // kBatch persons per group and iteration, software-pipelined one iteration deep: while batch k is
    // consumed, the edges of batch k+1 and the segment pointers of batch k+2 are in flight (all loads
    // unconditional / clamped, see kernels_tiled.hpp).  Without the batch a wavefront has one iteration's
    // loads in flight and the loop runs at one HBM latency per 8 persons.
    struct Seg {
        int s0[kBatch], s1[kBatch];
        bool bidder[kBatch];
    };
    struct Edges {
        int2 x[kBatch][kDepth];
    };
    constexpr int kStride = kGroups * kBatch;  // persons per workgroup iteration
    auto load_seg = [&](int ibase) {
        Seg sg_;
#pragma unroll
        for (int b = 0; b < kBatch; ++b) {
            const int i = ibase + b * kGroups;
            const int ic = min(i, i1 - 1);
            const int idx = tile_idx(ic, c, C, sa.rb);
            const int2 sp = sa.seg[idx];
            sg_.s0[b] = sp.x;
            sg_.s1[b] = sp.x + sp.y;
            sg_.bidder[b] = (i < i1) && (a.p2o[ic] == -1);  // unassigned persons bid (auction_.pyx:339-340)
        }
        return sg_;
    };
    auto load_edges = [&](const Seg &sg_) {
        Edges e;
#pragma unroll
        for (int b = 0; b < kBatch; ++b)
#pragma unroll
            for (int d = 0; d < kDepth; ++d) e.x[b][d] = sa.tiled[min(sg_.s0[b] + gl + 8 * d, last)];
        return e;
    };
    int ib = i0 + group;
    Seg seg_cur = load_seg(ib), seg_nxt = load_seg(ib + kStride);
    Edges e_cur = load_edges(seg_cur);
    for (; ib < i1; ib += kStride) {
        const Edges e_nxt = load_edges(seg_nxt);
        const Seg seg_nx2 = load_seg(ib + 2 * kStride);
        double v1[kBatch], w[kBatch];
        int g1[kBatch], s1[kBatch], qx[kBatch];
        bool more = false;
#pragma unroll
        for (int b = 0; b < kBatch; ++b) {
            v1[b] = ninf;
            w[b] = ninf;
            g1[b] = -1;
            const int s0 = seg_cur.s0[b];
            s1[b] = seg_cur.bidder[b] ? seg_cur.s1[b] : s0;
#pragma unroll
            for (int d = 0; d < kDepth; ++d) {  // auction_.pyx:350-358, branch-free (see kernels_tiled.hpp)
                const int2 x = e_cur.x[b][d];
                const int q = s0 + gl + 8 * d;
                const bool ok = q < s1[b];
                const double pr = s_price[ok ? x.x - c0 : sa.slice_cols];  // masked-off: +inf
                const double v = (double)__int_as_float(x.y) - pr;
                const bool ge = ok & (v >= v1[b]);
                w[b] = __builtin_fmax(w[b], __builtin_fmin(v, v1[b]));
                v1[b] = __builtin_fmax(v1[b], v);
                g1[b] = ge ? q : g1[b];
            }
            qx[b] = s0 + gl + 8 * kDepth;
            more |= qx[b] < s1[b];
        }
        // segments longer than 8 * kDepth edges: wave-uniform loop, the loads of the whole batch issued
        // together (a per-segment `for` with a load inside costs one HBM latency per segment of the batch)
        while (__any(more)) {
            int2 y[kBatch];
#pragma unroll
            for (int b = 0; b < kBatch; ++b) y[b] = sa.tiled[min(qx[b], last)];
            more = false;
#pragma unroll
            for (int b = 0; b < kBatch; ++b) {
                const bool ok = qx[b] < s1[b];
                const double pr = s_price[ok ? y[b].x - c0 : sa.slice_cols];
                const double v = (double)__int_as_float(y[b].y) - pr;
                const bool ge = ok & (v >= v1[b]);
                w[b] = __builtin_fmax(w[b], __builtin_fmin(v, v1[b]));
                v1[b] = __builtin_fmax(v1[b], v);
                g1[b] = ge ? qx[b] : g1[b];
                qx[b] += 8;
                more |= qx[b] < s1[b];
            }
        }
#pragma unroll
        for (int b = 0; b < kBatch; ++b) {
            const int i = ib + b * kGroups;
            const double V = group8_max_f64(v1[b]);
            const int G = group8_max_i32(v1[b] == V ? g1[b] : -1);
            const double W = group8_max_f64(g1[b] == G ? w[b] : v1[b]);
            if (gl == 0 && seg_cur.bidder[b]) {
                sa.part_v[pbase + i] = V;
                sa.part_w[pbase + i] = W;
                sa.part_g[pbase + i] = G;
            }
        }
        seg_cur = seg_nxt;
        seg_nxt = seg_nx2;
        e_cur = e_nxt;
    }
}

// One thread per bidder (list position): merge the C partial top-2s in slice order, form the bid.
__global__ __launch_bounds__(256) void k_merge2d(RoundArgs a, Scan2dArgs sa) {
    const Ctl *ctl = a.ctl;
    if (!round_live(ctl, a.thr) || ctl->K < sa.min_K) return;
    int lo, hi;
    shard_range(ctl->K, a.rank, a.world, a.shard_min_K, lo, hi);
    const double eps = (double)a.eps;
    const double ninf = -__builtin_huge_val();
    unsigned long long edges = 0;
    int nb = 0, err = 0;
    for (int n = lo + blockIdx.x * blockDim.x + threadIdx.x; n < hi; n += gridDim.x * blockDim.x) {
        const int i = a.U[n];
        double V = ninf, W = ninf;
        int G = -1;
        for (int cb = 0; cb < sa.C; cb += 4) {  // four slices' partials in flight (unconditional loads)
            double pv[4], pw[4];
            int pg[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const size_t k = (size_t)min(cb + u, sa.C - 1) * a.n_rows + i;
                pg[u] = sa.part_g[k];
                pv[u] = sa.part_v[k];
                pw[u] = sa.part_w[k];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (cb + u < sa.C && pg[u] >= 0) {
                    if (pv[u] >= V) {  // a later slice holds later stored indices: it wins equal values (:351)
                        W = V > pw[u] ? V : pw[u];
                        V = pv[u];
                        G = pg[u];
                    } else {
                        W = W > pv[u] ? W : pv[u];
                    }
                }
            }
        }
        const int2 best = sa.tiled[G];  // every row has at least one entry
        const double bid = ((double)__int_as_float(best.y) - W) + eps;  // :360
        if (bid_is_bad(bid)) err |= kErrNegativeBid;
        const unsigned long long key = bid_to_key(bid);
        a.bid_key[n] = key;
        a.bid_obj[n] = best.x;
        atomicMax(&a.best_key[best.x], key);
        edges += (unsigned long long)(a.row_ptr[i + 1] - a.row_ptr[i]);
        nb += 1;
    }
    __shared__ unsigned long long s_e[4];
    __shared__ int s_n[4];
    for (int off = 32; off >= 1; off >>= 1) {
        edges += ((unsigned long long)__shfl_xor((unsigned)(edges >> 32), off) << 32) |
                 (unsigned long long)__shfl_xor((unsigned)(edges & 0xffffffffull), off);
        nb += __shfl_xor(nb, off);
        err |= __shfl_xor(err, off);
    }
    if ((threadIdx.x & 63) == 0) {
        s_e[threadIdx.x >> 6] = edges;
        s_n[threadIdx.x >> 6] = nb;
        if (err) atomicOr(&a.ctl->err, err);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned long long te = s_e[0] + s_e[1] + s_e[2] + s_e[3];
        const int tb = s_n[0] + s_n[1] + s_n[2] + s_n[3];
        if (tb) {
            atomicAdd(&a.ctl->edges, te);
            if (a.world > 1 && a.ctl->K >= a.shard_min_K) atomicAdd(&a.ctl->shard_edges, te);
            atomicAdd(&a.ctl->bids, (unsigned long long)tb);
            if (a.launch_edges) atomicAdd(&a.launch_edges[a.launch_idx], te);
        }
    }
}

}  // namespace misslap
