// device_common.hpp -- shared device-side definitions of libmisslap (gfx950 only).
//
// Data layout in HBM (one solver handle):
//   edges   : 8 B/edge  = {int32 col, fp32 val}   interleaved (when every value is fp32-exact), or
//             12 B/edge = int32 col[] + fp64 val[] (structure of arrays) otherwise;
//             values already sign-flipped for 'min' (reference auction_.pyx:236-237)
//   row_ptr : int32[N+1]        (reference i_starts_stops, auction_.pyx:223)
//   price   : fp64[M]           (:220)
//   p2o     : int32[N], o2p : int32[M]   (:231-232)
//   U       : int32[N]  unassigned-person list, first K valid (:260).  The reference's
//             person_to_assignment_idx (:261) is NOT kept: it is only ever read for a bidder,
//             for which it equals the bidder's own position in U.
//   best_key: int64[M]  per-object maximum bid as (IEEE bits of the bid)+1, 0 = no bid (:255)
//   best_pos: int32[M]  position in U of the earliest bidder holding that maximum (:256)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace misslap {

constexpr int kWave = 64;
constexpr int kPosNone = 0x7fffffff;
constexpr int kTailMax = 1024;  // threads of the persistent tail workgroup = max K it handles

// sticky device-side error bits (Ctl::err)
constexpr int kErrNegativeBid = 1;   // a bid < 0 was formed (breaks the bits-as-integer ordering)
constexpr int kErrRowsUnsorted = 2;  // loc rows not ascending
constexpr int kErrRowGap = 4;        // a row index is missing (empty row)
constexpr int kErrNonFinite = 8;     // NaN / inf among the values
constexpr int kErrColNegative = 16;  // negative row / column index

// Device-resident control block: the scalar part of the reference's solver state.
struct Ctl {
    int K;                 // num_unassigned (auction_.pyx:198)
    int nholes;            // empty slots left in U[0,K) by this round's assignment phase
    int nleft;             // of which at positions < K' (filled by push_all_left)
    int err;               // sticky error bits
    long long nits;        // auction_.pyx:186
    long long max_iter;    // auction_.pyx:187
    unsigned long long edges;  // edges scanned so far
    unsigned long long bids;   // bids made so far
    int ece_fail;          // set by the eCE kernel when the condition is violated
    int dup_rows;          // rows whose assigned column is stored more than once
    long long grid_rounds;
    long long tail_rounds;
    unsigned long long tail_edges;
    double obj;            // objective accumulator (auction_.pyx:491)
};

// ---- edge storage ------------------------------------------------------------------------------
struct EdgesF32 {  // 8 B / edge
    const int2 *e;
    __device__ __forceinline__ void load(int g, int &col, double &val) const {
        const int2 x = e[g];
        col = x.x;
        val = (double)__int_as_float(x.y);  // exact widening
    }
};
struct EdgesF64 {  // 12 B / edge
    const int *col;
    const double *val;
    __device__ __forceinline__ void load(int g, int &c, double &v) const {
        c = col[g];
        v = val[g];
    }
};

// Bids are >= 0 (bid >= price >= 0), so their IEEE-754 bit patterns order like
// unsigned integers; +1 keeps 0 free as "no bid" (the reference's -1.0 sentinel, auction_.pyx:255)
// and the key also fits a signed int64 for the RCCL MAX all-reduce.
__device__ __forceinline__ unsigned long long bid_to_key(double bid) {
    return (unsigned long long)__double_as_longlong(bid) + 1ull;
}
__device__ __forceinline__ double key_to_bid(unsigned long long key) {
    return __longlong_as_double((long long)(key - 1ull));
}

__device__ __forceinline__ double shfl_xor_f64(double v, int off) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __shfl_xor(lo, off);
    hi = __shfl_xor(hi, off);
    return __hiloint2double(hi, lo);
}

// ---- the bid of one person, computed by one wavefront (auction_.pyx:339-365) ----------------------
// Row [s, e) of the CSR.  Per lane: running (v1 = best value, g1 = its stored index, w = second
// best value counting multiplicity) over its elements in ascending stored index, with the
// reference's ">=" rule (a later equal value replaces the best, :351).  Lanes are then merged by
// a butterfly with  A (+) B = B if (vB > vA) or (vB == vA and gB > gA) else A,
// second = max(loser's best, winner's second).  Every lane ends with the row's result.
// Returns bid key and the chosen object in (key, obj); valid in all lanes.
template <class E>
__device__ __forceinline__ void wave_bid(const E &ed, const double *price, int s, int e, double eps,
                                         unsigned long long &key, int &obj, int &err) {
    const int lane = threadIdx.x & (kWave - 1);
    const double ninf = -__builtin_huge_val();
    double v1 = ninf, w = ninf;
    int g1 = -1;
    for (int base = s; base < e; base += 4 * kWave) {
        int c[4];
        double a[4], pr[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int g = base + u * kWave + lane;
            c[u] = -1;
            a[u] = 0.0;
            if (g < e) ed.load(g, c[u], a[u]);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) pr[u] = (c[u] >= 0) ? price[c[u]] : 0.0;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (c[u] >= 0) {
                const double v = a[u] - pr[u];  // vi = cost - p[j]   (:350)
                if (v >= v1) {                  // :351
                    w = v1;
                    v1 = v;
                    g1 = base + u * kWave + lane;
                } else if (v > w) {             // :357
                    w = v;
                }
            }
        }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        const double v2 = shfl_xor_f64(v1, off);
        const double w2 = shfl_xor_f64(w, off);
        const int g2 = __shfl_xor(g1, off);
        const bool take = (v2 > v1) || (v2 == v1 && g2 > g1);
        const double lose_v = take ? v1 : v2;
        const double win_w = take ? w2 : w;
        w = lose_v > win_w ? lose_v : win_w;
        v1 = take ? v2 : v1;
        g1 = take ? g2 : g1;
    }
    int col;
    double cost;
    ed.load(g1, col, cost);  // g1 >= s: every row has at least one entry
    const double bid = (cost - w) + eps;  // bbest = costbest - wi + eps   (:360)
    if (!(bid >= 0.0)) err |= kErrNegativeBid;
    key = bid_to_key(bid);
    obj = col;
}

__device__ __forceinline__ int lane_id() { return threadIdx.x & (kWave - 1); }
__device__ __forceinline__ unsigned long long lanemask_lt() {
    return (1ull << lane_id()) - 1ull;
}

}  // namespace misslap
