// kernels_tail.hpp -- the long tail of tiny rounds inside ONE persistent workgroup.
//
// At the BASELINE sizes > 98 % of all rounds have <= 64 bidders and
// every round depends on the prices of the previous one, so the tail is a latency chain, not a
// bandwidth problem.  One 1024-thread workgroup (16 wavefronts, one CU) loops over rounds on the
// device: the unassigned list lives in LDS, each wavefront bids for one person at a time
// (auction_.pyx:339-365), conflicts are resolved in LDS (:375-385), winners are applied (:388-427)
// and the list is compacted (push_all_left, :137-162) without leaving the kernel.  K never grows
// inside an eps-phase (every winner evicts at most one owner), so once K <= threshold the whole
// rest of the phase runs here.  The kernel exits when K == 0 or nits == max_iter.
//
// Visibility: prices / o2p / p2o are written and re-read by this one workgroup only (same CU, same
// vector L1, __syncthreads() between phases); the CSR is read-only.  No other workgroup runs.
#pragma once
#include "device_common.hpp"

namespace misslap {

struct TailArgs {
    Ctl *ctl;
    const int *row_ptr;
    double *price;
    PriceRec *rec;  // {price, owner, owner's row start} per object, see device_common.hpp
    int *p2o;
    int *o2p;
    int *U;
    int thr;
    float eps;
};

constexpr int kHashSize = 2048;  // LDS open-addressing table for K > 64 (load factor <= 0.5)

__device__ __forceinline__ unsigned long long readlane_u64(unsigned long long v, int l) {
    const int lo = __builtin_amdgcn_readlane((int)(unsigned)(v & 0xffffffffull), l);
    const int hi = __builtin_amdgcn_readlane((int)(unsigned)(v >> 32), l);
    return ((unsigned long long)(unsigned)hi << 32) | (unsigned long long)(unsigned)lo;
}

// assignment of one winner (auction_.pyx:396-418); returns the new content of its U slot.  Inside the tail
// kernel the price record is the ONLY copy that is kept current: one 16-byte store per winner instead of
// five scattered ones, and price[] / o2p[] / p2o[] lines stay out of the CU's L2 working set.  k_sync_price /
// k_sync_p2o rebuild the three plain arrays from the records when the kernel has finished.
__device__ __forceinline__ int apply_winner(const TailArgs &a, int person, int pstart, int obj, int prev,
                                            unsigned long long key) {
    PriceRec r;
    r.price = key_to_bid(key);
    r.owner = person;
    r.ostart = pstart;
    a.rec[obj] = r;
    return prev;  // evicted owner inherits the slot (:409); -1 = hole (:412)
}

// After the tail kernel: price[j], o2p[j] from the records; p2o rebuilt as the inverse of o2p.
__global__ __launch_bounds__(256) void k_sync_clear_p2o(int *p2o, int n_rows) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n_rows; i += gridDim.x * blockDim.x) p2o[i] = -1;
}
__global__ __launch_bounds__(256) void k_sync_from_rec(const PriceRec *rec, double *price, int *o2p, int *p2o,
                                                       int n_cols) {
    for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < n_cols; j += gridDim.x * blockDim.x) {
        const PriceRec r = rec[j];
        price[j] = r.price;
        o2p[j] = r.owner;
        if (r.owner >= 0) p2o[r.owner] = j;
    }
}

// STAMP = diagnostic build: wavefront 0 accumulates s_memtime deltas of the four segments of a round
// (bid | barrier | resolve+assign+compact | barrier) into Ctl::dbg; never used for reported timings.
// ---- pair mode: K == 2 (30 % of the small rounds at C3: two eviction chains running side by side) -------------
// One wavefront runs both bidders of a round: their rows are requested together, their price records are gathered
// together (eight gathers in flight instead of four), and each next occupant's row is requested as soon as its
// winning lane is known -- so the round costs one row latency and one gather latency for both bids, with no
// barrier and no LDS traffic.  The reference's round is reproduced operation by operation: both bids use the
// prices of the previous round (records are stored after both bids are formed), RESOLVE keeps the earlier list
// position on equal bids (:379), the evicted owner inherits the winner's slot (:409), push_all_left moves slot 1
// into an emptied slot 0 (:137-162).  Returns with K < 2 (or nits == max_iter) and the list written back.
template <class E>
__device__ __forceinline__ void tail_pair_mode(const TailArgs &a, const E &ed, int *sU, int *sStart, int &K,
                                               long long &nits, const long long max_iter, const double eps,
                                               unsigned long long &edges, unsigned long long &bids, int &err) {
    const int lane = threadIdx.x & (kWave - 1);
    const double ninf = -__builtin_huge_val();
    int pi[2] = {sU[0], sU[1]}, ps[2] = {sStart[0], sStart[1]};
    int c[2][4], e[2];
    double av[2][4];
    auto request = [&](int X, int person, int start) {  // first four 64-edge chunks of a row + its end
#pragma unroll
        for (int u = 0; u < 4; ++u) ed.load_nt(start + u * kWave + lane, c[X][u], av[X][u]);
        e[X] = a.row_ptr[person + 1];
    };
    request(0, pi[0], ps[0]);
    request(1, pi[1], ps[1]);
    for (;;) {
        Top2 x[2];
        int c1[2], o1[2], os1[2];
        double a1[2];
        PriceRec r[2][4];
        int cc[2][4];
#pragma unroll
        for (int X = 0; X < 2; ++X) {
            x[X].v = ninf;
            x[X].w = ninf;
            x[X].g = -1;
            c1[X] = 0;
            o1[X] = -1;
            os1[X] = 0;
            a1[X] = 0.0;
#pragma unroll
            for (int u = 0; u < 4; ++u) {  // all eight gathers are issued before any of them is used
                const bool ok = ps[X] + u * kWave + lane < e[X];
                cc[X][u] = ok ? c[X][u] : -1;
                r[X][u] = a.rec[ok ? c[X][u] : 0];
            }
        }
#pragma unroll
        for (int X = 0; X < 2; ++X) {
            auto update = [&](int cu, double au, const PriceRec &ru, int g) {
                const bool ok = cu >= 0;
                const double v = ok ? au - ru.price : ninf;        // vi = cost - p[j]   (:350)
                const bool ge = ok & (v >= x[X].v);                // :351
                x[X].w = __builtin_fmax(x[X].w, __builtin_fmin(v, x[X].v));  // :353 / :357-358
                x[X].v = __builtin_fmax(x[X].v, v);
                x[X].g = ge ? g : x[X].g;
                c1[X] = ge ? cu : c1[X];
                a1[X] = ge ? au : a1[X];
                o1[X] = ge ? ru.owner : o1[X];
                os1[X] = ge ? ru.ostart : os1[X];
            };
#pragma unroll
            for (int u = 0; u < 4; ++u) update(cc[X][u], av[X][u], r[X][u], ps[X] + u * kWave + lane);
            // rows longer than 256 edges (wave-uniform, rare at the BASELINE densities): plain loads
            for (int base = ps[X] + 4 * kWave; base < e[X]; base += kWave) {
                const int g = base + lane;
                int cu;
                double au;
                ed.load_nt(min(g, e[X] - 1), cu, au);
                const PriceRec ru = a.rec[cu];
                update(g < e[X] ? cu : -1, au, ru, g);
            }
        }
        // winners first (see wave_bid_rec), then the next occupants' rows, then the second-best values
        int src[2], prev[2], pst[2];
        double W[2];
        bool fast[2];
        Top2 t2[2];
#pragma unroll
        for (int X = 0; X < 2; ++X) {
            const int hi = __double2hiint(x[X].v);
            const int k = hi ^ ((hi >> 31) & 0x7fffffff);
            const int kmax = wave_max_i32(k);
            const unsigned long long cand = __ballot(k == kmax);
            fast[X] = __popcll(cand) == 1;  // wave-uniform
            if (fast[X]) {
                src[X] = __ffsll((long long)cand) - 1;
            } else {
                const int g_mine = x[X].g;
                t2[X] = top2_wave_reduce(x[X]);
                src[X] = __ffsll((long long)__ballot(g_mine == t2[X].g)) - 1;
            }
            prev[X] = __builtin_amdgcn_readlane(o1[X], src[X]);
            pst[X] = __builtin_amdgcn_readlane(os1[X], src[X]);
        }
        const int len0 = e[0] - ps[0], len1 = e[1] - ps[1];
        const int col0 = __builtin_amdgcn_readlane(c1[0], src[0]), col1 = __builtin_amdgcn_readlane(c1[1], src[1]);
        // a bidder wins unless both bid on one object; which of the two wins is only known with the bids, so the
        // rows are requested for the common case (both win) and re-requested for a loser below
        request(0, prev[0], pst[0]);
        request(1, prev[1], pst[1]);
        unsigned long long key[2];
#pragma unroll
        for (int X = 0; X < 2; ++X) {
            W[X] = fast[X] ? wave_max_f64(lane == src[X] ? x[X].w : x[X].v) : t2[X].w;
            const double cost = readlane_f64(a1[X], src[X]);
            const double bid = (cost - W[X]) + eps;  // bbest = costbest - wi + eps   (:360)
            if (!(bid >= 0.0)) err |= kErrNegativeBid;
            key[X] = bid_to_key(bid);
        }
        edges += (unsigned long long)(len0 + len1);
        bids += 2;
        nits += 1;
        // RESOLVE (:375-385): strict ">" -- the earlier list position keeps an object on equal bids
        bool win0 = true, win1 = true;
        if (col0 == col1) {
            if (key[1] > key[0]) win0 = false;
            else win1 = false;
        }
        // ASSIGN (:396-418): a winner's slot goes to the evicted owner (or becomes a hole), a loser stays
        if (lane == 0) {
            if (win0) apply_winner(a, pi[0], ps[0], col0, prev[0], key[0]);
            if (win1) apply_winner(a, pi[1], ps[1], col1, prev[1], key[1]);
        }
        if (!win0) request(0, pi[0], ps[0]);  // (wave-uniform, rare) the loser bids again from its own row
        else {
            pi[0] = prev[0];
            ps[0] = pst[0];
        }
        if (!win1) request(1, pi[1], ps[1]);
        else {
            pi[1] = prev[1];
            ps[1] = pst[1];
        }
        // push_all_left (:137-162) on two slots
        if (pi[0] == -1 && pi[1] != -1) {
            pi[0] = pi[1];
            ps[0] = ps[1];
            e[0] = e[1];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                c[0][u] = c[1][u];
                av[0][u] = av[1][u];
            }
            pi[1] = -1;
        }
        K = (pi[0] != -1) + (pi[1] != -1);
        if (K < 2 || nits >= max_iter) break;
    }
    if (lane == 0) {
        sU[0] = pi[0];
        sU[1] = pi[1];
        sStart[0] = ps[0];
        sStart[1] = ps[1];
    }
}

// ---- team mode: 3 <= K <= 16 ------------------------------------------------------------------------------------
// Chain mode on every wavefront: wavefront w serves list slot w, keeps that slot's row in registers and requests the
// row of its next occupant (the owner its bidder evicts inherits the slot, :409 -- true for 99.9 % of the bids) as
// soon as the winning lane is known.  The bids of a round go to LDS; wavefront 0 runs RESOLVE / ASSIGN /
// push_all_left on lanes = slots and publishes the new list; two barriers per round, both ordering LDS traffic
// ONLY (s_waitcnt lgkmcnt(0); s_barrier), so that the row requests stay in flight across the resolve phase.
// Two traps: (1) a wave-uniform row_ptr[person + 1] would be a SCALAR load, which shares lgkmcnt with LDS and would
// make every barrier wait an L2 / HBM latency -- the row end is loaded through the vector path; (2) the winners'
// records are stored by wavefront 0 before the second barrier and gathered by everybody after it (one CU, one
// in-order vector L1), without waiting for the stores' acknowledgement.
constexpr int kTeamMax = 16;
__device__ __forceinline__ void tail_barrier_lds() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

template <class E>
__device__ __forceinline__ void tail_team_mode(const TailArgs &a, const E &ed, int *sU, int *sStart,
                                               unsigned long long *sKey, int *sObj, int *sPrev, int *sPst, int &K,
                                               long long &nits, const long long max_iter, const double eps,
                                               unsigned long long &edges, unsigned long long &bids, int &err) {
    const int lane = threadIdx.x & (kWave - 1), wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const double ninf = -__builtin_huge_val();
    __shared__ int mU[2][kTeamMax], mS[2][kTeamMax], mK[2];  // the list and K, double-buffered by round parity
    int pi, ps, c[4], e;
    double av[4];
    const int lane_zero = (int)__builtin_amdgcn_mbcnt_hi(0u, __builtin_amdgcn_mbcnt_lo(0u, 0u));  // 0, not uniform
    auto request = [&](int person, int start) {  // first four 64-edge chunks of a row + its end (vector loads)
#pragma unroll
        for (int u = 0; u < 4; ++u) ed.load_nt(start + u * kWave + lane, c[u], av[u]);
        e = a.row_ptr[person + 1 + lane_zero];
    };
    if ((int)threadIdx.x < kTeamMax) {
        mU[0][threadIdx.x] = (int)threadIdx.x < K ? sU[threadIdx.x] : -1;
        mS[0][threadIdx.x] = (int)threadIdx.x < K ? sStart[threadIdx.x] : 0;
    }
    pi = wave < K ? sU[wave] : -1;
    ps = wave < K ? sStart[wave] : 0;
    if (wave < K) request(pi, ps);
    __syncthreads();
    int par = 0;
    for (;;) {
        int prev = -1, pst = 0;
        if (wave < K) {  // wave-uniform: BID for my slot (slots < K are always occupied: the list is compact)
            Top2 x;
            x.v = ninf;
            x.w = ninf;
            x.g = -1;
            int c1 = 0, o1 = -1, os1 = 0;
            double a1 = 0.0;
            PriceRec r[4];
            int cc[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const bool ok = ps + u * kWave + lane < e;
                cc[u] = ok ? c[u] : -1;
                r[u] = a.rec[ok ? c[u] : 0];
            }
            auto update = [&](int cu, double au, const PriceRec &ru, int g) {
                const bool ok = cu >= 0;
                const double v = ok ? au - ru.price : ninf;        // vi = cost - p[j]   (:350)
                const bool ge = ok & (v >= x.v);                   // :351
                x.w = __builtin_fmax(x.w, __builtin_fmin(v, x.v)); // :353 / :357-358
                x.v = __builtin_fmax(x.v, v);
                x.g = ge ? g : x.g;
                c1 = ge ? cu : c1;
                a1 = ge ? au : a1;
                o1 = ge ? ru.owner : o1;
                os1 = ge ? ru.ostart : os1;
            };
#pragma unroll
            for (int u = 0; u < 4; ++u) update(cc[u], av[u], r[u], ps + u * kWave + lane);
            for (int base = ps + 4 * kWave; base < e; base += kWave) {  // rows longer than 256 edges
                const int g = base + lane;
                int cu;
                double au;
                ed.load_nt(min(g, e - 1), cu, au);
                const PriceRec ru = a.rec[cu];
                update(g < e ? cu : -1, au, ru, g);
            }
            const int len = e - ps;
            const int hi = __double2hiint(x.v);
            const int k = hi ^ ((hi >> 31) & 0x7fffffff);
            const int kmax = wave_max_i32(k);
            const unsigned long long cand = __ballot(k == kmax);
            const bool fast = __popcll(cand) == 1;  // wave-uniform
            int src;
            Top2 t2;
            if (fast) {
                src = __ffsll((long long)cand) - 1;
            } else {
                const int g_mine = x.g;
                t2 = top2_wave_reduce(x);
                src = __ffsll((long long)__ballot(g_mine == t2.g)) - 1;
            }
            prev = __builtin_amdgcn_readlane(o1, src);
            pst = __builtin_amdgcn_readlane(os1, src);
            const int col = __builtin_amdgcn_readlane(c1, src);
            request(prev, pst);  // the owner my bidder evicts if it wins (prev == -1: a valid, unused address)
            const double W = fast ? wave_max_f64(lane == src ? x.w : x.v) : t2.w;
            const double cost = readlane_f64(a1, src);
            const double bid = (cost - W) + eps;  // bbest = costbest - wi + eps   (:360)
            if (!(bid >= 0.0)) err |= kErrNegativeBid;
            edges += (unsigned long long)len;
            bids += 1;
            if (lane == 0) {
                sKey[wave] = bid_to_key(bid);
                sObj[wave] = col;
                sPrev[wave] = prev;
                sPst[wave] = pst;
            }
        }
        tail_barrier_lds();  // the bids are in LDS; row requests stay in flight
        if (wave == 0) {     // RESOLVE / ASSIGN / push_all_left on lanes = slots
            const bool act = lane < K;
            const unsigned long long lkey = act ? sKey[lane] : 0ull;
            const int lobj = act ? sObj[lane] : (-2 - lane);
            bool lose = false;
            for (int m = 0; m < K; ++m) {  // :375-385, all pairs via readlane
                const int om = __builtin_amdgcn_readlane(lobj, m);
                const bool same = (om == lobj) && (m != lane);
                if (__any(same)) {  // wave-uniform; two bidders on one object are the exception
                    const unsigned long long km = readlane_u64(lkey, m);
                    lose |= same && (km > lkey || (km == lkey && m < lane));
                }
            }
            const int me = act ? mU[par][lane] : -1;
            const int mst = act ? mS[par][lane] : 0;
            const bool won = act && !lose;
            const int lprev = sPrev[lane], lpst = sPst[lane];
            if (won) apply_winner(a, me, mst, lobj, lprev, lkey);  // :396-418
            int u = won ? lprev : me;  // the evicted owner inherits the slot (:409) / hole (:412) / a loser stays
            int st = won ? lpst : mst;
            const unsigned long long kmask = (1ull << K) - 1ull;
            const unsigned long long holes = __ballot(act && u == -1) & kmask;
            const int Kn = K - __popcll(holes);
            const unsigned long long lmask = (1ull << Kn) - 1ull;
            unsigned long long hl = holes & lmask;            // empty slots left of K'
            unsigned long long mv = ~holes & ~lmask & kmask;  // persons right of K'
            while (hl) {  // wave-uniform, rare: k-th hole <- k-th mover (:137-162)
                const int hk = __ffsll((long long)hl) - 1, mk = __ffsll((long long)mv) - 1;
                const int mu = __builtin_amdgcn_readlane(u, mk), ms = __builtin_amdgcn_readlane(st, mk);
                if (lane == hk) {
                    u = mu;
                    st = ms;
                }
                hl &= hl - 1;
                mv &= mv - 1;
            }
            if (lane < kTeamMax) {
                mU[par ^ 1][lane] = lane < Kn ? u : -1;
                mS[par ^ 1][lane] = st;
            }
            if (lane == 0) mK[par ^ 1] = Kn;
        }
        tail_barrier_lds();  // the new list is in LDS, the winners' records have been issued
        par ^= 1;
        const int Kold = K;
        K = mK[par];
        nits += 1;
        if (wave < Kold) {  // my slot's new occupant: usually exactly the row requested above
            const int np = wave < kTeamMax ? mU[par][wave] : -1, ns = wave < kTeamMax ? mS[par][wave] : 0;
            if (np >= 0 && !(prev >= 0 && np == prev && ns == pst)) request(np, ns);  // lost, or moved by push_all_left
            pi = np;
            ps = ns;
        }
        if (K <= 2 || nits >= max_iter) break;
    }
    __syncthreads();
    if (threadIdx.x < kTeamMax) {
        sU[threadIdx.x] = mU[par][threadIdx.x];
        sStart[threadIdx.x] = mS[par][threadIdx.x];
    }
    __syncthreads();
}

template <class E, bool STAMP>
__global__ __launch_bounds__(kTailMax) void k_tail(TailArgs a, E ed) {
    __shared__ int sU[kTailMax];
    __shared__ unsigned long long sKey[kTailMax];
    __shared__ int sObj[kTailMax];
    __shared__ int sPrev[kTailMax];
    __shared__ int sStart[kTailMax];  // row start of the person in slot n, kept next to sU: evicted owners bring
                                      // theirs with the price record, so no row_ptr load precedes a row fetch
    __shared__ int sPst[kTailMax];    // row start of the owner of the object slot n bid on
    __shared__ int sList[kTailMax];
    __shared__ int hObj[kHashSize];
    __shared__ unsigned long long hKey[kHashSize];
    __shared__ int hPos[kHashSize];
    __shared__ int sCnt[3][kTailMax / kWave];
    __shared__ int sK;
    __shared__ long long sNits;

    Ctl *ctl = a.ctl;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    constexpr int nwaves = kTailMax / kWave;
    int K = ctl->K;
    long long nits = ctl->nits;
    const long long max_iter = ctl->max_iter;
    if (K == 0 || K > a.thr || nits >= max_iter) return;  // uniform
    const int K0 = K;
    const long long nits0 = nits;
    sU[t] = (t < K) ? a.U[t] : -1;
    sStart[t] = (t < K) ? a.row_ptr[sU[t]] : 0;
    for (int h = t; h < kHashSize; h += kTailMax) {
        hObj[h] = -1;
        hKey[h] = 0ull;
        hPos[h] = kPosNone;
    }
    const double eps = (double)a.eps;
    unsigned long long edges = 0, bids = 0;
    int err = 0;
    unsigned long long st[5] = {0, 0, 0, 0, 0}, t_prev = 0;
    unsigned long long st2[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, t_prev2 = 0;
    auto stamp = [&](int k) {
        if (STAMP && wave == 0) {
            const unsigned long long t = __builtin_amdgcn_s_memtime();
            st[k] += t - t_prev;
            t_prev = t;
        }
    };
    __syncthreads();
    if (STAMP) t_prev = __builtin_amdgcn_s_memtime();

    for (;;) {
        if (K >= 3 && K <= kTeamMax && !STAMP) {
            // ---- team mode: every wavefront, until K <= 2 (or max_iter), see tail_team_mode
            tail_team_mode(a, ed, sU, sStart, sKey, sObj, sPrev, sPst, K, nits, max_iter, eps, edges, bids, err);
            if (K == 0 || nits >= max_iter) break;
            continue;
        }
        if (K == 2 && !STAMP) {
            // ---- pair mode: wavefront 0 runs the rounds alone until K < 2 (or max_iter), see tail_pair_mode
            if (wave == 0) {
                tail_pair_mode(a, ed, sU, sStart, K, nits, max_iter, eps, edges, bids, err);
                if (lane == 0) {
                    sK = K;
                    sNits = nits;
                }
            }
            __syncthreads();
            K = sK;
            nits = sNits;
            if (K == 0 || nits >= max_iter) break;
            continue;  // K == 1: chain mode
        }
        // ---- BID: one wavefront per bidder ---------------------------------------------------
        if (K == 1) {
            // ---- chain mode: one bidder per round until the phase ends (K never grows).  Wavefront 0 runs
            // the rounds alone, without barriers and without LDS; the other wavefronts wait at the barrier
            // below.  Round r+1's bidder is the owner evicted in round r, and its row start came with the
            // price record, so the only dependent global accesses per round are: the row (edges), the
            // price records of its columns, and row_ptr[i+1] in parallel with the edges.
            if (wave == 0) {
                int i = sU[0];
                int s = sStart[0];
                RowPrefetch pf;
                pf.row_ptr = a.row_ptr;
                pf.issue(ed, nullptr, i, s, lane);  // the first bidder's own row (row end = row_ptr[i + 1])
                pf.advance();
                for (;;) {
                    unsigned long long key;
                    int obj, prev, pstart, e;
                    if (STAMP) {
                        CycleStamp cs{st2, &t_prev2, true};
                        cs(15);
                        cs(0);
                        wave_bid_rec(ed, a.rec, s, a.row_ptr + i + 1, eps, key, obj, prev, pstart, e, err, cs, &pf);
                    } else {
                        wave_bid_rec(ed, a.rec, s, a.row_ptr + i + 1, eps, key, obj, prev, pstart, e, err, NoStamp(), &pf);
                    }
                    pf.advance();
                    edges += (unsigned long long)(e - s);
                    bids += 1;
                    nits += 1;
                    // ASSIGN (:396-418): the single bidder always wins
                    if (lane == 0) apply_winner(a, i, s, obj, prev, key);
                    if (STAMP) {
                        CycleStamp cs{st2, &t_prev2, true};
                        cs(5);
                    }
                    if (prev == -1) {  // nobody evicted: everybody is assigned
                        K = 0;
                        if (lane == 0) sU[0] = -1;
                        break;
                    }
                    i = prev;  // the evicted owner inherits the slot (:409) and bids next
                    s = pstart;
                    if (nits >= max_iter) {
                        if (lane == 0) {
                            sU[0] = i;
                            sStart[0] = s;
                        }
                        break;
                    }
                }
                if (lane == 0) {
                    sK = K;
                    sNits = nits;
                }
            }
            __syncthreads();
            K = sK;
            nits = sNits;
            break;
        }
        for (int n = wave; n < K; n += nwaves) {
            const int i = sU[n];
            const int s = sStart[n];
            unsigned long long key;
            int obj, prev, pstart, e;
            if (STAMP) {
                CycleStamp cs{st2, &t_prev2, wave == 0};
                cs(15);  // (re)arm
                cs(0);   // row pointers landed
                wave_bid_rec(ed, a.rec, s, a.row_ptr + i + 1, eps, key, obj, prev, pstart, e, err, cs);
            } else {
                wave_bid_rec(ed, a.rec, s, a.row_ptr + i + 1, eps, key, obj, prev, pstart, e, err);
            }
            if (lane == 0) {
                sKey[n] = key;
                sObj[n] = obj;
                // owner at the start of the round == what the assignment phase reads (:401): the record of
                // obj is only rewritten by this round's winner of obj, after every bid has been made.
                sPrev[n] = prev;
                sPst[n] = pstart;
            }
            if (STAMP) {
                CycleStamp cs{st2, &t_prev2, wave == 0};
                cs(5);  // LDS writes
            }
            edges += (unsigned long long)(e - s);
            bids += 1;
        }
        stamp(0);
        __syncthreads();
        stamp(1);

        if (K <= kWave) {
            // ---- fast path: the whole rest of the round in wavefront 0, no LDS atomics ------------
            if (wave == 0) {
                const bool act = lane < K;
                const unsigned long long key = act ? sKey[lane] : 0ull;
                const int obj = act ? sObj[lane] : (-2 - lane);
                bool lose = false;
                for (int m = 0; m < K; ++m) {  // RESOLVE (:375-385): K <= 64 all-pairs via readlane
                    const int om = __builtin_amdgcn_readlane(obj, m);
                    const bool same = (om == obj) && (m != lane);
                    if (__any(same)) {  // wave-uniform; two bidders on one object are the exception
                        const unsigned long long km = readlane_u64(key, m);
                        lose |= same && (km > key || (km == key && m < lane));
                    }
                }
                int u = act ? sU[lane] : -1;
                int st = act ? sStart[lane] : 0;  // row start travelling with the slot's person
                if (act && !lose) {
                    u = apply_winner(a, u, st, obj, sPrev[lane], key);
                    st = sPst[lane];
                }
                // push_all_left with ballots
                const unsigned long long kmask = (K >= 64) ? ~0ull : ((1ull << K) - 1ull);
                const unsigned long long holes = __ballot(act && u == -1) & kmask;
                const int Kn = K - __popcll(holes);
                const unsigned long long lmask = (Kn >= 64) ? ~0ull : ((1ull << Kn) - 1ull);
                const unsigned long long hl = holes & lmask;            // empty slots left of K'
                if (hl == 0ull) {  // wave-uniform: nothing to move (no hole, or only holes at the end)
                    if (act) {
                        sU[lane] = (lane < Kn) ? u : -1;
                        sStart[lane] = st;
                    }
                } else {
                    const unsigned long long mv = ~holes & ~lmask & kmask;  // persons right of K'
                    const bool is_hl = (hl >> lane) & 1ull, is_mv = (mv >> lane) & 1ull;
                    if (is_hl) sList[__popcll(hl & lanemask_lt())] = lane;
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                    if (is_mv) {
                        const int dst = sList[__popcll(mv & lanemask_lt())];
                        sU[dst] = u;
                        sStart[dst] = st;
                    }
                    if (act) {
                        if (lane >= Kn) sU[lane] = -1;
                        else if (!is_hl) {
                            sU[lane] = u;
                            sStart[lane] = st;
                        }
                    }
                }
                if (lane == 0) sK = Kn;
            }
        } else {
            // ---- general path (64 < K <= 1024): LDS hash table keyed by object ---------------------
            const bool act = t < K;
            unsigned long long key = 0ull;
            int obj = -1, h = 0;
            if (act) {
                key = sKey[t];
                obj = sObj[t];
                h = (int)(((unsigned)obj * 2654435761u) >> 21) & (kHashSize - 1);
                for (;;) {
                    const int old = atomicCAS(&hObj[h], -1, obj);
                    if (old == -1 || old == obj) break;
                    h = (h + 1) & (kHashSize - 1);
                }
                atomicMax(&hKey[h], key);
            }
            __syncthreads();
            if (act && hKey[h] == key) atomicMin(&hPos[h], t);
            __syncthreads();
            const bool win = act && hPos[h] == t;
            __syncthreads();
            if (act) {  // several threads may clear one slot: identical values
                hObj[h] = -1;
                hKey[h] = 0ull;
                hPos[h] = kPosNone;
            }
            int u = act ? sU[t] : -1;
            int st = act ? sStart[t] : 0;
            if (win) {
                u = apply_winner(a, u, st, obj, sPrev[t], key);
                st = sPst[t];
            }
            const bool hole = act && u == -1;
            const unsigned long long bh = __ballot(hole);
            if (lane == 0) sCnt[0][wave] = __popcll(bh);
            __syncthreads();
            int total = 0;
            for (int w2 = 0; w2 < nwaves; ++w2) total += sCnt[0][w2];
            const int Kn = K - total;
            const bool is_hl = hole && t < Kn;
            const bool is_mv = act && !hole && t >= Kn;
            const unsigned long long bl = __ballot(is_hl), bm = __ballot(is_mv);
            if (lane == 0) {
                sCnt[1][wave] = __popcll(bl);
                sCnt[2][wave] = __popcll(bm);
            }
            __syncthreads();
            int pl = __popcll(bl & lanemask_lt()), pm = __popcll(bm & lanemask_lt());
            for (int w2 = 0; w2 < wave; ++w2) {
                pl += sCnt[1][w2];
                pm += sCnt[2][w2];
            }
            if (is_hl) sList[pl] = t;
            __syncthreads();
            if (is_mv) {
                sU[sList[pm]] = u;
                sStart[sList[pm]] = st;
            }
            if (act) {
                if (t >= Kn) sU[t] = -1;
                else if (!is_hl) {
                    sU[t] = u;
                    sStart[t] = st;
                }
            }
            if (t == 0) sK = Kn;
        }
        stamp(2);
        __syncthreads();
        stamp(3);
        K = sK;
        nits += 1;
        if (K == 0 || nits >= max_iter) break;
    }
    if (STAMP && t == 0) {
        for (int k = 0; k < 4; ++k) atomicAdd(&ctl->dbg[k], st[k]);
        for (int k = 0; k < 6; ++k) atomicAdd(&ctl->dbg[4 + k], st2[k]);
        atomicAdd(&ctl->dbg[10], bids);  // bids made by wavefront 0
    }

    if (t < K0) a.U[t] = sU[t];
    if (lane == 0) {
        if (edges) {
            atomicAdd(&ctl->edges, edges);
            atomicAdd(&ctl->tail_edges, edges);
            atomicAdd(&ctl->bids, bids);
        }
        if (err) atomicOr(&ctl->err, err);
    }
    if (t == 0) {
        ctl->K = K;
        ctl->nits = nits;
        ctl->tail_rounds += nits - nits0;
    }
}

}  // namespace misslap
