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

#include <type_traits>

namespace misslap {

constexpr int kWave = 64;
constexpr int kPosNone = 0x7fffffff;
constexpr int kTailMax = 512;  // threads of the persistent tail workgroup (8 wavefronts: 256 VGPRs each) = max K it handles

// sticky device-side error bits (Ctl::err)
constexpr int kErrNegativeBid = 1;   // a bid < 0 was formed (breaks the bits-as-integer ordering)
constexpr int kErrRowsUnsorted = 2;  // loc rows not ascending
constexpr int kErrRowGap = 4;        // a row index is missing (empty row)
constexpr int kErrNonFinite = 8;     // NaN / inf among the values
constexpr int kErrColNegative = 16;  // negative row / column index
constexpr int kErrLdsBase = 32;      // k_bid_tiled: the price buffers do not start at LDS address 0
constexpr int kErrTooMany = 64;      // dense ingest: more valid entries than int32 row pointers can address
constexpr int kErrPriceFell = 128;   // a price update lowered a price: eps is below the rounding error of fl(fl(c - w) + eps)
                                     // -- candidate lines (which rely on prices only rising) may have answered wrongly

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
    unsigned long long shard_edges;  // edges scanned in sharded rounds (multi-GPU: this rank's share only)
    unsigned long long cand_hits;    // bids answered from the person's candidate line (no row scan)
    unsigned long long cand_edges;   // edges of those rows (counted in `edges` like every bid, but never read)
    double obj;            // objective accumulator (auction_.pyx:491)
    double obj_abs;        // sum of |contribution| (any order: only used as a bound, see k_obj_sum)
    int obj_minexp;        // smallest binary exponent of a lowest set bit among the contributions
    int n_need;            // entries of RoundArgs::need_list (long rows whose lines the maintenance pass found spent)
    int arrive;            // k_round_fused: workgroups of the launch that have handed their bids over (0 between launches)
    unsigned long long dbg[16]; // tail accounting: [0..2] rounds per mode, [3..5] 10-ns ticks, [12..15] bids / line hits /
                                // builds / hit edges; [6..11] cycles per segment in -DMISSLAP_TAIL_STAMP* builds
    unsigned long long val_cnt[4]; // k_validity: distinct owned objects, sol < 0, sol >= n_rows, invalid selected entries
};

// ---- live status ---------------------------------------------------------------------------------
// The kernel that closes a round (or a run of tail kernels) writes {K, error bits, nits}, each next to the TICKET the
// host gave the launch, into four words of PINNED HOST memory.  The host's loop then learns the state of the rounds it
// has enqueued by polling those words: no copy kernel behind every batch, no stream drain, no driver wake-up (a status
// read used to be hipMemcpyAsync + hipStreamSynchronize: 15-25 us of idle device per read; C4 has one in front of each
// of its 26 big rounds).  A word carries its ticket, so any four words with the same ticket are one consistent status
// whatever order they land in.
__device__ __forceinline__ void post_live_status(unsigned long long *live, unsigned ticket, int K, int err, long long nits) {
    if (!live) return;
    const unsigned long long t = (unsigned long long)ticket << 32;
    __hip_atomic_store(&live[0], t | (unsigned)K, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(&live[1], t | (unsigned)err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(&live[2], t | (unsigned)((unsigned long long)nits & 0xffffffffull), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(&live[3], t | (unsigned)((unsigned long long)nits >> 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// ---- edge storage ------------------------------------------------------------------------------
struct EdgesF32 {  // 8 B / edge
    static constexpr bool kCand = true;  // persons keep candidate lines (see below)
    typedef int2 Slot;                   // a line slot as held in a register: {col, fp32 cost}
    typedef int2 Raw;                    // one edge as loaded (a row requested ahead of its use stays in this form)
    const int2 *e;
    __device__ __forceinline__ Raw load_raw_nt(int g) const {
        typedef int v2i_t __attribute__((ext_vector_type(2)));
        const v2i_t x = __builtin_nontemporal_load(reinterpret_cast<const v2i_t *>(e) + g);
        return make_int2(x.x, x.y);
    }
    static __device__ __forceinline__ void decode(const Raw &x, int &col, double &val) {
        col = x.x;
        val = (double)__int_as_float(x.y);
    }
    __device__ __forceinline__ void load(int g, int &col, double &val) const {
        const int2 x = e[g];
        col = x.x;
        val = (double)__int_as_float(x.y);  // exact widening
    }
    // streaming variant for the tail kernel: rows are read once and must not evict the price records from L2
    __device__ __forceinline__ void load_nt(int g, int &col, double &val) const {
        typedef int v2i_t __attribute__((ext_vector_type(2)));
        const v2i_t x = __builtin_nontemporal_load(reinterpret_cast<const v2i_t *>(e) + g);
        col = x.x;
        val = (double)__int_as_float(x.y);
    }
};
// A line slot of the 12 B/edge layout as held in registers: the 8-byte slot {col, -} plus the candidate's fp64 cost,
// which lives in a second, parallel 256-byte line (the values of this layout do not fit the slot's fp32 field).
struct Slot64 {
    int x, y;
    double c;
};
struct EdgesF64 {  // 12 B / edge
    static constexpr bool kCand = true;  // persons keep candidate lines: 256 B of slots + 256 B of fp64 costs
    typedef Slot64 Slot;
    struct Raw {
        int c;
        double v;
    };
    const int *col;
    const double *val;
    __device__ __forceinline__ Raw load_raw_nt(int g) const {
        Raw x;
        x.c = __builtin_nontemporal_load(col + g);
        x.v = __builtin_nontemporal_load(val + g);
        return x;
    }
    static __device__ __forceinline__ void decode(const Raw &x, int &c, double &v) {
        c = x.c;
        v = x.v;
    }
    __device__ __forceinline__ void load(int g, int &c, double &v) const {
        c = col[g];
        v = val[g];
    }
    __device__ __forceinline__ void load_nt(int g, int &c, double &v) const {
        c = __builtin_nontemporal_load(col + g);
        v = __builtin_nontemporal_load(val + g);
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

// Classification by bit pattern: the library is built with -fno-honor-nans, so a floating-point comparison must
// never be what decides whether a NaN is present.
// "v >= 0" of the reference's dense scan (auction_.pyx:549): +0 .. +inf and -0; false for negatives and every NaN
__device__ __forceinline__ bool dense_entry_valid(double v) {
    const unsigned long long b = (unsigned long long)__double_as_longlong(v);
    return b <= 0x7ff0000000000000ull || b == 0x8000000000000000ull;
}
// a bid that would break the bits-as-integer ordering of bid keys: negative (sign bit set) or NaN
__device__ __forceinline__ bool bid_is_bad(double bid) {
    const unsigned long long b = (unsigned long long)__double_as_longlong(bid);
    return b > 0x7ff0000000000000ull;
}

__device__ __forceinline__ double shfl_xor_f64(double v, int off) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __shfl_xor(lo, off);
    hi = __shfl_xor(hi, off);
    return __hiloint2double(hi, lo);
}

// Diagnostic stamp hook: drains the memory counters, then adds the s_memtime delta to slot k.
struct NoStamp {
    __device__ __forceinline__ void operator()(int) const {}
    __device__ __forceinline__ void light(int) const {}
};
struct CycleStamp {
    unsigned long long *acc;
    unsigned long long *prev;
    bool on;
    __device__ __forceinline__ void operator()(int k) const {
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        const unsigned long long t = __builtin_amdgcn_s_memtime();
        if (on) acc[k] += t - *prev;
        *prev = t;
    }
    __device__ __forceinline__ void light(int k) const {  // without draining the memory counters
        const unsigned long long t = __builtin_amdgcn_s_memtime();
        if (on) acc[k] += t - *prev;
        *prev = t;
    }
};

// ---- cross-lane exchange without LDS: DPP lane permutations ----------------------------------------
constexpr int kDppXor1 = 0xB1;        // quad_perm [1,0,3,2]
constexpr int kDppXor2 = 0x4E;        // quad_perm [2,3,0,1]
constexpr int kDppHalfMirror = 0x141; // lane l <-> 7-l inside each 8 lanes
constexpr int kDppMirror = 0x140;     // lane l <-> 15-l inside each 16 lanes
constexpr int kDppBcast15 = 0x142;    // lane 15 of every row -> all lanes of the next row
constexpr int kDppBcast31 = 0x143;    // lane 31 -> all lanes of rows 2 and 3
template <int CTRL, int ROW_MASK = 0xF>
__device__ __forceinline__ int dpp_i32(int v) {
    return __builtin_amdgcn_update_dpp(v, v, CTRL, ROW_MASK, 0xF, false);  // disabled rows keep v
}
template <int CTRL, int ROW_MASK = 0xF>
__device__ __forceinline__ double dpp_f64(double v) {
    const int lo = dpp_i32<CTRL, ROW_MASK>(__double2loint(v));
    const int hi = dpp_i32<CTRL, ROW_MASK>(__double2hiint(v));
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double readlane_f64(double v, int l) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), l);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
    return __hiloint2double(hi, lo);
}

// all-reduce maximum over the 64 lanes, result uniform: 4 in-row DPP steps (xor 1, xor 2, mirrors),
// two row-broadcast steps that leave the total in lane 63, one v_readlane.  No LDS (ds_bpermute).
#define MISSLAP_WAVE_MAX_STEP(T, EXCH) \
    {                                  \
        const T o = EXCH;              \
        v = o > v ? o : v;             \
    }
__device__ __forceinline__ int wave_max_i32(int v) {
    // (mov_dpp form: folds into v_max_i32_dpp, one instruction per step)
    MISSLAP_WAVE_MAX_STEP(int, __builtin_amdgcn_mov_dpp(v, kDppXor1, 0xF, 0xF, true))
    MISSLAP_WAVE_MAX_STEP(int, __builtin_amdgcn_mov_dpp(v, kDppXor2, 0xF, 0xF, true))
    MISSLAP_WAVE_MAX_STEP(int, __builtin_amdgcn_mov_dpp(v, kDppHalfMirror, 0xF, 0xF, true))
    MISSLAP_WAVE_MAX_STEP(int, __builtin_amdgcn_mov_dpp(v, kDppMirror, 0xF, 0xF, true))
    MISSLAP_WAVE_MAX_STEP(int, (dpp_i32<kDppBcast15, 0xA>(v)))
    MISSLAP_WAVE_MAX_STEP(int, (dpp_i32<kDppBcast31, 0xC>(v)))
    return __builtin_amdgcn_readlane(v, 63);
}
__device__ __forceinline__ double wave_max_f64(double v) {  // no NaNs on this path: v_max_f64 is exact
    v = __builtin_fmax(v, dpp_f64<kDppXor1>(v));
    v = __builtin_fmax(v, dpp_f64<kDppXor2>(v));
    v = __builtin_fmax(v, dpp_f64<kDppHalfMirror>(v));
    v = __builtin_fmax(v, dpp_f64<kDppMirror>(v));
    v = __builtin_fmax(v, (dpp_f64<kDppBcast15, 0xA>(v)));
    v = __builtin_fmax(v, (dpp_f64<kDppBcast31, 0xC>(v)));
    return readlane_f64(v, 63);
}
#undef MISSLAP_WAVE_MAX_STEP

// running top-2 of a set of (value, stored index): best value v, its index g, second-best value w
// (counting multiplicity), with the reference's ">=" rule: among equal values the LAST stored index is
// the best one (auction_.pyx:351).
struct Top2 {
    double v, w;
    int g;
};
// Wave-wide top-2 from per-lane top-2s, as three scalar all-reduces:
//   V = max v;  G = max{ g : v == V } (last index wins ties);  W = max( w of G's lane, v of every other lane ).
// A lane other than G's whose best equals V contributes V to W: multiplicity is counted like the
// reference's sequential scan does (wi = vbest on a repeated maximum, :353).
__device__ __forceinline__ Top2 top2_wave_reduce(const Top2 &x) {
    Top2 r;
    r.v = wave_max_f64(x.v);
    r.g = wave_max_i32(x.v == r.v ? x.g : -1);
    r.w = wave_max_f64(x.g == r.g ? x.w : x.v);
    return r;
}

// Per-object record used by the latency-bound tail kernel: one 16-byte gather brings the price AND the
// current owner and the start of the owner's CSR row, so the next bidder of an eviction chain (the evicted
// owner) and the address of its row are known without two further dependent loads (o2p, row_ptr).
// `price[]` (plain fp64, used by the streaming kernels) and `rec[]` always hold the same prices.
struct __attribute__((aligned(16))) PriceRec {
    double price;
    int owner;   // == o2p[j]
    int ostart;  // row_ptr[owner] (undefined when owner == -1)
};

__device__ __forceinline__ int lane_id() { return threadIdx.x & (kWave - 1); }
__device__ __forceinline__ unsigned long long lanemask_lt() {
    return (1ull << lane_id()) - 1ull;
}

// ---- per-person candidate lines ---------------------------------------------------------------------------------
// Prices only rise (a bid is >= price + eps, auction_.pyx:360, and prices survive the eps-phase restart, :283-292),
// so the value  cost - p[j]  of every edge only falls.  When a wavefront has scanned a whole row it therefore knows a
// bound that holds for the rest of the solve: pick a threshold t, remember the row's edges with value >= t ("the
// candidates", at most kCandMax of them, in stored order) and tau = t; every OTHER edge of the row has value <= tau
// now and for ever.  A later bid of the same person is answered from the candidates alone -- 256 bytes and <= 30
// price look-ups instead of the whole row -- whenever that is provably the same bid:
//     v1 > tau   the best candidate beats every edge outside the line strictly, so it is the row's best, and among
//                equal candidates the later stored index wins exactly as in the full scan (:351)
//     v2 >= tau  no edge outside the line exceeds the second-best candidate, so it is the row's second best (:357)
// and bid = (cost1 - v2) + eps is formed with the same operations in the same order (:360).  Otherwise the row is
// scanned in full and the line is rebuilt.  Nothing is approximate: a line only decides WHICH edges are looked at.
//
// Line of person i = 32 slots of 8 bytes at cand[32 * i]: slot 0 = tau (fp64; +inf = no line yet), slots 1..30 =
// {int32 col, fp32 val} (col -1 = empty), slot 31 = {row length, 0} (the scanned-edge statistics count whole rows).
// In the 12 B/edge layout (values that are not fp32-exact) the slot's fp32 field is unused and the cost lives, as fp64,
// at the same index of a parallel line (Slot64 / LineIO).  Rows longer than kCandRowMax are never cached.
constexpr int kCandLanes = 32;
constexpr int kCandMax = 30;
constexpr int kCandMin = 24;            // the threshold search aims at [kCandMin, kCandMax] candidates
constexpr int kCandRowMax = 4 * kWave;  // the full scan keeps four 64-edge chunks in registers

// Lane permutation that the compiler can fold into the consuming VOP2 instruction (v_max_i32_dpp ...): every lane
// of these controls reads a valid source lane, so the "old" operand of update_dpp is not needed.
template <int CTRL>
__device__ __forceinline__ int dppf_i32(int v) {
    return __builtin_amdgcn_mov_dpp(v, CTRL, 0xF, 0xF, true);
}
template <int CTRL>
__device__ __forceinline__ double dppf_f64(double v) {
    return __hiloint2double(dppf_i32<CTRL>(__double2hiint(v)), dppf_i32<CTRL>(__double2loint(v)));
}
// maximum over each 32-lane half of the wavefront; valid in lanes 16..31 (half 0) and 48..63 (half 1)
__device__ __forceinline__ double half_max_f64(double v) {
    v = __builtin_fmax(v, dppf_f64<kDppXor1>(v));
    v = __builtin_fmax(v, dppf_f64<kDppXor2>(v));
    v = __builtin_fmax(v, dppf_f64<kDppHalfMirror>(v));
    v = __builtin_fmax(v, dppf_f64<kDppMirror>(v));
    v = __builtin_fmax(v, (dpp_f64<kDppBcast15, 0xA>(v)));
    return v;
}
__device__ __forceinline__ int half_max_i32(int v) {
    v = max(v, dppf_i32<kDppXor1>(v));
    v = max(v, dppf_i32<kDppXor2>(v));
    v = max(v, dppf_i32<kDppHalfMirror>(v));
    v = max(v, dppf_i32<kDppMirror>(v));
    v = max(v, (dpp_i32<kDppBcast15, 0xA>(v)));
    return v;
}

// the cost of a line slot's candidate (lane-wise / of lane `sl`, wave-uniform), and the lines in memory
__device__ __forceinline__ double slot_cost(const int2 &s) { return (double)__int_as_float(s.y); }
__device__ __forceinline__ double slot_cost(const Slot64 &s) { return s.c; }
__device__ __forceinline__ double slot_cost_at(const int2 &s, int sl) {
    return (double)__int_as_float(__builtin_amdgcn_readlane(s.y, sl));
}
__device__ __forceinline__ double slot_cost_at(const Slot64 &s, int sl) { return readlane_f64(s.c, sl); }
template <class Slot>
struct LineIO;
template <>
struct LineIO<int2> {
    static __device__ __forceinline__ int2 load(const int2 *cand, const double *, size_t idx) { return cand[idx]; }
    static __device__ __forceinline__ int2 make(int x, int y) { return make_int2(x, y); }
};
template <>
struct LineIO<Slot64> {
    static __device__ __forceinline__ Slot64 load(const int2 *cand, const double *cost, size_t idx) {
        const int2 s = cand[idx];
        Slot64 r;
        r.x = s.x;
        r.y = s.y;
        r.c = cost[idx];
        return r;
    }
    static __device__ __forceinline__ Slot64 make(int x, int y) {
        Slot64 r;
        r.x = x;
        r.y = y;
        r.c = 0.0;
        return r;
    }
};

// What one half-wavefront (32 lanes) knows after evaluating one person's line.  All members are wave-uniform.
struct CandBid {
    bool hit;
    unsigned long long key;  // bid as key
    int obj;                 // object bid on
    int prev, pstart;        // its owner and the owner's row start (only when the price source carries them)
    int len;                 // row length of the person
};

// price sources of the line evaluation / the full scan: the tail's 16-byte records or the plain price array
struct RecSource {
    const PriceRec *rec;
    __device__ __forceinline__ PriceRec get(int col) const { return rec[col]; }
};
struct PriceSource {
    const double *price;
    __device__ __forceinline__ PriceRec get(int col) const {
        PriceRec r;
        r.price = price[col];
        r.owner = -1;
        r.ostart = 0;
        return r;
    }
};

// Evaluate the lines of up to two persons at once: lanes 0..31 hold the slots of person A, lanes 32..63 those of
// person B (`slot` = the lane's 8-byte slot; act0 / act1 = the half holds a person at all, wave-uniform).  One gather
// serves both.  out[0] / out[1] are wave-uniform.
// Three 32-lane DPP reductions per half: the best value, the LAST slot holding it, the second-best value.  As soon as
// the winners are known -- and everything else has been read out of `slot` -- `early(out)` is called with obj / prev /
// pstart of both halves filled in: the owners of the winning objects are the next bidders if the bids win, so the
// caller can request their lines (into `slot` itself) before the rest of the round is computed.
// the lane's record gather of a two-person line evaluation (split off so that a caller with several lines in flight
// can issue all gathers before it evaluates any of them)
template <class Slot, class Src>
__device__ __forceinline__ PriceRec cand_gather2(const Slot slot, const bool act0, const bool act1, const Src &src) {
    const int lane = lane_id(), l32 = lane & (kCandLanes - 1);
    const bool active = lane < kCandLanes ? act0 : act1;
    const bool is_cand = active & (l32 >= 1) & (l32 <= kCandMax) & (slot.x >= 0);
    return src.get(is_cand ? slot.x : 0);
}
template <class Slot, class Early, class S = NoStamp>
__device__ __forceinline__ void cand_eval2_r(Slot &slot, const PriceRec r, const bool act0, const bool act1,
                                             const double eps, CandBid (&out)[2], int &err, Early &&early,
                                             const S &stamp = S(), int *alive = nullptr);
template <class Slot, class Src, class Early, class S = NoStamp>
__device__ __forceinline__ void cand_eval2(Slot &slot, const bool act0, const bool act1, const Src &src,
                                           const double eps, CandBid (&out)[2], int &err, Early &&early,
                                           const S &stamp = S(), int *alive = nullptr) {
    stamp(1);  // (diagnostic builds: drains the memory counters) the line has landed
    const PriceRec r = cand_gather2(slot, act0, act1, src);
    stamp(2);  // the records have landed
    cand_eval2_r(slot, r, act0, act1, eps, out, err, early, stamp, alive);
}
template <class Slot, class Early, class S>
__device__ __forceinline__ void cand_eval2_r(Slot &slot, const PriceRec r, const bool act0, const bool act1,
                                             const double eps, CandBid (&out)[2], int &err, Early &&early,
                                             const S &stamp, int *alive) {
    const int lane = lane_id(), l32 = lane & (kCandLanes - 1);
    const double ninf = -__builtin_huge_val();
    const bool active = lane < kCandLanes ? act0 : act1;
    const bool is_cand = active & (l32 >= 1) & (l32 <= kCandMax) & (slot.x >= 0);
    const double cost = slot_cost(slot);
    double tau[2];
    tau[0] = readlane_f64(__hiloint2double(slot.y, slot.x), 0);
    tau[1] = readlane_f64(__hiloint2double(slot.y, slot.x), kCandLanes);
    out[0].len = __builtin_amdgcn_readlane(slot.x, kCandLanes - 1);
    out[1].len = __builtin_amdgcn_readlane(slot.x, 2 * kCandLanes - 1);
    const double v = is_cand ? cost - r.price : ninf;  // vi = cost - p[j]   (:350)
    if (alive) {  // (callers that ask: k_bid) candidates still at or above tau: how much life the line has left
        const unsigned long long al = __ballot(is_cand & (v >= (lane < kCandLanes ? tau[0] : tau[1])));
        alive[0] = __popc((unsigned)al);
        alive[1] = __popc((unsigned)(al >> 32));
    }
    // (Straight-line on purpose: a "winner first" shortcut -- one 32-bit reduction on the values' high words, a ballot
    // and a wave-uniform branch -- was measured 3-4 % SLOWER here: in this latency-bound single-wavefront code the
    // scalar compare / branch / indexed-readlane chain costs more than the 64-bit DPP pass it saves.)
    double V[2], W[2];
    int G[2];
    {
        const double vm = half_max_f64(v);
        V[0] = readlane_f64(vm, 31);
        V[1] = readlane_f64(vm, 63);
        const double Vv = lane < kCandLanes ? V[0] : V[1];
        const int gm = half_max_i32((is_cand & (v == Vv)) ? l32 : -1);  // the LAST slot (= stored index) holding the best
        G[0] = __builtin_amdgcn_readlane(gm, 31);
        G[1] = __builtin_amdgcn_readlane(gm, 63);
    }
    double c1[2];
#pragma unroll
    for (int X = 0; X < 2; ++X) {
        const int sl = kCandLanes * X + max(G[X], 0);
        out[X].obj = __builtin_amdgcn_readlane(slot.x, sl);
        out[X].prev = __builtin_amdgcn_readlane(r.owner, sl);
        out[X].pstart = __builtin_amdgcn_readlane(r.ostart, sl);
        c1[X] = slot_cost_at(slot, sl);
    }
    early(out);  // `slot` is dead from here on
    stamp.light(3);  // winners known, next lines requested
    const int Gv = lane < kCandLanes ? G[0] : G[1];
    const double wm = half_max_f64(l32 == Gv ? ninf : v);  // second best, counting multiplicity
    W[0] = readlane_f64(wm, 31);
    W[1] = readlane_f64(wm, 63);
#pragma unroll
    for (int X = 0; X < 2; ++X) {
        out[X].hit = (G[X] >= 0) & (V[X] > tau[X]) & (W[X] >= tau[X]);
        const double bid = (c1[X] - W[X]) + eps;  // bbest = costbest - wi + eps   (:360)
        if (out[X].hit && bid_is_bad(bid)) err |= kErrNegativeBid;
        out[X].key = bid_to_key(bid);
    }
}
// The same for ONE person (lanes 0..31 hold its line, `cls` = the lane is one of them and its slot is a candidate
// slot, i.e. 1 <= lane <= kCandMax): the chain of single-bidder rounds is half of all rounds at C3 and two thirds
// at C5, and a round is bound by the length of this dependent instruction sequence, not by memory.
// the lane's record gather of a one-person line evaluation (split off: a caller may issue the gather of the NEXT round's
// line ahead of a barrier and evaluate it behind it, kernels_tail.hpp)
template <class Slot, class Src>
__device__ __forceinline__ PriceRec cand_gather1(const Slot &slot, const bool cls, const Src &src) {
    const bool is_cand = cls & (slot.x >= 0);
    return src.get(is_cand ? slot.x : 0);
}
template <class Slot, class Early, class S = NoStamp>
__device__ __forceinline__ void cand_eval1_r(Slot &slot, const PriceRec r, const bool cls, const double eps, CandBid &out,
                                             unsigned &bad_hi, Early &&early, const S &stamp = S());
template <class Slot, class Src, class Early, class S = NoStamp>
__device__ __forceinline__ void cand_eval1(Slot &slot, const bool cls, const Src &src, const double eps, CandBid &out,
                                           unsigned &bad_hi, Early &&early, const S &stamp = S()) {
    stamp(1);  // (diagnostic builds: drains the memory counters) the line has landed
    const PriceRec r = cand_gather1(slot, cls, src);
    stamp(2);  // the records have landed
    cand_eval1_r(slot, r, cls, eps, out, bad_hi, early, stamp);
}
template <class Slot, class Early, class S>
__device__ __forceinline__ void cand_eval1_r(Slot &slot, const PriceRec r, const bool cls, const double eps, CandBid &out,
                                             unsigned &bad_hi, Early &&early, const S &stamp) {
    const int lane = lane_id();
    const double ninf = -__builtin_huge_val();
    const bool is_cand = cls & (slot.x >= 0);
    const double cost = slot_cost(slot);
    const double tau = readlane_f64(__hiloint2double(slot.y, slot.x), 0);
    out.len = __builtin_amdgcn_readlane(slot.x, kCandLanes - 1);
    const double v = is_cand ? cost - r.price : ninf;  // vi = cost - p[j]   (:350)
    const int hi = __double2hiint(v);
    const int k = hi ^ ((hi >> 31) & 0x7fffffff);  // signed order of k == order of the doubles' high words
    const int km = __builtin_amdgcn_readlane(half_max_i32(k), 31);
    const unsigned eq = (unsigned)(__ballot(k == km) & 0xffffffffull);
    double V;
    int G;
    if (__popc(eq) == 1) {  // wave-uniform, the common case: the winner is known after one 32-bit reduction
        G = __ffs((int)eq) - 1;
        V = readlane_f64(v, G);
    } else {
        V = readlane_f64(half_max_f64(v), 31);
        G = __builtin_amdgcn_readlane(half_max_i32((is_cand & (v == V)) ? lane : -1), 31);  // the LAST slot holding it
    }
    const int sl = max(G, 0);
    out.obj = __builtin_amdgcn_readlane(slot.x, sl);
    out.prev = __builtin_amdgcn_readlane(r.owner, sl);
    out.pstart = __builtin_amdgcn_readlane(r.ostart, sl);
    const double c1 = slot_cost_at(slot, sl);
    early(out);  // `slot` is dead from here on
    stamp.light(3);
    const double W = readlane_f64(half_max_f64(lane == G ? ninf : v), 31);  // second best, counting multiplicity
    out.hit = (G >= 0) & (V > tau) & (W >= tau);
    const double bid = (c1 - W) + eps;  // bbest = costbest - wi + eps   (:360)
    // a bid that breaks the bits-as-integer order of the keys (negative: sign bit; NaN): the running maximum of the high
    // words of the bids that count, tested once when the kernel ends (bad_hi_is_error) -- two instructions per bid
    bad_hi = max(bad_hi, out.hit ? (unsigned)__double2hiint(bid) : 0u);
    out.key = bid_to_key(bid);
}
// (+inf = 0x7ff00000:00000000 is a legal bid; every NaN an operation produces and every negative value has a larger high word)
__device__ __forceinline__ bool bad_hi_is_error(unsigned bad_hi) { return bad_hi > 0x7ff00000u; }
struct NoEarly {
    __device__ __forceinline__ void operator()(const CandBid (&)[2]) const {}
    __device__ __forceinline__ void operator()(const CandBid &) const {}
};

// What a full scan leaves behind for the (re)build of the person's line: column, cost and value of the lane's
// element in each of the row's four 64-edge chunks, the row's best and second-best value.  Kept in registers so that
// the caller can publish its bid / request the next data first and build the line while it waits.
struct CandBuildArgs {
    int c[4];
    double a[4], v[4];
    double V, W;
    int len;
    bool want;  // the row fits the four chunks and the layout keeps lines
};

// (Re)build the line of `person` from a full scan of its row (len <= kCandRowMax).  The threshold t is searched
// below W (count(v >= W) >= 2): the first probe is W - hint, where `hint` is the distance the wavefront's previous
// build ended with (rows of one problem look alike, so one or two probes usually suffice); the distance then
// doubles / halves until the count has passed the window [kCandMin, kCandMax], followed by a short bisection.
// Every probe is four compares + popcounts on the wave's ballots.  Any t <= W with a count in [2, kCandMax]
// gives a valid line: the search only decides how full the line gets.
// `cost` = the parallel line of fp64 costs (12 B/edge layout; nullptr: the cost goes into the slot as an fp32).
__device__ __forceinline__ void cand_build(int2 *cand, double *cost, int person, const CandBuildArgs &ba, double eps,
                                           double &hint) {
    const int lane = lane_id();
    const double ninf = -__builtin_huge_val();
    const int len = ba.len;
    const double W = ba.W;
    unsigned long long okm[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int rem = len - kWave * u;
        okm[u] = rem >= kWave ? ~0ull : (rem <= 0 ? 0ull : ((1ull << rem) - 1ull));
    }
    auto count = [&](double t) {
        int n = 0;
#pragma unroll
        for (int u = 0; u < 4; ++u) n += __popcll(__ballot(ba.v[u] >= t) & okm[u]);
        return n;
    };
    double t = ninf;
    int n = len;
    if (len > kCandMax) {
        n = count(W);
        if (n > kCandMax) return;  // more than kCandMax ties at the top: the old line (if any) stays valid
        t = W;
        // distances: d_ok = largest known with count <= kCandMax, d_bad = smallest known with count > kCandMax
        double d_ok = 0.0, d_bad = __builtin_huge_val();
        double d = hint > 0.0 ? hint : 4.0 * __builtin_fmax(ba.V - W, eps);
        for (int k = 0; k < 40 && n < kCandMin; ++k) {
            const double t2 = W - d;
            const int n2 = count(t2);
            if (n2 <= kCandMax) {
                d_ok = d;
                t = t2;
                n = n2;
                d = d_bad == __builtin_huge_val() ? 2.0 * d : 0.5 * (d_ok + d_bad);
            } else {
                d_bad = d;
                d = 0.5 * (d_ok + d_bad);
            }
            if (!(d > d_ok) || !(d < d_bad)) break;  // the interval has collapsed (or d overflowed)
        }
        hint = d_ok > 0.0 ? d_ok : hint;
    }
    int2 *line = cand + (size_t)person * kCandLanes;
    int base = 0;
#pragma unroll
    for (int u = 0; u < 4; ++u) {  // candidates in stored order: rank = elements before it that qualify
        const unsigned long long m = __ballot(ba.v[u] >= t) & okm[u];
        if ((m >> lane) & 1ull) {
            const int at = 1 + base + __popcll(m & lanemask_lt());
            line[at] = make_int2(ba.c[u], cost ? 0 : __float_as_int((float)ba.a[u]));
            if (cost) cost[(size_t)person * kCandLanes + at] = ba.a[u];
        }
        base += __popcll(m);
    }
    if (lane < kCandLanes && (lane == 0 || lane == kCandLanes - 1 || lane > n)) {
        int2 x = make_int2(-1, 0);  // empty slot
        if (lane == 0) x = make_int2(__double2loint(t), __double2hiint(t));
        if (lane == kCandLanes - 1) x = make_int2(len, 0);
        line[lane] = x;
    }
}

// The bid of one person by a FULL scan of its row [s, e) by one wavefront (auction_.pyx:339-365):
// Per lane a running (v = best value, g = its stored index, w = second best value counting multiplicity) over its
// elements in ascending stored index, with the reference's ">=" rule (a later equal value replaces the best, :351);
// the lane also remembers the (col, cost, owner) of its own best element.  Lanes are merged winner-first: if exactly
// one lane holds the largest HIGH WORD of the per-lane best values, that lane holds the row's best element (no tie
// is possible), so one 32-bit wave maximum replaces the 64-bit value and index passes (top2_wave_reduce otherwise);
// the per-lane scan; `ba` receives what the (re)build of the person's candidate line needs.  `e` may still be in flight
// when the function is entered: the first four chunks are requested before it is used (the edge arrays are padded).
// kPre: the first four chunks were requested earlier and arrive in `pre`.  kNT: the row is read once and must not
// evict the price records from L2 (tail kernel).
template <class E, class Src, bool kNT, bool kPre>
__device__ __forceinline__ void wave_bid_full(const E &ed, const Src &src, int s, int e_in,
                                              const typename E::Raw (&pre)[4], double eps, CandBid &out,
                                              CandBuildArgs &ba, int &err) {
    const int lane = lane_id();
    const double ninf = -__builtin_huge_val();
    Top2 x;
    x.v = ninf;
    x.w = ninf;
    x.g = -1;
    int c1 = 0, o1 = -1, os1 = 0;
    double a1 = 0.0;
    int c[4];
    double a[4];
    auto load = [&](int g, int &cc, double &aa) {
        if (kNT) ed.load_nt(g, cc, aa);
        else ed.load(g, cc, aa);
    };
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        if (kPre) E::decode(pre[u], c[u], a[u]);
        else load(s + u * kWave + lane, c[u], a[u]);  // speculative: e may not have landed
    }
    const int e = __builtin_amdgcn_readfirstlane(e_in);  // (the same value in every lane)
    auto chunk = [&](int base, auto first_) {
        constexpr bool kFirst = decltype(first_)::value;
        PriceRec r[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const bool ok = base + u * kWave + lane < e;
            if (!ok) c[u] = -1;
            r[u] = src.get(ok ? c[u] : 0);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {  // branch-free: a masked-off element has value -inf and changes nothing
            const bool ok = c[u] >= 0;
            const double v = ok ? a[u] - r[u].price : ninf;     // vi = cost - p[j]   (:350)
            const bool ge = ok & (v >= x.v);                    // :351
            x.w = __builtin_fmax(x.w, __builtin_fmin(v, x.v));  // :353 / :357-358
            x.v = __builtin_fmax(x.v, v);
            x.g = ge ? base + u * kWave + lane : x.g;
            c1 = ge ? c[u] : c1;
            a1 = ge ? a[u] : a1;
            o1 = ge ? r[u].owner : o1;
            os1 = ge ? r[u].ostart : os1;
            if (kFirst && E::kCand) {
                ba.c[u] = c[u];
                ba.a[u] = a[u];
                ba.v[u] = v;
            }
        }
    };
    chunk(s, std::true_type{});
    for (int base = s + 4 * kWave; base < e; base += 4 * kWave) {
#pragma unroll
        for (int u = 0; u < 4; ++u) load(min(base + u * kWave + lane, e - 1), c[u], a[u]);
        chunk(base, std::false_type{});
    }
    // winner first
    const int hi = __double2hiint(x.v);
    const int k = hi ^ ((hi >> 31) & 0x7fffffff);
    const int kmax = wave_max_i32(k);
    const unsigned long long cnd = __ballot(k == kmax);
    int sl;
    double W;
    if (__popcll(cnd) == 1) {  // wave-uniform
        sl = __ffsll((long long)cnd) - 1;
        W = wave_max_f64(lane == sl ? x.w : x.v);
    } else {
        const int g_mine = x.g;
        const Top2 t2 = top2_wave_reduce(x);
        sl = __ffsll((long long)__ballot(g_mine == t2.g)) - 1;
        W = t2.w;
    }
    out.hit = false;
    out.obj = __builtin_amdgcn_readlane(c1, sl);
    out.prev = __builtin_amdgcn_readlane(o1, sl);
    out.pstart = __builtin_amdgcn_readlane(os1, sl);
    out.len = e - s;
    const double cost = readlane_f64(a1, sl);
    const double bid = (cost - W) + eps;  // bbest = costbest - wi + eps   (:360)
    if (bid_is_bad(bid)) err |= kErrNegativeBid;
    out.key = bid_to_key(bid);
    ba.want = E::kCand && e - s <= kCandRowMax;
    ba.len = e - s;
    ba.W = W;
    ba.V = readlane_f64(x.v, sl);
}

// The same bid for the rounds where THROUGHPUT counts (many bidders, K above the regime of the candidate lines): only
// the chunks the row has are requested (wave_bid_full asks for four up front, 2 KB, because in the tail kernel a row
// must arrive in ONE latency -- at a hundred edges per row that is 2.5x the bytes), nothing is kept for a line build,
// and the wavefront needs half the registers: more rows in flight per CU to hide the price gather behind.
template <class E, class Src>
__device__ __forceinline__ void wave_bid_lean(const E &ed, const Src &src, const int s, const int e, const double eps,
                                              CandBid &out, int &err) {
    const int lane = lane_id();
    const double ninf = -__builtin_huge_val();
    Top2 x;
    x.v = ninf;
    x.w = ninf;
    x.g = -1;
    int c1 = 0;
    double a1 = 0.0;
    for (int base = s; base < e; base += 2 * kWave) {
        int c[2];
        double a[2], pr[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) ed.load(min(base + u * kWave + lane, e - 1), c[u], a[u]);  // unconditional, clamped
#pragma unroll
        for (int u = 0; u < 2; ++u) pr[u] = src.get(c[u]).price;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int g = base + u * kWave + lane;
            const bool ok = g < e;
            const double v = ok ? a[u] - pr[u] : ninf;          // vi = cost - p[j]   (:350)
            const bool ge = ok & (v >= x.v);                    // :351
            x.w = __builtin_fmax(x.w, __builtin_fmin(v, x.v));  // :353 / :357-358
            x.v = __builtin_fmax(x.v, v);
            x.g = ge ? g : x.g;
            c1 = ge ? c[u] : c1;
            a1 = ge ? a[u] : a1;
        }
    }
    const int g_mine = x.g;
    const Top2 t2 = top2_wave_reduce(x);
    const int sl = __ffsll((long long)__ballot(g_mine == t2.g)) - 1;
    out.hit = false;
    out.obj = __builtin_amdgcn_readlane(c1, sl);
    out.prev = -1;
    out.pstart = 0;
    out.len = e - s;
    const double bid = (readlane_f64(a1, sl) - t2.w) + eps;  // bbest = costbest - wi + eps   (:360)
    if (bid_is_bad(bid)) err |= kErrNegativeBid;
    out.key = bid_to_key(bid);
}

// ---- the same bid through a single-precision FILTER (full scans of the wave-per-row kernel on big price tables) --------
// Where the prices do not fit an XCD's L2 (C5: 10^6 objects x 8 B = 8 MB against 4 MB) every 8-byte price gather that
// misses costs a 128-byte fabric line, and that traffic -- 2.9x the algorithmic bytes -- bounds the scan (PMC,
// profiles/r03_pmc_counters_C5_gather_scan.txt).  A 4-byte mirror of the prices (p32 = fl32(p), rebuilt in front of the
// launch: kernels_round.hpp, k_price_mirror) is half the table and fits.  The result stays BIT-EXACT because the fp32
// pass only decides WHICH two edges are looked at exactly:
//     a_e = fl32(fl32(c_e) - p32[col_e])       |a_e - v_e| <= delta  for  v_e = fl64(c_e - p[col_e]), the reference's value,
//                                              delta = 2^-22 (max|c| + max p)  (three roundings of <= 2^-24 relative each)
//     a1 >= a2 >= a3 the three largest a of the row (multiplicity counted), e1 / e2 the edges holding a1 / a2
//     if a2 - a3 > 2 delta:  v(e1), v(e2) >= a2 - delta > a3 + delta >= v(e) for every other edge e, so {e1, e2} ARE the
//     row's two best edges, strictly: the bid is formed from their exact values (two fp64 gathers), best = the larger,
//     the later stored index among equals (:351), w = the other (:357), bid = (cost - w) + eps (:360).
//     otherwise (ties or near-ties at the top, rows of fewer than two finite values): the exact scan (wave_bid_lean).
// `two_delta` = 2 delta as fp32 (+inf disables the filter for rows that have a finite third value).
template <class E>
__device__ __forceinline__ void wave_bid_filter(const E &ed, const double *price, const float *p32, const float two_delta,
                                                const int s, const int e, const double eps, CandBid &out, int &err) {
    const int lane = lane_id();
    const float ninf = -__builtin_huge_valf();
    float t1 = ninf, t2 = ninf, t3 = ninf;  // the lane's three largest a, and stored index / column / cost of the first two
    int g1 = -1, g2 = -1, k1 = 0, k2 = 0;
    double a1 = 0.0, a2 = 0.0;
    for (int base = s; base < e; base += 2 * kWave) {
        int c[2];
        double a[2];
        float pr[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) ed.load(min(base + u * kWave + lane, e - 1), c[u], a[u]);  // unconditional, clamped
#pragma unroll
        for (int u = 0; u < 2; ++u) pr[u] = p32[c[u]];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int g = base + u * kWave + lane;
            const float v = g < e ? (float)a[u] - pr[u] : ninf;
            const bool b1 = v > t1, b2 = v > t2, b3 = v > t3;
            t3 = b2 ? t2 : (b3 ? v : t3);
            g2 = b1 ? g1 : (b2 ? g : g2);
            k2 = b1 ? k1 : (b2 ? c[u] : k2);
            a2 = b1 ? a1 : (b2 ? a[u] : a2);
            t2 = b1 ? t1 : (b2 ? v : t2);
            g1 = b1 ? g : g1;
            k1 = b1 ? c[u] : k1;
            a1 = b1 ? a[u] : a1;
            t1 = b1 ? v : t1;
        }
    }
    auto wave_max_f32 = [](float v) {
        v = __builtin_fmaxf(v, __int_as_float(dppf_i32<kDppXor1>(__float_as_int(v))));
        v = __builtin_fmaxf(v, __int_as_float(dppf_i32<kDppXor2>(__float_as_int(v))));
        v = __builtin_fmaxf(v, __int_as_float(dppf_i32<kDppHalfMirror>(__float_as_int(v))));
        v = __builtin_fmaxf(v, __int_as_float(dppf_i32<kDppMirror>(__float_as_int(v))));
        v = __builtin_fmaxf(v, __int_as_float(dpp_i32<kDppBcast15, 0xA>(__float_as_int(v))));
        v = __builtin_fmaxf(v, __int_as_float(dpp_i32<kDppBcast31, 0xC>(__float_as_int(v))));
        return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
    };
    // the wave's three largest: M1 in lane L1, M2 in lane L2 (possibly the same lane), M3
    const float M1 = wave_max_f32(t1);
    const int L1 = __ffsll((long long)__ballot(t1 == M1)) - 1;
    const float c2 = lane == L1 ? t2 : t1;
    const float M2 = wave_max_f32(c2);
    const int L2 = __ffsll((long long)__ballot(c2 == M2)) - 1;
    const float c3 = lane == L1 ? (L2 == L1 ? t3 : t2) : (lane == L2 ? t2 : t1);
    const float M3 = wave_max_f32(c3);
    const bool decided = (M2 > ninf) && ((M3 == ninf) || (M2 - M3 > two_delta));  // wave-uniform
    if (!decided) {
        wave_bid_lean(ed, PriceSource{price}, s, e, eps, out, err);
        return;
    }
    // the two edges, from the registers of the lanes that hold them (no second look at the row)
    const bool same = L2 == L1;
    const int e1 = __builtin_amdgcn_readlane(g1, L1), col1 = __builtin_amdgcn_readlane(k1, L1);
    const int e2 = same ? __builtin_amdgcn_readlane(g2, L1) : __builtin_amdgcn_readlane(g1, L2);
    const int col2 = same ? __builtin_amdgcn_readlane(k2, L1) : __builtin_amdgcn_readlane(k1, L2);
    const double cost1 = readlane_f64(a1, L1), cost2 = same ? readlane_f64(a2, L1) : readlane_f64(a1, L2);
    const double v1 = cost1 - price[col1], v2 = cost2 - price[col2];  // vi = cost - p[j]   (:350), exact
    const bool second = (v2 > v1) | ((v2 == v1) & (e2 > e1));        // :351: the later stored index among equals
    const double cost = second ? cost2 : cost1, w = second ? v1 : v2;
    out.hit = false;
    out.obj = second ? col2 : col1;
    out.prev = -1;
    out.pstart = 0;
    out.len = e - s;
    const double bid = (cost - w) + eps;  // bbest = costbest - wi + eps   (:360)
    if (bid_is_bad(bid)) err |= kErrNegativeBid;
    out.key = bid_to_key(bid);
}

}  // namespace misslap
