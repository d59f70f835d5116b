// kernels_tail.hpp -- the long tail of tiny rounds inside ONE persistent workgroup.
//
// At the BASELINE sizes > 98 % of all rounds have <= 64 bidders and
// every round depends on the prices of the previous one, so the tail is a latency chain, not a
// bandwidth problem.  One workgroup (one CU) loops over rounds on the
// device: the unassigned list lives in LDS, bids are formed per person (auction_.pyx:339-365),
// conflicts are resolved in LDS (:375-385), winners are applied (:388-427) and the list is
// compacted (push_all_left, :137-162) without leaving the kernel.  K never grows inside an
// eps-phase (every winner evicts at most one owner), so once K <= threshold the whole rest of the
// phase runs here.  The kernels exit when K == 0 or nits == max_iter.
//
// A bid is first tried on the person's CANDIDATE LINE (device_common.hpp: 256 bytes, <= 30 price records, an
// exactness test) and only on a miss by a full scan of the row, which also rebuilds the line.  With the lines -- and
// k_bid refreshing the nearly spent ones in the grid rounds -- 99 % of the tail's bids need one 256-byte read and one
// 30-wide record gather instead of the whole row and ~200 gathers; the dependent chain of a round is
// line -> records -> 32-lane reduction -> next line.
//
// A round is a serial chain of dependent instructions per wavefront, so what makes it shorter is wavefronts: three
// instances of k_tail by role, launched back to back once per phase (K only falls: block -> team -> solo):
//   block (K > 16;  1024 threads): two list slots per wavefront and sweep (one per 32-lane half, one gather serves
//         both), misses queued and scanned in a second pass, wavefront 0 resolves (LDS hash table above 64 bidders);
//   team  (3 <= K <= 16; 1024 threads): wavefront w serves slot w alone; ONE LDS-only barrier per round, every
//         serving wavefront finishes the round for the whole list on lanes = slots;
//   duo / chain (K <= 2; 512 threads): K = 2: wavefronts 0 and 1, one bidder each, one LDS-only barrier per round --
//         the other six wavefronts END when K reaches 2, so that barrier is between two wavefronts; K = 1: wavefront 0
//         alone, no barrier, no LDS.  (Solo mode -- both bidders in the two halves of wavefront 0 -- is what the
//         layout without lines uses for K = 2.)
// The 512-thread instance still holds every mode (two slots per wavefront in team mode): a handle whose lines are
// switched off runs in it alone (plus the block instance).  Both edge layouts keep lines; in the 12 B/edge layout a
// slot's cost comes from a parallel line of fp64 costs (device_common.hpp: Slot64).
//
// Visibility: records / lines are written and re-read by this one workgroup only (same CU, same
// vector L1, barriers between phases); the CSR is read-only.  No other workgroup runs.
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
    int2 *cand;     // candidate lines (nullptr: none)
    double *cand64; // ... their fp64 costs (12 B/edge layout only, nullptr otherwise)
    int thr;
    int round_budget;  // > 0: a kernel instance returns to the host after so many rounds (the host then refreshes the
                       // lines of long rows, which the kernels cannot rebuild in place, and launches again); 0 = no limit
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
// five scattered ones, and price[] / o2p[] / p2o[] lines stay out of the CU's L2 working set.  k_sync_from_rec
// rebuilds the three plain arrays from the records when the kernel has finished.
__device__ __forceinline__ int apply_winner(const TailArgs &a, int person, int pstart, int obj, int prev,
                                            unsigned long long key) {
    PriceRec r;
    r.price = key_to_bid(key);
    r.owner = person;
    r.ostart = pstart;
    a.rec[obj] = r;
    return prev;  // evicted owner inherits the slot (:409); -1 = hole (:412)
}

// After the tail kernels: price[j], o2p[j] from the records, and p2o as the inverse of o2p.  p2o needs no clearing pass:
// every person is either the owner of exactly one record (written below) or unassigned, and the unassigned persons are
// the list U[0, K) -- K <= the tail threshold of them.
__device__ __forceinline__ void k_sync_from_rec_body(Ctl *ctl, const PriceRec *rec, double *price, int *o2p, int *p2o,
                                                       const int *U, int n_cols, int lines, unsigned long long *live,
                                                       unsigned ticket) {
    const int K = ctl->K;
    // (closes a run of tail kernels: K / nits are theirs; an error bit raised below reaches the host with the next status)
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        post_live_status(live, ticket, K, ctl->err, ctl->nits);
        ctl->n_need = 0;  // the maintenance pass's work list has been consumed
    }
    for (int n = blockIdx.x * blockDim.x + threadIdx.x; n < K; n += gridDim.x * blockDim.x) {
        const int i = U[n];
        if (i >= 0) p2o[i] = -1;
    }
    for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < n_cols; j += gridDim.x * blockDim.x) {
        const PriceRec r = rec[j];
        // price[] still holds the prices the tail kernels started from: a price may only have risen since (the
        // candidate lines rely on it; a net fall means eps is below the rounding error of a price update).  A handle
        // WITHOUT lines does not depend on the invariant -- there a falling price is what the reference computes too
        if (lines && r.price < price[j]) atomicOr(&ctl->err, kErrPriceFell);
        price[j] = r.price;
        o2p[j] = r.owner;
        if (r.owner >= 0) p2o[r.owner] = j;
    }
}
__global__ __launch_bounds__(256) void k_sync_from_rec(Ctl *ctl, const PriceRec *rec, double *price, int *o2p, int *p2o,
                                                       const int *U, int n_cols, int lines, unsigned long long *live,
                                                       unsigned ticket) { k_sync_from_rec_body(ctl, rec, price, o2p, p2o, U, n_cols, lines, live, ticket); }
struct F_k_sync_from_rec {  // (the body as a callable: what a batched launch runs per problem, csrc/host_batch.hpp)
    static __device__ __forceinline__ void run(Ctl *ctl, const PriceRec *rec, double *price, int *o2p, int *p2o, const int *U, int n_cols, int lines, unsigned long long *live, unsigned ticket) { k_sync_from_rec_body(ctl, rec, price, o2p, p2o, U, n_cols, lines, live, ticket); }
};


// per-wavefront statistics, flushed once when the kernel ends
// (what can be derived is not counted in the rounds: line hits = bids - misses, their edges = edges - miss_edges)
struct TailStats {
    unsigned long long edges, miss_edges, builds;
    unsigned bids, misses;
    unsigned bad_hi;  // largest high word of a bid made from a line (cand_eval1_r; bad_hi_is_error)
    int err;
    double hint;  // cand_build's search distance, carried from one build of this wavefront to the next
};

// zero in every lane, but not wave-uniform to the compiler: an index built with it goes through the VECTOR memory
// path.  A wave-uniform row_ptr[person + 1] would be a scalar load, which shares lgkmcnt with LDS and makes every
// LDS-only barrier wait an L2 / HBM latency for a value nobody needs yet.
__device__ __forceinline__ int lane_zero() {
    return (int)__builtin_amdgcn_mbcnt_hi(0u, __builtin_amdgcn_mbcnt_lo(0u, 0u));
}

// What a wavefront requests for its (up to) two persons ahead of their bids: the candidate lines -- person A's in
// lanes 0..31, person B's in lanes 32..63, they answer ~90 % of the bids -- and, speculatively, the first 256 edges of
// both rows plus the row ends, so that a miss costs one memory latency, not two.  The lines are requested FIRST: a
// wavefront's loads return in issue order, so a line can be used while the rows are still on their way, and a hit
// never waits for a row at all.
// the lane's slot of a line that never hits: tau = +inf in slot 0, every other slot empty
template <class Slot>
__device__ __forceinline__ Slot cand_no_line() {
    return (lane_id() & (kCandLanes - 1)) == 0 ? LineIO<Slot>::make(0, 0x7ff00000) : LineIO<Slot>::make(-1, 0);
}
// the lane's slot of person `p`'s line (l32 = lane & 31)
template <class E>
__device__ __forceinline__ typename E::Slot line_of(const TailArgs &a, int p, int l32) {
    return LineIO<typename E::Slot>::load(a.cand, a.cand64, (size_t)max(p, 0) * kCandLanes + l32);
}
template <class E>
struct TwoFetch {
    typename E::Slot slot;      // the lane's slot of its half's line
    typename E::Raw row[2][4];  // 12 B/edge layout (no lines): first four 64-edge chunks of each row, requested ahead
    int ev[2];                  // ... and the row ends (vector registers, the same value in every lane)
};
template <class E>
__device__ __forceinline__ void fetch_row(const TailArgs &a, const E &ed, int person, int start,
                                          typename E::Raw (&row)[4], int &e) {
    const int lane = lane_id();
#pragma unroll
    for (int u = 0; u < 4; ++u) row[u] = ed.load_raw_nt(start + u * kWave + lane);  // the edge arrays are padded
    e = a.row_ptr[max(person, 0) + 1 + lane_zero()];
}
// Request what the next bids of persons p0 / p1 (-1 = none: a valid, unused address is read) need first: with
// candidate lines the two lines (p0's in lanes 0..31, p1's in lanes 32..63); without (12 B/edge layout) the rows.
// A wavefront's loads return in issue order, so nothing is requested speculatively BEHIND which a later,
// more urgent load would have to wait: rows are only read when a line has not decided.
template <class E>
__device__ __forceinline__ void request_two(const TailArgs &a, const E &ed, int p0, int s0, int p1, int s1,
                                            TwoFetch<E> &tf) {
    if (E::kCand) {
        const int pme = lane_id() < kCandLanes ? p0 : p1;
        tf.slot = a.cand != nullptr ? line_of<E>(a, pme, lane_id() & (kCandLanes - 1)) : cand_no_line<typename E::Slot>();
    } else {
        fetch_row<E>(a, ed, p0, s0, tf.row[0], tf.ev[0]);
        fetch_row<E>(a, ed, p1, s1, tf.row[1], tf.ev[1]);
    }
}

__device__ __forceinline__ void tail_build(const TailArgs &a, int person, const CandBuildArgs &ba, double eps,
                                           TailStats &st) {
    cand_build(a.cand, a.cand64, person, ba, eps, st.hint);
    st.builds += 1;
}

// The bids of the wavefront's two persons (auction_.pyx:339-365): both lines by one gather, then a full scan for
// every person whose line did not decide.  `early(b)` is cand_eval2's hook (the winners' owners are known).  The
// rebuild of the (last) scanned person's line is left to the caller (bd / bd_person), who first publishes the bids
// and requests the next data.
template <class E, class Early, class S = NoStamp>
__device__ __forceinline__ void bid_two(const TailArgs &a, const E &ed, const int (&pi)[2], const int (&ps)[2],
                                        TwoFetch<E> &tf, double eps, CandBid (&b)[2], CandBuildArgs &bd,
                                        int &bd_person, TailStats &st, Early &&early, const S &stamp = S()) {
    const RecSource src{a.rec};
    b[0].hit = b[1].hit = false;
    if (E::kCand) cand_eval2(tf.slot, pi[0] >= 0, pi[1] >= 0, src, eps, b, st.err, early, stamp);
    stamp.light(4);  // lines evaluated
    bd_person = -1;
#pragma unroll
    for (int X = 0; X < 2; ++X) {
        if (pi[X] < 0) continue;  // wave-uniform
        if (!b[X].hit) {
            if (bd_person >= 0) tail_build(a, bd_person, bd, eps, st);  // two misses in one round: rare
            if (E::kCand) {
                const typename E::Raw none[4] = {};
                const int e = a.row_ptr[pi[X] + 1 + lane_zero()];
                wave_bid_full<E, RecSource, true, false>(ed, src, ps[X], e, none, eps, b[X], bd, st.err);
            } else {
                wave_bid_full<E, RecSource, true, true>(ed, src, ps[X], tf.ev[X], tf.row[X], eps, b[X], bd, st.err);
            }
            bd.want = bd.want && a.cand != nullptr;
            bd_person = bd.want ? pi[X] : -1;
            st.misses += 1;
            st.miss_edges += (unsigned long long)b[X].len;
        }
        st.edges += (unsigned long long)b[X].len;
        st.bids += 1;
    }
}

// rounds a mode may run, as 32 bits (a mode that stops short of K's range or max_iter is simply entered again: budgeted
// launches already work that way)
__device__ __forceinline__ int round_limit(long long nits, long long max_iter) {
    return (int)min(max_iter - nits, (long long)0x7fffffff);
}

// ---- chain mode: K == 1 --------------------------------------------------------------------------------------------
// One bidder per round until the phase ends (K never grows): the bidder always wins (:375-385 has nothing to
// resolve), the evicted owner inherits the only slot (:409) and bids next, and the chain ends when an unowned object
// is won.  The owner and its row start come with the winning price record, so the only dependent memory accesses of
// a round are the bidder's line and the records of its candidates; the next line is requested as soon as the winning
// candidate is known.  Everything a round does not need (second list slot, resolve, push_all_left) is left out: a
// round is bound by the length of its dependent instruction sequence.
template <class E>
__device__ __forceinline__ void tail_chain_mode(const TailArgs &a, const E &ed, int &pi, int &ps, int &K,
                                                long long &nits, const long long max_iter, const double eps,
                                                TailStats &st) {
    const int lane = lane_id(), l32 = lane & (kCandLanes - 1);
    const RecSource src{a.rec};
    const bool cls = (lane >= 1) & (lane <= kCandMax);
    const bool lines = E::kCand && a.cand != nullptr;
    typename E::Slot slot = cand_no_line<typename E::Slot>();
    typename E::Raw row[4];  // (no lines): the row, requested ahead
    int ev = 0;
    auto request = [&](int person, int start) {
        if (E::kCand) {
            if (lines) slot = line_of<E>(a, person, l32);
        } else {
            fetch_row<E>(a, ed, person, start, row, ev);
        }
    };
    request(pi, ps);
    const int rmax = round_limit(nits, max_iter);  // (rounds are counted in 32 bits inside a mode)
    int r = 0;
#ifdef MISSLAP_TAIL_STAMP
    // diagnostic build: cycles per segment of a chain round -> Ctl::dbg[6..11]: [6] wait for the line, [7] record
    // gather, [8] winner known + next line requested, [9] rest of the evaluation, [10] full scan of a missed person,
    // [11] store / re-request / line rebuild
    unsigned long long sacc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, sprev = __builtin_amdgcn_s_memtime();
    const CycleStamp stamp{sacc, &sprev, true};
#else
    const NoStamp stamp;
#endif
    for (;;) {
        CandBid b;
        b.hit = false;
        int sp = -2;  // the person whose line was requested early (-2: nothing requested)
        if (E::kCand)
            cand_eval1(slot, cls, src, eps, b, st.bad_hi, [&](const CandBid &w) {
                sp = w.prev;
                request(sp, 0);
            }, stamp);
        stamp.light(4);
        if (!b.hit) {  // wave-uniform (rare behind the maintenance pass: the scan and the line's rebuild stay inside this
                       // block -- what the rebuild needs must not be carried across the join with the common path)
            CandBuildArgs bd;
            if (E::kCand) {
                const typename E::Raw none[4] = {};
                const int e = a.row_ptr[pi + 1];
                wave_bid_full<E, RecSource, true, false>(ed, src, ps, e, none, eps, b, bd, st.err);
            } else {
                wave_bid_full<E, RecSource, true, true>(ed, src, ps, ev, row, eps, b, bd, st.err);
            }
            st.misses += 1;
            st.miss_edges += (unsigned long long)b.len;
            if (bd.want && lines) tail_build(a, pi, bd, eps, st);
        }
        stamp.light(5);
        st.edges += (unsigned long long)b.len;
        st.bids += 1;
        r += 1;
        if (lane == 0) apply_winner(a, pi, ps, b.obj, b.prev, b.key);  // ASSIGN (:396-418)
        pi = b.prev;  // the evicted owner inherits the slot (:409) and bids next; -1: everybody is assigned
        ps = b.pstart;
        const bool done = pi < 0 || r >= rmax;
        if (!done && sp != pi) request(pi, ps);  // (a scanned row decided differently from its line)
        stamp.light(6);
        if (done) break;
    }
    K = pi != -1;
    nits += r;
#ifdef MISSLAP_TAIL_STAMP
    if (lane == 0) {
        for (int k = 1; k <= 6; ++k) a.ctl->dbg[5 + k] += sacc[k];
    }
#endif
}

// ---- solo mode: K <= 2 ---------------------------------------------------------------------------------------------
// Wavefront 0 runs the rounds alone until the phase ends (K never grows), without barriers and without LDS.  The
// reference's round is reproduced operation by operation: both bids use the prices of the previous round (records
// are stored after both bids are formed), RESOLVE keeps the earlier list position on equal bids (:379), the evicted
// owner inherits the winner's slot (:409), push_all_left moves slot 1 into an emptied slot 0 (:137-162).  The next
// bidder of a slot is the owner its bidder evicts, and that owner comes with the winning price record -- no o2p /
// row_ptr look-ups between rounds: its line is requested as soon as the winning candidate is known, before the
// second-best reduction, the bid, the resolve step and the record stores of the current round.  A line that had to
// be rebuilt is rebuilt after the next round's data has been requested.
template <class E>
__device__ __forceinline__ void tail_solo_mode(const TailArgs &a, const E &ed, int *sU, int *sStart, int &K,
                                               long long &nits, const long long max_iter, const double eps,
                                               TailStats &st) {
    const int lane = lane_id();
    int pi[2], ps[2];
    pi[0] = __builtin_amdgcn_readfirstlane(sU[0]);
    ps[0] = __builtin_amdgcn_readfirstlane(sStart[0]);
    pi[1] = K > 1 ? __builtin_amdgcn_readfirstlane(sU[1]) : -1;
    ps[1] = K > 1 ? __builtin_amdgcn_readfirstlane(sStart[1]) : 0;
    if (K == 1) {
        tail_chain_mode(a, ed, pi[0], ps[0], K, nits, max_iter, eps, st);
        if (lane == 0) {
            sU[0] = pi[0];
            sU[1] = -1;
            sStart[0] = ps[0];
        }
        return;
    }
    TwoFetch<E> tf;
    tf.slot = cand_no_line<typename E::Slot>();
    request_two(a, ed, pi[0], ps[0], pi[1], ps[1], tf);
#ifdef MISSLAP_TAIL_STAMP_SOLO
    // diagnostic build: cycles of wavefront 0 per segment of a solo round -> Ctl::dbg[6..11] (+ dbg[15] = rounds):
    // [6] wait for the line, [7] record gather, [8] winner known + next line requested, [9] rest of the line
    // evaluation, [10] full scans of missed persons, [11] resolve / stores / re-request / line rebuild
    unsigned long long sacc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, sprev = __builtin_amdgcn_s_memtime();
    const CycleStamp stamp{sacc, &sprev, true};
#else
    const NoStamp stamp;
#endif
    for (;;) {
        CandBid b[2];
        CandBuildArgs bd;
        int bd_person;
        int sp[2] = {-2, -2};  // persons whose lines were requested early (-2: nothing requested)
        bid_two(a, ed, pi, ps, tf, eps, b, bd, bd_person, st, [&](const CandBid(&w)[2]) {
            sp[0] = pi[0] >= 0 ? w[0].prev : -1;
            sp[1] = pi[1] >= 0 ? w[1].prev : -1;
            request_two(a, ed, sp[0], 0, sp[1], 0, tf);  // (kCand only: the row starts are not needed)
        }, stamp);
        stamp.light(5);  // full scans of missed persons done
        nits += 1;
        // RESOLVE (:375-385): strict ">" -- the earlier list position keeps an object on equal bids
        bool win0 = true, win1 = pi[1] >= 0;
        if (win1 && b[0].obj == b[1].obj) {
            if (b[1].key > b[0].key) win0 = false;
            else win1 = false;
        }
        // ASSIGN (:396-418): a winner's slot goes to the evicted owner (or becomes a hole), a loser stays
        if (lane == 0) {
            if (win0) apply_winner(a, pi[0], ps[0], b[0].obj, b[0].prev, b[0].key);
            if (win1) apply_winner(a, pi[1], ps[1], b[1].obj, b[1].prev, b[1].key);
        }
        if (win0) {
            pi[0] = b[0].prev;
            ps[0] = b[0].pstart;
        }
        if (win1) {
            pi[1] = b[1].prev;
            ps[1] = b[1].pstart;
        }
        // push_all_left (:137-162) on two slots
        if (pi[0] == -1 && pi[1] != -1) {
            pi[0] = pi[1];
            ps[0] = ps[1];
            pi[1] = -1;
        }
        K = (pi[0] != -1) + (pi[1] != -1);
        const bool done = K <= 1 || nits >= max_iter;
        // the early request assumed "both bidders win, nobody moves"; otherwise (a scanned row, a lost bid, the
        // end of a chain) request again
        if (!done && (sp[0] != pi[0] || sp[1] != pi[1])) request_two(a, ed, pi[0], ps[0], pi[1], ps[1], tf);
        if (bd_person >= 0) tail_build(a, bd_person, bd, eps, st);
        stamp.light(6);
        if (done) break;
    }
    if (K == 1 && nits < max_iter) tail_chain_mode(a, ed, pi[0], ps[0], K, nits, max_iter, eps, st);
#ifdef MISSLAP_TAIL_STAMP_SOLO
    if (lane == 0)
        for (int k = 1; k <= 6; ++k) a.ctl->dbg[5 + k] += sacc[k];
#endif
    if (lane == 0) {
        sU[0] = pi[0];
        sU[1] = pi[1];
        sStart[0] = ps[0];
        sStart[1] = ps[1];
    }
}

// ---- team mode: 3 <= K <= 16 -----------------------------------------------------------------------------------
// Wavefront w serves list slots 2w and 2w + 1 (the two halves of its lines).  ONE barrier per round: the bids AND the
// bidders go to LDS (double-buffered by round parity), and behind the barrier EVERY wavefront finishes the round for
// the whole list redundantly on lanes = slots, so nobody waits for a resolver and no second barrier publishes a
// result.  Every serving wavefront stores the records of ALL winners itself (identical values from every wavefront):
// its own next gathers then follow its own stores in program order, which is the only ordering a wavefront's
// in-order memory path gives for free -- and the barrier guarantees that every wavefront has finished the gathers of
// the round before anybody stores.
// Almost every round is "clean": no object is bid on twice and no chain ends (no bidder wins an unowned object).
// Then every bidder wins, every slot passes to the owner its bidder evicts (:409), the list keeps its length and
// order, and a wavefront already knows its own next persons -- their lines were requested when the winning
// candidates were known.  The clean test is one LDS compare-and-swap per slot (a private 64-entry table per
// wavefront: an insert that meets its own object = a contested object) and one ballot; only an unclean round runs
// RESOLVE (:375-385) / push_all_left (:137-162) in full.
constexpr int kTeamMax = 16;
constexpr int kTeamTab = 64;  // entries of a wavefront's private duplicate-detection table (<= 16 inserts)
__device__ __forceinline__ void tail_barrier_lds() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

template <class E>
__device__ __forceinline__ void tail_team_mode(const TailArgs &a, const E &ed, int *sU, int *sStart, int &K,
                                               long long &nits, const long long max_iter, const double eps,
                                               TailStats &st) {
    const int lane = lane_id(), wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    static_assert(kTeamMax <= 2 * (kTailMax / kWave) && kTeamMax <= kWave, "two list slots per wavefront");
    __shared__ unsigned long long tKey[2][kTeamMax];
    __shared__ int tObj[2][kTeamMax], tPrev[2][kTeamMax], tPst[2][kTeamMax], tU[2][kTeamMax], tS[2][kTeamMax];
    __shared__ int wTab[kTailMax / kWave][kTeamTab];
    const int n0 = 2 * wave;  // my slots: n0, n0 + 1
    int pi[2], ps[2];
#pragma unroll
    for (int X = 0; X < 2; ++X) {
        pi[X] = n0 + X < K ? __builtin_amdgcn_readfirstlane(sU[min(n0 + X, kTailMax - 1)]) : -1;
        ps[X] = n0 + X < K ? __builtin_amdgcn_readfirstlane(sStart[min(n0 + X, kTailMax - 1)]) : 0;
    }
    wTab[wave][lane & (kTeamTab - 1)] = -1;
    TwoFetch<E> tf;
    tf.slot = cand_no_line<typename E::Slot>();
    if (n0 < K) request_two(a, ed, pi[0], ps[0], pi[1], ps[1], tf);
    __syncthreads();  // (sU / sStart have been read by everybody)
#ifdef MISSLAP_TAIL_STAMP_TEAM
    // diagnostic build: cycles of wavefront 0 per segment of a team round -> Ctl::dbg[6..9]: [6] bids of my slots,
    // [7] barrier, [8] clean test / resolve / assign, [9] re-request, line rebuild
    unsigned long long sacc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, sprev = __builtin_amdgcn_s_memtime();
    const CycleStamp stamp{sacc, &sprev, wave == 0};
#else
    const NoStamp stamp;
#endif
    int par = 0;
    for (;;) {
        int sp[2] = {-2, -2};  // persons whose lines were requested early (-2: nothing requested)
        CandBuildArgs bd;
        int bd_person = -1;
        stamp.light(0);
        if (n0 < K) {  // wave-uniform: BID for my slots (slots < K are always occupied: the list is compact)
            CandBid b[2];
            bid_two(a, ed, pi, ps, tf, eps, b, bd, bd_person, st, [&](const CandBid(&w)[2]) {
                sp[0] = pi[0] >= 0 ? w[0].prev : -1;
                sp[1] = pi[1] >= 0 ? w[1].prev : -1;
                request_two(a, ed, sp[0], 0, sp[1], 0, tf);  // the owners my bidders evict if they win
            });
            if (lane == 0) {
#pragma unroll
                for (int X = 0; X < 2; ++X)
                    if (pi[X] >= 0) {
                        tKey[par][n0 + X] = b[X].key;
                        tObj[par][n0 + X] = b[X].obj;
                        tPrev[par][n0 + X] = b[X].prev;
                        tPst[par][n0 + X] = b[X].pstart;
                        tU[par][n0 + X] = pi[X];
                        tS[par][n0 + X] = ps[X];
                    }
            }
        }
        stamp.light(1);
        tail_barrier_lds();  // the bids are in LDS and every gather of the round is done; requests stay in flight
        stamp.light(2);
        {
            const bool act = lane < K;
            const int ls = min(lane, kTeamMax - 1);
            const unsigned long long lkey = act ? tKey[par][ls] : 0ull;
            const int lobj = act ? tObj[par][ls] : (-2 - lane);
            const int lprev = act ? tPrev[par][ls] : 0, lpst = act ? tPst[par][ls] : 0;
            int u = act ? tU[par][ls] : -1;  // the list: lane l holds slot l
            int sx = act ? tS[par][ls] : 0;
            // clean round?  (a) no object bid on twice: one compare-and-swap per slot into my private table
            int hs = 0;
            bool dup = false;
            if (act) {
                hs = (int)(((unsigned)lobj * 2654435761u) >> 26) & (kTeamTab - 1);
                for (;;) {
                    const int old = atomicCAS(&wTab[wave][hs], -1, lobj);
                    if (old == -1) break;
                    if (old == lobj) {
                        dup = true;
                        break;
                    }
                    hs = (hs + 1) & (kTeamTab - 1);
                }
            }
            const bool contested = __any(dup);
            if (act && !dup) wTab[wave][hs] = -1;  // (LDS operations of a wavefront complete in order)
            // (b) no chain ends
            const bool ends = __any(act && lprev == -1);
#ifdef MISSLAP_TAIL_STAMP_TEAM
            if (threadIdx.x == 0) sacc[5] += (!contested && !ends), sacc[6] += contested, sacc[7] += 1;
#endif
            if (!contested && !ends) {
                // every bidder wins (:375-385 has nothing to resolve); ASSIGN (:396-418); the list keeps its shape
                if (act && n0 < K) apply_winner(a, u, sx, lobj, lprev, lkey);
#pragma unroll
                for (int X = 0; X < 2; ++X) {
                    const int w = __builtin_amdgcn_readlane(lprev, min(n0 + X, kWave - 1));
                    const int ws = __builtin_amdgcn_readlane(lpst, min(n0 + X, kWave - 1));
                    pi[X] = n0 + X < K ? w : -1;
                    ps[X] = n0 + X < K ? ws : 0;
                }
            } else {
                // RESOLVE / ASSIGN / push_all_left in full, on lanes = slots
                bool lose = false;
                for (int m = 0; m < K; ++m) {  // :375-385, all pairs via readlane
                    const int om = __builtin_amdgcn_readlane(lobj, m);
                    const bool same = (om == lobj) && (m != lane);
                    if (__any(same)) {  // wave-uniform
                        const unsigned long long km = readlane_u64(lkey, m);
                        lose |= same && (km > lkey || (km == lkey && m < lane));
                    }
                }
                const bool won = act && !lose;
                if (won && n0 < K) apply_winner(a, u, sx, lobj, lprev, lkey);  // :396-418 (serving wavefronts only)
                u = won ? lprev : u;  // the evicted owner inherits the slot (:409) / hole (:412) / a loser stays
                sx = won ? lpst : sx;
                const unsigned long long kmask = (1ull << K) - 1ull;
                const unsigned long long holes = __ballot(act && u == -1) & kmask;
                const int Kn = K - __popcll(holes);
                const unsigned long long lmask = (1ull << Kn) - 1ull;
                unsigned long long hl = holes & lmask;            // empty slots left of K'
                unsigned long long mv = ~holes & ~lmask & kmask;  // persons right of K'
                while (hl) {  // wave-uniform: k-th hole <- k-th mover (:137-162)
                    const int hk = __ffsll((long long)hl) - 1, mk = __ffsll((long long)mv) - 1;
                    const int mu = __builtin_amdgcn_readlane(u, mk), ms = __builtin_amdgcn_readlane(sx, mk);
                    if (lane == hk) {
                        u = mu;
                        sx = ms;
                    }
                    hl &= hl - 1;
                    mv &= mv - 1;
                }
                K = Kn;
#pragma unroll
                for (int X = 0; X < 2; ++X) {
                    pi[X] = n0 + X < K ? __builtin_amdgcn_readlane(u, min(n0 + X, kWave - 1)) : -1;
                    ps[X] = n0 + X < K ? __builtin_amdgcn_readlane(sx, min(n0 + X, kWave - 1)) : 0;
                }
            }
        }
        stamp.light(3);
        par ^= 1;
        nits += 1;
        const bool done = K <= 2 || nits >= max_iter;
        // my slots' new occupants are usually exactly the persons whose lines were requested early; otherwise (a
        // scanned row, a lost bid, a moved person) request now
        if (!done && n0 < K && (sp[0] != pi[0] || sp[1] != pi[1]))
            request_two(a, ed, pi[0], ps[0], pi[1], ps[1], tf);
        if (bd_person >= 0) tail_build(a, bd_person, bd, eps, st);
        stamp.light(4);
        if (done) break;
    }
#ifdef MISSLAP_TAIL_STAMP_TEAM
    if (threadIdx.x == 0)
        for (int k = 1; k <= 6; ++k) a.ctl->dbg[5 + k] += sacc[k];
#endif
    // hand the list back: every wavefront writes its own slots
    if (lane < 2 && n0 + lane < kTeamMax) {
        sU[n0 + lane] = lane == 0 ? pi[0] : pi[1];
        sStart[n0 + lane] = lane == 0 ? ps[0] : ps[1];
    }
    __syncthreads();
}

// ---- the team and duo rounds of handles with lines: few instructions, one barrier, the next gather issued ahead ---------
// A wavefront of the tail issues one instruction every ~8 cycles whatever the instruction is (tools/micro/
// exec_mask_bench.hip: dependent VALU 8.2 cycles, independent 6.1; profiles/r06_tail_overheads.txt), so a round is as long
// as the instructions on its wavefront plus the memory waits it cannot hide.  What a team / duo round does BESIDES the
// evaluation of its line is therefore written to be few instructions:
//   * a slot publishes TWO 16-byte LDS entries ahead of the barrier: the price record its winning bid would write
//     {bid, bidder, bidder's row start} and {object, its owner, the owner's row start};
//   * the clean test -- no object bid on twice, no chain ends (an unowned object won): 99.9 % of the rounds -- moves IN
//     FRONT of the barrier: each bidder swaps the round's number into a table slot keyed by its object's hash (one
//     ds_wrxchg per wavefront; meeting the round's own number = two bids on one hash) and writes the round's number into
//     the dirty word of the round's parity if that, or the end of a chain, is seen.  Behind the barrier the test is ONE
//     broadcast LDS read.  Round numbers only grow, so neither the table nor the dirty words are ever reset; a false
//     positive (two objects, one hash: < 1 % of the rounds) only sends the round through the exact RESOLVE / ASSIGN /
//     push_all_left (auction_.pyx:375-385 strict ">", :396-418, :137-162), which is the code every round ran before;
//   * in a clean round every bidder wins and every slot passes to the owner its bidder evicts (:409): a wavefront's next
//     bidder is in its own registers, and ALL winners' records are one 16-byte store on lanes = slots (every serving
//     wavefront stores them all itself: its later gathers follow them in program order);
//   * the record gather of the NEXT bid is issued AHEAD of the barrier, as soon as the next bidder's line has landed, and
//     made exact behind it: a candidate whose object was bid on in this round takes the PUBLISHED record (whether the
//     gather saw the old or the new one does not matter); every other record cannot have changed (:394-427 writes only
//     the objects bid on), and the next round's stores lie behind ITS barrier, which no wavefront reaches before its
//     gather has landed.  "Bid on in this round" is read off the same hash table (own object: patched from registers;
//     any other hit -- rare -- walks the published slots);
//   * rounds are counted in 32 bits inside a mode (a launch is bounded by TailArgs::round_budget / 2^31 rounds) and the
//     statistics that can be derived (line hits = bids - misses) are not counted;
//   * wavefronts whose slot lies beyond K (K never grows) END after one more barrier: the barrier is then between the
//     serving wavefronts only.
constexpr int kPipeTab = 4096;  // (K = 16: 3 % false positives of the clean test, K = 6: 0.4 %)
__device__ __forceinline__ int pipe_hash(int obj) { return (int)(((unsigned)obj * 2654435761u) >> 20); }
static_assert((1 << 12) == kPipeTab, "pipe_hash keeps the top 12 bits");
__device__ __forceinline__ void patch_rec(PriceRec &r, const bool take, const PriceRec &nw) {
    r.price = take ? nw.price : r.price;
    r.owner = take ? nw.owner : r.owner;
    r.ostart = take ? nw.ostart : r.ostart;
}
struct __attribute__((aligned(16))) PipeSlot {
    PriceRec rec;  // what the slot's winning bid writes: {bid, bidder, bidder's row start}
    int4 aux;      // {object bid on, its owner, the owner's row start, -}
};
struct __attribute__((aligned(16))) PipeLds {
    PipeSlot slot[2][kTeamMax];  // by round parity
    int dirty[2];                // number of the last round (of this parity) that needs the exact RESOLVE
    int pad[2];
    int tab[kPipeTab];           // hash of an object -> number of the last round in which it was bid on
};

// (a line's registers are "used" here so that the compiler's wait for the line load sits AT this point: what is issued
// behind it -- the winners' stores -- is then not in front of any later wait for the line)
__device__ __forceinline__ void touch_slot(const int2 &s) { asm volatile("" ::"v"(s.x), "v"(s.y)); }
__device__ __forceinline__ void touch_slot(const Slot64 &s) { asm volatile("" ::"v"(s.x), "v"(s.y), "v"(s.c)); }

// returns true when this wavefront has LEFT (its slot lies beyond K): the caller flushes its statistics and ends
template <class E>
__device__ __forceinline__ bool tail_team1_pipe(const TailArgs &a, const E &ed, int *sU, int *sStart, int &K,
                                                long long &nits, const long long max_iter, const double eps,
                                                TailStats &st) {
    const int lane = lane_id(), l32 = lane & (kCandLanes - 1);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    __shared__ PipeLds L;
    const RecSource src{a.rec};
    const bool cls = (lane >= 1) & (lane <= kCandMax);
    const int n0 = wave;  // my slot
    int pi = n0 < K ? __builtin_amdgcn_readfirstlane(sU[min(n0, kTailMax - 1)]) : -1;
    int ps = n0 < K ? __builtin_amdgcn_readfirstlane(sStart[min(n0, kTailMax - 1)]) : 0;
    for (int h = threadIdx.x; h < kPipeTab; h += blockDim.x) L.tab[h] = -1;  // (round numbers are >= 0)
    if (threadIdx.x < 2) L.dirty[threadIdx.x] = -1;
    typename E::Slot slot = cand_no_line<typename E::Slot>();
    auto request = [&](int person) { slot = line_of<E>(a, person, l32); };
    if (n0 < K) request(pi);
    __syncthreads();  // (sU / sStart have been read by everybody, the tables are set)
#ifdef MISSLAP_TAIL_STAMP_TEAM
    // diagnostic build: cycles of wavefront 0 per segment of a team round -> Ctl::dbg[6..11]: [6] wait for the line + record
    // gather, [7] evaluation of my slot up to the bid, [8] publish, [9] barrier, [10] LDS reads landed, [11] dirty word /
    // stores / re-request / loop
    unsigned long long sacc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, sprev = __builtin_amdgcn_s_memtime();
    const CycleStamp stamp{sacc, &sprev, wave == 0};
#else
    const NoStamp stamp;
#endif
    const int rmax = round_limit(nits, max_iter);
    int r = 0;  // rounds done here = the round's number
    int par = 0;
    const int ls = min(lane, kTeamMax - 1);
    PriceRec grec = PriceRec{0.0, -1, 0};
    bool have_g = false;  // the records of my slot's line have been requested (behind the previous round's stores)
    for (;;) {
        if (n0 >= K) {  // wave-uniform: my slot fell away (slots < K are always occupied: the list is compact)
            if (lane == 0) {
                sU[n0] = -1;  // (I end behind this barrier; the serving wavefronts hand their slots back)
                sStart[n0] = 0;
            }
            tail_barrier_lds();
            nits += r;
            return true;
        }
        int sp = -2;  // the person whose line was requested early (-2: nothing requested)
        CandBid b;
        b.hit = false;
        stamp.light(0);
        // BID for my slot (auction_.pyx:339-365).  The gather follows the previous round's stores in program order.
        if (!have_g) grec = cand_gather1(slot, cls, src);
#ifdef MISSLAP_TAIL_STAMP_TEAM
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        stamp.light(1);
#endif
        cand_eval1_r(slot, grec, cls, eps, b, st.bad_hi, [&](const CandBid &w) {
            sp = w.prev;
            request(sp);  // the owner my bidder evicts if it wins
        });
        if (!b.hit) {  // (rare behind the maintenance pass: the scan and the line's rebuild stay inside this block)
            CandBuildArgs bd;
            const typename E::Raw none[4] = {};
            const int e = a.row_ptr[pi + 1 + lane_zero()];
            wave_bid_full<E, RecSource, true, false>(ed, src, ps, e, none, eps, b, bd, st.err);
            st.misses += 1;
            st.miss_edges += (unsigned long long)b.len;
            if (bd.want) tail_build(a, pi, bd, eps, st);
        }
        st.edges += (unsigned long long)b.len;
        st.bids += 1;
        stamp.light(2);
        if (lane == 0) {
            const int old = atomicExch(&L.tab[pipe_hash(b.obj)], r);
            PriceRec mine;
            mine.price = key_to_bid(b.key);
            mine.owner = pi;
            mine.ostart = ps;
            L.slot[par][n0].rec = mine;
            L.slot[par][n0].aux = make_int4(b.obj, b.prev, b.pstart, 0);
            if (old == r || b.prev < 0) L.dirty[par] = r;
        }
        stamp.light(3);
        tail_barrier_lds();  // the bids are in LDS and every gather of the round has landed
        stamp.light(4);
        {
            const int dflag = L.dirty[par];
            const PriceRec lrec = L.slot[par][ls].rec;  // lanes = slots
            const int lobj = L.slot[par][ls].aux.x;
#ifdef MISSLAP_TAIL_STAMP_TEAM
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            stamp.light(5);
#endif
            if (__builtin_amdgcn_readfirstlane(dflag) != r) {
                // CLEAN: every bidder wins (:375-385 has nothing to resolve); ASSIGN (:396-418) of every slot on lanes =
                // slots; the list keeps its shape and my slot passes to the owner my bidder evicts (:409), whose line was
                // requested when the winner was known: it has been in flight behind the rest of the evaluation, the
                // publish, the barrier and these reads
                // (the next gather is issued right here, behind the stores and inside the block that has waited for the line:
                // at the loop's back edge the compiler would put a wait for EVERYTHING in front of it, the stores included)
                have_g = sp == b.prev;  // (wave-uniform; false only behind a scanned row that decided differently from its line)
                if (have_g) {
                    touch_slot(slot);
                    if (lane < K) a.rec[lobj] = lrec;
                    grec = cand_gather1(slot, cls, src);
                } else {
                    if (lane < K) a.rec[lobj] = lrec;
                }
                pi = b.prev;
                ps = b.pstart;
            } else {
                // RESOLVE / ASSIGN / push_all_left in full, on lanes = slots
                have_g = false;
                const bool act = lane < K;
                const int4 laux = L.slot[par][ls].aux;
                const unsigned long long lkey = act ? bid_to_key(lrec.price) : 0ull;
                const int lo = act ? laux.x : (-2 - lane);
                const int lprev = act ? laux.y : 0, lpst = act ? laux.z : 0;
                int u = act ? lrec.owner : -1;  // the list: lane l holds slot l
                int sx = act ? lrec.ostart : 0;
                bool lose = false;
                for (int m = 0; m < K; ++m) {  // :375-385, all pairs via readlane
                    const int om = __builtin_amdgcn_readlane(lo, m);
                    const bool same = (om == lo) && (m != lane);
                    if (__any(same)) {  // wave-uniform
                        const unsigned long long km = readlane_u64(lkey, m);
                        lose |= same && (km > lkey || (km == lkey && m < lane));
                    }
                }
                const bool won = act && !lose;
                if (won) apply_winner(a, u, sx, lo, lprev, lkey);  // :396-418
                u = won ? lprev : u;  // the evicted owner inherits the slot (:409) / hole (:412) / a loser stays
                sx = won ? lpst : sx;
                const unsigned long long kmask = (1ull << K) - 1ull;
                const unsigned long long holes = __ballot(act && u == -1) & kmask;
                const int Kn = K - __popcll(holes);
                const unsigned long long lmask = (1ull << Kn) - 1ull;
                unsigned long long hl = holes & lmask;            // empty slots left of K'
                unsigned long long mv = ~holes & ~lmask & kmask;  // persons right of K'
                while (hl) {  // wave-uniform: k-th hole <- k-th mover (:137-162)
                    const int hk = __ffsll((long long)hl) - 1, mk = __ffsll((long long)mv) - 1;
                    const int mu = __builtin_amdgcn_readlane(u, mk), ms = __builtin_amdgcn_readlane(sx, mk);
                    if (lane == hk) {
                        u = mu;
                        sx = ms;
                    }
                    hl &= hl - 1;
                    mv &= mv - 1;
                }
                K = Kn;
                pi = n0 < K ? __builtin_amdgcn_readlane(u, min(n0, kWave - 1)) : -1;
                ps = n0 < K ? __builtin_amdgcn_readlane(sx, min(n0, kWave - 1)) : 0;
            }
        }
        par ^= 1;
        r += 1;
        const bool done = K <= 2 || r >= rmax;
        // my slot's new occupant is usually exactly the person whose line was requested early; otherwise (a scanned
        // row, a lost bid, a moved person) request now
        if (!done && n0 < K && sp != pi) request(pi);
        stamp.light(6);
        if (done) break;
    }
    nits += r;
#ifdef MISSLAP_TAIL_STAMP_TEAM
    if (threadIdx.x == 0)
        for (int k = 1; k <= 6; ++k) a.ctl->dbg[5 + k] += sacc[k];
#endif
    // hand the list back: every wavefront that is still here writes its own slot
    if (lane == 0 && n0 < kTeamMax) {
        sU[n0] = pi;
        sStart[n0] = ps;
    }
    __syncthreads();
    return false;
}

// duo mode in the same form: two wavefronts, one bidder each; clean = two different objects, both owned
template <class E>
__device__ __forceinline__ void tail_duo_pipe(const TailArgs &a, const E &ed, int *sU, int *sStart, int &K,
                                              long long &nits, const long long max_iter, const double eps,
                                              TailStats &st) {
    const int lane = lane_id(), l32 = lane & (kCandLanes - 1);
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // 0 or 1: my slot
    const int o = w ^ 1;
    __shared__ PipeSlot D[2][2];
    const RecSource src{a.rec};
    const bool cls = (lane >= 1) & (lane <= kCandMax);
    int me = __builtin_amdgcn_readfirstlane(sU[w]), mys = __builtin_amdgcn_readfirstlane(sStart[w]);
    typename E::Slot slot = cand_no_line<typename E::Slot>();
    auto request = [&](int person) { slot = line_of<E>(a, person, l32); };
    request(me);
    tail_barrier_lds();  // (sU / sStart have been read by both)
#ifdef MISSLAP_TAIL_STAMP_DUO
    // diagnostic build: cycles of wavefront 0 per segment of a duo round -> Ctl::dbg[6..10]: [6] evaluation up to the bid
    // (incl. what is left of the gather's latency), [7] publish + wait for the next line + next gather issued, [8]
    // barrier (= the other wavefront), [9] exchange / patch / stores, [10] re-request, line rebuild
    unsigned long long sacc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, sprev = __builtin_amdgcn_s_memtime();
    const CycleStamp stamp{sacc, &sprev, w == 0};
#else
    const NoStamp stamp;
#endif
    const int rmax = round_limit(nits, max_iter);
    int r = 0;
    PriceRec grec = PriceRec{0.0, -1, 0};
    bool have_g = false;
    int par = 0;
    // (the two stores of a clean round are issued in the NEXT round, behind its evaluation: see tail_team1_pipe)
    bool pend = false;
    PriceRec prec = PriceRec{0.0, -1, 0};
    int pobj = 0;
    auto flush_store = [&]() {
        if (pend) {  // wave-uniform
            if (lane < 2) a.rec[pobj] = prec;  // ASSIGN (:396-418): lane 0 my record, lane 1 the other's
            pend = false;
        }
    };
    for (;;) {
        int sp = -2;  // the person whose line was requested early (-2: nothing requested)
        CandBid b;
        b.hit = false;
        stamp.light(0);
        if (!have_g) {
            flush_store();
            grec = cand_gather1(slot, cls, src);
        }
        cand_eval1_r(slot, grec, cls, eps, b, st.bad_hi, [&](const CandBid &x) {
            sp = x.prev;
            request(sp);  // the owner my bidder evicts if it wins
        });
        if (!b.hit) {  // (rare behind the maintenance pass: the scan and the line's rebuild stay inside this block)
            flush_store();
            CandBuildArgs bd;
            const typename E::Raw none[4] = {};
            const int e = a.row_ptr[me + 1 + lane_zero()];
            wave_bid_full<E, RecSource, true, false>(ed, src, mys, e, none, eps, b, bd, st.err);
            st.misses += 1;
            st.miss_edges += (unsigned long long)b.len;
            if (bd.want) tail_build(a, me, bd, eps, st);
            b.hit = false;
        }
        flush_store();  // (the common case: the previous round's stores, behind this round's evaluation)
        st.edges += (unsigned long long)b.len;
        st.bids += 1;
        stamp.light(1);
        PriceRec mine;
        mine.price = key_to_bid(b.key);
        mine.owner = me;
        mine.ostart = mys;
        if (lane == 0) {
            D[par][w].rec = mine;
            D[par][w].aux = make_int4(b.obj, b.prev, b.pstart, 0);
        }
        // the gather of the NEXT round, ahead of the barrier (into grec itself: its old contents are dead, and a second
        // variable is a copy at the loop's back edge with a wait for every outstanding store in front of it)
        have_g = b.hit && b.prev >= 0;
        if (have_g) grec = cand_gather1(slot, cls, src);
        stamp.light(2);
        tail_barrier_lds();  // both bids are in LDS and both gathers of the round have been issued against the old records
        stamp.light(3);
        const PriceRec orec = D[par][o].rec;  // (one address for the whole wavefront)
        const int4 oaux = D[par][o].aux;
        r += 1;
        const int unclean = __builtin_amdgcn_readfirstlane((int)(oaux.x == b.obj) | (int)(oaux.y < 0)) | (int)(b.prev < 0);
        if (!unclean) {
            // both bidders win different owned objects (:375-385 has nothing to resolve) and both slots pass to the
            // evicted owners.  The records gathered ahead are made exact FIRST, the stores (ASSIGN, :396-418, by both
            // wavefronts alike: lane 0 my record, lane 1 the other's) are issued behind that
            if (have_g) {
                const int c = slot.x;
                patch_rec(grec, c == b.obj, mine);
                patch_rec(grec, c == oaux.x, orec);
            }
            pend = true;
            prec = mine;
            patch_rec(prec, lane == 1, orec);
            pobj = lane == 1 ? oaux.x : b.obj;
            me = b.prev;
            mys = b.pstart;
        } else {
            // the two bids by slot, RESOLVE (:375-385: strict ">", the earlier list position keeps an object on equal
            // bids), ASSIGN (:396-418) by both wavefronts alike, push_all_left (:137-162) on two slots
            const unsigned long long okey = bid_to_key(readlane_f64(orec.price, 0));
            const int oobj = __builtin_amdgcn_readfirstlane(oaux.x), oprev = __builtin_amdgcn_readfirstlane(oaux.y);
            const int opst = __builtin_amdgcn_readfirstlane(oaux.z);
            const int ome = __builtin_amdgcn_readfirstlane(orec.owner), omys = __builtin_amdgcn_readfirstlane(orec.ostart);
            int pi[2], ps[2];
            pi[0] = w ? ome : me, pi[1] = w ? me : ome;
            ps[0] = w ? omys : mys, ps[1] = w ? mys : omys;
            const unsigned long long key0 = w ? okey : b.key, key1 = w ? b.key : okey;
            const int obj0 = w ? oobj : b.obj, obj1 = w ? b.obj : oobj;
            const int prev0 = w ? oprev : b.prev, prev1 = w ? b.prev : oprev;
            const int pst0 = w ? opst : b.pstart, pst1 = w ? b.pstart : opst;
            bool win0 = true, win1 = true;
            if (obj0 == obj1) {
                if (key1 > key0) win0 = false;
                else win1 = false;
            }
            if (lane == 0) {
                if (win0) apply_winner(a, pi[0], ps[0], obj0, prev0, key0);
                if (win1) apply_winner(a, pi[1], ps[1], obj1, prev1, key1);
            }
            if (win0) {
                pi[0] = prev0;
                ps[0] = pst0;
            }
            if (win1) {
                pi[1] = prev1;
                ps[1] = pst1;
            }
            if (pi[0] == -1 && pi[1] != -1) {
                pi[0] = pi[1];
                ps[0] = ps[1];
                pi[1] = -1;
            }
            K = (pi[0] != -1) + (pi[1] != -1);
            me = w ? pi[1] : pi[0];
            mys = w ? ps[1] : ps[0];
            have_g = false;
        }
        stamp.light(4);
        par ^= 1;
        const bool done = K <= 1 || r >= rmax;
        // the early request assumed "my bidder wins, nobody moves"; otherwise request again
        if (!done && sp != me) request(me);
        stamp.light(5);
        if (done) break;
    }
    flush_store();
    nits += r;
#ifdef MISSLAP_TAIL_STAMP_DUO
    if (w == 0 && lane == 0)
        for (int k = 1; k <= 5; ++k) a.ctl->dbg[5 + k] += sacc[k];
#endif
    if (lane == 0) {  // every wavefront hands its own slot back
        sU[w] = me;
        sStart[w] = mys;
    }
    tail_barrier_lds();
}

// kThreads = kTailMax (512): every mode.  kThreads = 1024 ("block only"): the rounds with more than kTeamMax bidders
// with SIXTEEN wavefronts -- half the sweeps per wavefront in pass A, which is where a block round spends its time --
// and nothing else: the solo / team code needs more than the 128 registers a 1024-thread workgroup leaves a
// wavefront.  The host launches it ahead of the 512-thread kernel, which then finds K <= kTeamMax.
// kTeamOnly (1024 threads as well): the rounds with 3..kTeamMax bidders, one slot per wavefront (tail_team1_pipe).
// kThreads = 128 ("duo / chain only", handles with lines): the rounds with K <= 2 and nothing else -- two wavefronts
// for duo mode, wavefront 0 alone for the chain; the instance that carries every mode needs 185 VGPRs and spills 14
// SGPRs, each spill a v_writelane / v_readlane pair inside a chain that is bound by its instruction count.
template <class E, int kThreads, bool kTeamOnly>
__device__ __forceinline__ void k_tail_body(TailArgs a, E ed) {
    static_assert(!kTeamOnly || (kThreads == 2 * kTailMax && kThreads / kWave == kTeamMax), "one slot per wavefront");
    constexpr bool kBlockOnly = kThreads > kTailMax && !kTeamOnly;
    constexpr bool kDuoOnly = kThreads == 2 * kWave;
    static_assert(!kDuoOnly || !kTeamOnly, "roles are exclusive");
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
    __shared__ int sCnt[3][kThreads / kWave];
    __shared__ int sK, sMissCnt;
    __shared__ long long sNits;

    Ctl *ctl = a.ctl;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    constexpr int nwaves = kThreads / kWave;
    int K = ctl->K;
    long long nits = ctl->nits;
    const long long max_iter =
        a.round_budget > 0 ? min(ctl->max_iter, nits + (long long)(kBlockOnly ? max(a.round_budget / 4, 1) : a.round_budget))
                           : ctl->max_iter;  // (block rounds have the most bidders, i.e. spend the most lines)
    if (K == 0 || K > a.thr || nits >= max_iter) return;  // uniform
    if (kBlockOnly && K <= kTeamMax) return;
    if (kTeamOnly && (K <= 2 || K > kTeamMax)) return;
    if (kDuoOnly && (K > 2 || !E::kCand || a.cand == nullptr)) return;
    // (budgeted launches: an instance that ran out of rounds leaves K to the NEXT launch's instance of the right role)
    if (!kBlockOnly && !kTeamOnly && a.round_budget > 0 && K > 2 && E::kCand && a.cand != nullptr) return;
    const int K0 = K;
    const long long nits0 = nits;
    for (int n = t; n < kTailMax; n += kThreads) {
        sU[n] = (n < K) ? a.U[n] : -1;
        sStart[n] = (n < K) ? a.row_ptr[sU[n]] : 0;
    }
    if (!kDuoOnly)
        for (int h = t; h < kHashSize; h += kThreads) {
            hObj[h] = -1;
            hKey[h] = 0ull;
            hPos[h] = kPosNone;
        }
    const double eps = (double)a.eps;
    TailStats st;
    st.edges = st.miss_edges = st.builds = 0ull;
    st.bids = st.misses = st.bad_hi = 0u;
    st.err = 0;
    st.hint = 0.0;
    if (t == 0) sMissCnt = 0;
    __syncthreads();

    // per-mode accounting (always on: two s_memrealtime reads per mode entry, 100 MHz ticks), Ctl::dbg:
    //   [0..2] rounds in chain + solo / team / block mode, [3..5] ticks
    unsigned long long md[6] = {0, 0, 0, 0, 0, 0};
#ifdef MISSLAP_TAIL_STAMP_BLOCK
    unsigned long long bacc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
    auto mode_begin = [&](int m) {
        md[3 + m] -= __builtin_amdgcn_s_memrealtime();
        md[m] -= (unsigned long long)nits;
    };
    auto mode_end = [&](int m) {
        md[3 + m] += __builtin_amdgcn_s_memrealtime();
        md[m] += (unsigned long long)nits;
    };
    auto flush_stats = [&]() {  // a wavefront's statistics, once, when it leaves the kernel
        if (lane == 0) {
            if (st.bids) {
                const unsigned long long hits = (unsigned long long)(st.bids - st.misses), hit_edges = st.edges - st.miss_edges;
                atomicAdd(&ctl->edges, st.edges);
                atomicAdd(&ctl->tail_edges, st.edges);
                atomicAdd(&ctl->bids, (unsigned long long)st.bids);
                if (hits) {
                    atomicAdd(&ctl->cand_hits, hits);
                    atomicAdd(&ctl->cand_edges, hit_edges);
                }
                atomicAdd(&ctl->dbg[12], (unsigned long long)st.bids);  // the tail's own totals: bids, line hits, line builds
                atomicAdd(&ctl->dbg[13], hits);
                atomicAdd(&ctl->dbg[14], st.builds);
                atomicAdd(&ctl->dbg[15], hit_edges);
            }
            if (bad_hi_is_error(st.bad_hi)) st.err |= kErrNegativeBid;
            if (st.err) atomicOr(&ctl->err, st.err);
        }
    };
    for (;;) {
        if (kBlockOnly && K <= kTeamMax) break;  // the next kernel takes over
        if (kDuoOnly) {
            // ---- K <= 2, lines: duo mode while two bidders are left (wavefronts 0 and 1, one bidder each), then
            // wavefront 0 runs the single-bidder chain alone
            mode_begin(0);
            if (K == 2) tail_duo_pipe(a, ed, sU, sStart, K, nits, max_iter, eps, st);
            if (wave == 1) {
                flush_stats();
                return;
            }
            if (K == 1 && nits < max_iter) {
                int pi = __builtin_amdgcn_readfirstlane(sU[0]), ps = __builtin_amdgcn_readfirstlane(sStart[0]);
                tail_chain_mode(a, ed, pi, ps, K, nits, max_iter, eps, st);
                if (lane == 0) {
                    sU[0] = pi;
                    sU[1] = -1;
                    sStart[0] = ps;
                }
            }
            mode_end(0);
            break;  // K == 0 or nits == max_iter
        }
        if (kTeamOnly) {
            if (K > 2 && nits < max_iter) {
                mode_begin(1);
                if (tail_team1_pipe(a, ed, sU, sStart, K, nits, max_iter, eps, st)) {  // (this wavefront's slot fell away)
                    flush_stats();
                    return;
                }
                mode_end(1);
            }
            break;
        }
        if (!kBlockOnly && !kTeamOnly && K == 2 && K0 <= 2 && E::kCand && a.cand != nullptr) {
            // ---- duo mode: wavefronts 0 and 1 run the two-bidder rounds, one bidder each; the other wavefronts have
            // nothing left to do in this phase (K never grows) and END here, so the barriers of duo mode are between
            // two wavefronts.  Then wavefront 0 runs the single-bidder chain alone.
            if (wave >= 2) {
                flush_stats();
                return;
            }
            mode_begin(0);
            tail_duo_pipe(a, ed, sU, sStart, K, nits, max_iter, eps, st);
            if (wave == 1) {
                flush_stats();
                return;
            }
            if (K > 0 && nits < max_iter) tail_solo_mode(a, ed, sU, sStart, K, nits, max_iter, eps, st);
            mode_end(0);
            break;  // K == 0 or nits == max_iter
        }
        if (!kBlockOnly && K <= 2) {
            // ---- solo mode: wavefront 0 runs the rest of the phase alone, see tail_solo_mode
            if (wave == 0) {
                mode_begin(0);
                tail_solo_mode(a, ed, sU, sStart, K, nits, max_iter, eps, st);
                mode_end(0);
                if (lane == 0) {
                    sK = K;
                    sNits = nits;
                }
            }
            __syncthreads();
            K = sK;
            nits = sNits;
            break;  // K == 0 or nits == max_iter
        }
        if (!kBlockOnly && K <= kTeamMax) {
            // ---- team mode: every wavefront, until K <= 2 (or max_iter), see tail_team_mode
            mode_begin(1);
            tail_team_mode(a, ed, sU, sStart, K, nits, max_iter, eps, st);
            mode_end(1);
            if (K == 0 || nits >= max_iter) break;
            continue;
        }
        // ---- block mode, BID in two passes.  Pass A: the lines, two list slots per wavefront (one per 32-lane
        // half, one gather serves both); a hit goes straight to LDS, a miss is queued.  Pass B: the queued persons,
        // one wavefront each, by a full scan of their rows (+ line rebuild).
        mode_begin(2);
#ifdef MISSLAP_TAIL_STAMP_BLOCK
        // diagnostic build: cycles of wavefront 0 per segment of a block round -> Ctl::dbg[6..11]: [6] lines landed,
        // [7] records landed, [8] evaluated + barrier, [9] scan pass + barrier, [10] resolve / assign / compaction,
        // [11] closing barrier
        unsigned long long sprev_b = __builtin_amdgcn_s_memtime();
        const CycleStamp bstamp{bacc, &sprev_b, wave == 0};
#else
        const NoStamp bstamp;
#endif
        {
            const RecSource src{a.rec};
            // kBlockDepth sweeps of 2 * nwaves slots are in flight together: all their lines are requested first,
            // then all record gathers are issued (each as its line lands), then they are evaluated one after the
            // other -- the two memory latencies are paid once per group of sweeps, not once per sweep
            // (sixteen wavefronts -- four per SIMD -- hide the two latencies by themselves: measured per block round
            // 3.51 us with one sweep in flight, 3.69 with two, 3.96 with three, 4.06 with four)
            constexpr int kBlockDepth = kBlockOnly ? 1 : 3;
            for (int base = 0; base < K; base += kBlockDepth * 2 * nwaves) {
                typename E::Slot sl[kBlockDepth];
                PriceRec rr[kBlockDepth];
#pragma unroll
                for (int c = 0; c < kBlockDepth; ++c) {
                    const int nme = base + c * 2 * nwaves + 2 * wave + (lane >> 5);
                    const int pme = nme < K ? sU[min(nme, kTailMax - 1)] : -1;
                    sl[c] = cand_no_line<typename E::Slot>();
                    if (E::kCand && a.cand != nullptr) sl[c] = line_of<E>(a, pme, lane & (kCandLanes - 1));
                }
                bstamp(1);
#pragma unroll
                for (int c = 0; c < kBlockDepth; ++c) {
                    const int n0 = base + c * 2 * nwaves + 2 * wave;
                    if (E::kCand) rr[c] = cand_gather2(sl[c], n0 < K, n0 + 1 < K, src);
                }
                bstamp(2);
#pragma unroll
                for (int c = 0; c < kBlockDepth; ++c) {
                    const int n0 = base + c * 2 * nwaves + 2 * wave;
                    if (n0 >= K) continue;  // wave-uniform
                    CandBid b[2];
                    b[0].hit = b[1].hit = false;
                    if (E::kCand) cand_eval2_r(sl[c], rr[c], n0 < K, n0 + 1 < K, eps, b, st.err, NoEarly());
#pragma unroll
                    for (int X = 0; X < 2; ++X) {
                        const int n = n0 + X;
                        if (n >= K) continue;  // wave-uniform
                        if (b[X].hit) {
                            if (lane == 0) {
                                sKey[n] = b[X].key;
                                sObj[n] = b[X].obj;
                                // owner at the start of the round == what the assignment phase reads (:401): the
                                // record of obj is only rewritten by this round's winner of obj, after every bid
                                // has been made.
                                sPrev[n] = b[X].prev;
                                sPst[n] = b[X].pstart;
                            }
                            st.edges += (unsigned long long)b[X].len;
                            st.bids += 1;
                        } else if (lane == 0) {
                            sList[atomicAdd(&sMissCnt, 1)] = n;
                        }
                    }
                }
            }
            __syncthreads();
            bstamp(3);
            const int nmiss = sMissCnt;
            for (int m = wave; m < nmiss; m += nwaves) {
                const int n = __builtin_amdgcn_readfirstlane(sList[m]);
                const int i = __builtin_amdgcn_readfirstlane(sU[n]);
                const int s0 = __builtin_amdgcn_readfirstlane(sStart[n]);
                const typename E::Raw none[4] = {};
                const int ev = a.row_ptr[i + 1 + lane_zero()];
                CandBid bm;
                CandBuildArgs ba;
                wave_bid_full<E, RecSource, true, false>(ed, src, s0, ev, none, eps, bm, ba, st.err);
                ba.want = ba.want && a.cand != nullptr;
                if (lane == 0) {
                    sKey[n] = bm.key;
                    sObj[n] = bm.obj;
                    sPrev[n] = bm.prev;
                    sPst[n] = bm.pstart;
                }
                st.edges += (unsigned long long)bm.len;
                st.miss_edges += (unsigned long long)bm.len;
                st.bids += 1;
                st.misses += 1;
                if (ba.want) tail_build(a, i, ba, eps, st);
            }
        }
        __syncthreads();
        bstamp(4);

        if (K <= kWave) {
            // ---- fast path: the whole rest of the round in wavefront 0, no LDS atomics ------------
            if (wave == 0) {
                const bool act = lane < K;
                const unsigned long long key = act ? sKey[lane] : 0ull;
                const int obj = act ? sObj[lane] : (-2 - lane);
                // RESOLVE (:375-385).  Two bidders on one object are the exception: every bidder inserts its object
                // into the (empty) LDS hash table with one compare-and-swap; only if some insert meets its own object
                // -- a contested object -- is the all-pairs loop run.  The table is emptied again right away.
                bool lose = false;
                int hs = 0;
                bool dup = false;
                if (act) {
                    hs = (int)(((unsigned)obj * 2654435761u) >> 21) & (kHashSize - 1);
                    for (;;) {
                        const int old = atomicCAS(&hObj[hs], -1, obj);
                        if (old == -1) break;
                        if (old == obj) {
                            dup = true;
                            break;
                        }
                        hs = (hs + 1) & (kHashSize - 1);
                    }
                }
                const bool contested = __any(dup);
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                if (act && !dup) hObj[hs] = -1;
                if (contested) {
                    for (int m = 0; m < K; ++m) {  // all pairs via readlane
                        const int om = __builtin_amdgcn_readlane(obj, m);
                        const bool same = (om == obj) && (m != lane);
                        if (__any(same)) {  // wave-uniform
                            const unsigned long long km = readlane_u64(key, m);
                            lose |= same && (km > key || (km == key && m < lane));
                        }
                    }
                }
                int u = act ? sU[lane] : -1;
                int sx = act ? sStart[lane] : 0;  // row start travelling with the slot's person
                if (act && !lose) {
                    u = apply_winner(a, u, sx, obj, sPrev[lane], key);
                    sx = sPst[lane];
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
                        sStart[lane] = sx;
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
                        sStart[dst] = sx;
                    }
                    if (act) {
                        if (lane >= Kn) sU[lane] = -1;
                        else if (!is_hl) {
                            sU[lane] = u;
                            sStart[lane] = sx;
                        }
                    }
                }
                if (lane == 0) {
                    sK = Kn;
                    sMissCnt = 0;
                }
            }
        } else {
            // ---- general path (64 < K <= 512): LDS hash table keyed by object ---------------------
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
            int sx = act ? sStart[t] : 0;
            if (win) {
                u = apply_winner(a, u, sx, obj, sPrev[t], key);
                sx = sPst[t];
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
                sStart[sList[pm]] = sx;
            }
            if (act) {
                if (t >= Kn) sU[t] = -1;
                else if (!is_hl) {
                    sU[t] = u;
                    sStart[t] = sx;
                }
            }
            if (t == 0) {
                sK = Kn;
                sMissCnt = 0;
            }
        }
        bstamp(5);
        __syncthreads();
        bstamp(6);
        K = sK;
        nits += 1;
        mode_end(2);
        if (K == 0 || nits >= max_iter) break;
    }
    if (t == 0)
        for (int k = 0; k < 6; ++k) ctl->dbg[k] += md[k];
#ifdef MISSLAP_TAIL_STAMP_BLOCK
    if (t == 0)
        for (int k = 1; k <= 6; ++k) ctl->dbg[5 + k] += bacc[k];
#endif

    for (int n = t; n < K0; n += kThreads) a.U[n] = sU[n];
    flush_stats();
    if (t == 0) {
        ctl->K = K;
        ctl->nits = nits;
        ctl->tail_rounds += nits - nits0;
    }
}
template <class E, int kThreads, bool kTeamOnly = false>
__global__ __launch_bounds__(kThreads) void k_tail(TailArgs a, E ed) { k_tail_body<E, kThreads, kTeamOnly>(a, ed); }
template <class E, int kThreads, bool kTeamOnly>
struct F_k_tail {  // (the body as a callable: what a batched launch runs per problem, csrc/host_batch.hpp)
    static __device__ __forceinline__ void run(TailArgs a, E ed) { k_tail_body<E, kThreads, kTeamOnly>(a, ed); }
};


}  // namespace misslap
