// kernels_check.hpp -- full-CSR reductions that run once per eps-phase / once per solve: the
// eps-complementary-slackness test (auction_.pyx:443-485), the objective (:489-523) and the validity flags of the
// reference's benchmark harness (benchmarking.py:56-64).
#pragma once
#include "device_common.hpp"

namespace misslap {

// ---- what a pass over the rows of the CSR has to find, per person ------------------------------------------------
// eCE_satisfied (auction_.pyx:443-485), get_obj (:489-523) and the validity flags of the reference's benchmark harness
// (benchmarking.py:56-64) all look for the same stored entry -- (i, p2o[i]), the LAST one if it is stored more than
// once -- and the eCE test needs the row maximum of (val_k - p[col_k]) on top.  So one pass serves all three:
//   choice_cost = value of the LAST stored entry of row i whose column is p2o[i]      (:467-471)
//   LHS = (choice_cost - p[j]) + tol                                                  (:475)
//   violated if for some entry k of the row  LHS < (val_k - p[col_k]) - eps           (:482)
// x -> fl(x - eps) is monotone, so "some k" is equivalent to testing the row maximum only.
//
// Where the answer comes from.  The test is ONE bit over all rows, and at the end of every eps-phase but the last it
// fails for 40-100 % of them (counted on C1-C4 at every end of a phase: the first failing row is row 0 or 1 every time).  So
// the host first launches the pass on a small SAMPLE of the rows (k_ece on the first kEceSampleRows persons, a few
// microseconds); the full pass that follows starts by reading the flag and returns at once when the sample has
// already failed.  Only a check that passes -- the last phase's and the one behind meta['eCE'] -- pays a full scan,
// and that scan runs on the bandwidth engine where the handle has the tile-major layout (k_bid_tiled, MODE 1).
constexpr int kEceSampleRows = 512;

// Results of the final pass per WORKGROUP (plain stores, no contention: thousands of wavefronts ending together on the
// same seven words cost the pass more than its scan -- 5 ns per same-address atomic); k_obj_sum adds the slots up.
struct __attribute__((aligned(64))) FinSlot {
    unsigned long long distinct, n_neg, n_big, n_inv, dups;
    long long min_exp;
    double abs_sum;
    unsigned long long pad;
};

// Sums of the final pass, kept per lane and flushed once per wavefront (flush_final).
struct FinalAcc {
    int distinct = 0, n_neg = 0, n_big = 0, n_inv = 0, dups = 0, min_exp = 1 << 20;
    double abs_sum = 0.0;
    bool bad = false;
};
struct FinalOut {
    int fin;          // 0: eCE test only; 1: also the objective's contributions and the validity counters
    int maximize;
    const int *o2p;
    double *contrib;  // [n_rows]
    int *nmatch;      // [n_rows]
    int n_rows, n_cols;
    FinSlot *slots;   // [gridDim.x] of the launch (fin = 1)
};
// The column the passes look for: p2o[i], with numpy's index wrap-around for the validity flags (sol[i] = -1 selects
// the LAST column: benchmarking.py:59 indexes mat[arange(size), sol]); -1 when that is no column at all.
__device__ __forceinline__ int wanted_column(int j, int n_cols) {
    const int c = j < 0 ? n_cols + j : j;
    return (c >= 0 && c < n_cols) ? c : -1;
}
// One person, by the ONE lane that knows the result of the row pass: j = p2o[i]; found / cnt / cost = the last stored
// entry (i, wanted_column(j)), how often it is stored, its (sign-flipped) stored value; vmax = the row maximum of
// val - price; pj = price of the wanted column.
__device__ __forceinline__ void final_person(FinalAcc &acc, const FinalOut &fo, int i, int j, bool found, int cnt,
                                             double cost, double vmax, double pj, double eps) {
    const double tol = 1e-7;  // auction_.pyx:16
    // an assigned column that is not in the row cannot happen (the reference would reuse the previous row's cost)
    acc.bad |= !found || ((cost - pj) + tol) < (vmax - eps);
    if (!fo.fin) return;
    // get_obj (:508-521): unassigned persons are skipped; the stored values are sign-flipped for 'min'
    const int n = j != -1 ? cnt : 0;
    const double cv = (n == 1) ? (fo.maximize ? cost : -cost) : 0.0;  // rows with n > 1 are re-added by k_obj_sum
    fo.contrib[i] = cv;
    fo.nmatch[i] = n;
    acc.dups += n > 1;
    if (cv != 0.0) {  // binary exponent of the lowest set bit of cv
        const unsigned long long b = (unsigned long long)__double_as_longlong(cv) & 0x7fffffffffffffffull;
        const int ex = (int)(b >> 52);
        const unsigned long long mant = (b & 0xfffffffffffffull) | (ex ? (1ull << 52) : 0ull);
        const int q = (ex ? ex - 1075 : -1074) + (__ffsll((long long)mant) - 1);
        acc.min_exp = q < acc.min_exp ? q : acc.min_exp;
        acc.abs_sum += cv < 0.0 ? -cv : cv;
    }
    // validity flags (benchmarking.py:56-64), counters in Ctl::val_cnt:
    //   [0] persons i with an object j >= 0 that names i as its owner: the number of DISTINCT objects in sol whenever
    //       the two maps are consistent -- np.unique(sol).size minus the -1 value
    //   [1] persons with sol[i] < 0        [2] persons with sol[i] >= n_rows
    //   [3] persons whose selected entry is missing or negative in the caller's sign (of several stored entries
    //       (i, c) the last one counts: a dense matrix built from loc / val keeps the last assignment)
    acc.n_inv += !(found && dense_entry_valid(fo.maximize ? cost : -cost));
    acc.n_neg += j < 0;
    acc.n_big += j >= fo.n_rows;
    acc.distinct += (j >= 0 && j < fo.n_cols && fo.o2p[j] == i);
}
// The one bit of the eCE test: set by whoever finds a violated row and does not see it set already (a plain store of
// 1 by any number of writers; every wavefront of a failing pass adding an atomic to the same word cost the sample
// pass 20 us).
__device__ __forceinline__ void flag_ece_failure(Ctl *ctl, bool bad) {
    if (__ballot(bad) && lane_id() == 0 && !__hip_atomic_load(&ctl->ece_fail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
        __hip_atomic_store(&ctl->ece_fail, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// wave-wide sums of the accumulators (valid in every lane)
__device__ __forceinline__ void wave_sum_final(FinalAcc &acc) {
    for (int off = 32; off >= 1; off >>= 1) {
        acc.distinct += __shfl_xor(acc.distinct, off);
        acc.n_neg += __shfl_xor(acc.n_neg, off);
        acc.n_big += __shfl_xor(acc.n_big, off);
        acc.n_inv += __shfl_xor(acc.n_inv, off);
        acc.dups += __shfl_xor(acc.dups, off);
        acc.min_exp = min(acc.min_exp, __shfl_xor(acc.min_exp, off));
        acc.abs_sum += shfl_xor_f64(acc.abs_sum, off);  // (any order: only a bound, see k_obj_sum)
    }
}
// ... of a workgroup: every wavefront (all of them must call) leaves its sums in `scratch` (LDS, 8 doubles per
// wavefront, free to be overwritten), thread 0 writes the workgroup's slot
__device__ __forceinline__ void flush_final_wg(const FinalOut &fo, FinalAcc acc, double *scratch) {
    wave_sum_final(acc);
    const int wave = threadIdx.x >> 6, nw = (int)blockDim.x >> 6;
    if (lane_id() == 0) {
        unsigned long long *w = reinterpret_cast<unsigned long long *>(scratch) + 8 * wave;
        w[0] = (unsigned long long)acc.distinct;
        w[1] = (unsigned long long)acc.n_neg;
        w[2] = (unsigned long long)acc.n_big;
        w[3] = (unsigned long long)acc.n_inv;
        w[4] = (unsigned long long)acc.dups;
        w[5] = (unsigned long long)(long long)acc.min_exp;
        scratch[8 * wave + 6] = acc.abs_sum;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        FinSlot r{0, 0, 0, 0, 0, 1 << 20, 0.0, 0};
        for (int k = 0; k < nw; ++k) {
            const unsigned long long *w = reinterpret_cast<const unsigned long long *>(scratch) + 8 * k;
            r.distinct += w[0];
            r.n_neg += w[1];
            r.n_big += w[2];
            r.n_inv += w[3];
            r.dups += w[4];
            r.min_exp = min(r.min_exp, (long long)w[5]);
            r.abs_sum += scratch[8 * k + 6];
        }
        fo.slots[blockIdx.x] = r;
    }
}

// The pass on the row-major CSR, one wavefront per person (handles without the tile-major layout, and the sample).
// Branch-free: every load is unconditional (clamped index), a masked-off element has value -inf and matches nothing.
// fo.fin = 0 (eCE only): every wavefront stops as soon as any row has failed.
template <class E>
__device__ __forceinline__ void k_ece_body(Ctl *ctl, E ed, const int *row_ptr, const double *price,
                                             const int *p2o, int n_rows, float eps_f, FinalOut fo) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const double eps = (double)eps_f;
    const double ninf = -__builtin_huge_val();
    FinalAcc acc;
    for (int i = blockIdx.x * 4 + wave; i < n_rows; i += gridDim.x * 4) {
        if (!fo.fin && __atomic_load_n(&ctl->ece_fail, __ATOMIC_RELAXED)) break;  // wave-uniform
        const int s = row_ptr[i], e = row_ptr[i + 1];
        const int j = p2o[i];
        const int want = wanted_column(j, fo.n_cols);
        const double pj = price[max(want, 0)];
        double vmax = ninf, asel = 0.0;
        int gsel = -1, cnt = 0;
        for (int base = s; base < e; base += 4 * kWave) {  // four 64-edge chunks in flight, like wave_bid
            int c[4];
            double a[4], pr[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) ed.load(min(base + u * kWave + lane, e - 1), c[u], a[u]);
#pragma unroll
            for (int u = 0; u < 4; ++u) pr[u] = price[c[u]];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int g = base + u * kWave + lane;
                const bool ok = g < e;
                const double v = ok ? a[u] - pr[u] : ninf;
                vmax = __builtin_fmax(vmax, v);
                const bool m = ok & (c[u] == want);  // ascending g per lane: keeps the last match
                gsel = m ? g : gsel;
                asel = m ? a[u] : asel;
                cnt += m;
            }
        }
        vmax = wave_max_f64(vmax);
        const int gmax = wave_max_i32(gsel);
        for (int off = 32; off >= 1; off >>= 1) cnt += __shfl_xor(cnt, off);
        const int sl = __ffsll((long long)__ballot(gsel == gmax)) - 1;  // (gmax = -1: any lane, the value is unused)
        const double cost = readlane_f64(asel, sl);
        if (lane == 0) final_person(acc, fo, i, j, gmax >= 0, cnt, cost, vmax, pj, eps);
    }
    flag_ece_failure(ctl, acc.bad);
    __shared__ double s_fin[8 * 4];
    if (fo.fin) flush_final_wg(fo, acc, s_fin);  // (uniform over the launch)
}
template <class E>
__global__ __launch_bounds__(256) void k_ece(Ctl *ctl, E ed, const int *row_ptr, const double *price,
                                             const int *p2o, int n_rows, float eps_f, FinalOut fo) { k_ece_body<E>(ctl, ed, row_ptr, price, p2o, n_rows, eps_f, fo); }
template <class E>
struct F_k_ece {  // (the body as a callable: what a batched launch runs per problem, csrc/host_batch.hpp)
    static __device__ __forceinline__ void run(Ctl *ctl, E ed, const int *row_ptr, const double *price, const int *p2o, int n_rows, float eps_f, FinalOut fo) { k_ece_body<E>(ctl, ed, row_ptr, price, p2o, n_rows, eps_f, fo); }
};


__global__ void k_final_reset(Ctl *ctl) { ctl->ece_fail = 0; }

// get_obj, step 2: the reference adds in person order into ONE double (:491, :519-521); floating-point
// addition is not associative, so the sum is reproduced sequentially by a single lane.
//
// Exception that costs nothing in exactness: if every contribution is an integer multiple of 2^q and the sum of
// their magnitudes stays below 2^(q+52), EVERY partial sum in ANY order is exactly representable, no addition
// rounds, and the result does not depend on the order -- then 1024 threads add in parallel (fp32-exact values of
// similar magnitude, the usual case, qualify: q >= -20 or so against sums below 2^30).
template <class E>
__global__ __launch_bounds__(1024) void k_obj_sum(Ctl *ctl, E ed, const int *row_ptr, const int *p2o, int n_rows,
                                                  int maximize, const double *contrib, const int *nmatch,
                                                  const FinSlot *slots, int n_slots) {
    if (blockIdx.x != 0) return;
    // the workgroups' results of the final pass (kernels_check.hpp, FinSlot) -> the control block
    __shared__ double s_fin[8 * 16];
    __shared__ FinSlot s_tot;
    {
        unsigned long long d = 0, ng = 0, nb = 0, ni = 0, du = 0;  // (64-bit sums: the per-lane accumulators are ints)
        long long me = 1 << 20;
        double ab = 0.0;
        for (int k = threadIdx.x; k < n_slots; k += 1024) {
            const FinSlot r = slots[k];
            d += r.distinct;
            ng += r.n_neg;
            nb += r.n_big;
            ni += r.n_inv;
            du += r.dups;
            me = min(me, r.abs_sum != 0.0 ? r.min_exp : (long long)(1 << 20));  // (a zeroed slot: a workgroup without rows)
            ab += r.abs_sum;
        }
        for (int off = 32; off >= 1; off >>= 1) {
            auto sx = [&](unsigned long long v) {
                return ((unsigned long long)__shfl_xor((unsigned)(v >> 32), off) << 32) | (unsigned long long)__shfl_xor((unsigned)(v & 0xffffffffull), off);
            };
            d += sx(d);
            ng += sx(ng);
            nb += sx(nb);
            ni += sx(ni);
            du += sx(du);
            me = min(me, (long long)sx((unsigned long long)me));
            ab += shfl_xor_f64(ab, off);
        }
        const int wave = threadIdx.x >> 6;
        if ((threadIdx.x & 63) == 0) {
            unsigned long long *w = reinterpret_cast<unsigned long long *>(s_fin) + 8 * wave;
            w[0] = d, w[1] = ng, w[2] = nb, w[3] = ni, w[4] = du, w[5] = (unsigned long long)me;
            s_fin[8 * wave + 6] = ab;
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            FinSlot r{0, 0, 0, 0, 0, 1 << 20, 0.0, 0};
            for (int k = 0; k < 16; ++k) {
                const unsigned long long *w = reinterpret_cast<const unsigned long long *>(s_fin) + 8 * k;
                r.distinct += w[0], r.n_neg += w[1], r.n_big += w[2], r.n_inv += w[3], r.dups += w[4];
                r.min_exp = min(r.min_exp, (long long)w[5]);
                r.abs_sum += s_fin[8 * k + 6];
            }
            s_tot = r;
            ctl->val_cnt[0] = r.distinct;
            ctl->val_cnt[1] = r.n_neg;
            ctl->val_cnt[2] = r.n_big;
            ctl->val_cnt[3] = r.n_inv;
            ctl->dup_rows = (int)min(r.dups, 0x7fffffffull);
            ctl->obj_minexp = (int)r.min_exp;
            ctl->obj_abs = r.abs_sum;
        }
        __syncthreads();
    }
    const bool no_dups = s_tot.dups == 0;
    if (no_dups) {  // uniform
        const int q = (int)s_tot.min_exp;
        const bool order_free = q >= (1 << 20) || (q > -1000 && 2.0 * s_tot.abs_sum < __builtin_ldexp(1.0, q + 52));
        if (order_free) {
            __shared__ double s_part[16];
            double part = 0.0;
            for (int i = threadIdx.x; i < n_rows; i += 1024) part += contrib[i];
            for (int off = 32; off >= 1; off >>= 1) {
                const int lo = __shfl_xor(__double2loint(part), off), hi = __shfl_xor(__double2hiint(part), off);
                part += __hiloint2double(hi, lo);
            }
            if ((threadIdx.x & 63) == 0) s_part[threadIdx.x >> 6] = part;
            __syncthreads();
            if (threadIdx.x == 0) {
                double obj = 0.0;
                for (int w = 0; w < 16; ++w) obj += s_part[w];
                ctl->obj = obj;
            }
            return;
        }
    }
    if (threadIdx.x >= kWave) return;  // one wavefront
    const int lane = threadIdx.x;
    double obj = 0.0;
    if (no_dups) {
        // 64 contributions per coalesced load, then added one by one in person order (v_readlane): the
        // additions stay sequential, only the loads are parallel
        double nxt = (lane < n_rows) ? contrib[lane] : 0.0;
        for (int base = 0; base < n_rows; base += kWave) {
            const double cur = nxt;
            const int k = base + kWave + lane;
            nxt = (k < n_rows) ? contrib[k] : 0.0;
            const int cnt = min(kWave, n_rows - base);
            if (cnt == kWave) {
#pragma unroll
                for (int l = 0; l < kWave; ++l) obj += readlane_f64(cur, l);  // adding +0.0 for skipped rows is exact
            } else {
                for (int l = 0; l < cnt; ++l) obj += readlane_f64(cur, l);
            }
        }
    } else if (lane == 0) {
        for (int i = 0; i < n_rows; ++i) {
            if (nmatch[i] <= 1) {
                obj += contrib[i];
            } else {
                const int j = p2o[i];
                for (int g = row_ptr[i]; g < row_ptr[i + 1]; ++g) {
                    int c;
                    double v;
                    ed.load(g, c, v);
                    if (c == j) obj += maximize ? v : -v;
                }
            }
        }
    }
    if (lane != 0) return;
    ctl->obj = obj;
}

}  // namespace misslap
