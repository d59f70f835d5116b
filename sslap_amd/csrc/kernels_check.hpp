// kernels_check.hpp -- full-CSR reductions that run once per eps-phase / once per solve:
// the eps-complementary-slackness test (auction_.pyx:443-485) and the objective (:489-523).
#pragma once
#include "device_common.hpp"

namespace misslap {

// eCE_satisfied(eps), one wavefront per person (only called with K == 0: everybody assigned).
//   choice_cost = value of the LAST stored entry of row i whose column is p2o[i]      (:467-471)
//   LHS = (choice_cost - p[j]) + tol                                                  (:475)
//   violated if for some entry k of the row  LHS < (val_k - p[col_k]) - eps           (:482)
// x -> fl(x - eps) is monotone, so "some k" is equivalent to testing the row maximum of
// (val_k - p[col_k]) only; choice and maximum are found in ONE pass over the row.
template <class E>
__global__ __launch_bounds__(256) void k_ece(Ctl *ctl, E ed, const int *row_ptr, const double *price,
                                             const int *p2o, int n_rows, float eps_f) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const double tol = 1e-7;  // auction_.pyx:16
    const double eps = (double)eps_f;
    const double ninf = -__builtin_huge_val();
    for (int i = blockIdx.x * 4 + wave; i < n_rows; i += gridDim.x * 4) {
        // The answer is one bit.  Once any row has failed, the rest of the pass is pointless -- and at the end of
        // every eps-phase but the last the test fails for most rows: 200 000 atomicOr on one word cost 1.1 ms.
        if (__atomic_load_n(&ctl->ece_fail, __ATOMIC_RELAXED)) return;  // wave-uniform
        const int s = row_ptr[i], e = row_ptr[i + 1];
        const int j = p2o[i];
        double vmax = ninf;
        int gsel = -1;
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
                if (g < e) {
                    const double v = a[u] - pr[u];
                    vmax = v > vmax ? v : vmax;
                    if (c[u] == j) gsel = g;  // ascending g per lane: keeps the last match
                }
            }
        }
        vmax = wave_max_f64(vmax);
        gsel = wave_max_i32(gsel);
        if (lane == 0) {
            bool bad = true;  // an assigned column that is not in the row cannot happen
            if (gsel >= 0) {
                int c;
                double choice_cost;
                ed.load(gsel, c, choice_cost);
                const double lhs = (choice_cost - price[j]) + tol;
                bad = lhs < (vmax - eps);
            }
            if (bad) atomicOr(&ctl->ece_fail, 1);
        }
    }
}

// Validity flags of the returned assignment, as the reference's benchmark harness forms them on the host
// (benchmarking.py:56-64: complete_assignment = (np.unique(sol).size == size, (sol >= 0).all(), (sol < size).all()),
// valid_assignment = (mat[arange(size), sol] >= 0).all(), size = number of rows), reduced here so that `sol` is the
// only O(N) copy-out.  One wavefront per person; counters in Ctl::val_cnt (zeroed by the host before the launch):
//   [0] persons i with an object j = sol[i] >= 0 that names i as its owner (o2p[j] == i): the number of DISTINCT
//       objects in sol whenever the two maps are consistent -- np.unique(sol).size minus the -1 value
//   [1] persons with sol[i] < 0        [2] persons with sol[i] >= n_rows
//   [3] persons whose selected entry is missing or negative in the caller's sign.  numpy wraps a negative index
//       around (sol[i] = -1 selects the LAST column), and so does this kernel; of several stored entries (i, c) the
//       last one counts (a dense matrix built from loc / val keeps the last assignment).
template <class E>
__global__ __launch_bounds__(256) void k_validity(Ctl *ctl, E ed, const int *row_ptr, const int *p2o, const int *o2p,
                                                  int n_rows, int n_cols, int maximize) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    int distinct = 0, n_neg = 0, n_big = 0, n_inv = 0;
    for (int i = blockIdx.x * 4 + wave; i < n_rows; i += gridDim.x * 4) {
        const int j = p2o[i];
        const int c = j < 0 ? n_cols + j : j;  // numpy index wrap-around
        int gsel = -1;
        const int s = row_ptr[i], e = row_ptr[i + 1];
        for (int g = s + lane; g < e; g += kWave) {
            int cc;
            double v;
            ed.load(g, cc, v);
            if (cc == c) gsel = g;
        }
        gsel = wave_max_i32(gsel);
        if (lane == 0) {
            bool ok = false;
            if (gsel >= 0) {
                int cc;
                double v;
                ed.load(gsel, cc, v);
                const double orig = maximize ? v : -v;  // the stored values are sign-flipped for 'min'
                ok = dense_entry_valid(orig);
            }
            n_inv += !ok;
            n_neg += j < 0;
            n_big += j >= n_rows;
            distinct += (j >= 0 && j < n_cols && o2p[j] == i);
        }
    }
    if (lane == 0) {
        if (distinct) atomicAdd(&ctl->val_cnt[0], (unsigned long long)distinct);
        if (n_neg) atomicAdd(&ctl->val_cnt[1], (unsigned long long)n_neg);
        if (n_big) atomicAdd(&ctl->val_cnt[2], (unsigned long long)n_big);
        if (n_inv) atomicAdd(&ctl->val_cnt[3], (unsigned long long)n_inv);
    }
}

__global__ void k_obj_reset(Ctl *ctl) {
    ctl->obj_abs = 0.0;
    ctl->obj_minexp = 1 << 20;
}

// get_obj, step 1 (parallel): the contribution of every person, in row order inside the row:
// contrib[i] = +val / -val of the stored entry (i, p2o[i]) ('max' / 'min'; val is the sign-flipped
// copy, so obj -= val restores the caller's sign, :518-521).  Rows whose assigned column is stored
// more than once are flagged and re-added sequentially in step 2.
template <class E>
__global__ __launch_bounds__(256) void k_obj_rows(Ctl *ctl, E ed, const int *row_ptr, const int *p2o,
                                                  int n_rows, int maximize, double *contrib, int *nmatch) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    double abs_sum = 0.0;      // lane 0: bound material for the order-independence test of k_obj_sum
    int min_exp = 1 << 20;
    for (int i = blockIdx.x * 4 + wave; i < n_rows; i += gridDim.x * 4) {
        const int j = p2o[i];
        int cnt = 0, gsel = -1;
        if (j != -1) {  // :508
            const int s = row_ptr[i], e = row_ptr[i + 1];
            for (int g = s + lane; g < e; g += kWave) {
                int c;
                double v;
                ed.load(g, c, v);
                if (c == j) {
                    cnt += 1;
                    gsel = g;
                }
            }
        }
        for (int off = 32; off >= 1; off >>= 1) {
            cnt += __shfl_xor(cnt, off);
            const int g2 = __shfl_xor(gsel, off);
            gsel = g2 > gsel ? g2 : gsel;
        }
        if (lane == 0) {
            double cv = 0.0;
            if (cnt == 1) {
                int c;
                double v;
                ed.load(gsel, c, v);
                cv = maximize ? v : -v;
            }
            contrib[i] = cv;
            nmatch[i] = cnt;
            if (cnt > 1) atomicAdd(&ctl->dup_rows, 1);
            if (cv != 0.0) {  // binary exponent of the lowest set bit of cv
                const unsigned long long b = (unsigned long long)__double_as_longlong(cv) & 0x7fffffffffffffffull;
                const int ex = (int)(b >> 52);
                const unsigned long long mant = (b & 0xfffffffffffffull) | (ex ? (1ull << 52) : 0ull);
                const int q = (ex ? ex - 1075 : -1074) + (__ffsll((long long)mant) - 1);
                min_exp = q < min_exp ? q : min_exp;
                abs_sum += cv < 0.0 ? -cv : cv;
            }
        }
    }
    if (lane == 0 && abs_sum != 0.0) {
        atomicMin(&ctl->obj_minexp, min_exp);
        atomicAdd(&ctl->obj_abs, abs_sum);
    }
}

// get_obj, step 2: the reference adds in person order into ONE double (:491, :519-521); floating-point
// addition is not associative, so the sum is reproduced sequentially by a single lane.
//
// Exception that costs nothing in exactness: if every contribution is an integer multiple of 2^q and the sum of
// their magnitudes stays below 2^(q+52), EVERY partial sum in ANY order is exactly representable, no addition
// rounds, and the result does not depend on the order -- then 1024 threads add in parallel (fp32-exact values of
// similar magnitude, the usual case, qualify: q >= -20 or so against sums below 2^30).
template <class E>
__global__ __launch_bounds__(1024) void k_obj_sum(Ctl *ctl, E ed, const int *row_ptr, const int *p2o, int n_rows,
                                                  int maximize, const double *contrib, const int *nmatch) {
    if (blockIdx.x != 0) return;
    if (ctl->dup_rows == 0) {  // uniform
        const int q = ctl->obj_minexp;
        const bool order_free = q >= (1 << 20) || (q > -1000 && 2.0 * ctl->obj_abs < __builtin_ldexp(1.0, q + 52));
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
    if (ctl->dup_rows == 0) {
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
