// kernels_ingest.hpp -- COO / dense input -> device CSR (reference AuctionSolver.__init__,
// auction_.pyx:202-265, and the dense scan of _from_matrix, :546-557).  All O(nnz) / O(N*M) work
// runs on the GPU; the host only sequences the kernels and reads back a few scalars.
#pragma once
#include "device_common.hpp"

namespace misslap {

struct IngestStats {
    unsigned long long max_abs_bits;  // bits of max |val| (non-negative doubles order like integers)
    int max_col;
    int err;          // kErr* bits
    int not_f32;      // some value is not exactly representable in fp32
    int max_row_len;  // longest row (k_max_row_len): decides whether the long-row line builder has work
    int long_rows;    // rows of more than kCandRowMax edges (the scans of the bid kernels cannot rebuild their lines)
    long long dense_total;  // dense ingest: number of valid entries, counted in 64 bits
};

// cumulative_idxs (auction_.pyx:33-48) for the valid input class (rows ascending, no gaps): row_ptr[r] =
// index of the first entry of row r.  Also the column maximum (M = max + 1, :210).
__global__ __launch_bounds__(256) void k_ingest_rows(const int *loc, long long nnz, int n_rows, int *row_ptr,
                                                     IngestStats *st) {
    int err = 0, mc = -1;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x; g < nnz; g += stride) {
        const int r = loc[2 * g], c = loc[2 * g + 1];
        const int rp = g ? loc[2 * (g - 1)] : -1;
        if (r < 0 || c < 0) {  // malformed entry: reported, and nothing is derived from it (no row_ptr[-1] store)
            err |= kErrColNegative;
            continue;
        }
        if (r < rp) err |= kErrRowsUnsorted;
        else if (r > rp) {
            if (r != rp + 1 || r >= n_rows) err |= kErrRowGap;
            else row_ptr[r] = (int)g;
        }
        mc = c > mc ? c : mc;
    }
    for (int off = 32; off >= 1; off >>= 1) {
        err |= __shfl_xor(err, off);
        const int m2 = __shfl_xor(mc, off);
        mc = m2 > mc ? m2 : mc;
    }
    // (an atomic only where it can still change the result: 8192 wavefronts each adding one to the same word are 45 us
    // during which the launch cannot end -- same-address atomics retire one after the other)
    if ((threadIdx.x & 63) == 0) {
        if (err) atomicOr(&st->err, err);
        if (mc > __hip_atomic_load(&st->max_col, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(&st->max_col, mc);
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) row_ptr[n_rows] = (int)nnz;  // :47
}

// max_val (auction_.pyx:123-134) + the fp32-exactness / finiteness checks that pick the edge layout.
__global__ __launch_bounds__(256) void k_ingest_vals(const double *val, long long nnz, IngestStats *st) {
    unsigned long long mx = 0ull;
    int notf = 0, err = 0;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x; g < nnz; g += stride) {
        const double v = val[g];
        const unsigned long long b = (unsigned long long)__double_as_longlong(v) & 0x7fffffffffffffffull;
        if (b >= 0x7ff0000000000000ull) err |= kErrNonFinite;
        mx = b > mx ? b : mx;
        notf |= ((double)(float)v != v);
    }
    for (int off = 32; off >= 1; off >>= 1) {
        const unsigned lo = __shfl_xor((unsigned)(mx & 0xffffffffull), off);
        const unsigned hi = __shfl_xor((unsigned)(mx >> 32), off);
        const unsigned long long m2 = ((unsigned long long)hi << 32) | lo;
        mx = m2 > mx ? m2 : mx;
        notf |= __shfl_xor(notf, off);
        err |= __shfl_xor(err, off);
    }
    if ((threadIdx.x & 63) == 0) {
        if (mx > __hip_atomic_load(&st->max_abs_bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(&st->max_abs_bits, mx);
        if (notf && !__hip_atomic_load(&st->not_f32, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicOr(&st->not_f32, 1);
        if (err) atomicOr(&st->err, err);
    }
}

// flat_j copy (:229) + sign flip for 'min' (:236-237) into the streaming layout.
__global__ __launch_bounds__(256) void k_build_edges_f32(const int *loc, const double *val, long long nnz, int flip,
                                                         int2 *edges) {
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x; g < nnz; g += stride) {
        const double v = flip ? val[g] * -1 : val[g];
        edges[g] = make_int2(loc[2 * g + 1], __float_as_int((float)v));
    }
}
__global__ __launch_bounds__(256) void k_build_edges_f64(const int *loc, const double *val, long long nnz, int flip,
                                                         int *col, double *v64) {
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x; g < nnz; g += stride) {
        col[g] = loc[2 * g + 1];
        v64[g] = flip ? val[g] * -1 : val[g];
    }
}

__global__ __launch_bounds__(256) void k_init_state(Ctl *ctl, double *price, PriceRec *rec, int *p2o, int *o2p, int *U,
                                                    unsigned long long *best_key, int *best_pos, int2 *cand,
                                                    int n_rows, int n_cols, long long max_iter) {
    const int stride = gridDim.x * blockDim.x;
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    for (int i = t; i < n_rows; i += stride) {
        p2o[i] = -1;  // :231
        U[i] = i;     // :260
    }
    if (cand) {  // candidate lines: tau = +inf (no line yet), every slot empty
        const double inf = __builtin_huge_val();
        for (long long k = t; k < (long long)n_rows * kCandLanes; k += stride)
            cand[k] = (k & (kCandLanes - 1)) == 0 ? make_int2(__double2loint(inf), __double2hiint(inf))
                                                  : make_int2(-1, 0);
    }
    for (int j = t; j < n_cols; j += stride) {
        price[j] = 0.0;        // :220
        PriceRec r;
        r.price = 0.0;
        r.owner = -1;
        r.ostart = 0;
        rec[j] = r;
        o2p[j] = -1;           // :232
        best_key[j] = 0ull;    // :255
        best_pos[j] = kPosNone;  // :256
    }
    if (t == 0) {
        ctl->K = n_rows;  // :259
        ctl->nholes = 0;
        ctl->nleft = 0;
        ctl->err = 0;
        ctl->nits = 0;
        ctl->max_iter = max_iter;
        ctl->edges = 0;
        ctl->bids = 0;
        ctl->ece_fail = 0;
        ctl->dup_rows = 0;
        ctl->grid_rounds = 0;
        ctl->tail_rounds = 0;
        ctl->cand_hits = 0;
        ctl->cand_edges = 0;
        ctl->tail_edges = 0;
        ctl->shard_edges = 0;
        ctl->obj_abs = 0.0;
        ctl->obj_minexp = 1 << 20;
        ctl->n_need = 0;
        ctl->arrive = 0;
        ctl->obj = 0.0;
        for (int k = 0; k < 16; ++k) ctl->dbg[k] = 0;
    }
}

__global__ __launch_bounds__(256) void k_max_row_len(const int *row_ptr, int n_rows, IngestStats *st) {
    int m = 0, nl = 0;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n_rows; i += gridDim.x * blockDim.x) {
        const int len = row_ptr[i + 1] - row_ptr[i];
        m = max(m, len);
        nl += len > kCandRowMax;
    }
    for (int off = 32; off >= 1; off >>= 1) {
        m = max(m, __shfl_xor(m, off));
        nl += __shfl_xor(nl, off);
    }
    if ((threadIdx.x & 63) == 0 && m > __hip_atomic_load(&st->max_row_len, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
        atomicMax(&st->max_row_len, m);
    if ((threadIdx.x & 63) == 0 && nl) atomicAdd(&st->long_rows, nl);
}

// ---- dense ingest (_from_matrix, auction_.pyx:546-557): keep v >= 0 in row-major order -----------------
// pass 1: valid entries per row
__global__ __launch_bounds__(256) void k_dense_count(const double *mat, int n_rows, int n_cols, int *row_cnt) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int r = blockIdx.x * 4 + wave; r < n_rows; r += gridDim.x * 4) {
        int cnt = 0;
        const double *row = mat + (size_t)r * n_cols;
        for (int c = lane; c < n_cols; c += kWave) cnt += dense_entry_valid(row[c]);
        for (int off = 32; off >= 1; off >>= 1) cnt += __shfl_xor(cnt, off);
        if (lane == 0) row_cnt[r] = cnt;
    }
}
// pass 2: exclusive scan of the row counts -> row_ptr (single workgroup; N is at most a few 1e5
// for a dense input that fits in memory at all)
__global__ __launch_bounds__(1024) void k_dense_scan(const int *row_cnt, int n_rows, int *row_ptr,
                                                     IngestStats *st) {
    __shared__ int s_w[16];
    __shared__ long long s_carry;  // 64 bits: a dense input can hold more valid entries than an int32 can count
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    if (t == 0) s_carry = 0;
    __syncthreads();
    int empty = 0;
    for (int base = 0; base < n_rows; base += 1024) {
        const int i = base + t;
        const int v = (i < n_rows) ? row_cnt[i] : 0;
        if (i < n_rows && v == 0) empty = 1;
        int x = v;
        for (int off = 1; off < 64; off <<= 1) {
            const int y = __shfl_up(x, off);
            if (lane >= off) x += y;
        }
        if (lane == 63) s_w[wave] = x;
        __syncthreads();
        int wpre = 0;
        for (int w2 = 0; w2 < wave; ++w2) wpre += s_w[w2];
        const long long carry = s_carry;
        if (i < n_rows) row_ptr[i] = (int)(carry + wpre + x - v);  // (garbage beyond 2^31: the host rejects the input)
        __syncthreads();
        if (t == 1023) s_carry = carry + wpre + x;
        __syncthreads();
    }
    if (t == 0) {
        row_ptr[n_rows] = (int)s_carry;
        st->dense_total = s_carry;
        if (s_carry >= 0x7fffffffLL) atomicOr(&st->err, kErrTooMany);
    }
    if (empty) atomicOr(&st->err, kErrRowGap);
}
// pass 3: ordered compaction of each row into (loc, val) COO, the reference's scan order
__global__ __launch_bounds__(256) void k_dense_fill(const double *mat, int n_rows, int n_cols, const int *row_ptr,
                                                    int *loc, double *val) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int r = blockIdx.x * 4 + wave; r < n_rows; r += gridDim.x * 4) {
        long long out = row_ptr[r];
        const double *row = mat + (size_t)r * n_cols;
        for (int base = 0; base < n_cols; base += kWave) {
            const int c = base + lane;
            const double v = (c < n_cols) ? row[c] : -1.0;
            const bool ok = dense_entry_valid(v);  // :549 (NaN fails the test like in the reference); the same
                                                   // predicate as k_dense_count, or the fill overruns its segment
            const unsigned long long b = __ballot(ok);
            if (ok) {
                const long long o = out + __popcll(b & lanemask_lt());
                loc[2 * o] = r;
                loc[2 * o + 1] = c;
                val[o] = v;
            }
            out += __popcll(b);
        }
    }
}

}  // namespace misslap
