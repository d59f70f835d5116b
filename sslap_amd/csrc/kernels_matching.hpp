// kernels_matching.hpp -- maximum bipartite matching on the GPU: the feasibility guard of the reference's front-end
// (Hopcroft-Karp, sslap/feasibility_.pyx:95-225, called at auction_.pyx:562-566 / :608-612) for graphs where the
// host version is the slow part of `cardinality_check=True`.  Only the CARDINALITY reaches the auction path
// (:565, :611), so this matcher need not -- and does not -- reproduce the reference's pairings (the host C++
// restatement in host_matching.hpp does); it returns a maximum matching of the same size.
//
// Same phase structure as the reference (:199-211): a breadth-first layering from all free left vertices
// (:128-168), then augmentation along shortest alternating paths (:170-197) -- with the sequential DFS replaced by
// what a level-synchronous BFS gives for free:
//   * one kernel launch per BFS layer; a frontier row claims an unvisited column with atomicCAS on pred_col[v], so
//     every column (and the row matched to it) has exactly ONE parent: the layers form a forest whose trees are
//     vertex-disjoint and rooted at the free rows;
//   * the layering stops at the first layer that reaches a free column (shortest augmenting paths, as in :138-160);
//     a tree keeps at most one such end point (atomicCAS on end_of_root[root]);
//   * one thread per tree walks its path back to the root and flips it.  Paths of different trees share no vertex,
//     so all flips of a phase are independent.
// Every phase augments at least one path while one exists (a BFS from ALL free rows reaches a free column iff an
// augmenting path exists), so the loop ends with a maximum matching (Berge).  A greedy pass first matches most rows.
#pragma once
#include "device_common.hpp"

namespace misslap {

struct MatchArgs {
    const int *row_ptr;  // int32[N + 1]
    const int *col;      // adjacency of the rows in stored order: entry g at col[g * col_stride]
    int col_stride;      // 1: a plain int32[nnz]; 2: the {col, val} pairs of a solver handle's 8 B/edge layout
    int *match_row;      // int32[N]: column matched to row u, -1 = free
    int *match_col;      // int32[M]: row matched to column v, -1 = free
    int *level;          // int32[N]: BFS layer of row u in this phase, -1 = not reached
    int *root;           // int32[N]: free row at the root of u's tree
    int *pred_col;       // int32[M]: frontier row that reached column v in this phase, -1 = not reached
    int *end_of_root;    // int32[N]: free column that ends the path of the tree rooted at r, -1 = none
    int *counters;       // [0] free columns reached in this phase, [1] rows put on the next layer, [2] matched rows,
                         // [3] free rows with edges at the start of the phase
    int n_rows, n_cols;
};

// row_ptr of a row-sorted edge list that may skip rows (feasibility_.pyx:22-46 tolerates gaps, unlike the auction's
// cumulative_idxs): row_ptr[r] = first edge with row >= r.
__global__ __launch_bounds__(256) void k_m_row_ptr(const int *loc, long long nnz, int n_rows, int *row_ptr, int *col,
                                                   int *err) {
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x; g < nnz; g += stride) {
        const int r = loc[2 * g], c = loc[2 * g + 1];
        const int rp = g ? loc[2 * (g - 1)] : -1;
        col[g] = c;
        if (r < rp || r < 0) atomicOr(err, 1);
        else
            for (int q = max(rp + 1, 0); q <= min(r, n_rows - 1); ++q) row_ptr[q] = (int)g;
        if (g == nnz - 1)
            for (int q = max(r + 1, 0); q <= n_rows; ++q) row_ptr[q] = (int)nnz;
    }
}

__global__ __launch_bounds__(256) void k_m_init(MatchArgs a) {
    const int stride = gridDim.x * blockDim.x, t = blockIdx.x * blockDim.x + threadIdx.x;
    for (int u = t; u < a.n_rows; u += stride) a.match_row[u] = -1;
    for (int v = t; v < a.n_cols; v += stride) a.match_col[v] = -1;
    if (t < 4) a.counters[t] = 0;
}

// greedy start: a free row takes the first free column of its adjacency list
__global__ __launch_bounds__(256) void k_m_greedy(MatchArgs a) {
    for (int u = blockIdx.x * blockDim.x + threadIdx.x; u < a.n_rows; u += gridDim.x * blockDim.x) {
        for (int g = a.row_ptr[u]; g < a.row_ptr[u + 1]; ++g) {
            const int v = a.col[(size_t)g * a.col_stride];
            if (a.match_col[v] == -1 && atomicCAS(&a.match_col[v], -1, u) == -1) {
                a.match_row[u] = v;
                break;
            }
        }
    }
}

// start of a phase: the free rows are layer 0 and the roots of the trees (:128-136)
__global__ __launch_bounds__(256) void k_m_phase_init(MatchArgs a) {
    const int stride = gridDim.x * blockDim.x, t = blockIdx.x * blockDim.x + threadIdx.x;
    int nfree = 0;
    for (int u = t; u < a.n_rows; u += stride) {
        const bool fr = a.match_row[u] == -1 && a.row_ptr[u + 1] > a.row_ptr[u];
        a.level[u] = fr ? 0 : -1;
        a.root[u] = u;
        a.end_of_root[u] = -1;
        nfree += fr;
    }
    for (int v = t; v < a.n_cols; v += stride) a.pred_col[v] = -1;
    for (int off = 32; off >= 1; off >>= 1) nfree += __shfl_xor(nfree, off);
    if ((threadIdx.x & 63) == 0 && nfree) atomicAdd(&a.counters[3], nfree);  // free rows that have edges
}

// one BFS layer (:138-166): one wavefront per frontier row, lanes over its adjacency list
__global__ __launch_bounds__(256) void k_m_bfs_layer(MatchArgs a, int L) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    int reached = 0, ends = 0;
    for (int u = blockIdx.x * 4 + wave; u < a.n_rows; u += gridDim.x * 4) {
        if (a.level[u] != L) continue;  // wave-uniform
        const int r = a.root[u];
        for (int g = a.row_ptr[u] + lane; g < a.row_ptr[u + 1]; g += kWave) {
            const int v = a.col[(size_t)g * a.col_stride];
            if (a.pred_col[v] != -1 || atomicCAS(&a.pred_col[v], -1, u) != -1) continue;  // somebody's already
            const int w = a.match_col[v];
            if (w == -1) {  // a free column: the end of a shortest augmenting path of tree r (one per tree)
                if (atomicCAS(&a.end_of_root[r], -1, v) == -1) ends += 1;
            } else {        // follow the matched edge v -- w: w joins the next layer (:160-164)
                a.root[w] = r;
                a.level[w] = L + 1;
                reached += 1;
            }
        }
    }
    for (int off = 32; off >= 1; off >>= 1) {
        reached += __shfl_xor(reached, off);
        ends += __shfl_xor(ends, off);
    }
    if (lane == 0) {
        if (ends) atomicAdd(&a.counters[0], ends);
        if (reached) atomicAdd(&a.counters[1], reached);
    }
}

// flip the path of every tree that reached a free column (:170-197): from the end point back to the root along
// pred_col (column -> the row that reached it) and the OLD matching (row -> the column it was reached through)
__global__ __launch_bounds__(256) void k_m_augment(MatchArgs a) {
    for (int r = blockIdx.x * blockDim.x + threadIdx.x; r < a.n_rows; r += gridDim.x * blockDim.x) {
        int v = a.end_of_root[r];
        if (v < 0) continue;
        for (int guard = 0; guard <= a.n_rows; ++guard) {  // (a path has at most n_rows rows)
            const int u = a.pred_col[v];
            const int v_old = a.match_row[u];
            a.match_row[u] = v;
            a.match_col[v] = u;
            if (v_old == -1) break;  // u is the free root
            v = v_old;
        }
    }
}

__global__ __launch_bounds__(256) void k_m_count(MatchArgs a) {
    int n = 0;
    for (int u = blockIdx.x * blockDim.x + threadIdx.x; u < a.n_rows; u += gridDim.x * blockDim.x) n += a.match_row[u] >= 0;
    for (int off = 32; off >= 1; off >>= 1) n += __shfl_xor(n, off);
    if ((threadIdx.x & 63) == 0 && n) atomicAdd(&a.counters[2], n);
}

}  // namespace misslap
