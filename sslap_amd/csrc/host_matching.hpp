// host_matching.hpp -- the feasibility guard of the reference's front-end: maximum bipartite matching
// (Hopcroft-Karp), host C++.  Reference: cdef class HopcroftKarpSolverCython, sslap/feasibility_.pyx:95-225,
// called from _from_matrix / _from_sparse (auction_.pyx:562-566, :608-612) -- there only the cardinality reaches
// the auction path -- and exported as sslap.hopcroft_solve (feasibility_.pyx:227-283), whose pairings this
// restatement reproduces exactly: same phase structure (BFS layering from the free left vertices, then one DFS
// per free left vertex in ascending order), same neighbour order (stored order), same Dist bookkeeping.
// Differences: the queue holds N entries instead of N^2 ints (a vertex is enqueued at most once, :128-148), the
// recursion of :155-181 is an explicit stack (a 200 000-deep recursion overflows the C stack), and the Dist
// array is int with INT_MAX for the reference's double infinity.
#pragma once
#include <climits>
#include <cstdint>
#include <vector>

namespace misslap {

struct HopcroftKarp {
    int n_rows, n_cols;
    std::vector<int> row_ptr, col;   // adjacency of the left vertices, stored order (cumulative_idxs, :21-44)
    std::vector<int> pair_u, pair_v; // :112-113
    std::vector<int> dist;           // :114
    int dist_nil = INT_MAX;          // :116
    int matching = 0;

    // loc: int32[nnz][2], rows ascending (the reference's cumulative_idxs assumes it as well)
    HopcroftKarp(const int32_t *loc, int64_t nnz, int n_rows_, int n_cols_)
        : n_rows(n_rows_), n_cols(n_cols_), row_ptr((size_t)n_rows_ + 1, 0), col((size_t)nnz),
          pair_u((size_t)n_rows_, -1), pair_v((size_t)n_cols_, -1), dist((size_t)n_rows_, 0) {
        for (int64_t k = 0; k < nnz; ++k) {
            row_ptr[(size_t)loc[2 * k] + 1] += 1;
            col[(size_t)k] = loc[2 * k + 1];
        }
        for (int i = 0; i < n_rows; ++i) row_ptr[(size_t)i + 1] += row_ptr[(size_t)i];
    }

    // The same graph from a CSR (row pointers + column of every entry, `stride` ints apart): the fallback of the GPU
    // matcher on the graph a solver handle holds in device memory.
    HopcroftKarp(const int *row_ptr_, const int *col_, int stride, int n_rows_, int n_cols_)
        : n_rows(n_rows_), n_cols(n_cols_), row_ptr(row_ptr_, row_ptr_ + n_rows_ + 1), col((size_t)row_ptr_[n_rows_]),
          pair_u((size_t)n_rows_, -1), pair_v((size_t)n_cols_, -1), dist((size_t)n_rows_, 0) {
        for (size_t k = 0; k < col.size(); ++k) col[k] = col_[k * (size_t)stride];
    }

    // Start from a matching found elsewhere (the GPU matcher's, when it gave up): match_row[u] = v / match_col[v] = u,
    // -1 = free.  Inconsistent pairs are dropped.
    void seed(const int *match_row, const int *match_col) {
        for (int u = 0; u < n_rows; ++u) {
            const int v = match_row[u];
            if (v >= 0 && v < n_cols && match_col[v] == u && pair_v[v] == -1) {
                pair_u[u] = v;
                pair_v[v] = u;
                matching += 1;
            }
        }
    }

    void bfs(std::vector<int> &queue) {  // breadth_first_search, :119-150
        size_t front = 0, back = 0;
        for (int u = 0; u < n_rows; ++u) {
            if (pair_u[u] == -1) {
                dist[u] = 0;
                queue[back++] = u;
            } else {
                dist[u] = INT_MAX;
            }
        }
        int nil = INT_MAX;
        while (front < back) {
            const int u = queue[front++];
            if (dist[u] < nil) {
                for (int g = row_ptr[u]; g < row_ptr[u + 1]; ++g) {
                    const int pu = pair_v[col[g]];
                    if (pu == -1) {
                        if (nil == INT_MAX) nil = dist[u] + 1;
                    } else if (dist[pu] == INT_MAX) {
                        dist[pu] = dist[u] + 1;
                        queue[back++] = pu;
                    }
                }
            }
        }
        dist_nil = nil;
    }

    // depth_first_search(u), :152-181, with an explicit stack of (vertex, next neighbour)
    bool dfs(int root, std::vector<int> &st_u, std::vector<int> &st_g) {
        size_t depth = 0;
        st_u[0] = root;
        st_g[0] = row_ptr[root];
        bool found = false;  // result of the call that has just returned
        for (;;) {
            const int u = st_u[depth];
            if (found) {  // the child call returned 1: finish `if self.depth_first_search(pairu)` (:169-174)
                const int v = col[st_g[depth] - 1];
                pair_v[v] = u;
                pair_u[u] = v;
                if (depth == 0) return true;
                depth -= 1;
                continue;  // propagate 1 upwards
            }
            bool descended = false;
            while (st_g[depth] < row_ptr[u + 1]) {
                const int v = col[st_g[depth]++];
                const int pu = pair_v[v];
                const int d = pu == -1 ? dist_nil : dist[pu];
                if (d != INT_MAX && d == dist[u] + 1) {  // :168 (inf == finite + 1 is never true)
                    if (pu == -1) {  // depth_first_search(-1) returns 1 (:181)
                        pair_v[v] = u;
                        pair_u[u] = v;
                        found = true;
                    } else {
                        depth += 1;
                        st_u[depth] = pu;
                        st_g[depth] = row_ptr[pu];
                        descended = true;
                    }
                    break;
                }
            }
            if (descended) continue;
            if (found) {
                if (depth == 0) return true;
                depth -= 1;
                continue;
            }
            dist[u] = INT_MAX;  // :176
            if (depth == 0) return false;
            depth -= 1;  // the parent goes on with its next neighbour; `found` stays false
        }
    }

    int solve() {  // :183-196
        std::vector<int> queue((size_t)n_rows), st_u((size_t)n_rows + 1), st_g((size_t)n_rows + 1);
        for (;;) {
            bfs(queue);
            if (dist_nil == INT_MAX) break;
            for (int u = 0; u < n_rows; ++u)
                if (pair_u[u] == -1 && dfs(u, st_u, st_g)) matching += 1;
        }
        return matching;
    }
};

}  // namespace misslap
