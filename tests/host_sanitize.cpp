// host_sanitize.cpp -- the host-only entry points of libmisslap under AddressSanitizer + UBSan (CPU box, no GPU).
//
// Built and run by tests/test_host_sanitize.py:
//   hipcc -fsanitize=address,undefined -fno-gpu-sanitize ... sslap_amd/csrc/misslap.hip -> libmisslap_asan.so
//   hipcc -fsanitize=address,undefined tests/host_sanitize.cpp -L... -lmisslap_asan    -> host_sanitize
// Exercises: misslap_hopcroft_karp (paths, ladders, random graphs, bad input), misslap_drive_sharded over C++
// stand-ins for the round kernels with a two-rank in-process communicator (threads), option validation of
// misslap_create / misslap_create_dense (every error return ahead of the first HIP call, and the no-device return),
// communicator construction / destruction, misslap_trim_caches, the error string.  Any sanitizer report aborts the
// process (halt_on_error), which the test sees as a non-zero exit code.  Never run on the GPU box.
#include <algorithm>
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <condition_variable>
#include <cstring>
#include <functional>
#include <mutex>
#include <random>
#include <thread>
#include <vector>

#include "misslap.h"

#define CHECK(cond)                                                          \
    do {                                                                     \
        if (!(cond)) {                                                       \
            std::fprintf(stderr, "CHECK failed: %s (line %d): %s\n", #cond, __LINE__, misslap_last_error()); \
            std::exit(1);                                                    \
        }                                                                    \
    } while (0)

// ---- maximum matching ---------------------------------------------------------------------------------------------
static int brute_force_matching(const std::vector<int32_t> &loc, int n, int m) {  // Kuhn's algorithm, tiny graphs
    std::vector<std::vector<int>> adj((size_t)n);
    for (size_t k = 0; k < loc.size() / 2; ++k) adj[(size_t)loc[2 * k]].push_back(loc[2 * k + 1]);
    std::vector<int> mt((size_t)m, -1);
    std::vector<char> seen;
    std::function<bool(int)> aug = [&](int u) {
        for (int v : adj[(size_t)u]) {
            if (seen[(size_t)v]) continue;
            seen[(size_t)v] = 1;
            if (mt[(size_t)v] == -1 || aug(mt[(size_t)v])) {
                mt[(size_t)v] = u;
                return true;
            }
        }
        return false;
    };
    int res = 0;
    for (int u = 0; u < n; ++u) {
        seen.assign((size_t)m, 0);
        res += aug(u);
    }
    return res;
}

static void test_matching() {
    // a path: row k -> columns k, k + 1 (the last row only k): ONE augmenting path through every vertex
    const int n = 20000;
    std::vector<int32_t> loc;
    for (int k = 0; k < n; ++k) {
        loc.push_back(k), loc.push_back(k);
        if (k + 1 < n) loc.push_back(k), loc.push_back(k + 1);
    }
    int32_t size = -1;
    std::vector<int32_t> left((size_t)n), right((size_t)n);
    CHECK(misslap_hopcroft_karp(loc.data(), (int64_t)loc.size() / 2, n, n, &size, left.data(), right.data()) == MISSLAP_OK);
    CHECK(size == n);
    for (int k = 0; k < n; ++k) CHECK(left[(size_t)k] >= 0 && right[(size_t)left[(size_t)k]] == k);
    // random small graphs against an independent matcher, with and without the pairing outputs
    std::mt19937 rng(12345);
    for (int rep = 0; rep < 200; ++rep) {
        const int a = 1 + (int)(rng() % 12), b = 1 + (int)(rng() % 12);
        std::vector<int32_t> g;
        for (int i = 0; i < a; ++i)
            for (int j = 0; j < b; ++j)
                if (rng() % 3 == 0) g.push_back(i), g.push_back(j);
        int32_t sz = -1;
        std::vector<int32_t> l((size_t)a), r((size_t)b);
        CHECK(misslap_hopcroft_karp(g.empty() ? nullptr : g.data(), (int64_t)g.size() / 2, a, b, &sz,
                                    rep % 2 ? l.data() : nullptr, rep % 3 ? r.data() : nullptr) == MISSLAP_OK);
        CHECK(sz == brute_force_matching(g, a, b));
    }
    // input contract: out-of-range entries, unsorted rows, null arguments
    int32_t bad1[] = {0, 0, 5, 1};
    CHECK(misslap_hopcroft_karp(bad1, 2, 2, 2, &size, nullptr, nullptr) == MISSLAP_ERR_INVALID);
    int32_t bad2[] = {1, 0, 0, 1};
    CHECK(misslap_hopcroft_karp(bad2, 2, 2, 2, &size, nullptr, nullptr) == MISSLAP_ERR_INVALID);
    int32_t bad3[] = {0, -1};
    CHECK(misslap_hopcroft_karp(bad3, 1, 1, 1, &size, nullptr, nullptr) == MISSLAP_ERR_INVALID);
    CHECK(misslap_hopcroft_karp(bad1, 2, 2, 2, nullptr, nullptr, nullptr) == MISSLAP_ERR_INVALID);
    CHECK(std::strlen(misslap_last_error()) > 0);
}

// ---- the solve loop over stand-ins ---------------------------------------------------------------------------------
// A toy "auction": K falls by a data-dependent fraction per round, a phase ends at K == 0, three phases.  Two ranks
// (threads) hold replicated state and exchange through an in-process all-reduce; the loop must issue the same
// sequence of collectives on both and end with identical state.
struct Rendezvous {
    std::mutex m;
    std::condition_variable cv;
    int arrived = 0, phase = 0;  // phase 0: contributing, 1: everybody has contributed, results are being read
    std::vector<int64_t> acc64;
    std::vector<int32_t> acc32;
};
struct ToyRank {
    int rank = 0, world = 1;
    int64_t K = 0, its = 0, N = 0;
    int phase = 0;
    int64_t max_iter = 0;
    std::vector<int64_t> best_key;
    std::vector<int32_t> best_pos;
    Rendezvous *rv = nullptr;
    int64_t collectives = 0, tails = 0, posted = 0;
    int64_t stat_K[2] = {0, 0}, stat_its[2] = {0, 0};
};
template <class T>
static int toy_allreduce(ToyRank *t, std::vector<T> &acc, T *buf, int64_t count, bool is_max) {
    Rendezvous &rv = *t->rv;
    std::unique_lock<std::mutex> lk(rv.m);
    rv.cv.wait(lk, [&] { return rv.phase == 0; });  // the previous collective has been left by everybody
    if (rv.arrived == 0) acc.assign(buf, buf + count);
    else
        for (int64_t k = 0; k < count; ++k)
            acc[(size_t)k] = is_max ? std::max(acc[(size_t)k], buf[k]) : std::min(acc[(size_t)k], buf[k]);
    if (++rv.arrived == t->world) {
        rv.phase = 1;
        rv.cv.notify_all();
    } else {
        rv.cv.wait(lk, [&] { return rv.phase == 1; });
    }
    std::copy(acc.begin(), acc.begin() + count, buf);
    if (--rv.arrived == 0) {
        rv.phase = 0;
        rv.cv.notify_all();
    }
    t->collectives += 1;
    return 0;
}
static void test_drive_sharded() {
    const int world = 2;
    Rendezvous rv;
    std::vector<ToyRank> ranks((size_t)world);
    std::vector<std::thread> threads;
    std::atomic<int> failures{0};
    for (int r = 0; r < world; ++r) {
        threads.emplace_back([&, r] {
            ToyRank &t = ranks[(size_t)r];
            t.rank = r, t.world = world, t.N = t.K = 5000, t.max_iter = 100000, t.rv = &rv;
            t.best_key.assign(64, 0);
            t.best_pos.assign(64, 0x7fffffff);
            misslap_comm_ops cops;
            std::memset(&cops, 0, sizeof cops);
            cops.struct_size = (int32_t)sizeof cops;
            cops.rank = r, cops.world = world, cops.ctx = &t;
            cops.allreduce_max_i64 = [](void *c, void *buf, int64_t n, void *) {
                ToyRank *x = static_cast<ToyRank *>(c);
                return toy_allreduce<int64_t>(x, x->rv->acc64, static_cast<int64_t *>(buf), n, true);
            };
            cops.allreduce_min_i32 = [](void *c, void *buf, int64_t n, void *) {
                ToyRank *x = static_cast<ToyRank *>(c);
                return toy_allreduce<int32_t>(x, x->rv->acc32, static_cast<int32_t *>(buf), n, false);
            };
            misslap_comm *comm = nullptr;
            if (misslap_comm_init_custom(&comm, &cops) != MISSLAP_OK) {
                failures += 1;
                return;
            }
            misslap_round_ops o;
            std::memset(&o, 0, sizeof o);
            o.struct_size = (int32_t)sizeof o;
            o.tail_threshold = 64, o.shard_min_K = 1500, o.rounds_per_sync = 4, o.max_iter = t.max_iter;
            o.large_round_K = 2048, o.rounds_per_sync_large = 2;
            o.ctx = &t;
            o.status = [](void *c, int64_t *K, int64_t *its) {
                ToyRank *x = static_cast<ToyRank *>(c);
                *K = x->K, *its = x->its;
                return 0;
            };
            o.round_bid = [](void *c) {
                ToyRank *x = static_cast<ToyRank *>(c);
                if (x->K > 64 && x->its < x->max_iter)  // shard-local maximum in a sharded round, replicated otherwise
                    x->best_key[(size_t)(x->its % 64)] = x->K * 7 + (x->K >= 1500 ? x->rank : 0);
                return 0;
            };
            o.round_tiebreak = [](void *c) {
                ToyRank *x = static_cast<ToyRank *>(c);
                if (x->K > 64 && x->its < x->max_iter)
                    x->best_pos[(size_t)(x->its % 64)] = (int32_t)(x->K % 1000) + (x->K >= 1500 ? x->rank : 0);
                return 0;
            };
            o.round_apply = [](void *c) {
                ToyRank *x = static_cast<ToyRank *>(c);
                if (x->K > 64 && x->its < x->max_iter) {  // a round that is not live is a no-op, like the kernels
                    x->K -= std::max<int64_t>(1, x->K / 3);
                    x->its += 1;
                }
                return 0;
            };
            o.run_tail = [](void *c) {
                ToyRank *x = static_cast<ToyRank *>(c);
                x->its += x->K, x->K = 0, x->tails += 1;
                return 0;
            };
            o.phase_end = [](void *c, int32_t *fin) {
                ToyRank *x = static_cast<ToyRank *>(c);
                if (x->its >= x->max_iter || ++x->phase == 3) *fin = 1;
                else x->K = x->N, *fin = 0;
                return 0;
            };
            if (r == 0) {  // one rank with the trailing status reads, one without: both forms of the batch loop
                o.status_post = [](void *c, int32_t slot) {
                    ToyRank *x = static_cast<ToyRank *>(c);
                    x->stat_K[slot & 1] = x->K, x->stat_its[slot & 1] = x->its, x->posted += 1;
                    return 0;
                };
                o.status_take = [](void *c, int32_t slot, int64_t *K, int64_t *its) {
                    ToyRank *x = static_cast<ToyRank *>(c);
                    *K = x->stat_K[slot & 1], *its = x->stat_its[slot & 1];
                    return 0;
                };
            }
            o.best_key = t.best_key.data(), o.best_pos = t.best_pos.data(), o.n_objects = 64;
            if (misslap_drive_sharded(&o, comm) != MISSLAP_OK) failures += 1;
            misslap_comm_destroy(comm);
        });
    }
    for (auto &th : threads) th.join();
    CHECK(failures == 0);
    CHECK(ranks[0].K == 0 && ranks[1].K == 0 && ranks[0].its == ranks[1].its && ranks[0].phase == 3);
    CHECK(ranks[0].collectives == ranks[1].collectives && ranks[0].collectives > 0 && ranks[0].collectives % 2 == 0);
    CHECK(ranks[0].tails == 3 && ranks[0].posted > 0);
    CHECK(ranks[0].best_key == ranks[1].best_key && ranks[0].best_pos == ranks[1].best_pos);
    // malformed operation tables
    misslap_round_ops bad;
    std::memset(&bad, 0, sizeof bad);
    CHECK(misslap_drive_sharded(&bad, nullptr) == MISSLAP_ERR_INVALID);
    CHECK(misslap_drive_sharded(nullptr, nullptr) == MISSLAP_ERR_INVALID);
    misslap_comm_ops cbad;
    std::memset(&cbad, 0, sizeof cbad);
    misslap_comm *c = nullptr;
    CHECK(misslap_comm_init_custom(&c, &cbad) == MISSLAP_ERR_INVALID);
    CHECK(misslap_comm_destroy(nullptr) == MISSLAP_OK);
}

// ---- option validation (everything ahead of the first HIP call) and the no-device return ----------------------------
static void test_options() {
    int32_t loc[] = {0, 0, 1, 1};
    double val[] = {1.0, 2.0};
    misslap_solver *h = nullptr;
    misslap_options o;
    std::memset(&o, 0, sizeof o);
    CHECK(misslap_create(&h, 2, loc, val, &o) == MISSLAP_ERR_INVALID);  // struct_size 0
    o.struct_size = 90;
    CHECK(misslap_create(&h, 2, loc, val, &o) == MISSLAP_ERR_INVALID);
    o.struct_size = (int32_t)sizeof o;
    o.reserved[6] = 1;
    CHECK(misslap_create(&h, 2, loc, val, &o) == MISSLAP_ERR_INVALID);
    o.reserved[6] = 0;
    o.cand_mode = 3;
    CHECK(misslap_create(&h, 2, loc, val, &o) == MISSLAP_ERR_INVALID);
    o.cand_mode = 0;
    o.cand_refresh_min = 99;
    CHECK(misslap_create(&h, 2, loc, val, &o) == MISSLAP_ERR_INVALID);
    o.cand_refresh_min = 0;
    o.tiled_shape = 1000;
    CHECK(misslap_create(&h, 2, loc, val, &o) == MISSLAP_ERR_INVALID);
    o.tiled_shape = 0;
    CHECK(misslap_create(&h, 2, nullptr, val, &o) == MISSLAP_ERR_INVALID);
    CHECK(misslap_create(&h, 0, loc, val, &o) == MISSLAP_ERR_INVALID);
    CHECK(misslap_create(&h, 2, loc, val, nullptr) == MISSLAP_ERR_INVALID);
    o.nnz_limit = 2;
    CHECK(misslap_create(&h, 2, loc, val, &o) == MISSLAP_ERR_INVALID);  // nnz >= limit
    o.nnz_limit = 0;
    // a version-1 struct (88 bytes, knobs in reserved[8]) is accepted up to the device check
    unsigned char v1[88];
    std::memset(v1, 0, sizeof v1);
    const int32_t sz = 88;
    std::memcpy(v1, &sz, 4);
    const int rc1 = misslap_create(&h, 2, loc, val, reinterpret_cast<const misslap_options *>(v1));
    const int rc2 = misslap_create(&h, 2, loc, val, &o);
    CHECK(rc1 == rc2);
    if (rc2 == MISSLAP_OK) {  // (a GPU is present after all)
        misslap_destroy(h);
    } else {
        CHECK(rc2 == MISSLAP_ERR_NO_DEVICE);
        CHECK(std::strstr(misslap_last_error(), "no CPU fallback") != nullptr);
    }
    int64_t nnz = -1;
    double mat[] = {1, 2, 3, 4};
    CHECK(misslap_create_dense(&h, 0, 2, mat, &o, &nnz) == MISSLAP_ERR_INVALID);
    CHECK(misslap_create_dense(&h, 2, 2, nullptr, &o, &nnz) == MISSLAP_ERR_INVALID);
    CHECK(misslap_destroy(nullptr) == MISSLAP_OK);
    CHECK(misslap_dims(nullptr, nullptr, nullptr, nullptr) == MISSLAP_ERR_INVALID);
    CHECK(misslap_solve(nullptr, nullptr, nullptr) == MISSLAP_ERR_INVALID);
    CHECK(misslap_get_status(nullptr, nullptr) == MISSLAP_ERR_INVALID);
    int64_t freed = -1;
    CHECK(misslap_trim_caches(&freed) == MISSLAP_OK && freed == 0);
    CHECK(misslap_trim_caches(nullptr) == MISSLAP_OK);
    CHECK(misslap_abi_version() == MISSLAP_ABI_VERSION);
}

int main() {
    test_matching();
    test_drive_sharded();
    test_options();
    std::puts("host_sanitize ok");
    return 0;
}
