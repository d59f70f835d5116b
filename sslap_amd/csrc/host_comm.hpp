// host_comm.hpp -- the multi-GPU exchange step behind the C ABI: communicators and the sharded solve loop.
//
// The path shards within a round: bidders are independent given the common price vector
// (reference auction_.pyx:339-365 reads prices that are only written at :397), so the one exchange of a round is the
// per-object arg-max of the bids (:375-385) -- all-reduce MAX over the bid keys, then all-reduce MIN over the list
// positions of the bidders holding the maximum (the earliest bidder wins equal bids, strict '>' of :379).  Both are
// issued on the solver's own stream between k_bid / k_tiebreak / k_apply: no host status read inside a sharded
// round, no Python or torch in the loop.
//
// Two communicator kinds: RCCL (librccl.so.1 is opened at run time with dlopen, so that the library itself carries
// no link-time dependency on it and binds to the copy the process already has, e.g. a PyTorch wheel's) and "custom"
// (caller-provided all-reduce callbacks: other transports, and the tests' gloo / in-process stand-ins).
#pragma once
#include <dlfcn.h>
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>

#include "../../include/misslap.h"

struct misslap_comm {
    int rank = 0, world = 1;
    // custom
    misslap_comm_ops ops{};
    bool custom = false;
    // RCCL
    void *nccl_comm = nullptr;
    int device = 0;
    long long sharded_rounds = 0;  // rounds of the current solve that went through the exchange step
};

namespace misslap {

// the few RCCL entry points the exchange needs (rccl.h: ncclInt32 = 2, ncclInt64 = 4, ncclMax = 2, ncclMin = 3)
struct RcclApi {
    typedef struct {
        char internal[128];
    } UniqueId;
    int (*GetUniqueId)(UniqueId *) = nullptr;
    int (*CommInitRank)(void **, int, UniqueId, int) = nullptr;
    int (*CommDestroy)(void *) = nullptr;
    int (*AllReduce)(const void *, void *, size_t, int, int, void *, hipStream_t) = nullptr;
    int (*CommCount)(void *, int *) = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
    void *handle = nullptr;
    std::string error;
};

inline RcclApi &rccl_api() {
    static RcclApi api;
    static std::once_flag once;
    std::call_once(once, [] {
        for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            api.handle = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (api.handle) break;
        }
        if (!api.handle) {
            api.error = std::string("librccl.so.1 cannot be opened: ") + (dlerror() ? dlerror() : "?");
            return;
        }
        auto sym = [&](const char *n) {
            void *p = dlsym(api.handle, n);
            if (!p && api.error.empty()) api.error = std::string("librccl: missing symbol ") + n;
            return p;
        };
        api.GetUniqueId = reinterpret_cast<decltype(api.GetUniqueId)>(sym("ncclGetUniqueId"));
        api.CommInitRank = reinterpret_cast<decltype(api.CommInitRank)>(sym("ncclCommInitRank"));
        api.CommDestroy = reinterpret_cast<decltype(api.CommDestroy)>(sym("ncclCommDestroy"));
        api.AllReduce = reinterpret_cast<decltype(api.AllReduce)>(sym("ncclAllReduce"));
        api.GetErrorString = reinterpret_cast<decltype(api.GetErrorString)>(sym("ncclGetErrorString"));
        api.CommCount = reinterpret_cast<decltype(api.CommCount)>(sym("ncclCommCount"));
    });
    return api;
}

constexpr int kNcclInt32 = 2, kNcclInt64 = 4, kNcclMax = 2, kNcclMin = 3;

// (The rounds of the full-scan regime are issued one at a time behind a status read, K known exactly, also WITHOUT a
// communicator.  Sending them through the batched path -- status read trailing by one round -- was measured in round 4:
// C4 7.99 -> 7.71 ms per solve, C3 unchanged, but on the stale upper bound of K the two or three rounds behind the end of
// the regime launch the scan engine for nothing: C3 all-launch fraction 0.419 -> 0.404.  Not kept.)

// The solve loop of AuctionSolver.solve() (auction_.pyx:268-306) over the ranks of a communicator, written against the
// round operations of misslap_round_ops (the GPU handle's, or a test's stand-ins).  Control decisions are taken from
// replicated state, so all ranks issue the same sequence of collectives.  Only rounds with K >= shard_min_K (and
// above the tail threshold) are sharded and exchanged; smaller rounds are replicated: every rank bids for every list
// position (or runs the persistent tail kernel), no communication -- an all-reduce over M keys costs more than such
// a round, and the replicas stay bit-identical because every step is deterministic.
template <class Fail>
int drive_sharded(const misslap_round_ops *o, misslap_comm *c, Fail &&fail) {
    auto exchange = [&](bool max64) -> int {
        if (!c) return MISSLAP_OK;
        void *buf = max64 ? o->best_key : o->best_pos;
        if (c->custom) {
            auto fn = max64 ? c->ops.allreduce_max_i64 : c->ops.allreduce_min_i32;
            const int rc = fn(c->ops.ctx, buf, o->n_objects, o->stream);
            return rc ? fail(MISSLAP_ERR_HIP, "custom all-reduce callback failed (%d)", rc) : MISSLAP_OK;
        }
        RcclApi &api = rccl_api();
        const int rc = api.AllReduce(buf, buf, (size_t)o->n_objects, max64 ? kNcclInt64 : kNcclInt32,
                                     max64 ? kNcclMax : kNcclMin, c->nccl_comm, (hipStream_t)o->stream);
        return rc ? fail(MISSLAP_ERR_HIP, "ncclAllReduce failed: %s", api.GetErrorString(rc)) : MISSLAP_OK;
    };
    int rc;
    for (;;) {
        for (;;) {  // rounds of one eps-phase
            int64_t K = 0, its = 0;
            if ((rc = o->status(o->ctx, &K, &its))) return rc;
            if (K == 0 || its >= o->max_iter) break;
            if (K >= o->shard_min_K && K > o->tail_threshold) {
                // a big round: bidders sharded over the ranks, per-object arg-max exchanged.  K is exact here, so
                // the device-side decision "K >= shard_min_K" is the same on every rank.  No host read until the
                // round is complete.  (The same without a communicator: nothing is exchanged.)
                if (c) c->sharded_rounds += 1;
                if ((rc = o->round_bid(o->ctx))) return rc;
                if ((rc = exchange(true))) return rc;
                if ((rc = o->round_tiebreak(o->ctx))) return rc;
                if ((rc = exchange(false))) return rc;
                if ((rc = o->round_apply(o->ctx))) return rc;
            } else if (K > o->tail_threshold) {
                // K never grows inside a phase: from here on every rank bids for everybody (replicated,
                // deterministic), no exchange; several rounds per status read.  While K is still large it falls
                // fast (by a third or more per round): short batches there, so that a stale upper bound of K does
                // not keep the big-round launch shapes alive for rounds that have long become small.
                auto batch_of = [&](int64_t k) {
                    const int rps = o->rounds_per_sync > 0 ? o->rounds_per_sync : 1;
                    return k > o->large_round_K && o->rounds_per_sync_large > 0 ? std::min(rps, o->rounds_per_sync_large)
                                                                                 : rps;
                };
                auto issue = [&](int n) -> int {
                    for (int r = 0; r < n; ++r) {
                        if ((rc = o->round_bid(o->ctx))) return rc;
                        if ((rc = o->round_tiebreak(o->ctx))) return rc;
                        if ((rc = o->round_apply(o->ctx))) return rc;
                    }
                    return MISSLAP_OK;
                };
                if (!o->status_post || !o->status_take) {
                    if ((rc = issue(batch_of(K)))) return rc;
                } else {
                    // The status read trails the rounds by one batch: these rounds are a few microseconds each, and
                    // a queue drained after every batch idles the device for longer than a batch runs.  A round that
                    // is not live is a no-op, so the batch issued on a stale "go on" costs its launches only.
                    int slot = 0;
                    bool outstanding = false;
                    for (bool stop = false; !stop; slot ^= 1) {
                        if ((rc = issue(batch_of(K)))) return rc;
                        if ((rc = o->status_post(o->ctx, slot))) return rc;
                        if (outstanding) {
                            if ((rc = o->status_take(o->ctx, slot ^ 1, &K, &its))) return rc;
                            stop = K <= o->tail_threshold || its >= o->max_iter;
                        }
                        outstanding = true;
                    }
                    // (the status read at the top of the loop drains the batch that is still in flight)
                }
            } else {
                if ((rc = o->run_tail(o->ctx))) return rc;
            }
        }
        int32_t fin = 0;
        if ((rc = o->phase_end(o->ctx, &fin))) return rc;
        if (fin) break;
    }
    return MISSLAP_OK;
}

}  // namespace misslap
