// host_batch.hpp -- many independent problems solved in LOCKSTEP on one HIP stream (misslap_solve_batch).
// (part of the single translation unit misslap.hip; included in the order given there)
//
// Why.  During 92 % of a solve one problem occupies ONE of the 256 CUs (the tail kernels), so independent problems
// overlap almost freely -- but a solve is also ~1 400 small launches, and a GPU that is fed from many queues retires
// only ~90 000 launches per second over all of them (round 4: every small launch takes 47-67 us next to 15 busy queues):
// 16 solves at a time from 16 host threads / streams stall at 6-8x the single-solve throughput.  The reference's own use
// is a loop over problems (benchmarking.py:84-142).  Here the problems of a GROUP share one stream and every launch of
// the solve loop that several of them issue at the same point is ONE launch: the kernels' bodies are device functions
// (k_X_body, functor F_k_X), and k_batched<F> runs F for problem blockIdx.y with that problem's own arguments, passed
// by value in the kernel-argument block (a few hundred bytes per problem: up to kMax of them per launch).
//
// How.  Nothing of the per-handle driver is re-derived: every handle of a group runs the ordinary solve loop
// (drive_sharded + misslap_finish) on a FIBER of its own (ucontext), and the launch sites of host_rounds.hpp go through
// MISSLAP_LAUNCH*: outside a batch they launch as before; inside, they RECORD the call in the handle's pending list and
// return.  A fiber runs until it needs something from the device -- a status word that has not been posted yet
// (live_poll), or a drained stream (read_ctl) -- and yields.  When every fiber of the group has yielded, the scheduler
// flushes: it pops the head call of every pending list, issues the heads that name the same kernel as one batched
// launch (anything else -- the full-scan engine, asynchronous copies / fills -- goes out by itself, in order), and
// repeats until the lists are empty; then it drains the stream for the fibers that asked for it and resumes everybody.
// Per-handle order on the stream is the recording order, so every handle sees exactly the sequence of kernels it would
// have launched alone: the results are bit-identical by construction (and by test: tests/test_gpu_parity.py).
#pragma once
#include <sys/mman.h>
#include <ucontext.h>

#include <functional>
#include <memory>
#include <tuple>

namespace misslap {

// a trivially copyable tuple (what a recorded launch keeps of its arguments, and what k_batched gets per problem)
template <class... A>
struct ArgPack;
template <>
struct ArgPack<> {};
template <class H, class... T>
struct ArgPack<H, T...> {
    H h;
    ArgPack<T...> t;
    ArgPack() = default;
    ArgPack(const H &hh, const T &...tt) : h(hh), t(tt...) {}
};
template <class F, class... B>
__device__ __forceinline__ void pack_call(const ArgPack<> &, const B &...b) {
    F::run(b...);
}
template <class F, class H, class... T, class... B>
__device__ __forceinline__ void pack_call(const ArgPack<H, T...> &p, const B &...b) {
    pack_call<F>(p.t, b..., p.h);
}
template <class... A>
struct BatchSlots {
    // (a kernel-argument block holds 4 KB; 16 problems per launch at most)
    static constexpr int kMaxRaw = (int)((4096 - 64) / sizeof(ArgPack<A...>));
    static constexpr int kMax = kMaxRaw < 1 ? 1 : kMaxRaw > 16 ? 16 : kMaxRaw;
    int n;
    ArgPack<A...> s[kMax];
};
// problem blockIdx.y of the launch runs F with its own arguments; blockIdx.x / gridDim.x are what F's body expects (the
// grid is the LARGEST of the merged launches' grids: every kernel of the solve loop tolerates a grid above the one the
// host computed -- the host's K is an upper bound anyway)
template <class F, int kBounds, class... A>
__global__ __launch_bounds__(kBounds) void k_batched(BatchSlots<A...> p) {
    if ((int)blockIdx.y < p.n) pack_call<F>(p.s[blockIdx.y]);
}
// ... the same for more problems than a kernel-argument block carries: their arguments staged in device memory (a ring of
// the group, filled by an asynchronous copy on the same stream right in front of the launch)
template <class F, int kBounds, class... A>
__global__ __launch_bounds__(kBounds) void k_batched_ptr(const ArgPack<A...> *slots, int n) {
    if ((int)blockIdx.y < n) pack_call<F>(slots[blockIdx.y]);
}

}  // namespace misslap

namespace {

constexpr size_t kBatchArgBytes = 640;  // the largest by-value argument pack of a mergeable launch

struct BatchGroup;
struct BatchCall {
    // the identity of a mergeable launch (same pointer = same kernel body, bounds and argument types); launches
    // calls[0 .. n) as one batched launch per kMax of them.  nullptr: `single` goes onto the stream by itself
    void (*merge)(BatchGroup &, BatchCall *const *calls, int n) = nullptr;
    std::function<void(hipStream_t)> single;
    // the first call of a handle's tail sequence: it stays in the list until EVERY unfinished problem of the group has
    // reached its own (batch_flush).  The tail kernels run for milliseconds on one workgroup each: issued as the problems
    // get there, one after the other on the group's in-order stream, they would serialise; issued together they overlap
    bool hold = false;
    dim3 grid, block;
    alignas(16) unsigned char args[kBatchArgBytes];
};

struct BatchGroup;
struct BatchCall;
// A fiber's stack: its own mapping with an inaccessible page below it, so that running out of it is a fault at the
// guard page and not a silent write into whatever the heap had next to it.
struct FiberStack {
    void *map = nullptr;
    size_t map_bytes = 0;
    static constexpr size_t kGuard = 16384;
    bool alloc(size_t bytes) {
        map_bytes = bytes + kGuard;
        map = mmap(nullptr, map_bytes, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_STACK, -1, 0);
        if (map == MAP_FAILED) {
            map = nullptr;
            return false;
        }
        return mprotect(map, kGuard, PROT_NONE) == 0;  // (stacks grow downwards: the low end)
    }
    char *base() const { return static_cast<char *>(map) + kGuard; }
    size_t size() const { return map_bytes - kGuard; }
    FiberStack() = default;
    FiberStack(const FiberStack &) = delete;
    FiberStack &operator=(const FiberStack &) = delete;
    ~FiberStack() {
        if (map) munmap(map, map_bytes);
    }
};

struct BatchFiber {
    enum State { kRunnable, kPolling, kWantsSync, kDone };
    misslap_solver *h = nullptr;
    BatchGroup *grp = nullptr;
    ucontext_t ctx;
    FiberStack stack;
    State state = kRunnable;
    int rc = MISSLAP_OK;
    std::string err;
    int32_t *sol = nullptr;
    misslap_meta *meta = nullptr;
    std::vector<BatchCall> pending;
    size_t at = 0;           // pending[at ..) have not been issued yet
    bool hold_next = false;  // the next recorded call is the head of a tail sequence
    hipStream_t own_stream = nullptr;  // the handle's stream outside the batch
    bool own_own_stream = false;
};
struct BatchGroup {
    hipStream_t stream = nullptr;
    ucontext_t sched;
    std::vector<std::unique_ptr<BatchFiber>> fibers;
    long long launches_merged = 0, launches_issued = 0;  // calls recorded / launches that went out
    double ms_fibers = 0, ms_flush = 0, ms_wait = 0;     // where this group's host thread spent its time
    // argument ring for launches over more problems than a kernel-argument block carries (k_batched_ptr): a pinned host
    // buffer the packs are written to, and its device twin they are copied to on the stream in front of the launch.  A
    // region is reused only after the stream has been drained (once per kBatchRingBytes of arguments)
    char *h_ring = nullptr, *d_ring = nullptr;
    size_t ring_at = 0;
};
constexpr size_t kBatchRingBytes = (size_t)32 << 20;
// n packs of `bytes` each -> device memory, ordered on the group's stream; returns the device address (nullptr: failed)
inline const void *batch_stage_args(BatchGroup &g, BatchCall *const *calls, int n, size_t bytes) {
    const size_t need = ((size_t)n * bytes + 255) & ~(size_t)255;
    if (!g.h_ring) {
        if (hipHostMalloc((void **)&g.h_ring, kBatchRingBytes) != hipSuccess || hipMalloc((void **)&g.d_ring, kBatchRingBytes) != hipSuccess) return nullptr;
    }
    if (need > kBatchRingBytes) return nullptr;
    if (g.ring_at + need > kBatchRingBytes) {
        if (hipStreamSynchronize(g.stream) != hipSuccess) return nullptr;  // every earlier copy and launch has read its region
        g.ring_at = 0;
    }
    char *hp = g.h_ring + g.ring_at, *dp = g.d_ring + g.ring_at;
    for (int k = 0; k < n; ++k) std::memcpy(hp + (size_t)k * bytes, calls[k]->args, bytes);
    if (hipMemcpyAsync(dp, hp, (size_t)n * bytes, hipMemcpyHostToDevice, g.stream) != hipSuccess) return nullptr;
    g.ring_at += need;
    return dp;
}

// ---- recording (called from the launch sites through MISSLAP_LAUNCH*) -------------------------------------------------
template <class F, int kBounds, class... A>
void batch_merge(BatchGroup &g, BatchCall *const *calls, int n) {
    using Slots = BatchSlots<A...>;
    unsigned gx = 1;
    for (int k = 0; k < n; ++k) gx = std::max(gx, calls[k]->grid.x);
    if (n > Slots::kMax) {  // through the device ring: ONE launch whatever n
        const void *d = batch_stage_args(g, calls, n, sizeof(ArgPack<A...>));
        if (d) {
            hipLaunchKernelGGL((k_batched_ptr<F, kBounds, A...>), dim3(gx, (unsigned)n), calls[0]->block, 0, g.stream,
                               static_cast<const ArgPack<A...> *>(d), n);
            return;
        }
    }
    for (int k0 = 0; k0 < n; k0 += Slots::kMax) {  // (n <= kMax: the arguments travel in the kernel-argument block)
        Slots p;
        p.n = std::min(Slots::kMax, n - k0);
        for (int k = 0; k < p.n; ++k) std::memcpy(static_cast<void *>(&p.s[k]), calls[k0 + k]->args, sizeof(ArgPack<A...>));
        hipLaunchKernelGGL((k_batched<F, kBounds, A...>), dim3(gx, (unsigned)p.n), calls[k0]->block, 0, g.stream, p);
    }
}
template <class F, int kBounds, class... A>
void batch_record(misslap_solver *h, dim3 g, dim3 b, const A &...a) {
    static_assert(sizeof(ArgPack<A...>) <= kBatchArgBytes, "raise kBatchArgBytes");
    static_assert(std::is_trivially_copyable<ArgPack<A...>>::value, "kernel arguments are copied as bytes");
    BatchFiber *f = h->batch;
    f->pending.emplace_back();
    BatchCall &c = f->pending.back();
    c.merge = &batch_merge<F, kBounds, A...>;
    c.grid = g;
    c.block = b;
    c.hold = f->hold_next;
    f->hold_next = false;
    new (c.args) ArgPack<A...>(a...);
}
inline void batch_record_plain(misslap_solver *h, std::function<void(hipStream_t)> fn) {
    BatchFiber *f = h->batch;
    f->pending.emplace_back();
    f->pending.back().single = std::move(fn);
    f->pending.back().hold = f->hold_next;
    f->hold_next = false;
}
// the fiber gives the thread back to its group's scheduler
inline void batch_yield(misslap_solver *h, BatchFiber::State why) {
    BatchFiber *f = h->batch;
    f->state = why;
    swapcontext(&f->ctx, &f->grp->sched);
}

#define MISSLAP_UNPAREN(...) __VA_ARGS__
// a launch of the solve loop that several problems of a batch can share: KERNEL = the __global__ wrapper (a batch-less
// launch looks exactly as before: same kernel name in a trace), FUNCTOR = its body as a callable, BOUNDS = its block size
#define MISSLAP_LAUNCH(H, KERNEL, FUNCTOR, BOUNDS, GRID, BLOCK, ...)                                    \
    do {                                                                                               \
        if (!(H)->batch) hipLaunchKernelGGL(KERNEL, GRID, BLOCK, 0, (H)->stream, __VA_ARGS__);         \
        else batch_record<MISSLAP_UNPAREN FUNCTOR, BOUNDS>((H), GRID, BLOCK, __VA_ARGS__);             \
    } while (0)
// ... and one that goes out by itself also inside a batch (the full-scan engine: dynamic LDS, one workgroup per CU)
#define MISSLAP_LAUNCH_PLAIN(H, KERNEL, GRID, BLOCK, LDS, ...)                                                          \
    do {                                                                                                               \
        if (!(H)->batch) {                                                                                             \
            hipLaunchKernelGGL(KERNEL, GRID, BLOCK, LDS, (H)->stream, __VA_ARGS__);                                    \
        } else {                                                                                                       \
            const dim3 g_ = (GRID), b_ = (BLOCK);                                                                      \
            const unsigned l_ = (unsigned)(LDS);                                                                       \
            batch_record_plain((H), [g_, b_, l_, args_ = std::make_tuple(__VA_ARGS__)](hipStream_t st_) {              \
                std::apply([&](const auto &...x_) { hipLaunchKernelGGL(KERNEL, g_, b_, l_, st_, x_...); }, args_);     \
            });                                                                                                        \
        }                                                                                                              \
    } while (0)

// asynchronous fills / copies of the solve loop, in stream order
inline hipError_t stream_memset(misslap_solver *h, void *p, int v, size_t n) {
    if (!h->batch) return hipMemsetAsync(p, v, n, h->stream);
    batch_record_plain(h, [p, v, n](hipStream_t st) { (void)hipMemsetAsync(p, v, n, st); });
    return hipSuccess;
}
inline hipError_t stream_memcpy(misslap_solver *h, void *dst, const void *src, size_t n, hipMemcpyKind kind) {
    if (!h->batch) return hipMemcpyAsync(dst, src, n, kind, h->stream);
    batch_record_plain(h, [dst, src, n, kind](hipStream_t st) { (void)hipMemcpyAsync(dst, src, n, kind, st); });
    return hipSuccess;
}
// everything this handle has enqueued (recorded) so far has completed
inline hipError_t stream_sync(misslap_solver *h) {
    if (!h->batch) return hipStreamSynchronize(h->stream);
    batch_yield(h, BatchFiber::kWantsSync);
    return hipSuccess;
}

}  // namespace
