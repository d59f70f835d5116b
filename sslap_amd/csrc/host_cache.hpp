// host_cache.hpp -- device-memory helpers and the process-wide caches: blocks carved from one hipMalloc, scratch released on every path, parked streams / pinned mirrors, the device-block cache.
// (part of the single translation unit misslap.hip; included in the order given there)
#pragma once

namespace {

template <class T>
int dev_alloc(T **p, size_t n) {
    HIP_TRY(hipMalloc((void **)p, (n ? n : 1) * sizeof(T)));
    return MISSLAP_OK;
}

// Device temporaries of a constructor: freed when the scope is left, on every path.
struct DevScratch {
    std::vector<Blk> blks;
    int device = 0;
    bool drained = false;  // set by the owner after it has synchronised the stream(s) that used the blocks
    DevScratch() { (void)hipGetDevice(&device); }
    DevScratch(const DevScratch &) = delete;
    DevScratch &operator=(const DevScratch &) = delete;
    ~DevScratch() {
        // The blocks go back to a process-wide cache (not through hipFree, which would synchronise): on an error
        // return kernels may still be running on them, and another thread's handle could be handed that memory.
        if (!drained && !blks.empty()) (void)hipDeviceSynchronize();
        for (const Blk &b : blks) block_free(device, b.p, b.bytes);
    }
    template <class T>
    int alloc(T **p, size_t n) {
        Blk b;
        const int rc = block_alloc(&b.p, (n ? n : 1) * sizeof(T), &b.bytes);
        if (rc == MISSLAP_OK) {
            *p = static_cast<T *>(b.p);
            blks.push_back(b);
        }
        return rc;
    }
};

// Several device arrays carved from ONE hipMalloc (256-byte aligned): hipMalloc / hipFree cost tens of microseconds
// each and a handle holds some thirty arrays -- allocated one by one they are a fifth of the time it takes to set a
// 40 M-edge problem up.  `want` registers an array, `commit` allocates and hands the pointers out; the block is
// released as a whole (by the handle: misslap_solver::blocks, or by a DevScratch).
struct DevBlock {
    struct Item {
        void **target;
        size_t bytes;
    };
    std::vector<Item> items;
    template <class T>
    void want(T **p, size_t n) {
        items.push_back({reinterpret_cast<void **>(p), (n ? n : 1) * sizeof(T)});
    }
    int commit(Blk *out) {
        size_t total = 0;
        for (const Item &it : items) total += (it.bytes + 255) & ~(size_t)255;
        const int rc = block_alloc(&out->p, total ? total : 256, &out->bytes);
        if (rc) return rc;
        char *base = static_cast<char *>(out->p);
        size_t off = 0;
        for (const Item &it : items) {
            *it.target = base + off;
            off += (it.bytes + 255) & ~(size_t)255;
        }
        items.clear();
        return MISSLAP_OK;
    }
};

// Host-side resources are kept across handles: creating a stream (a hardware queue) takes several milliseconds -- more
// than everything else a handle's setup does --, the pinned mirror of the control block and the two status events
// another tenth of a millisecond.  A destroyed handle parks its idle bundle here; the next handle on that device takes it.
struct HostRes {
    hipStream_t stream = nullptr;
    Ctl *h_ctl = nullptr;  // pinned, 3 blocks: the mirror and the two trailing status copies
    hipEvent_t ev[2] = {nullptr, nullptr};
};
struct HostResPool {
    std::mutex m;
    std::vector<std::pair<int, HostRes>> idle;
    static constexpr size_t kMaxIdle = 8;
    bool take(int device, HostRes *out) {
        std::lock_guard<std::mutex> g(m);
        for (size_t k = 0; k < idle.size(); ++k)
            if (idle[k].first == device) {
                *out = idle[k].second;
                idle.erase(idle.begin() + (long)k);
                return true;
            }
        return false;
    }
    bool park(int device, const HostRes &r) {
        std::lock_guard<std::mutex> g(m);
        if (idle.size() >= kMaxIdle) return false;
        idle.emplace_back(device, r);
        return true;
    }
};
HostResPool &host_pool() {
    static HostResPool *pool = new HostResPool();  // never destroyed: the HIP runtime may be gone at static teardown
    return *pool;
}

// ... and so are small device blocks: hipMalloc + hipFree of a handle's four blocks cost a quarter of a millisecond,
// which is what a 20 x 20 problem takes to SOLVE.  Freed blocks of at most kMaxEach bytes wait here (at most
// kMaxEntries, kMaxHeld bytes in total) for a request they fit within a factor of two.
struct BlockCache {
    struct Ent {
        int device;
        size_t bytes;
        void *p;
    };
    std::mutex m;
    std::vector<Ent> idle;
    size_t held = 0;
    // limits (misslap_set_cache_limits; MISSLAP_BLOCK_CACHE_MB in the environment sets the first two at start-up).  The
    // defaults -- 4 GB of a 288 GB device, blocks of up to 1 GB -- hold the blocks of one or two problems of the
    // BASELINE sizes (C3: 0.9 GB per handle): hipMalloc + hipFree of those cost a millisecond per create / destroy pair
    // (C4: setup 4.0 -> 3.0 ms), and hipFree waits for every stream of the device, i.e. for other solves' kernels.  An
    // application that solves many large problems at a time raises them further
    size_t kMaxHeld = (size_t)4 << 30, kMaxEach = (size_t)1 << 30, kMaxEntries = 64;
    bool explicit_limits = false;  // set by the caller (misslap_set_cache_limits / MISSLAP_BLOCK_CACHE_MB)
    bool sized = false;            // the default total has been bounded by the device's memory (first block parked)
    BlockCache() {
        if (const char *e = std::getenv("MISSLAP_BLOCK_CACHE_MB")) {
            const long long mb = std::atoll(e);
            if (mb >= 0) {
                kMaxHeld = (size_t)mb << 20;
                kMaxEach = kMaxHeld;
                kMaxEntries = 4096;
                explicit_limits = true;
            }
        }
    }
    // the DEFAULT total never exceeds 1 / 64 of the device's memory (4 GB of an MI355X's 288 GB; 1 GB of a 64 GB part)
    void size_default(size_t device_bytes) {
        std::lock_guard<std::mutex> g(m);
        if (explicit_limits || sized) return;
        sized = true;
        kMaxHeld = std::min(kMaxHeld, device_bytes / 64);
        kMaxEach = std::min(kMaxEach, kMaxHeld / 4);
    }
    bool needs_sizing() {
        std::lock_guard<std::mutex> g(m);
        return !sized && !explicit_limits;
    }
    void *take(int device, size_t bytes, size_t *got) {
        std::lock_guard<std::mutex> g(m);
        size_t best = idle.size();
        for (size_t k = 0; k < idle.size(); ++k)
            if (idle[k].device == device && idle[k].bytes >= bytes && idle[k].bytes <= 2 * bytes + 4096 &&
                (best == idle.size() || idle[k].bytes < idle[best].bytes))
                best = k;
        if (best == idle.size()) return nullptr;
        void *p = idle[best].p;
        *got = idle[best].bytes;
        held -= idle[best].bytes;
        idle.erase(idle.begin() + (long)best);
        return p;
    }
    bool give(int device, void *p, size_t bytes) {
        std::lock_guard<std::mutex> g(m);  // (the limits are read under the lock: size_default / misslap_set_cache_limits write them)
        if (bytes == 0 || bytes > kMaxEach) return false;
        if (idle.size() >= kMaxEntries || held + bytes > kMaxHeld) return false;
        idle.push_back({device, bytes, p});
        held += bytes;
        return true;
    }
};
BlockCache &block_cache() {
    static BlockCache *c = new BlockCache();
    return *c;
}
// a device block of at least `bytes` on the current device, from the cache if one fits; *got = its real size
// MISSLAP_DEBUG_POISON=<byte> (debugging): every block handed out is filled with that byte first (0xFF: NaN / -1 patterns),
// so that a read of device memory nobody has written shows up whatever the process has run before
int block_alloc(void **p, size_t bytes, size_t *got) {
    int dev = 0;
    HIP_TRY(hipGetDevice(&dev));
    static const int poison = [] {
        const char *e = std::getenv("MISSLAP_DEBUG_POISON");
        return e ? (int)std::strtol(e, nullptr, 0) : -1;
    }();
    *p = block_cache().take(dev, bytes, got);
    if (!*p) {
        hipError_t e = hipMalloc(p, bytes);
        if (e == hipErrorOutOfMemory) {
            // the parked blocks are memory no other allocator of the process can see as free: give them back and retry once
            (void)hipGetLastError();
            std::vector<BlockCache::Ent> take;
            {
                BlockCache &bc = block_cache();
                std::lock_guard<std::mutex> g(bc.m);
                take.swap(bc.idle);
                bc.held = 0;
            }
            for (const BlockCache::Ent &en : take) {
                if (hipSetDevice(en.device) == hipSuccess) {
                    (void)hipDeviceSynchronize();  // nothing may still be running on a parked block
                    (void)hipFree(en.p);
                }
            }
            (void)hipSetDevice(dev);
            e = hipMalloc(p, bytes);
        }
        HIP_TRY(e);
        *got = bytes;
    }
    if (poison >= 0) {
        HIP_TRY(hipMemset(*p, poison & 0xff, *got));
        HIP_TRY(hipDeviceSynchronize());
    }
    return MISSLAP_OK;
}
void block_free(int device, void *p, size_t bytes) {
    BlockCache &bc = block_cache();
    if (p && bc.needs_sizing()) {
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) == hipSuccess && total_b > 0) bc.size_default(total_b);
    }
    if (p && !bc.give(device, p, bytes)) (void)hipFree(p);
}
}  // namespace
