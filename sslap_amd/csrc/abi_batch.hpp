// abi_batch.hpp -- C ABI: misslap_solve_batch, many independent problems in lockstep (the mechanism: host_batch.hpp).
// (part of the single translation unit misslap.hip; included in the order given there)
#pragma once

namespace {

constexpr size_t kBatchFiberStack = 512 * 1024;

// what a fiber runs: the ordinary solve of its handle (the loop of host_comm.hpp without a communicator, then
// misslap_finish), with every launch recorded instead of issued
void batch_fiber_main(unsigned lo, unsigned hi) {
    BatchFiber *f = reinterpret_cast<BatchFiber *>(((uintptr_t)hi << 32) | (uintptr_t)lo);
    try {  // (an exception must not unwind out of a makecontext entry: e.g. bad_alloc of the list of recorded calls)
        f->rc = misslap_solve_sharded(f->h, nullptr, f->sol, f->meta);
        if (f->rc) f->err = g_err;
    } catch (const std::exception &e) {
        f->rc = MISSLAP_ERR_STATE;
        f->err = std::string("exception inside a batched solve: ") + e.what();
    } catch (...) {
        f->rc = MISSLAP_ERR_STATE;
        f->err = "exception inside a batched solve";
    }
    f->state = BatchFiber::kDone;
    swapcontext(&f->ctx, &f->grp->sched);  // (never resumed)
}

// issue what the fibers of the group have recorded: the heads of all pending lists at a time, equal kernels as one
// launch.  A held head (the start of a tail sequence) stops its list unless `release` says that everybody is there.
// Returns the first HIP error an issued launch / copy / fill raised (the fibers checked theirs at RECORD time, when
// nothing had been issued yet): the caller aborts the group with it instead of running into a poll timeout.
hipError_t batch_flush(BatchGroup &g, bool release) {
    hipError_t first = hipSuccess;
    auto note = [&]() {
        const hipError_t e = hipGetLastError();
        if (e != hipSuccess && first == hipSuccess) first = e;
    };
    std::vector<BatchCall *> heads, same;
    for (;;) {
        heads.clear();
        for (auto &fp : g.fibers) {
            BatchFiber &f = *fp;
            if (f.at >= f.pending.size()) continue;
            BatchCall &c = f.pending[f.at];
            if (c.hold && !release) continue;
            c.hold = false;
            heads.push_back(&c);
            f.at += 1;
        }
        if (heads.empty()) break;
        g.launches_merged += (long long)heads.size();
        std::vector<bool> done(heads.size(), false);
        for (size_t a = 0; a < heads.size(); ++a) {
            if (done[a]) continue;
            BatchCall *c = heads[a];
            if (!c->merge) {
                c->single(g.stream);
                note();
                g.launches_issued += 1;
                continue;
            }
            same.clear();
            for (size_t b = a; b < heads.size(); ++b)
                if (!done[b] && heads[b]->merge == c->merge && heads[b]->block.x == c->block.x) {
                    same.push_back(heads[b]);
                    done[b] = true;
                }
            c->merge(g, same.data(), (int)same.size());
            note();
            g.launches_issued += 1;
        }
    }
    for (auto &fp : g.fibers)
        if (fp->at == fp->pending.size()) {
            fp->pending.clear();
            fp->at = 0;
        }
    return first;
}

// one group: its handles' solves on fibers of this thread, one stream
int batch_run_group(BatchGroup &g, int device) {
    if (hipSetDevice(device) != hipSuccess) return fail(MISSLAP_ERR_HIP, "hipSetDevice failed");
    if (hipStreamCreateWithFlags(&g.stream, hipStreamNonBlocking) != hipSuccess) return fail(MISSLAP_ERR_HIP, "hipStreamCreate failed");
    for (auto &fp : g.fibers) {
        BatchFiber &f = *fp;
        f.grp = &g;
        if (!f.stack.alloc(kBatchFiberStack)) {
            (void)hipStreamDestroy(g.stream);
            g.stream = nullptr;
            for (auto &fq : g.fibers)  // (the ones already set up)
                if (fq->h->batch == fq.get()) {
                    fq->h->batch = nullptr;
                    fq->h->stream = fq->own_stream;
                    fq->h->own_stream = fq->own_own_stream;
                }
            return fail(MISSLAP_ERR_STATE, "no memory for the stack of a batched solve (%zu bytes)", kBatchFiberStack);
        }
        getcontext(&f.ctx);
        f.ctx.uc_stack.ss_sp = f.stack.base();
        f.ctx.uc_stack.ss_size = f.stack.size();
        f.ctx.uc_link = &g.sched;
        const uintptr_t p = reinterpret_cast<uintptr_t>(&f);
        makecontext(&f.ctx, reinterpret_cast<void (*)()>(batch_fiber_main), 2, (unsigned)(p & 0xffffffffu), (unsigned)(p >> 32));
        f.own_stream = f.h->stream;
        f.own_own_stream = f.h->own_stream;
        f.h->stream = g.stream;  // (everything of the handle's earlier life on its own stream has completed: create drains it)
        f.h->batch = &f;
        f.state = BatchFiber::kRunnable;
    }
    // How long the problems that HAVE their next launches wait for the ones still polling for a status (passes over the
    // fibers, a few microseconds each): launches recorded together merge, and the stream is in order anyway -- what a
    // polling problem waits for is queued in front of whatever the others would add
    constexpr int kPatience = 64;
    const double t_end = now_ms() + 3600e3;
    int rc = MISSLAP_OK, idle = 0;
    double t_mark = now_ms();
    auto lap = [&](double &acc) {
        const double t = now_ms();
        acc += t - t_mark;
        t_mark = t;
    };
    for (;;) {
        bool active = false, progressed = false;
        for (auto &fp : g.fibers) {
            BatchFiber &f = *fp;
            if (f.state == BatchFiber::kDone) continue;
            active = true;
            if (f.state == BatchFiber::kWantsSync) continue;
            const size_t before = f.pending.size();
            swapcontext(&g.sched, &f.ctx);  // runs the fiber until it needs the device (or ends)
            progressed = progressed || f.pending.size() != before || f.state != BatchFiber::kPolling;
        }
        lap(progressed ? g.ms_fibers : g.ms_wait);
        if (!active) break;
        int n_ready = 0, n_hold = 0, n_wait = 0, n_sync = 0;
        for (auto &fp : g.fibers) {
            const BatchFiber &f = *fp;
            if (f.state == BatchFiber::kDone && f.at >= f.pending.size()) continue;
            if (f.at < f.pending.size()) (f.pending[f.at].hold ? n_hold : n_ready) += 1;
            else if (f.state == BatchFiber::kWantsSync) n_sync += 1;
            else n_wait += 1;
        }
        const bool release = n_hold > 0 && n_ready == 0 && n_wait == 0 && n_sync == 0;
        if (release || (n_ready > 0 && (n_wait == 0 || ++idle > kPatience))) {
            const hipError_t fe = batch_flush(g, release);
            if (fe != hipSuccess) {
                rc = fail(MISSLAP_ERR_HIP, "a launch of a batched solve failed: %s", hipGetErrorString(fe));
                break;
            }
            lap(g.ms_flush);
            idle = 0;
            continue;  // (fibers whose calls just went out may have asked for a drained stream: next pass)
        }
        if (n_sync > 0 && n_ready == 0) {
            if (hipStreamSynchronize(g.stream) != hipSuccess) {
                rc = fail(MISSLAP_ERR_HIP, "hipStreamSynchronize failed inside a batched solve: %s", hipGetErrorString(hipGetLastError()));
                break;
            }
            for (auto &fp : g.fibers)
                if (fp->state == BatchFiber::kWantsSync && fp->at >= fp->pending.size()) fp->state = BatchFiber::kRunnable;
            lap(g.ms_wait);
            continue;
        }
        __builtin_ia32_pause();  // everybody is waiting for a status word (or for the patience above)
        if (now_ms() > t_end) {
            rc = fail(MISSLAP_ERR_STATE, "batched solve timed out");
            break;
        }
    }
    (void)hipStreamSynchronize(g.stream);
    for (auto &fp : g.fibers) {
        BatchFiber &f = *fp;
        f.h->batch = nullptr;
        f.h->stream = f.own_stream;
        f.h->own_stream = f.own_own_stream;
        if (!rc && f.state != BatchFiber::kDone) rc = fail(MISSLAP_ERR_STATE, "a batched solve did not finish");
        if (!rc && f.rc) rc = fail(f.rc, "%s", f.err.c_str());
    }
    (void)hipStreamDestroy(g.stream);
    g.stream = nullptr;
    if (g.h_ring) (void)hipHostFree(g.h_ring);
    if (g.d_ring) (void)hipFree(g.d_ring);
    g.h_ring = g.d_ring = nullptr;
    return rc;
}

}  // namespace

MISSLAP_API int misslap_solve_batch(misslap_solver *const *handles, int32_t n, int32_t *const *person_to_object_out,
                                    misslap_meta *const *meta_out, int32_t group_size, misslap_batch_info *info) {
    if (!handles || n <= 0) return fail(MISSLAP_ERR_INVALID, "bad argument");
    if (group_size <= 0) group_size = 12;  // (what one kernel-argument block carries of the widest launch, host_batch.hpp)
    if (!handles[0]) return fail(MISSLAP_ERR_INVALID, "null handle in the batch");
    const int device = handles[0]->device;
    for (int k = 0; k < n; ++k) {
        const misslap_solver *h = handles[k];
        if (!h) return fail(MISSLAP_ERR_INVALID, "null handle in the batch");
        if (h->device != device) return fail(MISSLAP_ERR_INVALID, "the handles of a batch live on one device");
        if (h->world != 1) return fail(MISSLAP_ERR_INVALID, "sharded handles cannot be batched");
        if (h->profile) return fail(MISSLAP_ERR_INVALID, "profiled handles (options.profile) cannot be batched");
        if (h->live_off) return fail(MISSLAP_ERR_STATE, "a batched solve needs the live status words (MISSLAP_LIVE_STATUS=0 is set, or they failed)");
        // (merged launches run on the largest of their grids; what is sized by the number of PERSONS -- chunk counters,
        // per-workgroup result slots -- is indexed by workgroup, so that number is the same for every problem of a batch;
        // everything sized by the objects is walked with grid-stride loops)
        if (h->n_rows != handles[0]->n_rows)
            return fail(MISSLAP_ERR_INVALID, "the problems of a batch have the same number of persons (%d here, %d in handle %d): merged "
                        "launches run on the largest grid", handles[0]->n_rows, h->n_rows, k);
        for (int j = 0; j < k; ++j)
            if (handles[j] == h) return fail(MISSLAP_ERR_INVALID, "handle %d appears twice in the batch", k);
        // (an array of POINTERS: the caller's misslap_meta may be shorter or longer than this build's, struct_size says)
        if (meta_out && meta_out[k] && h->abi >= 2 &&
            (meta_out[k]->struct_size < (int32_t)offsetof(misslap_meta, edges_scanned) || meta_out[k]->struct_size > 65536))
            return fail(MISSLAP_ERR_INVALID, "meta_out[%d]->struct_size = %d: set it to sizeof(misslap_meta) before the call", k, meta_out[k]->struct_size);
    }
    const double t0 = now_ms();
    const int n_groups = (n + group_size - 1) / group_size;
    std::vector<std::unique_ptr<BatchGroup>> groups;
    for (int gi = 0; gi < n_groups; ++gi) {
        groups.emplace_back(new BatchGroup());
        for (int k = gi * group_size; k < std::min(n, (gi + 1) * group_size); ++k) {
            std::unique_ptr<BatchFiber> f(new BatchFiber());
            f->h = handles[k];
            f->sol = person_to_object_out ? person_to_object_out[k] : nullptr;
            f->meta = meta_out ? meta_out[k] : nullptr;
            groups.back()->fibers.push_back(std::move(f));
        }
    }
    // one host thread and one stream per group: the groups overlap on the GPU (their tail kernels are one workgroup per
    // problem), the problems of a group share every launch
    std::vector<int> rcs((size_t)n_groups, MISSLAP_OK);
    std::vector<std::string> errs((size_t)n_groups);
    if (n_groups == 1) {
        rcs[0] = batch_run_group(*groups[0], device);
        if (rcs[0]) errs[0] = g_err;
    } else {
        std::vector<std::thread> th;
        for (int gi = 0; gi < n_groups; ++gi)
            th.emplace_back([&, gi] {
                rcs[(size_t)gi] = batch_run_group(*groups[(size_t)gi], device);
                if (rcs[(size_t)gi]) errs[(size_t)gi] = g_err;  // (thread-local: carried back by hand)
            });
        for (auto &t : th) t.join();
    }
    if (info) {
        info->groups = n_groups;
        info->calls_recorded = 0;
        info->launches_issued = 0;
        for (auto &g : groups) {
            info->calls_recorded += g->launches_merged;
            info->launches_issued += g->launches_issued;
        }
        info->wall_ms = now_ms() - t0;
        info->host_ms_fibers = info->host_ms_flush = info->host_ms_wait = 0.0;
        for (auto &g : groups) {
            info->host_ms_fibers += g->ms_fibers;
            info->host_ms_flush += g->ms_flush;
            info->host_ms_wait += g->ms_wait;
        }
    }
    for (int gi = 0; gi < n_groups; ++gi)
        if (rcs[(size_t)gi]) return fail(rcs[(size_t)gi], "%s", errs[(size_t)gi].c_str());
    return MISSLAP_OK;
}
