// host_rounds.hpp -- the launch sequence of a round on one HIP stream: status reads (live words or copy), bid / tie-break / apply / tail launches, the pass over all rows (eCE, objective, validity).
// (part of the single translation unit misslap.hip; included in the order given there)
#pragma once

namespace {
// The start of an eps-phase (create: the first; misslap_phase_end: every later one): are the candidate lines still exact
// at this phase's eps?  (fp32 eps promoted to double exactly as the bid does, auction_.pyx:360.)
void begin_phase(misslap_solver *h) {
    if (h->cand != nullptr && !h->lines_dropped && (double)h->eps < h->lines_safe_eps) h->lines_dropped = true;
    h->phases_run += 1;
    h->phases_with_lines += h->lines_live() ? 1 : 0;
}

RoundArgs round_args(misslap_solver *h) {
    RoundArgs a;
    a.ctl = h->ctl;
    a.row_ptr = h->row_ptr;
    a.price = h->price;
    a.rec = h->rec;
    a.p2o = h->p2o;
    a.o2p = h->o2p;
    a.U = h->U;
    a.bid_key = h->bid_key;
    a.bid_obj = h->bid_obj;
    a.bid_rec = h->bid_rec;
    a.best_key = h->best_key;
    a.best_pos = h->best_pos;
    a.cnt = h->cnt;
    a.hole_list = h->hole_list;
    a.mover_list = h->mover_list;
    a.launch_edges = h->profile ? h->launch_edges : nullptr;
    a.n_rows = h->n_rows;
    a.n_cols = h->n_cols;
    a.thr = h->thr;
    a.rank = h->rank;
    a.world = h->world;
    a.shard_min_K = h->world > 1 ? h->shard_min_K : 0;
    a.eps = h->eps;
    a.launch_idx = 0;
    a.wg_stats = h->wg_stats;
    a.need_list = h->need_list;
    a.live = nullptr;
    a.ticket = 0;
    a.gather_max_K = h->tiled_ok ? h->tiled_min_K : 0;
    a.cand = h->lines_live() ? h->cand : nullptr;
    a.cand64 = h->lines_live() ? h->cand64 : nullptr;
    a.cand_build_max_K = h->cand_build_max_K;
    // (small rounds leave the rebuild of a spent line to the maintenance pass: same box, 2048 vs 0: C3 400.1 vs 401.4 ms,
    // C2 132.0 vs 132.5, C1 9.69 vs 9.85; without that pass nobody else rebuilds)
    a.cand_build_min_K = (h->thr > 0 && h->line_maintenance) ? kRoundSmallMax : 0;
    a.cand_refresh_min = h->cand_refresh_min;
    a.price32 = nullptr;  // (set by launch_bid for the launches that scan through the filter)
    a.pmax_bits = h->pmax_bits;
    a.cmax = h->cmax32;
    return a;
}

int blocks_for(long long items, int per_block) {
    long long b = (items + per_block - 1) / per_block;
    if (b < 1) b = 1;
    if (b > kMaxGridBlocks) b = kMaxGridBlocks;
    return (int)b;
}

ProfRec *prof_next(misslap_solver *h, int kind) {
    if (h->prof_used == h->prof.size()) {
        ProfRec r{};
        if (hipEventCreate(&r.start) != hipSuccess || hipEventCreate(&r.stop) != hipSuccess) return nullptr;
        h->prof.push_back(r);
    }
    ProfRec *r = &h->prof[h->prof_used++];
    r->kind = kind;
    r->fullscan = 0;
    r->launch_idx = -1;
    return r;
}

// (A status read is a stream drain: ~20 us.  A solve of a small problem is a few hundred rounds of ~1 us inside one
// tail launch per eps-phase and was spending most of its time in the five reads per phase; with the mirror reused
// while nothing has been enqueued since the last read, two remain.)
// (behind every status read) the rounds of the tail launches enqueued before it
void count_tail_rounds(misslap_solver *h) {
    if (h->tail_nits0 < 0) return;
    h->tail_rounds_host += h->h_ctl->nits - h->tail_nits0;
    h->tail_nits0 = -1;
}

int read_ctl(misslap_solver *h) {
    if (h->ctl_fresh == 2) {
        if (h->h_ctl->err)
            return fail(MISSLAP_ERR_STATE, "device-side invariant violated (error bits 0x%x)", h->h_ctl->err);
        return MISSLAP_OK;
    }
    HIP_TRY(stream_memcpy(h, h->h_ctl, h->ctl, sizeof(Ctl), hipMemcpyDeviceToHost));
    HIP_TRY(stream_sync(h));
    h->ctl_fresh = 2;
    h->K_ub = h->h_ctl->K;
    h->K_exact = true;
    count_tail_rounds(h);
    if (h->h_ctl->err)
        return fail(MISSLAP_ERR_STATE, "device-side invariant violated (error bits 0x%x)", h->h_ctl->err);
    return MISSLAP_OK;
}

// Wait for the status a round-closing launch posts (post_live_status): exact = the ticket `want` itself, otherwise any
// ticket at or behind it.  Returns false on a timeout (the caller falls back to a copy + drain and stops using the words).
bool live_poll(misslap_solver *h, unsigned want, bool exact, int *K, int *err, long long *nits) {
    volatile unsigned long long *w = h->live;
    double t_end = now_ms() + 20000.0;
    for (unsigned spins = 0;; ++spins) {
        const unsigned long long a = w[0], b = w[1], c = w[2], d = w[3];
        const unsigned t = (unsigned)(a >> 32);
        if ((unsigned)(b >> 32) == t && (unsigned)(c >> 32) == t && (unsigned)(d >> 32) == t && w[0] == a &&
            (exact ? t == want : (int)(t - want) >= 0)) {
            *K = (int)(unsigned)(a & 0xffffffffull);
            *err = (int)(unsigned)(b & 0xffffffffull);
            *nits = (long long)((c & 0xffffffffull) | ((d & 0xffffffffull) << 32));
            return true;
        }
        if (h->batch) {  // (a fiber of a batch: let the other problems run; the scheduler comes back to this poll)
            batch_yield(h, BatchFiber::kPolling);
            // The deadline bounds a DEVICE stall.  While calls this fiber has recorded are still unissued -- the head of
            // its tail sequence is held until every problem of the group has reached its own -- the wait is the
            // scheduler's, however long the other problems' big rounds take: the clock starts when the last call is out.
            if (h->batch->at < h->batch->pending.size()) t_end = now_ms() + 20000.0;
            else if ((spins & 255) == 255 && now_ms() > t_end) return false;
        } else if (spins < 4000) {
            __builtin_ia32_pause();
        } else {
            std::this_thread::yield();  // (a tail kernel runs for milliseconds: do not burn a core another solve needs)
            if ((spins & 1023) == 0 && now_ms() > t_end) return false;
        }
    }
}
// K / nits / error bits of everything enqueued so far, into the mirror's fields: from the live words where the last
// thing enqueued that changes them was a ticketed launch, by a full read otherwise.
// the live words cover everything enqueued: if the last launch that changed K / nits carried no ticket, one that only
// posts the status is enqueued behind it
void ensure_posted(misslap_solver *h) {
    if (h->live_valid || h->live_off) return;
    MISSLAP_LAUNCH(h, k_post_status, (F_k_post_status), 1, dim3(1), dim3(1), (const Ctl *)h->ctl, h->live_dev, ++h->ticket);
    h->live_valid = true;
}
int read_status(misslap_solver *h) {
    if (h->ctl_fresh == 2 || h->live_off) return read_ctl(h);
    if (h->ctl_fresh == 1) {  // (K, nits and the error bits of the mirror are current)
        if (h->h_ctl->err) return fail(MISSLAP_ERR_STATE, "device-side invariant violated (error bits 0x%x)", h->h_ctl->err);
        return MISSLAP_OK;
    }
    ensure_posted(h);
    int K = 0, err = 0;
    long long nits = 0;
    if (!live_poll(h, h->ticket, true, &K, &err, &nits)) {
        h->live_off = true;
        return read_ctl(h);
    }
    h->h_ctl->K = K;
    h->h_ctl->nits = nits;
    h->h_ctl->err = err;
    h->K_ub = K;
    h->K_exact = true;
    count_tail_rounds(h);
    if (err) return fail(MISSLAP_ERR_STATE, "device-side invariant violated (error bits 0x%x)", err);
    return MISSLAP_OK;
}

// Status of the round loop WITHOUT draining the stream: a copy of the control block is enqueued behind a batch of
// rounds and read while the next batch runs.  K never grows inside an eps-phase, so a status that is one batch old
// is still an upper bound for the launch grids, and every round kernel is a no-op once the round is not live: a
// batch enqueued on a stale "go on" costs its launches and nothing else.
int status_enqueue(misslap_solver *h, int slot) {
    h->slot_live[slot] = !h->live_off;
    if (h->slot_live[slot]) {  // no copy: the closing kernel of the batch's last round has posted, or k_post_status does
        ensure_posted(h);
        h->slot_ticket[slot] = h->ticket;
        return MISSLAP_OK;
    }
    h->ctl_fresh = false;
    if (h->batch) return fail(MISSLAP_ERR_STATE, "a batched solve needs the live status words");
    HIP_TRY(hipMemcpyAsync(&h->h_stat[slot], h->ctl, sizeof(Ctl), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipEventRecord(h->stat_ev[slot], h->stream));
    return MISSLAP_OK;
}
int status_wait(misslap_solver *h, int slot) {
    if (h->slot_live[slot]) {
        int K = 0, err = 0;
        long long nits = 0;
        if (live_poll(h, h->slot_ticket[slot], false, &K, &err, &nits)) {
            h->h_stat[slot].K = K;
            h->h_stat[slot].nits = nits;
            h->h_stat[slot].err = err;
            h->K_ub = K;
            h->K_exact = false;
            if (err) return fail(MISSLAP_ERR_STATE, "device-side invariant violated (error bits 0x%x)", err);
            return MISSLAP_OK;
        }
        h->live_off = true;  // timed out: drain the stream and read the control block
        int rc = read_ctl(h);
        h->h_stat[slot] = *h->h_ctl;
        h->K_exact = false;
        return rc;
    }
    HIP_TRY(hipEventSynchronize(h->stat_ev[slot]));
    const Ctl &c = h->h_stat[slot];
    h->K_ub = c.K;
    h->K_exact = false;  // rounds have been enqueued behind this copy
    if (c.err) return fail(MISSLAP_ERR_STATE, "device-side invariant violated (error bits 0x%x)", c.err);
    return MISSLAP_OK;
}

// Does this engine launch walk the column tiles backwards?  The first scan of an eps-phase (K = N: tail kernels lie in
// front of it, the cache holds nothing of the copy) walks forwards, every further launch of the phase in the direction
// opposite to the one before it.  Column-keyed formats 0 / 1 in the production shapes.
bool walk_backwards(misslap_solver *h, const int *shp) {
    const bool can = MISSLAP_TILED_KEYCOL_ON && h->tiled_fmt <= 1 && shp[7] == 1 &&
                     (h->tiled_shape == 0 || h->tiled_shape == 8 || h->tiled_shape == 9);
    if (h->phase_fresh) h->walk_rev_next = false;
    const bool rev = can && h->walk_rev_next;
    h->walk_rev_next = !h->walk_rev_next;
    return rev;
}

int launch_bid_tiled(misslap_solver *h) {
    h->ctl_fresh = false;
    RoundArgs a = round_args(h);
    // K_ub is only an upper bound unless the host has just read K: the device decides sharded / replicated from
    // the exact K, so the smaller sharded grid is used only when the host knows the same K
    const bool sharded = h->world > 1 && h->K_exact && h->K_ub >= h->shard_min_K;
    const long long share = sharded ? ((long long)h->K_ub + h->world - 1) / h->world : h->K_ub;
    const int *shp = kTiledShapes[h->tiled_shape];
    const int groups = (shp[0] - 64 * shp[5]) / shp[6];  // lane groups; loader wavefronts own no persons
    const int per_wg_max = groups * shp[1];
    const int cs = shp[7];  // column split: `cs` workgroups share a slice of bidders, each with 1 / cs of the tiles
    long long grid = (share + per_wg_max - 1) / per_wg_max;
    const long long resident = h->n_cus / cs;  // one workgroup per CU: its two price tiles take the whole LDS
    const long long spread = std::min<long long>(resident, (share + groups - 1) / groups);
    if (grid < spread) grid = spread;
    grid *= cs;
    TiledArgs ta{h->tiled, h->tcol, h->seg4, h->T, h->tiled_min_K, h->n_tiled,
                 nullptr, nullptr, h->ovf_ptr, h->ovf_q, h->ovf_cap, h->part_vw, h->part_g, h->n_rows, h->split_cnt, FinalOut{}};
    // A partial round whose K the host knows: bidders in person order (kernels_tiled.hpp, k_order_*).  Scratch that
    // is idle during a bid phase: the compaction lists (the tie-break reads order_pos before they are rewritten),
    // the chunk counters, the objective's match counters.
    // (K < N in every round of a phase but the first: every winner of the first round takes an unowned object.  The
    // ordering kernels and the scan take K from the device, so the host need not know it exactly.)
    h->round_ordered = !h->phase_fresh;
    if (h->round_ordered) {
        int *pos_of = h->nmatch, *order_person = h->hole_list, *order_pos = h->mover_list, *sums = h->cnt;
        const int nchunks = (h->n_rows + kScanChunk - 1) / kScanChunk;
        // (each returns at once when the scan itself will: a round enqueued on a stale upper bound of K)
        MISSLAP_LAUNCH(h, k_order_prepare, (F_k_order_prepare), 1024, dim3(std::max(nchunks, blocks_for(h->K_ub, 1024))), dim3(1024),
                       (const Ctl *)h->ctl, (const int *)h->U, pos_of, (const int *)h->p2o, h->n_rows, nchunks, sums, h->thr, h->tiled_min_K);
        MISSLAP_LAUNCH(h, k_order_scatter, (F_k_order_scatter), 1024, dim3(nchunks), dim3(1024), (const Ctl *)h->ctl, (const int *)h->p2o,
                       h->n_rows, (const int *)sums, (const int *)pos_of, order_person, order_pos, h->thr, h->tiled_min_K);
        ta.order_person = order_person;
        ta.order_pos = order_pos;
    }
    if (grid > h->wg_stats_slots) return fail(MISSLAP_ERR_STATE, "scan grid %lld exceeds the statistics slots (%d)", grid, h->wg_stats_slots);
    ProfRec *pr = nullptr;
    if (h->profile) {
        if (h->launch_idx >= h->launch_edges_cap)
            return fail(MISSLAP_ERR_STATE, "profile buffer exhausted (%d bid launches)", h->launch_idx);
        pr = prof_next(h, 2);
        if (!pr) return fail(MISSLAP_ERR_HIP, "hipEventCreate failed");
        pr->fullscan = h->phase_fresh;  // K == N; with several ranks: this rank's share of the full scan
        pr->launch_idx = a.launch_idx = h->launch_idx++;
    }
    const size_t lds = tiled_lds_bytes(shp[4]);
    const dim3 g((unsigned)grid);
    if (walk_backwards(h, shp)) {
        // the launches of an eps-phase alternate their direction over the column tiles (kernels_tiled.hpp, kRev): this one
        // reads first what the one before it touched last
        const int key = h->tiled_fmt * 100 + shp[6];
        switch (key) {
#define X(FMT, GL) \
    case FMT * 100 + GL:                                                                                               \
        if (h->batch) MISSLAP_LAUNCH_PLAIN(h, (MISSLAP_BID_KERNEL_REV(GL, FMT)), g, dim3(1024), lds, a, ta);                  \
        else MISSLAP_LAUNCH_TIMED(pr, (MISSLAP_BID_KERNEL_REV(GL, FMT)), g, dim3(1024), (unsigned)lds, h->stream, a, ta);     \
        break;
            X(0, 4) X(0, 8) X(0, 16) X(1, 4) X(1, 8) X(1, 16)
#undef X
            default: return fail(MISSLAP_ERR_STATE, "no backward instance for format %d with %d lanes per person", h->tiled_fmt, shp[6]);
        }
    } else if (h->tiled_fmt == 0) {
        switch (h->tiled_shape) {
#define X(I, TH, R, B, D, TC, LD, GL, CS) \
    case I:                                                                                                            \
        if (h->batch) MISSLAP_LAUNCH_PLAIN(h, (k_bid_tiled<TH, R, B, D, TC, LD, 0, GL, CS>), g, dim3(TH), lds, a, ta);        \
        else MISSLAP_LAUNCH_TIMED(pr, (k_bid_tiled<TH, R, B, D, TC, LD, 0, GL, CS>), g, dim3(TH), (unsigned)lds, h->stream, a, ta); \
        break;
            MISSLAP_FOR_TILED_SHAPES(X)
#undef X
            default: return fail(MISSLAP_ERR_STATE, "bad tiled shape");
        }
    } else {  // formats 1..3 (fp64 values / unsorted rows): the three production shapes, 4 / 8 / 16 lanes per person
        const int key = h->tiled_fmt * 100 + shp[6];
        switch (key) {
#define X(FMT, GL) \
    case FMT * 100 + GL:                                                                                               \
        if (h->batch) MISSLAP_LAUNCH_PLAIN(h, (MISSLAP_BID_KERNEL_FMT(GL, FMT)), g, dim3(1024), lds, a, ta);                  \
        else MISSLAP_LAUNCH_TIMED(pr, (MISSLAP_BID_KERNEL_FMT(GL, FMT)), g, dim3(1024), (unsigned)lds, h->stream, a, ta);     \
        break;
            MISSLAP_FOR_FMT_LANES(X)
#undef X
            default: return fail(MISSLAP_ERR_STATE, "no full-scan instance for format %d with %d lanes per person", h->tiled_fmt, shp[6]);
        }
    }
    if (pr) {  // (the round's k_tiebreak adds the workgroups' counts up: no launch of its own inside a timed solve)
        h->take_edges_n = (int)grid;
        h->take_edges_out = h->launch_edges + 2 * (size_t)pr->launch_idx;
    }
    HIP_TRY(hipGetLastError());
    return MISSLAP_OK;
}

// rounds with few bidders that are not sharded over GPUs: tiebreak, apply and compaction in one launch
bool use_round_small(const misslap_solver *h) {
    return h->K_ub <= kRoundSmallMax && (h->world == 1 || h->K_ub < h->shard_min_K);
}

int launch_bid(misslap_solver *h) {
    h->ctl_fresh = false;
    // (not behind a full-scan engine launch: the engines always feed best_key, which k_round_small ignores)
    h->round_small = use_round_small(h) && !(h->tiled_ok && h->K_ub >= h->tiled_min_K);
    if (h->tiled_ok && h->K_ub >= h->tiled_min_K) {
        int rc = launch_bid_tiled(h);  // no-op on the device when K < tiled_min_K
        if (rc) return rc;
        if (h->K_exact) {  // the host has just read K: k_bid would be a no-op
            h->phase_fresh = false;
            h->K_exact = false;
            return MISSLAP_OK;
        }
    }
    h->K_exact = false;
    RoundArgs a = round_args(h);
    const long long share = h->K_ub;  // upper bound: unsharded rounds bid for every list position
    // a round with few bidders is ONE launch (k_round_fused: bids by 16-wavefront workgroups, the rest by the last of them)
    const bool fused = h->round_small && h->round_fused;
    const int grid = blocks_for(share, (fused ? 1024 : kBidBlock) / kWave);
    ProfRec *pr = nullptr;
    // profile 1 times the full scans only (two event records around each of the ~3000 small launches of a solve
    // cost more host time than the launches themselves); profile 2 / 3 time every launch
    const bool fullscan = h->phase_fresh && !(h->tiled_ok && h->K_ub >= h->tiled_min_K);
    if (!(h->profile && (h->profile_all || fullscan))) a.launch_edges = nullptr;
    if (h->profile && (h->profile_all || fullscan)) {
        if (h->launch_idx >= h->launch_edges_cap)
            return fail(MISSLAP_ERR_STATE, "profile buffer exhausted (%d bid launches)", h->launch_idx);
        pr = prof_next(h, 0);
        if (!pr) return fail(MISSLAP_ERR_HIP, "hipEventCreate failed");
        pr->fullscan = fullscan;
        pr->launch_idx = a.launch_idx = h->launch_idx++;
    }
    const EdgesF32 e32{h->edges32};
    const EdgesF64 e64{h->col, h->val64};
    const dim3 g(grid), b(fused ? 1024 : kBidBlock);
    if (fused) {  // the launch closes the round: it carries the round's ticket (launch_apply has nothing left to do)
        a.live = (h->live_off || !h->live_every_round) ? nullptr : h->live_dev;
        a.ticket = ++h->ticket;
        h->live_valid = a.live != nullptr;
    }
    // variant: 2 = lines used and rebuilt; 1 = lines used, lean scan, nothing built (the full-scan regime); 0 = no lines
    const int variant = !h->lines_live() ? 0 : h->K_ub > h->cand_build_max_K ? 1 : 2;
    // big rounds of a handle whose price table does not fit an XCD's L2: the lean scans go through the fp32 filter
    // (wave_bid_filter); the mirror is rebuilt from the prices in front of the launch (12 bytes per object)
    if (h->price32 && !h->round_small && variant != 2 && (long long)h->K_ub * 8 >= h->n_rows) {
        HIP_TRY(stream_memset(h, h->pmax_bits, 0, sizeof(int)));
        MISSLAP_LAUNCH(h, k_price_mirror, (F_k_price_mirror), 1024, dim3(std::min(blocks_for(h->n_cols, 1024 * 4), h->n_cus)), dim3(1024),
                       (const Ctl *)h->ctl, (const double *)h->price, h->price32, h->n_cols, h->pmax_bits, h->thr, a.gather_max_K);
        a.price32 = h->price32;
    }
#define MISSLAP_LAUNCH_BID(E, ED)                                                                                   \
    do {                                                                                                            \
        if (h->batch) {  /* (no profiling inside a batch) */                                                        \
            if (fused) MISSLAP_LAUNCH(h, (k_round_fused<E>), (F_k_round_fused<E>), 1024, g, b, a, ED);              \
            else if (h->round_small) MISSLAP_LAUNCH(h, (k_bid<E, RecSource, 2>), (F_k_bid<E, RecSource, 2>), kBidBlock, g, b, a, ED); \
            else if (variant == 0) MISSLAP_LAUNCH(h, (k_bid<E, PriceSource, 0>), (F_k_bid<E, PriceSource, 0>), kBidBlock, g, b, a, ED); \
            else if (variant == 1) MISSLAP_LAUNCH(h, (k_bid<E, PriceSource, 1>), (F_k_bid<E, PriceSource, 1>), kBidBlock, g, b, a, ED); \
            else MISSLAP_LAUNCH(h, (k_bid<E, PriceSource, 2>), (F_k_bid<E, PriceSource, 2>), kBidBlock, g, b, a, ED); \
        } else if (fused) MISSLAP_LAUNCH_TIMED(pr, (k_round_fused<E>), g, b, 0, h->stream, a, ED);                  \
        else if (h->round_small) MISSLAP_LAUNCH_TIMED(pr, (k_bid<E, RecSource, 2>), g, b, 0, h->stream, a, ED);     \
        else if (variant == 0) MISSLAP_LAUNCH_TIMED(pr, (k_bid<E, PriceSource, 0>), g, b, 0, h->stream, a, ED);     \
        else if (variant == 1) MISSLAP_LAUNCH_TIMED(pr, (k_bid<E, PriceSource, 1>), g, b, 0, h->stream, a, ED);     \
        else MISSLAP_LAUNCH_TIMED(pr, (k_bid<E, PriceSource, 2>), g, b, 0, h->stream, a, ED);                       \
    } while (0)
    if (h->f32) MISSLAP_LAUNCH_BID(EdgesF32, e32);  // (rounds that k_round_small finishes: bids with the owners)
    else MISSLAP_LAUNCH_BID(EdgesF64, e64);
#undef MISSLAP_LAUNCH_BID
    if (pr && !h->round_small) {
        h->take_edges_n = std::max(h->take_edges_n, grid);  // (a full-scan engine launch of the same round may be pending too)
        h->take_edges_out = h->launch_edges + 2 * (size_t)pr->launch_idx;
    } else if (pr) {  // (no k_tiebreak in a round that k_round_small finishes; such launches are profiled at level 2 / 3 only)
        hipLaunchKernelGGL(k_take_launch_edges, dim3(1), dim3(1024), 0, h->stream, h->wg_stats, grid, h->launch_edges + 2 * (size_t)pr->launch_idx);
    }
    HIP_TRY(hipGetLastError());
    h->phase_fresh = false;
    h->round_done = fused;
    return MISSLAP_OK;
}

int launch_tiebreak(misslap_solver *h) {
    h->ctl_fresh = false;
    if (h->round_small) return MISSLAP_OK;  // k_round_small (launch_apply) resolves the ties itself
    RoundArgs a = round_args(h);
    const long long share = h->K_ub;
    MISSLAP_LAUNCH(h, k_tiebreak, (F_k_tiebreak), 256, dim3(blocks_for(share, 256)), dim3(256), a,
                   (const int *)(h->round_ordered ? h->mover_list : nullptr), h->tiled_min_K, h->take_edges_n, h->take_edges_out);
    h->take_edges_n = 0;
    h->take_edges_out = nullptr;
    HIP_TRY(hipGetLastError());
    return MISSLAP_OK;
}

int launch_apply(misslap_solver *h) {
    h->ctl_fresh = false;
    if (h->round_small && h->round_done) {  // (k_round_fused has closed the round)
        h->round_small = h->round_done = false;
        h->K_exact = false;
        return MISSLAP_OK;
    }
    RoundArgs a = round_args(h);
    // (a round that k_round_small closes posts nothing: four stores to host memory are 1.5 us on a 3-5 us kernel that
    // runs thousands of times per solve -- a batch of such rounds is followed by k_post_status instead, ensure_posted)
    a.live = (h->live_off || (h->round_small && !h->live_every_round)) ? nullptr : h->live_dev;
    a.ticket = ++h->ticket;
    h->live_valid = a.live != nullptr;
    if (h->round_small) {
        h->round_small = false;
        h->K_exact = false;
        MISSLAP_LAUNCH(h, k_round_small, (F_k_round_small), 1024, dim3(1), dim3(1024), a);
        HIP_TRY(hipGetLastError());
        return MISSLAP_OK;
    }
    h->K_exact = false;
    h->round_ordered = false;
    // by the bidders where they are few against the objects (every rank holds every bid only in unsharded rounds)
    if ((h->world == 1 || h->K_ub < h->shard_min_K) && (long long)h->K_ub * h->apply_bidders_ratio <= h->n_cols)
        MISSLAP_LAUNCH(h, k_apply_bidders, (F_k_apply_bidders), 256, dim3(blocks_for(h->K_ub, 256)), dim3(256), a);
    else
        MISSLAP_LAUNCH(h, k_apply, (F_k_apply), 256, dim3(blocks_for(h->n_cols, 256)), dim3(256), a);
    if (h->K_ub <= kCompactSmallMax) {
        MISSLAP_LAUNCH(h, k_compact_small, (F_k_compact_small), 1024, dim3(1), dim3(1024), a);
    } else {
        const int cb = blocks_for(h->K_ub, kChunk);
        MISSLAP_LAUNCH(h, k_compact_count, (F_k_compact_count), 256, dim3(cb), dim3(256), a);
        MISSLAP_LAUNCH(h, k_compact_scatter, (F_k_compact_scatter), 256, dim3(cb), dim3(256), a);
        MISSLAP_LAUNCH(h, k_compact_fill, (F_k_compact_fill), 256, dim3(blocks_for(h->K_ub, 256)), dim3(256), a);
    }
    HIP_TRY(hipGetLastError());
    return MISSLAP_OK;
}

int launch_tail(misslap_solver *h) {
    if (h->thr <= 0) return MISSLAP_OK;
    h->ctl_fresh = false;
    // Rows of a few hundred edges keep no lines until the solve has shown that its tail is long: that many tail rounds
    // (a tail round without a line is a row scan by one wavefront, 1.5-4 us at 300-1000 edges; the pass that builds the
    // lines of every row costs milliseconds at C4's 100 000 rows and pays for itself within a phase at a dense
    // 1000 x 1000).  The tail kernels of such a handle return after as many rounds, so that a first phase with thousands
    // of tail rounds does not run to its end without lines (dense 1000^2: 13 of 16 ms were its first two tail launches).
    const long long long_after = std::max<long long>(kLongRowsAfterTailRoundsMin, h->n_rows / 64);
    if (h->long_rows_later && h->tail_rounds_host >= long_after) {  // (status read just before)
        h->long_rows = true;
        h->long_rows_later = false;
    }
    TailArgs a;
    a.ctl = h->ctl;
    a.row_ptr = h->row_ptr;
    a.price = h->price;
    a.rec = h->rec;
    a.p2o = h->p2o;
    a.o2p = h->o2p;
    a.U = h->U;
    const bool lines = h->lines_live();
    a.cand = lines ? h->cand : nullptr;
    a.cand64 = lines ? h->cand64 : nullptr;
    a.round_budget = h->long_rows && lines && h->line_maintenance ? h->tail_round_budget
                     : h->long_rows_later                            ? (int)std::min<long long>(std::max<long long>(long_after, 1), 1 << 30)
                                                                     : 0;
    if (a.round_budget == 0 || a.round_budget > h->tail_launch_rounds) a.round_budget = h->tail_launch_rounds;
    a.thr = h->thr;
    a.eps = h->eps;
    ProfRec *pr = nullptr;
    if (h->profile) {
        pr = prof_next(h, 1);
        if (!pr) return fail(MISSLAP_ERR_HIP, "hipEventCreate failed");
        HIP_TRY(hipEventRecord(pr->start, h->stream));
    }
    const EdgesF32 e32{h->edges32};
    const EdgesF64 e64{h->col, h->val64};
    // rows the long-row builder takes (it runs right behind the pass over all lines, on the list that pass leaves)
    const int long_max = !(lines && h->line_maintenance && h->long_rows) ? 0 : (h->max_row_len <= 256 * kLongPer ? 256 : 512) * kLongPer;
    const int min_alive_long = kLongRowMinAlive;
    // every line checked at today's prices (kernels_round.hpp); then the rounds with more than kTeamMax bidders, with
    // sixteen wavefronts (kernels_tail.hpp); then -- lines only -- the rounds with 3..kTeamMax bidders, one list slot
    // per wavefront; then the rest: with lines the two-wavefront duo / chain instance, without them the 512-thread
    // instance that holds every mode
    // (inside a batch all three instances are launched whatever K: an instance that finds K outside its range returns
    // at once, and the problems of a group then issue the same sequence of kernels, i.e. share every launch)
    const bool in_batch = h->batch != nullptr;
    if (in_batch) h->batch->hold_next = true;  // ... and the sequence waits for the other problems of the group (host_batch.hpp)
#define MISSLAP_LAUNCH_TAIL(E, ED)                                                                                       \
    do {                                                                                                                 \
        if (lines && h->line_maintenance)                                                                                \
            MISSLAP_LAUNCH(h, k_refresh_lines<E>, (F_k_refresh_lines<E>), kBidBlock,                                     \
                           dim3(blocks_for((h->n_rows + 1) / 2, kBidBlock / kWave)), dim3(kBidBlock), round_args(h), ED, \
                           (int)kCandMaintenanceMin, long_max, min_alive_long);                                          \
        if (lines && h->line_maintenance && h->long_rows) {                                                              \
            if (h->max_row_len <= 256 * kLongPer)                                                                        \
                MISSLAP_LAUNCH(h, (k_refresh_long<E, 256>), (F_k_refresh_long<E, 256>), 256, dim3(blocks_for(h->n_rows, 1)), \
                               dim3(256), round_args(h), ED);                                                            \
            else                                                                                                         \
                MISSLAP_LAUNCH(h, (k_refresh_long<E, 512>), (F_k_refresh_long<E, 512>), 512, dim3(blocks_for(h->n_rows, 1)), \
                               dim3(512), round_args(h), ED);                                                            \
        }                                                                                                                \
        if (h->K_ub > kTeamMax || in_batch)                                                                              \
            MISSLAP_LAUNCH(h, (k_tail<E, 2 * kTailMax>), (F_k_tail<E, 2 * kTailMax, false>), 2 * kTailMax, dim3(1),      \
                           dim3(2 * kTailMax), a, ED);                                                                   \
        if ((h->K_ub > 2 || in_batch) && lines)                                                                          \
            MISSLAP_LAUNCH(h, (k_tail<E, 2 * kTailMax, true>), (F_k_tail<E, 2 * kTailMax, true>), 2 * kTailMax, dim3(1), \
                           dim3(2 * kTailMax), a, ED);                                                                   \
        if (lines) MISSLAP_LAUNCH(h, (k_tail<E, 2 * kWave>), (F_k_tail<E, 2 * kWave, false>), 2 * kWave, dim3(1),        \
                                  dim3(2 * kWave), a, ED);                                                               \
        else MISSLAP_LAUNCH(h, (k_tail<E, kTailMax>), (F_k_tail<E, kTailMax, false>), kTailMax, dim3(1), dim3(kTailMax), a, ED); \
    } while (0)
    if (h->f32) MISSLAP_LAUNCH_TAIL(EdgesF32, e32);
    else MISSLAP_LAUNCH_TAIL(EdgesF64, e64);
#undef MISSLAP_LAUNCH_TAIL
    if (pr) HIP_TRY(hipEventRecord(pr->stop, h->stream));
    // the tail keeps only the price records current: rebuild price / o2p / p2o from them
    h->live_valid = !h->live_off && h->live_dev != nullptr;
    MISSLAP_LAUNCH(h, k_sync_from_rec, (F_k_sync_from_rec), 256, dim3(blocks_for(h->n_cols, 256)), dim3(256), h->ctl,
                   (const PriceRec *)h->rec, h->price, h->o2p, h->p2o, (const int *)h->U, h->n_cols,
                   (h->cand != nullptr && !h->lines_dropped) ? 1 : 0, h->live_valid ? h->live_dev : (unsigned long long *)nullptr,
                   ++h->ticket);
    HIP_TRY(hipGetLastError());
    h->phase_fresh = false;
    if (h->tail_nits0 < 0) h->tail_nits0 = h->h_ctl->nits;  // (the status read in front of this launch)
    return MISSLAP_OK;
}

// The pass over all rows behind eCE_satisfied / get_obj / the validity flags (kernels_check.hpp) runs on the
// full-scan engine where the handle has the tile-major copy in a shape the check instances cover: lanes per person of
// that shape (the overflow lists are built for 2 x lanes x 2 loads per segment), 0 = the pass on the row-major CSR.
int check_lanes(const misslap_solver *h) {
    if (!h->tiled_ok) return 0;
    const int *shp = kTiledShapes[h->tiled_shape];
    return (shp[3] == 2 && shp[4] == kTileColsHalf) ? shp[6] : 0;
}
#define MISSLAP_FOR_CHECK_LANES(X) X(4) X(8) X(16)
#define MISSLAP_CHECK_KERNEL(GL) k_bid_tiled<1024, 4, 2, 2, kTileColsHalf, 3, 0, GL, 1, 1>
#define MISSLAP_CHECK_KERNEL_FMT(GL, FMT) k_bid_tiled<1024, 4, 2, 2, kTileColsHalf, 3, 0, GL, 1, 1, FMT>

// rows [0, n_rows) on the row-major CSR (the sample of run_ece; every row where there is no tile-major copy)
// the largest grid launch_rows_all can ask for on n_rows persons (over every lanes-per-person shape of the check pass)
size_t final_pass_grid_max(size_t n_rows, int n_cus) {
    const size_t per_wg_min = (size_t)((1024 - 64 * 3) / 16) * 4;  // 16 lanes per person
    return std::max<size_t>((n_rows + per_wg_min - 1) / per_wg_min, (size_t)n_cus);
}
int launch_rows_gather(misslap_solver *h, float eps, const FinalOut &fo, int n_rows, int *n_blocks = nullptr) {
    const int grid = blocks_for(n_rows, 4);
    if (n_blocks) *n_blocks = grid;
    if (h->f32) {
        EdgesF32 ed{h->edges32};
        MISSLAP_LAUNCH(h, k_ece<EdgesF32>, (F_k_ece<EdgesF32>), 256, dim3(grid), dim3(256), h->ctl, ed, (const int *)h->row_ptr,
                       (const double *)h->price, (const int *)h->p2o, n_rows, eps, fo);
    } else {
        EdgesF64 ed{h->col, h->val64};
        MISSLAP_LAUNCH(h, k_ece<EdgesF64>, (F_k_ece<EdgesF64>), 256, dim3(grid), dim3(256), h->ctl, ed, (const int *)h->row_ptr,
                       (const double *)h->price, (const int *)h->p2o, n_rows, eps, fo);
    }
    HIP_TRY(hipGetLastError());
    return MISSLAP_OK;
}
// every row, on the engine the handle has
int launch_rows_all(misslap_solver *h, float eps, const FinalOut &fo, int *n_blocks = nullptr) {
    const int gl = check_lanes(h);
    if (!gl) return launch_rows_gather(h, eps, fo, h->n_rows, n_blocks);
    RoundArgs a = round_args(h);
    a.eps = eps;
    a.launch_edges = nullptr;
    TiledArgs ta{h->tiled, h->tcol, h->seg4, h->T, 0, h->n_tiled,
                 nullptr, nullptr, h->ovf_ptr, h->ovf_q, h->ovf_cap, nullptr, nullptr, h->n_rows, nullptr, fo};
    const int groups = (1024 - 64 * 3) / gl, per_wg_max = groups * 4;
    long long grid = ((long long)h->n_rows + per_wg_max - 1) / per_wg_max;
    const long long spread = std::min<long long>(h->n_cus, ((long long)h->n_rows + groups - 1) / groups);
    if (grid < spread) grid = spread;
    if (fo.fin && grid > h->fin_slots_n) return fail(MISSLAP_ERR_STATE, "final pass: grid %lld exceeds its result slots (%d)", grid, h->fin_slots_n);
    if (n_blocks) *n_blocks = (int)grid;
    const size_t lds = tiled_lds_bytes(kTileColsHalf);
    if (h->tiled_fmt == 0) {
        switch (gl) {
#define X(GL) \
    case GL: MISSLAP_LAUNCH_PLAIN(h, (MISSLAP_CHECK_KERNEL(GL)), dim3((unsigned)grid), dim3(1024), lds, a, ta); break;
            MISSLAP_FOR_CHECK_LANES(X)
#undef X
        }
    } else {
        switch (h->tiled_fmt * 100 + gl) {
#define X(FMT, GL) \
    case FMT * 100 + GL: MISSLAP_LAUNCH_PLAIN(h, (MISSLAP_CHECK_KERNEL_FMT(GL, FMT)), dim3((unsigned)grid), dim3(1024), lds, a, ta); break;
            MISSLAP_FOR_FMT_LANES(X)
#undef X
        }
    }
    HIP_TRY(hipGetLastError());
    return MISSLAP_OK;
}

// eCE_satisfied(eps), auction_.pyx:443-485.  The sample pass first (kernels_check.hpp: a failing test fails within the
// first few rows), then every row -- a launch that returns at once when the sample has set the flag.
int run_ece(misslap_solver *h, float eps, int *ok) {
    int rc = read_status(h);
    if (rc) return rc;
    if (h->h_ctl->K > 0) {  // auction_.pyx:446-447
        *ok = 0;
        return MISSLAP_OK;
    }
    h->ctl_fresh = false;
    // (the flag is clear behind the state initialisation and behind every k_reset_phase: one runtime fill kernel less
    // per phase; a second test on the same state -- misslap_check_ece -- clears it itself)
    if (!h->ece_flag_clear) HIP_TRY(stream_memset(h, &h->ctl->ece_fail, 0, sizeof(int)));
    h->ece_flag_clear = false;
    const FinalOut fo{0, h->maximize, h->o2p, h->contrib, h->nmatch, h->n_rows, h->n_cols, nullptr};
    const int sample = std::min(h->n_rows, kEceSampleRows);
    if ((rc = launch_rows_gather(h, eps, fo, sample))) return rc;
    if (sample < h->n_rows && (rc = launch_rows_all(h, eps, fo))) return rc;
    if (!h->live_off) {  // the verdict through the live words: no copy of the control block, no stream drain
        MISSLAP_LAUNCH(h, k_post_ece, (F_k_post_ece), 1, dim3(1), dim3(1), (const Ctl *)h->ctl, h->live_dev, ++h->ticket);
        h->live_valid = true;
        int K = 0, err = 0;
        long long nits = 0;
        if (live_poll(h, h->ticket, true, &K, &err, &nits)) {
            // (k_post_ece stores the verdict word right behind the four status words; the wait is bounded by time,
            // like live_poll's)
            volatile unsigned long long *w = h->live + 4;
            unsigned long long v = *w;
            const double t_end = now_ms() + 2000.0;
            for (unsigned spins = 0; (unsigned)(v >> 32) != h->ticket; ++spins) {
                if (h->batch) batch_yield(h, BatchFiber::kPolling);
                else __builtin_ia32_pause();
                if ((spins & (h->batch ? 63 : 4095)) == (h->batch ? 63u : 4095u) && now_ms() > t_end) break;
                v = *w;
            }
            if ((unsigned)(v >> 32) == h->ticket) {
                h->h_ctl->K = K;
                h->h_ctl->nits = nits;
                h->h_ctl->err = err;
                h->h_ctl->ece_fail = (int)(unsigned)(v & 0xffffffffull);
                h->K_ub = K;
                h->K_exact = true;
                h->ctl_fresh = 1;
                count_tail_rounds(h);
                if (err) return fail(MISSLAP_ERR_STATE, "device-side invariant violated (error bits 0x%x)", err);
                *ok = h->h_ctl->ece_fail ? 0 : 1;
                return MISSLAP_OK;
            }
        }
        h->live_off = true;  // timed out: from here on by copy + drain
    }
    rc = read_ctl(h);
    if (rc) return rc;
    *ok = h->h_ctl->ece_fail ? 0 : 1;
    return MISSLAP_OK;
}
}  // namespace
