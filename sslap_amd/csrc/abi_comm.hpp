// abi_comm.hpp -- C ABI: communicators (RCCL opened at run time, custom callbacks) and the sharded solve.
// (part of the single translation unit misslap.hip; included in the order given there)
#pragma once

// ---- multi-GPU: communicators and the sharded solve (host_comm.hpp) ------------------------------------------------
MISSLAP_API int misslap_rccl_unique_id(void *id_out) {
    if (!id_out) return fail(MISSLAP_ERR_INVALID, "null argument");
    RcclApi &api = rccl_api();
    if (!api.error.empty()) return fail(MISSLAP_ERR_HIP, "%s", api.error.c_str());
    RcclApi::UniqueId id;
    const int rc = api.GetUniqueId(&id);
    if (rc) return fail(MISSLAP_ERR_HIP, "ncclGetUniqueId failed: %s", api.GetErrorString(rc));
    std::memcpy(id_out, &id, sizeof(id));
    return MISSLAP_OK;
}

MISSLAP_API int misslap_rccl_selfcheck(int32_t *n_symbols, int32_t enums[6], char *lib_path, int32_t lib_path_len) {
    RcclApi &api = rccl_api();
    if (!api.handle) return fail(MISSLAP_ERR_HIP, "%s", api.error.c_str());
    int n = 0;
    n += api.GetUniqueId != nullptr;
    n += api.CommInitRank != nullptr;
    n += api.CommDestroy != nullptr;
    n += api.AllReduce != nullptr;
    n += api.GetErrorString != nullptr;
    n += api.CommCount != nullptr;
    if (n_symbols) *n_symbols = n;
    if (enums) {
        enums[0] = kNcclInt32;
        enums[1] = kNcclInt64;
        enums[2] = kNcclMax;
        enums[3] = kNcclMin;
        enums[4] = (int32_t)sizeof(RcclApi::UniqueId);
        enums[5] = 0;
        if (auto ver = reinterpret_cast<int (*)(int *)>(dlsym(api.handle, "ncclGetVersion"))) {
            int v = 0;
            if (ver(&v) == 0) enums[5] = v;
        }
    }
    if (lib_path && lib_path_len > 0) {
        lib_path[0] = 0;
        Dl_info di;
        if (api.AllReduce && dladdr(reinterpret_cast<void *>(api.AllReduce), &di) && di.dli_fname)
            snprintf(lib_path, (size_t)lib_path_len, "%s", di.dli_fname);
    }
    if (n != MISSLAP_RCCL_SYMBOLS) return fail(MISSLAP_ERR_HIP, "%s", api.error.empty() ? "librccl: a symbol is missing" : api.error.c_str());
    return MISSLAP_OK;
}

MISSLAP_API int misslap_comm_init_rccl(misslap_comm **out, const void *unique_id, int32_t rank, int32_t world,
                                       int32_t device) {
    if (!out || !unique_id || world < 1 || rank < 0 || rank >= world) return fail(MISSLAP_ERR_INVALID, "bad argument");
    RcclApi &api = rccl_api();
    if (!api.error.empty()) return fail(MISSLAP_ERR_HIP, "%s", api.error.c_str());
    HIP_TRY(hipSetDevice(device));
    RcclApi::UniqueId id;
    std::memcpy(&id, unique_id, sizeof(id));
    misslap_comm *c = new misslap_comm();
    c->rank = rank;
    c->world = world;
    c->device = device;
    const int rc = api.CommInitRank(&c->nccl_comm, world, id, rank);
    if (rc) {
        delete c;
        return fail(MISSLAP_ERR_HIP, "ncclCommInitRank failed: %s", api.GetErrorString(rc));
    }
    *out = c;
    return MISSLAP_OK;
}

MISSLAP_API int misslap_comm_init_custom(misslap_comm **out, const misslap_comm_ops *ops) {
    if (!out || !ops || ops->struct_size != (int32_t)sizeof(misslap_comm_ops) || !ops->allreduce_max_i64 ||
        !ops->allreduce_min_i32 || ops->world < 1 || ops->rank < 0 || ops->rank >= ops->world)
        return fail(MISSLAP_ERR_INVALID, "bad misslap_comm_ops");
    misslap_comm *c = new misslap_comm();
    c->rank = ops->rank;
    c->world = ops->world;
    c->ops = *ops;
    c->custom = true;
    *out = c;
    return MISSLAP_OK;
}

MISSLAP_API int misslap_comm_info(const misslap_comm *c, int32_t *kind, int32_t *rank, int32_t *world, int32_t *transport_ranks) {
    if (!c) return fail(MISSLAP_ERR_INVALID, "null communicator");
    if (kind) *kind = c->custom ? 0 : 1;
    if (rank) *rank = c->rank;
    if (world) *world = c->world;
    if (transport_ranks) {
        *transport_ranks = c->ops.world;
        if (!c->custom) {
            RcclApi &api = rccl_api();
            int n = 0;
            if (!api.CommCount) return fail(MISSLAP_ERR_HIP, "librccl: ncclCommCount is missing");
            const int rc = api.CommCount(c->nccl_comm, &n);
            if (rc) return fail(MISSLAP_ERR_HIP, "ncclCommCount failed: %s", api.GetErrorString(rc));
            *transport_ranks = n;
        }
    }
    return MISSLAP_OK;
}

MISSLAP_API int misslap_comm_destroy(misslap_comm *c) {
    if (!c) return MISSLAP_OK;
    if (!c->custom && c->nccl_comm) (void)rccl_api().CommDestroy(c->nccl_comm);
    delete c;
    return MISSLAP_OK;
}

MISSLAP_API int misslap_drive_sharded(const misslap_round_ops *ops, misslap_comm *comm) {
    if (!ops || ops->struct_size != (int32_t)sizeof(misslap_round_ops) || !ops->status || !ops->round_bid ||
        !ops->round_tiebreak || !ops->round_apply || !ops->run_tail || !ops->phase_end)
        return fail(MISSLAP_ERR_INVALID, "bad misslap_round_ops");
    return drive_sharded(ops, comm, fail);
}

MISSLAP_API int misslap_solve_sharded(misslap_solver *h, misslap_comm *comm, int32_t *person_to_object_out,
                                      misslap_meta *meta) {
    if (!h) return fail(MISSLAP_ERR_INVALID, "null handle");
    if (comm && (comm->world != h->world || comm->rank != h->rank))
        return fail(MISSLAP_ERR_INVALID, "communicator is rank %d of %d, the handle was created as shard %d of %d",
                    comm->rank, comm->world, h->rank, h->world);
    if (!comm && h->world != 1) return fail(MISSLAP_ERR_INVALID, "a handle of %d shards needs a communicator", h->world);
    if (comm && !comm->custom && comm->device != h->device)  // an all-reduce enqueued on another device's stream fails late or hangs
        return fail(MISSLAP_ERR_INVALID, "the RCCL communicator lives on device %d, the handle on device %d", comm->device,
                    h->device);
    // (before any work: a caller that forgot the size must not pay for a solve to learn it)
    if (meta && h->abi >= 2 && (meta->struct_size < (int32_t)offsetof(misslap_meta, edges_scanned) || meta->struct_size > 65536))
        return fail(MISSLAP_ERR_INVALID, "misslap_meta.struct_size = %d: set it to sizeof(misslap_meta) before the call", meta->struct_size);
    HIP_TRY(hipSetDevice(h->device));
    const double t0 = now_ms();
    const misslap_round_ops o = handle_round_ops(h);
    if (comm) comm->sharded_rounds = 0;
    int rc = drive_sharded(&o, comm, fail);
    h->sharded_rounds = comm ? comm->sharded_rounds : 0;
    if (rc) return rc;
    HIP_TRY(stream_sync(h));
    h->solve_ms += now_ms() - t0;
    return misslap_finish(h, person_to_object_out, meta);
}
