"""ctypes binding of libmisslap.so (include/misslap.h).  There is no fallback: if the HIP library
is missing or no GPU is usable, every solver call raises."""
import ctypes as C
import os

_PKG = os.path.dirname(os.path.abspath(__file__))
# MISSLAP_LIB selects another build of the same library (A/B timing of kernel variants on one GPU box)
LIB_PATH = os.environ.get("MISSLAP_LIB") or os.path.join(_PKG, "libmisslap.so")

MISSLAP_OK, ERR_INVALID, ERR_HIP, ERR_NO_DEVICE, ERR_STATE = 0, 1, 2, 3, 4


class Options(C.Structure):
    _fields_ = [
        ("struct_size", C.c_int32), ("device", C.c_int32), ("maximize", C.c_int32), ("eps_start", C.c_float),
        ("max_iter", C.c_int64), ("input_on_device", C.c_int32), ("tail_threshold", C.c_int32),
        ("force_f64_values", C.c_int32), ("profile", C.c_int32), ("shard_rank", C.c_int32),
        ("shard_world", C.c_int32), ("rounds_per_sync", C.c_int32),
        # tuning knobs (0 = library default)
        ("tiled_min_K", C.c_int32), ("tiled_shape", C.c_int32), ("tiled_force", C.c_int32),
        ("shard_min_K", C.c_int32), ("cand_mode", C.c_int32), ("partial_in_list_order", C.c_int32),
        ("nnz_limit", C.c_int32), ("cand_build_max_K", C.c_int32), ("cand_refresh_min", C.c_int32),
        ("reserved", C.c_int32 * 7),
        ("input_stream", C.c_void_p),
    ]


class Meta(C.Structure):
    """misslap_meta (ABI 2).  Create with new_meta(): struct_size must be set before a call."""
    _fields_ = [
        ("struct_size", C.c_int32), ("abi_version", C.c_int32),
        ("start_eps", C.c_float), ("final_eps", C.c_float), ("target_eps", C.c_float),
        ("eCE", C.c_int32), ("soln_found", C.c_int32), ("nreductions", C.c_int32),
        ("its", C.c_int64), ("n_assigned", C.c_int64),
        ("n_rows", C.c_int64), ("n_cols", C.c_int64), ("nnz", C.c_int64),
        ("obj_f32", C.c_float), ("obj_f64", C.c_double), ("setup_ms", C.c_double), ("solve_ms", C.c_double),
        ("edges_scanned", C.c_uint64), ("bids_made", C.c_uint64),
        ("grid_rounds", C.c_int64), ("tail_rounds", C.c_int64),
        ("bytes_per_edge", C.c_int32), ("profiled", C.c_int32),
        ("bid_launches", C.c_int64), ("bid_ms", C.c_double), ("bid_edges", C.c_uint64),
        ("fullscan_launches", C.c_int64), ("fullscan_ms", C.c_double), ("fullscan_edges", C.c_uint64),
        ("tail_launches", C.c_int64), ("tail_ms", C.c_double), ("tail_edges", C.c_uint64),
        ("tiled_launches", C.c_int64), ("tiled_ms", C.c_double), ("tiled_edges", C.c_uint64),
        ("tiled_active", C.c_int32), ("tiled_min_K", C.c_int32),
        ("bid_edges_read", C.c_uint64), ("fullscan_edges_read", C.c_uint64),
        ("shard_edges", C.c_uint64), ("cand_hits", C.c_uint64), ("cand_edges", C.c_uint64),
        ("tail_stats", C.c_double * 12),
        ("complete_assignment", C.c_int32), ("valid_assignment", C.c_int32), ("lines_active", C.c_int32),
        ("reserved_i", C.c_int32), ("sharded_rounds", C.c_int64),
        ("tiled_format", C.c_int32), ("phases_with_lines", C.c_int32), ("eps_phases", C.c_int32), ("filter_undecided", C.c_int32),
    ]


class BatchInfo(C.Structure):
    _fields_ = [("groups", C.c_int32), ("reserved", C.c_int32), ("calls_recorded", C.c_int64),
                ("launches_issued", C.c_int64), ("wall_ms", C.c_double), ("host_ms_fibers", C.c_double),
                ("host_ms_flush", C.c_double), ("host_ms_wait", C.c_double)]


def new_meta():
    m = Meta()
    m.struct_size = C.sizeof(Meta)
    return m


class Status(C.Structure):
    _fields_ = [
        ("its", C.c_int64), ("K", C.c_int32), ("nreductions", C.c_int32), ("eps", C.c_float),
        ("target_eps", C.c_float), ("finished", C.c_int32), ("error_bits", C.c_int32),
        ("tail_threshold", C.c_int32), ("rounds_per_sync", C.c_int32),
        ("shard_min_K", C.c_int32), ("reserved", C.c_int32),
    ]


_AR = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p)


class CommOps(C.Structure):
    _fields_ = [("struct_size", C.c_int32), ("rank", C.c_int32), ("world", C.c_int32), ("reserved", C.c_int32),
                ("ctx", C.c_void_p), ("allreduce_max_i64", _AR), ("allreduce_min_i32", _AR)]


_OP0 = C.CFUNCTYPE(C.c_int, C.c_void_p)
_OP_STATUS = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_int64))
_OP_PHASE = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_int32))
_OP_POST = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int32)
_OP_TAKE = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int32, C.POINTER(C.c_int64), C.POINTER(C.c_int64))


class RoundOps(C.Structure):
    _fields_ = [("struct_size", C.c_int32), ("tail_threshold", C.c_int32), ("shard_min_K", C.c_int32),
                ("rounds_per_sync", C.c_int32), ("max_iter", C.c_int64), ("ctx", C.c_void_p),
                ("status", _OP_STATUS), ("round_bid", _OP0), ("round_tiebreak", _OP0), ("round_apply", _OP0),
                ("run_tail", _OP0), ("phase_end", _OP_PHASE), ("best_key", C.c_void_p), ("best_pos", C.c_void_p),
                ("n_objects", C.c_int64), ("stream", C.c_void_p),
                ("status_post", _OP_POST), ("status_take", _OP_TAKE), ("large_round_K", C.c_int32),
                ("rounds_per_sync_large", C.c_int32)]


# every symbol include/misslap.h declares: (name, restype, argtypes)
_VP, _I32P = C.c_void_p, C.POINTER(C.c_int32)
SYMBOLS = {
    "misslap_create": (C.c_int, [C.POINTER(_VP), C.c_int64, _VP, _VP, C.POINTER(Options)]),
    "misslap_create_dense": (C.c_int, [C.POINTER(_VP), C.c_int64, C.c_int64, _VP, C.POINTER(Options),
                                       C.POINTER(C.c_int64)]),
    "misslap_destroy": (C.c_int, [_VP]),
    "misslap_dims": (C.c_int, [_VP, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "misslap_solve": (C.c_int, [_VP, _VP, C.POINTER(Meta)]),
    "misslap_solve_batch": (C.c_int, [C.POINTER(_VP), C.c_int32, C.POINTER(_VP), C.POINTER(C.POINTER(Meta)), C.c_int32,
                                      C.POINTER(BatchInfo)]),
    "misslap_round_bid": (C.c_int, [_VP]),
    "misslap_round_tiebreak": (C.c_int, [_VP]),
    "misslap_round_apply": (C.c_int, [_VP]),
    "misslap_run_tail": (C.c_int, [_VP]),
    "misslap_get_status": (C.c_int, [_VP, C.POINTER(Status)]),
    "misslap_phase_end": (C.c_int, [_VP, _I32P]),
    "misslap_check_ece": (C.c_int, [_VP, C.c_float, _I32P]),
    "misslap_finish": (C.c_int, [_VP, _VP, C.POINTER(Meta)]),
    "misslap_exchange_buffers": (C.c_int, [_VP, C.POINTER(_VP), C.POINTER(_VP), C.POINTER(C.c_int64)]),
    "misslap_set_stream": (C.c_int, [_VP, _VP]),
    "misslap_get_state": (C.c_int, [_VP, _VP, _VP, _VP, _VP]),
    "misslap_device_info": (C.c_int, [C.c_int32, C.c_char_p, C.c_int32, _I32P, C.POINTER(C.c_int64)]),
    "misslap_matching_of": (C.c_int, [_VP, _I32P, _I32P]),
    "misslap_rccl_unique_id": (C.c_int, [_VP]),
    "misslap_rccl_selfcheck": (C.c_int, [C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.c_char_p, C.c_int32]),
    "misslap_comm_init_rccl": (C.c_int, [C.POINTER(_VP), _VP, C.c_int32, C.c_int32, C.c_int32]),
    "misslap_comm_init_custom": (C.c_int, [C.POINTER(_VP), C.POINTER(CommOps)]),
    "misslap_comm_destroy": (C.c_int, [_VP]),
    "misslap_comm_info": (C.c_int, [_VP, _I32P, _I32P, _I32P, _I32P]),
    "misslap_device_uuid": (C.c_int, [C.c_int32, C.c_char_p, C.c_int32]),
    "misslap_measure_hbm": (C.c_int, [C.c_int32, C.c_int64, C.c_int32, C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "misslap_solve_sharded": (C.c_int, [_VP, _VP, _VP, C.POINTER(Meta)]),
    "misslap_drive_sharded": (C.c_int, [C.POINTER(RoundOps), _VP]),
    "misslap_hopcroft_karp": (C.c_int, [_VP, C.c_int64, C.c_int32, C.c_int32, _I32P, _VP, _VP]),
    "misslap_matching_gpu": (C.c_int, [_VP, C.c_int64, C.c_int32, C.c_int32, C.c_int32, _I32P, _VP, _VP, _I32P]),
    "misslap_trim_caches": (C.c_int, [C.POINTER(C.c_int64)]),
    "misslap_set_cache_limits": (C.c_int, [C.c_int64, C.c_int64, C.c_int32]),
    "misslap_last_error": (C.c_char_p, []),
    "misslap_abi_version": (C.c_int, []),
}

_LIB = None


def _preload_hip_runtime():
    """One HIP runtime per process, whatever the import order.

    libmisslap.so names its runtime by SONAME (libamdhip64.so.7) and the dynamic linker binds that to a copy the
    process has ALREADY loaded.  A PyTorch-ROCm wheel ships a private copy and loads it by path, so "libmisslap
    first, torch later" would end with two runtimes, the second of which cannot open the GPU (measured on the GPU
    box: torch then reports "No HIP GPUs are available").  Therefore: if a torch wheel with a bundled runtime is
    INSTALLED, its copy is loaded here by path -- torch itself is not imported -- and both libmisslap.so and a later
    `import torch` bind to it; without torch the system ROCm runtime is used.  Covered in both orders by
    tests/test_gpu_parity.py::test_hip_runtime_is_shared_in_either_import_order."""
    import importlib.util
    import sys
    if "torch" in sys.modules:
        return  # torch's runtime is already in the process
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    for d in (spec.submodule_search_locations if spec and spec.submodule_search_locations else ()):
        cand = os.path.join(d, "lib", "libamdhip64.so")
        if os.path.exists(cand):
            C.CDLL(cand, mode=C.RTLD_GLOBAL)
            return


def load():
    """Load libmisslap.so and bind every exported symbol; raises if the library is absent."""
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: build it with `python -m sslap_amd.build` "
                "(hipcc --offload-arch=gfx950).  sslap_amd has no CPU fallback.")
        _preload_hip_runtime()
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in SYMBOLS.items():
            f = getattr(lib, name)
            f.restype, f.argtypes = res, args
        if lib.misslap_abi_version() != 2:
            raise RuntimeError("libmisslap.so ABI version mismatch")
        _LIB = lib
    return _LIB


def load_diag():
    """The diagnostics build (libmisslap_diag.so, include/misslap_diag.h): tools/ only.  Select it for the whole
    process with MISSLAP_LIB=<path> BEFORE the first load(); this only binds the extra entry point."""
    lib = load()
    if not hasattr(lib, "misslap_debug_time_bid"):
        raise RuntimeError("not a diagnostics build: run `python -m sslap_amd.build diag` and set "
                           "MISSLAP_LIB=sslap_amd/libmisslap_diag.so")
    f = lib.misslap_debug_time_bid
    f.restype, f.argtypes = C.c_int, [_VP, C.c_int32, C.c_int32, C.POINTER(C.c_float)]
    return lib


def check(rc):
    """Map a C status code to the Python exception the reference's callers expect."""
    if rc == MISSLAP_OK:
        return
    msg = load().misslap_last_error().decode("utf-8", "replace")
    if rc == ERR_INVALID:
        raise ValueError(msg)
    raise RuntimeError(f"libmisslap error {rc}: {msg}")
