/*
 * misslap.h -- C ABI of libmisslap.so, the MI355X (gfx950) auction LAP solver.
 *
 * This is the drop-in boundary for the reference's bid/assign hot path
 * (OllieBoyne/sslap v0.2.5).  The reference has no C ABI: its boundary is the
 * Python -> Cython `cpdef` layer.  Each entry point below names the reference
 * interface it replaces (file:line relative to the reference checkout).  Plain
 * pointers and sizes only; no torch / numpy types.  All functions return
 * MISSLAP_OK (0) or an error code; misslap_last_error() gives the text.
 *
 * Threading: a handle is single-owner (not thread-safe); distinct handles are
 * independent.  Host arrays are borrowed for the duration of a call only.
 * There is NO CPU fallback: every entry point that computes needs a gfx950 GPU
 * and fails with MISSLAP_ERR_NO_DEVICE / MISSLAP_ERR_HIP otherwise.
 */
#ifndef MISSLAP_H
#define MISSLAP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MISSLAP_ABI_VERSION 2
/* ABI history.
 *   1  (rounds 1-2) misslap_options = 88 bytes: the tuning knobs travelled in `reserved[8]`; misslap_meta had no size
 *      field.  Still ACCEPTED: a caller that passes struct_size == 88 to misslap_create* is served with the version-1
 *      meaning of reserved[] and receives the version-1 misslap_meta layout (376 bytes) from misslap_solve /
 *      misslap_finish / misslap_solve_sharded on that handle (INTEGRATION.md section 5 lists both layouts).
 *   2  the knobs are named fields, misslap_meta starts with `struct_size` (the library writes min(struct_size,
 *      sizeof) bytes: a caller built against a shorter version-2 header keeps working when fields are appended),
 *      validity flags of the assignment, misslap_trim_caches. */

#define MISSLAP_OK 0
#define MISSLAP_ERR_INVALID 1    /* malformed arguments / input contract violated */
#define MISSLAP_ERR_HIP 2        /* a HIP runtime call failed */
#define MISSLAP_ERR_NO_DEVICE 3  /* no usable GPU */
#define MISSLAP_ERR_STATE 4      /* call not valid in the handle's current state */

typedef struct misslap_solver misslap_solver;

/* Options of misslap_create.  Zero-initialise, set struct_size = sizeof, then fill. */
typedef struct misslap_options {
    int32_t struct_size;
    int32_t device;          /* HIP device ordinal */
    int32_t maximize;        /* 1 = problem 'max', 0 = 'min' (auction_.pyx:236-237) */
    float eps_start;         /* > 0 overrides eps0 = C/2 (auction_.pyx:251-252) */
    int64_t max_iter;        /* rounds, auction_.pyx:204,:308 */
    int32_t input_on_device; /* loc / val are device pointers already resident in HBM */
    int32_t tail_threshold;  /* rounds with K <= this run in the persistent one-workgroup kernels (launched once per
                                eps-phase: > 16 bidders, 3..16, <= 2); < 0 = library default (192; 40 without candidate lines); 0 = grid kernels
                                only; max 512 */
    int32_t force_f64_values;/* keep 12 B/edge (int32 col + fp64 val) even when values are fp32-exact */
    int32_t profile;         /* 1: record HIP events around the full-scan bid launches, every launch of the full-scan
                                engine and every tail-kernel launch; 2 / 3: around every bid-kernel launch as well */
    int32_t shard_rank;      /* multi-GPU: this process bids for U positions of its shard only */
    int32_t shard_world;     /* number of shards (1 = single GPU) */
    int32_t rounds_per_sync; /* grid rounds enqueued between host status reads; <= 0 = default */
    /* ---- tuning knobs (all 0 = library default; none of them changes a single bit of the result) ---- */
    int32_t tiled_min_K;     /* LDS-tiled full-scan bid kernel: 0 = default threshold (0.7 N where the rows keep candidate
                                lines, 0.3 N otherwise), < 0 = never, > 0 = minimum K */
    int32_t tiled_shape;     /* its launch shape: 0 = chosen from the average (person, tile) segment length, k + 1 =
                                shape k of misslap.hip:kTiledShapes */
    int32_t tiled_force;     /* != 0 together with tiled_min_K > 0: build the tile-major copy whatever the size and
                                density of the problem (tests) */
    int32_t shard_min_K;     /* multi-GPU shard threshold: 0 = default (the full-scan threshold), > 0 = minimum K of a
                                sharded round, < 0 = shard every grid round */
    int32_t cand_mode;       /* candidate lines: 0 = on, 1 = off (every bid scans its whole row; A/B timing, parity
                                tests), 2 = on, but no maintenance pass ahead of the tail kernels (k_refresh_lines) */
    int32_t partial_in_list_order; /* ignored since round 6 (partial rounds of the full-scan engine always take their bidders
                                in person order: the list-order form lost every A/B); the slot keeps the layout */
    int32_t nnz_limit;       /* > 0 lowers the entry limit of a handle (default 2^31 - 1: int32 row pointers), for
                                tests of that guard */
    int32_t cand_build_max_K;/* > 0: k_bid (re)builds lines only in rounds with at most so many bidders */
    int32_t cand_refresh_min;/* r + 1: a line hit with fewer than r live candidates is rebuilt by k_bid (0 = library
                                default, 1 = never) */
    int32_t reserved[7];     /* must be zero */
    void *input_stream;      /* input_on_device only: the hipStream_t the caller's buffers were produced on.  The library
                                then orders its own stream behind that one with an event instead of waiting for the
                                whole device (NULL: hipDeviceSynchronize before the inputs are read -- always safe, but
                                it serialises every other stream of the caller) */
} misslap_options;

/* Result block of misslap_finish: the reference's `meta` dict (auction_.pyx:264,:297-304)
 * plus GPU-side measurements.  The caller sets struct_size first (see the ABI history above). */
typedef struct misslap_meta {
    int32_t struct_size; /* IN: set to sizeof(misslap_meta) before the call; the library writes that many bytes at most */
    int32_t abi_version; /* OUT: MISSLAP_ABI_VERSION of the library */
    float start_eps;     /* auction_.pyx:264 (unrounded fp32) */
    float final_eps;     /* :303 */
    float target_eps;    /* :247 */
    int32_t eCE;         /* :297 */
    int32_t soln_found;  /* :300 */
    int32_t nreductions; /* :299 */
    int64_t its;         /* :298 */
    int64_t n_assigned;  /* :301 */
    int64_t n_rows, n_cols, nnz;
    float obj_f32;       /* get_obj() returns a C float, :489 */
    double obj_f64;      /* the same sum before the cast */
    double setup_ms;     /* :206-207,:265 */
    double solve_ms;     /* :270,:294 */
    /* GPU-side counters */
    uint64_t edges_scanned;      /* sum over all bids of the bidder's row length */
    uint64_t bids_made;
    int64_t grid_rounds;         /* rounds executed by grid kernels */
    int64_t tail_rounds;         /* rounds executed inside the persistent kernel */
    int32_t bytes_per_edge;      /* 8 (int32 col + fp32 val) or 12 (int32 col + fp64 val) */
    int32_t profiled;
    /* valid when options.profile != 0 (HIP-event timings, ms) */
    int64_t bid_launches;        /* grid bid-kernel launches (no-op launches past the end of a phase included) */
    double bid_ms;               /* their summed duration */
    uint64_t bid_edges;          /* edges they scanned */
    int64_t fullscan_launches;   /* of which K == n_rows (every row scanned) */
    double fullscan_ms;
    uint64_t fullscan_edges;
    int64_t tail_launches;
    double tail_ms;
    uint64_t tail_edges;
    int64_t tiled_launches;      /* launches of the LDS-tiled bid kernel (k_bid_tiled), no-ops included */
    double tiled_ms;
    uint64_t tiled_edges;
    int32_t tiled_active;        /* full-scan engine for big rounds: 0 none (k_bid only), 1 k_bid_tiled */
    int32_t tiled_min_K;         /* rounds with K >= this use it */
    uint64_t bid_edges_read;     /* of bid_edges: edges of the rows those launches actually streamed -- the rest are rows a
                                    candidate line answered, counted like the reference counts them but never read.
                                    (These two 8-byte fields were merge_launches / merge_ms, always 0, until round 5.) */
    uint64_t fullscan_edges_read;/* the same for fullscan_edges (the LDS-tiled engine reads every edge it counts) */
    uint64_t shard_edges;        /* multi-GPU: edges scanned in sharded rounds (this rank's share); the rest of
                                    edges_scanned is replicated work, identical on every rank */
    uint64_t cand_hits;          /* bids answered from the person's candidate line (exactly the same bid, no row scan) */
    uint64_t cand_edges;         /* edges of those bidders' rows: part of edges_scanned (reference-equivalent count),
                                    never read from memory */
    double tail_stats[12];       /* tail kernel accounting: [0..2] rounds in chain+solo / team / block mode, [3..5] their
                                    duration in 10-ns ticks, [6] bids, [7] bids answered by candidate lines, [8] line
                                    (re)builds, [9] edges of the rows those lines answered, [10..11] reserved */
    /* validity of the returned assignment, reduced on the device so that `sol` is the only O(N) copy-out (the flags
     * the reference's benchmark harness forms on the host, benchmarking.py:56-64, with size = n_rows and numpy's
     * wrap-around for sol = -1):
     *   complete_assignment  bit 0: np.unique(sol).size == n_rows, bit 1: (sol >= 0).all(), bit 2: (sol < n_rows).all()
     *   valid_assignment     1 if every selected entry (i, sol[i]) exists and its value -- in the caller's sign -- is >= 0 */
    int32_t complete_assignment;
    int32_t valid_assignment;
    int32_t lines_active;        /* 1 = the handle kept candidate lines (0: switched off by option, by row length, or
                                    because eps could fall below the rounding error of a price update, see create) */
    int32_t reserved_i;
    int64_t sharded_rounds;      /* multi-GPU: rounds whose bidders were sharded over the ranks -- each of them issued the
                                    two all-reduces of the exchange step on this rank (0 on a single GPU) */
    /* appended in round 5 */
    int32_t tiled_format;        /* record format of the full-scan engine's tile-major copy (tiled_active = 1): 0 = 6 B/edge
                                    {u16 price slot, f32 value}, 1 = 10 B/edge {slot, f64 value}, 2 / 3 = the same + the u16
                                    stored index of every edge (rows whose columns are not ascending) */
    int32_t phases_with_lines;   /* eps-phases of the solve that ran WITH candidate lines (all of them unless eps fell below
                                    the rounding error of a price update on the way: see misslap_create) */
    int32_t eps_phases;          /* eps-phases of the solve (nreductions + 1 when it ran to its end) */
    int32_t filter_undecided;    /* reserved, always -1 (the counter of round 5's opt-in fp32-tile filter scans, which were
                                    measured slower than the exact scans and removed in round 6; the slot keeps the layout) */
} misslap_meta;

/* Snapshot of the round state (tests / multi-GPU driver). */
typedef struct misslap_status {
    int64_t its;          /* rounds done (auction_.pyx:273) */
    int32_t K;            /* num_unassigned (auction_.pyx:198) */
    int32_t nreductions;
    float eps;
    float target_eps;
    int32_t finished;     /* solve loop has broken out (auction_.pyx:275-281) */
    int32_t error_bits;   /* device-side invariant violations (0 = none) */
    int32_t tail_threshold;   /* effective value (library default resolved) */
    int32_t rounds_per_sync;  /* effective value */
    int32_t shard_min_K;      /* multi-GPU: rounds with K >= this are sharded + exchanged, smaller ones replicated */
    int32_t reserved;
} misslap_status;

/* ---- construction: replaces AuctionSolver.__init__ (auction_.pyx:202-265) as reached through
 * _from_sparse (auction_.pyx:575-617) / _from_matrix (:528-571).
 * loc: int32[nnz][2] (row, col), rows ascending with no empty row (the input contract of
 * cumulative_idxs, auction_.pyx:33-48; violations -> MISSLAP_ERR_INVALID instead of the reference's
 * silent garbage); val: double[nnz].  N = max row + 1, M = max col + 1 (:209-210).  The CSR build,
 * |val| maximum (eps0), sign flip for 'min' and the (col, val) edge layout are done on the GPU.
 * The caller's arrays are never written (the Python front-end reproduces the reference's in-place
 * negation of `val` itself). */
int misslap_create(misslap_solver **out, int64_t nnz, const int32_t *loc, const double *val,
                   const misslap_options *opt);

/* ---- dense ingest: replaces the row-major `v >= 0` scan of _from_matrix (auction_.pyx:546-557).
 * mat: double[n_rows][n_cols] on the host; entries < 0 (and NaN) are invalid.  The scan / stream
 * compaction runs on the GPU.  *nnz_out receives the number of valid entries (for the
 * "Fewer than N valid values" guard of :559 the caller compares it with n_rows BEFORE solving;
 * create fails with MISSLAP_ERR_INVALID when a row is empty, or when the matrix holds 2^31 - 1 or more valid
 * entries -- they are counted in 64 bits on the device, *nnz_out is exact).
 * options.input_on_device: `mat` (here) / `loc`, `val` (misslap_create) are device pointers.  The library then
 * waits for the whole device (hipDeviceSynchronize) before reading them, so buffers still being produced on any
 * stream of the caller are safe to pass -- or, with options.input_stream set, only for the work already enqueued on
 * that stream (an event the solver's stream waits for). */
int misslap_create_dense(misslap_solver **out, int64_t n_rows, int64_t n_cols, const double *mat,
                         const misslap_options *opt, int64_t *nnz_out);

int misslap_destroy(misslap_solver *h);

/* N, M, nnz as the solver sees them (auction_.pyx:209-212). */
int misslap_dims(const misslap_solver *h, int64_t *n_rows, int64_t *n_cols, int64_t *nnz);

/* ---- the whole solve loop: replaces AuctionSolver.solve() (auction_.pyx:268-306).
 * person_to_object_out: int32[n_rows] on the host (-1 = unassigned), may be NULL. */
int misslap_solve(misslap_solver *h, int32_t *person_to_object_out, misslap_meta *meta);

/* ---- stepwise interface (multi-GPU driver, round-level parity tests).  One grid round is
 *   misslap_round_bid      bid phase (auction_.pyx:339-365) over this shard's bidders + per-object
 *                          maximum of the shard's bids (first half of :375-385).  Only rounds with
 *                          K >= status.shard_min_K are sharded (and need the two exchanges); in smaller
 *                          rounds every rank bids for every list position and no exchange is needed.
 *   [exchange: all-reduce MAX over the best-key buffer, see misslap_exchange_buffers]
 *   misslap_round_tiebreak earliest bidder in list order wins equal bids (strict '>' of :379)
 *   [exchange: all-reduce MIN over the best-position buffer]
 *   misslap_round_apply    assignment phase (:388-429) + push_all_left (:137-162, :430); its += 1
 * Every call is a no-op once K == 0, K <= tail_threshold or its >= max_iter, so a driver may
 * enqueue rounds blindly and read the status afterwards. */
int misslap_round_bid(misslap_solver *h);
int misslap_round_tiebreak(misslap_solver *h);
int misslap_round_apply(misslap_solver *h);
/* All remaining rounds with 0 < K <= tail_threshold, inside the persistent kernels (no host round trip until K == 0). */
int misslap_run_tail(misslap_solver *h);
/* Synchronise and read the round state. */
int misslap_get_status(misslap_solver *h, misslap_status *st);
/* Loop control of solve() after a round (auction_.pyx:275-292): terminate(), eps reduction and
 * assignment reset.  Sets *finished. */
int misslap_phase_end(misslap_solver *h, int32_t *finished);
/* eps-complementary-slackness test, eCE_satisfied (auction_.pyx:443-485), on the current state. */
int misslap_check_ece(misslap_solver *h, float eps, int32_t *satisfied);
/* meta + result copy-out (auction_.pyx:297-306). */
int misslap_finish(misslap_solver *h, int32_t *person_to_object_out, misslap_meta *meta);

/* Device pointers of the two per-object exchange buffers (int64 best key = bid bits + 1, 0 = none;
 * int32 best position, INT32_MAX = none), n_objects entries each, for RCCL all-reduces. */
int misslap_exchange_buffers(misslap_solver *h, void **best_key, void **best_pos, int64_t *n_objects);
/* Run all of the handle's GPU work on this hipStream_t (default: a private stream). */
int misslap_set_stream(misslap_solver *h, void *hip_stream);

/* ---- multi-GPU: the exchange step behind the C ABI (one process per GPU, persons of a round sharded over the
 * ranks; the reference has no counterpart: it is single-threaded, auction_.pyx:6).
 * A communicator provides the two collectives a sharded round needs -- all-reduce MAX over the int64 best-key buffer
 * and all-reduce MIN over the int32 best-position buffer (misslap_exchange_buffers) -- issued on the solver's HIP
 * stream between the round kernels, with no host read inside a round.
 *   RCCL over xGMI:  rank 0 calls misslap_rccl_unique_id, the 128 bytes are handed to every rank by any means the
 *                    application has (file, socket, MPI, torch.distributed object broadcast ...), every rank calls
 *                    misslap_comm_init_rccl.  librccl.so.1 is opened at run time (dlopen).
 *   custom:          caller-provided callbacks (another transport; the tests' gloo / in-process stand-ins).
 * Every rank creates its handle on the same input with options.shard_rank / shard_world set and calls
 * misslap_solve_sharded; every rank receives the same assignment. */
typedef struct misslap_comm misslap_comm;
#define MISSLAP_RCCL_ID_BYTES 128
typedef struct misslap_comm_ops {
    int32_t struct_size;
    int32_t rank, world;
    int32_t reserved;
    void *ctx;
    /* in-place all-reduce of `count` elements of a DEVICE buffer, ordered after the work already enqueued on
     * `hip_stream` and before work enqueued afterwards; return 0 on success */
    int (*allreduce_max_i64)(void *ctx, void *buf, int64_t count, void *hip_stream);
    int (*allreduce_min_i32)(void *ctx, void *buf, int64_t count, void *hip_stream);
} misslap_comm_ops;
int misslap_rccl_unique_id(void *id_out /* MISSLAP_RCCL_ID_BYTES */);
/* Self-check of the RCCL binding, needs no GPU: opens librccl exactly as misslap_comm_init_rccl does and reports how many
 * of the entry points the exchange uses were resolved (*n_symbols of MISSLAP_RCCL_SYMBOLS: ncclGetUniqueId,
 * ncclCommInitRank, ncclCommDestroy, ncclAllReduce, ncclGetErrorString, ncclCommCount), the enumerator VALUES the
 * library passes to ncclAllReduce -- enums[0..3] = ncclInt32, ncclInt64, ncclMax, ncclMin as compiled in (rccl.h:
 * ncclDataType_t, ncclRedOp_t) --, enums[4] = the size of the unique id it assumes, enums[5] = ncclGetVersion() of the
 * copy it bound to, and that copy's path.  A test compares them with the installed rccl.h (tests/test_cabi.py). */
#define MISSLAP_RCCL_SYMBOLS 6
int misslap_rccl_selfcheck(int32_t *n_symbols, int32_t enums[6], char *lib_path, int32_t lib_path_len);
int misslap_comm_init_rccl(misslap_comm **out, const void *unique_id, int32_t rank, int32_t world, int32_t device);
int misslap_comm_init_custom(misslap_comm **out, const misslap_comm_ops *ops);
int misslap_comm_destroy(misslap_comm *comm);
/* What a communicator is, as the transport itself reports it: *kind = 1 RCCL / 0 custom; *rank, *world as given at
 * creation; *transport_ranks = ncclCommCount of the RCCL communicator (the number of ranks RCCL itself sees -- a
 * multi-GPU run proves with it that N ranks took part), for a custom one the world size of its ops.  Any pointer
 * may be NULL. */
int misslap_comm_info(const misslap_comm *comm, int32_t *kind, int32_t *rank, int32_t *world, int32_t *transport_ranks);
/* AuctionSolver.solve() (auction_.pyx:268-306) over all ranks of `comm` (NULL: no exchange, a single rank).  Only
 * rounds with K >= status.shard_min_K are sharded and exchanged; all others are replicated. */
int misslap_solve_sharded(misslap_solver *h, misslap_comm *comm, int32_t *person_to_object_out, misslap_meta *meta);

/* The same loop over caller-provided round operations: what misslap_solve_sharded runs with the handle's own
 * operations.  Exposed so that the loop -- shard / replicate decision, collective sequence, loop control -- can be
 * driven without a GPU (tests/test_dist_gloo.py: numpy stand-ins for the round kernels, gloo for the exchange). */
typedef struct misslap_round_ops {
    int32_t struct_size;
    int32_t tail_threshold;
    int32_t shard_min_K;
    int32_t rounds_per_sync;
    int64_t max_iter;
    void *ctx;
    int (*status)(void *ctx, int64_t *K, int64_t *its); /* synchronise and read the round state */
    int (*round_bid)(void *ctx);
    int (*round_tiebreak)(void *ctx);
    int (*round_apply)(void *ctx);
    int (*run_tail)(void *ctx);
    int (*phase_end)(void *ctx, int32_t *finished);
    void *best_key;    /* int64[n_objects] exchange buffer (device memory for the GPU operations) */
    void *best_pos;    /* int32[n_objects] */
    int64_t n_objects;
    void *stream;      /* handed to the communicator's callbacks */
    /* Optional (both or neither): a status read that does not drain the queue.  status_post enqueues a copy of the
     * round state behind everything issued so far into slot 0 / 1; status_take waits for THAT copy only.  With them
     * the replicated rounds are issued in batches whose status trails by one batch (K never grows inside an
     * eps-phase and a round that is not live is a no-op, so a batch issued on a stale "go on" is harmless). */
    int (*status_post)(void *ctx, int32_t slot);
    int (*status_take)(void *ctx, int32_t slot, int64_t *K, int64_t *its);
    int32_t large_round_K;          /* while K > this ... */
    int32_t rounds_per_sync_large;  /* ... a batch has so many rounds (<= 0: rounds_per_sync) */
} misslap_round_ops;
int misslap_drive_sharded(const misslap_round_ops *ops, misslap_comm *comm);

/* ---- state copy-out for parity tests (all host buffers, any may be NULL):
 * prices double[M] (auction_.pyx:169), unassigned list int32[N] (first K valid, :199),
 * person_to_object int32[N] (:177), object_to_person int32[M] (:178). */
int misslap_get_state(misslap_solver *h, double *prices, int32_t *unassigned, int32_t *person_to_object,
                      int32_t *object_to_person);

/* Device properties of the GPU the handle runs on (name buffer >= 128 bytes). */
int misslap_device_info(int32_t device, char *name, int32_t name_len, int32_t *compute_units,
                        int64_t *hbm_bytes);
/* The device's UUID as 32 hex digits + NUL (buffer >= 33 bytes): tells two ranks that share a GPU from two that do not. */
int misslap_device_uuid(int32_t device, char *uuid_hex, int32_t len);
/* What the HBM of THIS device delivers to a streaming kernel of the library, measured now: a read-only pass (every
 * workgroup walks one contiguous chunk, four 16-byte non-temporal loads in flight per lane) and a copy of the same
 * shape over `bytes` bytes, `reps` timed launches each; GB/s of bytes read, and of bytes read + written.  The "achievable" peak next to the data sheet's 8 TB/s that
 * the roofline figures of bench.py are quoted against. */
int misslap_measure_hbm(int32_t device, int64_t bytes, int32_t reps, double *read_GBs, double *copy_GBs);

/* Feasibility guard of the front-end: maximum bipartite matching (Hopcroft-Karp) on the host, the reference's
 * c_hopcroft_solve / sslap.hopcroft_solve (feasibility_.pyx:95-283; called at auction_.pyx:562-566, :608-612).
 * loc: int32[nnz][2] (row, col), rows ascending.  size: cardinality; left_pairings int32[n_rows] / right_pairings
 * int32[n_cols] (-1 = unmatched, either may be NULL) reproduce the reference's result() arrays.  Needs no GPU. */
int misslap_hopcroft_karp(const int32_t *loc, int64_t nnz, int32_t n_rows, int32_t n_cols, int32_t *size,
                          int32_t *left_pairings, int32_t *right_pairings);

/* The same maximum matching computed on the GPU (BFS-layered, csrc/kernels_matching.hpp): equal cardinality -- the only
 * thing that reaches the auction path (auction_.pyx:565, :611) -- but not necessarily the same pairings.  loc is a
 * host array; *phases (may be NULL) receives the number of augmentation phases.  Needs a GPU. */
int misslap_matching_gpu(const int32_t *loc, int64_t nnz, int32_t n_rows, int32_t n_cols, int32_t device, int32_t *size,
                         int32_t *left_pairings, int32_t *right_pairings, int32_t *phases);
/* The same matcher on the graph a solver handle already holds in device memory (no host copy of the entries, no second
 * upload): what the front-end's feasibility guard (auction_.pyx:562-566, :608-612) uses once the handle exists. */
int misslap_matching_of(misslap_solver *h, int32_t *size, int32_t *phases);

/* Host-side caches: a destroyed handle parks its idle HIP stream, pinned status mirror and events (at most 8 bundles)
 * and its freed device blocks for the next handle -- BY DEFAULT up to 4 GB in total (never more than 1 / 64 of the
 * device's memory), blocks of up to 1 GB, at most 64 blocks: i.e. after misslap_destroy the process may still hold up to
 * 4 GB of HBM that an embedding application (a PyTorch process, say) does not see as free until it calls
 * misslap_trim_caches or lowers the limits (misslap_set_cache_limits(0, 0, 0) = park nothing).  This releases all of them
 * (streams destroyed, device and pinned memory freed); *freed_bytes (may be NULL) receives the device bytes returned.
 * Call it when the embedding application needs the memory back; never while another thread creates / destroys handles
 * on a stream that may still use a parked block (the entry point synchronises every device it frees on). */
int misslap_trim_caches(int64_t *freed_bytes);
/* Limits of the device-block cache: total bytes parked, largest block parked, number of blocks.  Defaults: min(4 GB,
 * device memory / 64) / a quarter of that / 64 blocks -- the blocks of one or two problems of the BASELINE sizes (0.9 GB
 * per C3 handle), because hipMalloc + hipFree of those cost a millisecond per create / destroy pair and hipFree waits
 * for EVERY stream of the device; MISSLAP_BLOCK_CACHE_MB in the environment sets the first two at start-up (0 = park
 * nothing).  An application that solves many LARGE problems at a time raises them (bench.py --concurrent: C3, 16 at a
 * time: 4 GB per solve in flight); one that must not lose HBM to the library lowers them. */
int misslap_set_cache_limits(int64_t max_total_bytes, int64_t max_block_bytes, int32_t max_blocks);

/* ---- many independent problems at a time (round 5; no counterpart in the reference, whose own harness solves its
 * problems in a loop: benchmarking.py:84-142).  During ~90 % of a solve one problem occupies ONE of the GPU's 256 compute
 * units, and a GPU fed from many queues retires only ~90 000 launches per second over all of them, so solves issued from
 * many host threads stall at 6-8x the single-solve throughput.  misslap_solve_batch solves the n handles in LOCKSTEP:
 * groups of `group_size` problems (0 = default, 12) share one HIP stream and every launch of the solve loop that several of
 * them issue at the same point is ONE launch (csrc/host_batch.hpp).  Each handle goes through exactly the sequence of
 * kernels misslap_solve would have launched for it: person_to_object_out[k] / meta_out[k] are what misslap_solve(handles[k])
 * returns, bit for bit -- except the TIMING fields of the meta (solve_ms and the per-kernel times): inside a batch they are
 * wall time of the handle's fiber, which includes the other problems of its group.  The handles must be unsolved, unsharded, unprofiled, on one device, and have the same number of
 * persons (rows; objects and entries may differ).  meta_out: n pointers to structs with struct_size set (the array and
 * any of its entries may be NULL, like the output pointers); *info (may be NULL) says how many launches went out for how many recorded. */
typedef struct misslap_batch_info {
    int32_t groups;            /* streams / host threads used */
    int32_t reserved;
    int64_t calls_recorded;    /* launches + asynchronous copies / fills the n solve loops asked for */
    int64_t launches_issued;   /* ... and what went onto the streams after merging */
    double wall_ms;
    /* where the host threads of the groups spent their time (summed over groups): running the problems' solve loops up to
     * their next wait, merging + issuing launches, waiting for the device (polling status words / draining the stream) */
    double host_ms_fibers, host_ms_flush, host_ms_wait;
} misslap_batch_info;
int misslap_solve_batch(misslap_solver *const *handles, int32_t n, int32_t *const *person_to_object_out,
                        misslap_meta *const *meta_out, int32_t group_size, misslap_batch_info *info);

const char *misslap_last_error(void);
int misslap_abi_version(void);

#ifdef __cplusplus
}
#endif
#endif /* MISSLAP_H */
