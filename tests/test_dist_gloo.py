"""CPU suite, part 3: the sharded solve loop of the LIBRARY (misslap_drive_sharded, csrc/host_comm.hpp -- the loop
misslap_solve_sharded runs on a GPU handle) with world_size 2, 3 and 8 on the gloo backend (uneven shard ranges
[K r / W, K (r + 1) / W), ranks with an EMPTY range when K < W).  The per-rank round operations
are the numpy stand-ins of tests/_numpy_backend.py, handed to the C loop as callbacks; the exchange is a custom
communicator whose callbacks all-reduce the (host) buffers with gloo.  Under test: shard ranges, MAX / MIN exchange
sequence, replicated apply, loop control -- the result must be bit-identical to the single-process oracle, and
both ranks must agree."""
import ctypes as C
import os
import socket
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def drive_with_c_loop(b, comm, trailing=False):
    """Run misslap_drive_sharded over a NumpyBackend (`b`) and a sslap_amd.dist.Comm (or None).  trailing: also hand
    the loop the optional status_post / status_take pair (status reads that trail the rounds by one batch)."""
    from sslap_amd import _lib
    lib = _lib.load()
    calls = []

    def status(_ctx, K, its):
        K[0], its[0] = b.status()
        return 0

    def op(fn):
        def cb(_ctx):
            calls.append(fn.__name__)
            fn()
            return 0
        return _lib._OP0(cb)

    def phase_end(_ctx, fin):
        fin[0] = 1 if b.phase_end() else 0
        return 0

    o = _lib.RoundOps()
    o.struct_size = C.sizeof(_lib.RoundOps)
    o.tail_threshold, o.shard_min_K, o.rounds_per_sync, o.max_iter = b.thr, b.shard_min_K, b.rounds_per_sync, b.max_iter
    keep = (_lib._OP_STATUS(status), op(b.round_bid), op(b.round_tiebreak), op(b.round_apply), op(b.run_tail),
            _lib._OP_PHASE(phase_end))
    o.status, o.round_bid, o.round_tiebreak, o.round_apply, o.run_tail, o.phase_end = keep
    o.best_key, o.best_pos = b.best_key.data_ptr(), b.best_pos.data_ptr()
    o.n_objects = b.M
    if trailing:
        slots = {}

        def post(_ctx, slot):
            slots[slot] = b.status()  # the stand-in is synchronous: the "copy" is taken at once
            calls.append("status_post")
            return 0

        def take(_ctx, slot, K, its):
            K[0], its[0] = slots.pop(slot)
            return 0
        keep += (_lib._OP_POST(post), _lib._OP_TAKE(take))
        o.status_post, o.status_take = keep[-2:]
        o.large_round_K, o.rounds_per_sync_large = 24, 1
    _lib.check(lib.misslap_drive_sharded(C.byref(o), comm._c if comm is not None else None))
    return b.finish(), calls


def host_gloo_comm(rank, world, counter):
    """Custom communicator over HOST buffers: the callbacks all-reduce them with gloo."""
    import torch
    import torch.distributed as dist
    from sslap_amd.dist import Comm

    def reduce(op, dtype, torch_dtype):
        def fn(ptr, count, _stream):
            arr = np.ctypeslib.as_array(C.cast(ptr, C.POINTER(dtype)), (count,))
            t = torch.from_numpy(arr)
            assert t.dtype == torch_dtype
            dist.all_reduce(t, op=op)
            counter.append(op)
        return fn
    return Comm.custom(rank, world, reduce(dist.ReduceOp.MAX, C.c_int64, torch.int64),
                       reduce(dist.ReduceOp.MIN, C.c_int32, torch.int32))


def _worker(rank, world, port, spec, prob, max_iter, shard_min_K, out):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    import torch.distributed as dist
    import cases
    from _numpy_backend import NumpyBackend
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    loc, val = cases.synth_inputs(spec)
    b = NumpyBackend(loc, val, prob, rank, world, max_iter=max_iter, shard_min_K=shard_min_K)
    exchanges = []
    comm = host_gloo_comm(rank, world, exchanges)
    sol, calls = drive_with_c_loop(b, comm)
    out.put((rank, sol.tolist(), b.its, b.nreductions, b.p.tobytes(), len(exchanges), calls.count("round_bid")))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,spec,prob,max_iter,shard_min_K", [
    (2, dict(kind="sparse", n=64, m=64, density=0.1), "max", 10**8, 0),            # every round exchanged
    (2, dict(kind="sparse", n=60, m=60, density=0.15, ints=3), "max", 10**8, 0),   # cross-rank equal bids
    (2, dict(kind="sparse", n=40, m=60, density=0.2), "min", 10**8, 0),            # rectangular
    (2, dict(kind="sparse", n=64, m=64, density=0.1), "max", 7, 0),                # stops at max_iter
    (2, dict(kind="sparse", n=64, m=64, density=0.1), "max", 10**8, 20),           # big rounds sharded, rest replicated
    (2, dict(kind="sparse", n=60, m=60, density=0.15, ints=3), "min", 10**8, 30),
    # three ranks: K r / 3 is uneven in almost every round
    (3, dict(kind="sparse", n=64, m=64, density=0.1), "max", 10**8, 0),
    (3, dict(kind="sparse", n=61, m=61, density=0.15, ints=3), "min", 10**8, 0),   # ties across uneven shards
    (3, dict(kind="sparse", n=64, m=64, density=0.1), "max", 10**8, 20),
    # eight ranks (BASELINE config 5's world size): shards of 8 positions and fewer, EMPTY shards as soon as K < 8
    (8, dict(kind="sparse", n=64, m=64, density=0.1), "max", 10**8, 0),
    (8, dict(kind="sparse", n=60, m=60, density=0.15, ints=3), "max", 10**8, 0),
    (8, dict(kind="sparse", n=6, m=6, density=0.7), "max", 10**8, 0),              # K < W from the first round on
    (8, dict(kind="sparse", n=40, m=60, density=0.2), "min", 10**8, 25),           # rectangular, mixed regime
])
def test_sharded_c_loop_matches_oracle(world, spec, prob, max_iter, shard_min_K, built_lib):
    import cases
    from oracle import oracle as orc
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, spec, prob, max_iter, shard_min_K, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=240) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    loc, val = cases.synth_inputs(spec)
    o = orc.from_sparse(loc, val.copy(), problem=prob, max_iter=max_iter, cardinality_check=False)
    sol = o.solve()
    st = o.state()
    for rank, s, its, nred, pbytes, n_exchanges, n_bids in res:
        assert s == sol.tolist(), f"rank {rank}"
        assert its == o.meta["its"] and nred == o.meta["nreductions"]
        assert pbytes == st["p"].tobytes()
        assert n_exchanges % 2 == 0 and n_exchanges > 0          # MAX and MIN come in pairs ...
        if shard_min_K == 0:
            assert n_exchanges == 2 * n_bids                       # ... one pair per (sharded) round
        else:
            assert n_exchanges < 2 * n_bids                        # the small rounds are replicated, not exchanged
    assert len({r[5] for r in res}) == 1 and len(res) == world     # every rank issued the same collectives


def test_single_rank_c_loop_matches_oracle(built_lib):
    """world_size 1, no communicator: the loop alone."""
    import cases
    from _numpy_backend import NumpyBackend
    from oracle import oracle as orc
    spec = dict(kind="sparse", n=80, m=80, density=0.1, ints=5)
    loc, val = cases.synth_inputs(spec)
    b = NumpyBackend(loc, val, "max", 0, 1)
    sol, calls = drive_with_c_loop(b, None)
    ref = orc.auction_solve(loc=loc, val=val.copy(), problem="max", cardinality_check=False)
    assert np.array_equal(sol, ref["sol"]) and b.its == ref["meta"]["its"]
    assert calls.count("round_bid") == calls.count("round_apply") == ref["meta"]["its"]


@pytest.mark.parametrize("rps", [1, 3, 5])
def test_single_rank_c_loop_trailing_status(rps, built_lib):
    """The replicated rounds in batches whose status read trails by one batch (what the GPU handle's operations do):
    rounds issued on a stale "go on" are no-ops, the result and the round count do not change."""
    import cases
    from _numpy_backend import NumpyBackend
    from oracle import oracle as orc
    spec = dict(kind="sparse", n=80, m=80, density=0.1, ints=5)
    loc, val = cases.synth_inputs(spec)
    b = NumpyBackend(loc, val, "max", 0, 1, rounds_per_sync=rps, shard_min_K=60)
    sol, calls = drive_with_c_loop(b, None, trailing=True)
    ref = orc.auction_solve(loc=loc, val=val.copy(), problem="max", cardinality_check=False)
    assert np.array_equal(sol, ref["sol"]) and b.its == ref["meta"]["its"]
    assert calls.count("status_post") > 0
    assert calls.count("round_bid") >= ref["meta"]["its"]  # live rounds + the no-ops behind the end of a phase


def test_comm_argument_validation(built_lib):
    from sslap_amd import _lib
    lib = _lib.load()
    h = C.c_void_p()
    ops = _lib.CommOps()
    assert lib.misslap_comm_init_custom(C.byref(h), C.byref(ops)) == _lib.ERR_INVALID  # struct_size / callbacks unset
    assert lib.misslap_drive_sharded(None, None) == _lib.ERR_INVALID
    assert lib.misslap_comm_destroy(None) == _lib.MISSLAP_OK


def test_thread_group_collectives():
    """sslap_amd.dist.ThreadGroup (the ranks of an in-process rehearsal are threads): barrier, all-gather and all-reduce
    give every rank the same result in rank order; a rank that fails breaks the barrier instead of hanging the others."""
    import threading
    from sslap_amd.dist import ThreadGroup
    W = 8
    g = ThreadGroup(W, timeout_s=30.0)
    out = [None] * W

    def rank_main(r):
        got = []
        for it in range(50):
            a = np.arange(16, dtype=np.int64) * (r + 1) + it
            got.append((g.all_reduce(r, a, "max").tolist(), g.all_reduce(r, a.astype(np.int32), "min").tolist(),
                        g.all_gather(r, (r, it)), int(g.all_reduce(r, np.int64(r + it), "sum"))))
            g.barrier()
        out[r] = got
    th = [threading.Thread(target=rank_main, args=(r,)) for r in range(W)]
    for t in th:
        t.start()
    for t in th:
        t.join(60)
    assert all(o == out[0] for o in out) and len(out[0]) == 50
    mx, mn, gathered, sm = out[0][3]
    assert mx == (np.arange(16) * W + 3).tolist() and mn == (np.arange(16) + 3).tolist()
    assert gathered == [(r, 3) for r in range(W)] and sm == sum(r + 3 for r in range(W))
    # a failing rank: the others raise BrokenBarrierError within the timeout instead of waiting for ever
    g2 = ThreadGroup(2, timeout_s=5.0)
    res = []

    def waiter():
        try:
            g2.barrier()
            res.append("passed")
        except threading.BrokenBarrierError:
            res.append("broken")
    t = threading.Thread(target=waiter)
    t.start()
    g2.abort()
    t.join(10)
    assert res == ["broken"]
