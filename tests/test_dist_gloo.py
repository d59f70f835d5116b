"""CPU suite, part 3: the multi-GPU driver (sslap_amd/dist.py) with world_size 2 on the gloo backend.
The per-rank compute is the numpy stand-in of tests/_numpy_backend.py; what is under test is the
driver: shard ranges, MAX / MIN exchange sequence, replicated apply, loop control -- the result must be
bit-identical to the single-process oracle, and both ranks must agree."""
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


def _worker(rank, world, port, spec, prob, max_iter, shard_min_K, out):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    import torch.distributed as dist
    import cases
    from _numpy_backend import NumpyBackend
    from sslap_amd.dist import solve_sharded
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    loc, val = cases.synth_inputs(spec)
    b = NumpyBackend(loc, val, prob, rank, world, max_iter=max_iter, shard_min_K=shard_min_K)
    sol = solve_sharded(b)
    out.put((rank, sol.tolist(), b.its, b.nreductions, b.p.tobytes()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("spec,prob,max_iter,shard_min_K", [
    (dict(kind="sparse", n=64, m=64, density=0.1), "max", 10**8, 0),            # every round exchanged
    (dict(kind="sparse", n=60, m=60, density=0.15, ints=3), "max", 10**8, 0),   # cross-rank equal bids
    (dict(kind="sparse", n=40, m=60, density=0.2), "min", 10**8, 0),            # rectangular
    (dict(kind="sparse", n=64, m=64, density=0.1), "max", 7, 0),                # stops at max_iter
    (dict(kind="sparse", n=64, m=64, density=0.1), "max", 10**8, 20),           # big rounds sharded, rest replicated
    (dict(kind="sparse", n=60, m=60, density=0.15, ints=3), "min", 10**8, 30),
])
def test_sharded_driver_world2_matches_oracle(spec, prob, max_iter, shard_min_K):
    import cases
    from oracle import oracle as orc
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, spec, prob, max_iter, shard_min_K, q)) for r in range(2)]
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
    for rank, s, its, nred, pbytes in res:
        assert s == sol.tolist(), f"rank {rank}"
        assert its == o.meta["its"] and nred == o.meta["nreductions"]
        assert pbytes == st["p"].tobytes()


def test_single_process_driver_matches_oracle():
    """world_size 1 (no process group): the driver alone."""
    import cases
    from _numpy_backend import NumpyBackend
    from oracle import oracle as orc
    from sslap_amd.dist import solve_sharded
    spec = dict(kind="sparse", n=80, m=80, density=0.1, ints=5)
    loc, val = cases.synth_inputs(spec)
    b = NumpyBackend(loc, val, "max", 0, 1)
    sol = solve_sharded(b)
    ref = orc.auction_solve(loc=loc, val=val.copy(), problem="max", cardinality_check=False)
    assert np.array_equal(sol, ref["sol"]) and b.its == ref["meta"]["its"]
