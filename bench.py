#!/usr/bin/env python3
"""bench.py -- the hot path (device COO -> CSR build -> eps-scaling auction -> assignment) on the
BASELINE.json headline workload (C3: 200k x 200k, density 0.1 %, fp32-exact costs, problem='max').

Contract (one JSON line on rank 0):
  a "step" = one complete solve of the workload, inputs already resident in HBM;
  value    = edges scanned by all ranks / wall time of the K timed steps, in Medges/s
             (edges scanned = sum over every bid of the bidder's CSR row length);
  roofline = the bid kernel (k_bid, the "bid-phase CSR scan"): algorithmic 8 B/edge over the summed
             HIP-event durations of its launches inside the timed steps, against 8 TB/s HBM;
  cpu_baseline = the C oracle (oracle/, a port of the reference's algorithm) on a bounded sample of the
             same workload, one host core, rank 0, N = 1 only.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--config C3] [--no-cpu] [--mode sharded|replicas]
N > 1: one rank per GPU.  Under torch.distributed.run (RANK / WORLD_SIZE in the environment) this process IS a rank;
without a launcher `python bench.py --gpus N` starts the N ranks itself -- fresh child processes, spawned before this
process has imported torch or touched the GPU; the parent only relays rank 0's JSON line.  MISSLAP_DIST_BACKEND=gloo lets
the rank processes share ONE GPU (rehearsal; a one-GPU box admits about four of them on its card), MISSLAP_DIST_BACKEND=
threads runs the N ranks as threads of this process (rehearsal of N = 8 on one GPU).  Mode 'sharded': persons of
each big round sharded over the ranks, RCCL all-reduces on the per-object best bids issued by the library itself
(misslap_solve_sharded, include/misslap.h; sslap_amd/dist.py only creates the communicator); mode 'replicas': one
independent solve per GPU, no collective.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X spec (MI355X_MICROARCH.md); measured copy peak is ~6300


def cpu_baseline(cfg, budget_rounds, whole, args=None):
    """Oracle (port of the reference, single thread like the reference, faithful O(M) assignment walk) on the same
    workload: the first `budget_rounds` rounds as the bounded sample, and -- `whole` -- the complete solve in the
    same run (about 30-50 s at C3; the sample over-weights the big bandwidth-bound rounds of the first phase)."""
    import numpy as np
    from oracle import oracle as orc
    from sslap_amd import synth
    loc, val = workload(args, synth) if args is not None else synth.gen_config(cfg)
    s = orc.from_sparse(loc, val, problem="max", max_iter=10**8, cardinality_check=False)
    s.set_timing(True)
    t0 = time.perf_counter()
    sample = None
    rounds = 0
    while True:
        done = s.step()
        rounds += 1
        if rounds == budget_rounds or (done and sample is None):
            m = s.raw_meta()  # (one O(nnz) pass for the eCE / objective fields: outside the clock)
            t1 = time.perf_counter()
            sample = (int(m.edges_scanned), t1 - t0, float(m.t_bid), rounds)
            t0 += time.perf_counter() - t1
            if not whole:
                break
        if done:
            break
    dt = time.perf_counter() - t0
    e_s, dt_s, tbid_s, r_s = sample
    out = {
        "value": round(e_s / dt_s / 1e6, 2), "unit": "Medges/s", "cores": 1, "kind": "port",
        "sample": f"first {r_s} rounds of {cfg} (oracle/auction_oracle.c, faithful O(M) assignment walk), "
                  f"{e_s} edges in {dt_s:.2f} s",
        "bid_phase_only_medges_s": round(e_s / max(tbid_s, 1e-9) / 1e6, 2),
        "host_cpu": _cpu_name(), "host_cores_available": os.cpu_count(),
    }
    if whole:
        m = s.raw_meta()
        out.update(whole_solve_medges_s=round(int(m.edges_scanned) / dt / 1e6, 2), whole_solve_s=round(dt, 2),
                   whole_solve_rounds=int(m.its), whole_solve_edges=int(m.edges_scanned))
        # BASELINE.md section 3's second CPU figure: the same port with the assignment phase visiting only the objects that
        # received a bid (O(#bids) per round) instead of the reference's walk over all M objects (auction_.pyx:394) --
        # what a maintainer's first optimisation of the CPU path would give; same assignment (checked below and in
        # tests/test_oracle_golden.py), never the parity oracle
        sol_faithful = np.ctypeslib.as_array(orc.lib().oracle_person_to_object(s._h), (s.N,)).copy()
        loc2, val2 = workload(args, synth) if args is not None else synth.gen_config(cfg)
        s2 = orc.from_sparse(loc2, val2, problem="max", max_iter=10**8, cardinality_check=False)
        s2.set_assign_by_bidders(True)
        t2 = time.perf_counter()
        sol2 = s2.solve()
        dt2 = time.perf_counter() - t2
        out.update(optimised_whole_solve_s=round(dt2, 2),
                   optimised_whole_solve_medges_s=round(int(s2.raw_meta().edges_scanned) / dt2 / 1e6, 2),
                   optimised_kind="port, assignment phase through the round's bidders (O(#bids)) instead of the O(M) walk",
                   optimised_same_assignment=bool(np.array_equal(sol2, sol_faithful)))
    return out


def workload(args, synth):
    """The synthetic input of the run: a BASELINE config (sslap_amd.synth, seed 1), optionally with values that are not
    fp32-exact and / or a random stored order inside the rows."""
    loc, val = synth.gen_config(args.config)
    if args.values == "f64":
        val = val * (1.0 + 2.0 ** -30)  # no longer fp32-representable: the 12 B/edge layout
    if args.shuffle_rows:
        loc, val = synth.shuffle_within_rows(loc, val, 5)
    return loc, val


def launch_ranks(n, argv):
    """`--gpus N` without a launcher: start the N ranks as fresh child processes of this script (RANK / LOCAL_RANK /
    WORLD_SIZE / MASTER_* set, rendezvous on 127.0.0.1), relay rank 0's stdout, exit with the worst return code.
    Nothing here imports torch or calls into HIP: a process that has initialised the GPU must never be replaced or
    forked, and the children must be the first to open the device."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    if os.environ.get("MISSLAP_BENCH_TRACE"):
        print(f"bench.py parent: spawning {n} ranks, torch_imported_in_parent={'torch' in sys.modules}", file=sys.stderr)
    import tempfile
    procs = []
    with tempfile.TemporaryFile() as out0:  # rank 0's stdout (a file, not a pipe: nothing to drain while we wait)
        for r in range(n):
            env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                       MASTER_PORT=str(port))
            procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                          stdout=out0 if r == 0 else subprocess.DEVNULL))
        deadline = time.time() + float(os.environ.get("MISSLAP_BENCH_TIMEOUT_S", 3000))
        try:
            while any(p.poll() is None for p in procs):
                if any(p.poll() not in (None, 0) for p in procs) or time.time() > deadline:
                    break  # one rank failed (or the run hangs): the others would wait in a collective for ever
                time.sleep(0.2)
        finally:
            for p in procs:
                if p.poll() is None:
                    p.kill()
        rcs = [p.wait() for p in procs]
        out0.seek(0)
        sys.stdout.write(out0.read().decode())
        sys.stdout.flush()
    raise SystemExit(0 if all(rc == 0 for rc in rcs) else next(rc for rc in rcs if rc != 0) or 1)


class TorchRanks:
    """The ranks are processes of a torch.distributed group (RCCL, or gloo for a one-GPU rehearsal)."""

    def __init__(self, rank, world, dist, torch):
        self.rank, self.world, self._dist, self._torch = rank, world, dist, torch

    def barrier(self):
        if self.world > 1:
            self._dist.barrier()
        self._torch.cuda.synchronize()

    def _reduce(self, x, dtype, op):
        if self.world == 1:
            return x
        t = self._torch.tensor([x], dtype=dtype, device="cuda")
        self._dist.all_reduce(t, op=op)
        return t.item()

    def max_f(self, x):
        return float(self._reduce(x, self._torch.float64, self._dist.ReduceOp.MAX))

    def sum_i(self, x):
        return int(self._reduce(x, self._torch.int64, self._dist.ReduceOp.SUM))

    def gather(self, obj):
        if self.world == 1:
            return [obj]
        out = [None] * self.world
        self._dist.all_gather_object(out, obj)
        return out


class ThreadRanks:
    """The ranks are threads of this process (sslap_amd.dist.ThreadGroup): same interface."""

    def __init__(self, rank, group, torch):
        self.rank, self.world, self._g, self._torch = rank, group.world, group, torch

    def barrier(self):
        self._g.barrier()
        self._torch.cuda.synchronize()
        self._g.barrier()

    def max_f(self, x):
        return float(max(self._g.all_gather(self.rank, x)))

    def sum_i(self, x):
        return int(sum(self._g.all_gather(self.rank, x)))

    def gather(self, obj):
        return self._g.all_gather(self.rank, obj)


def measured_hbm_peaks(device):
    """What the HBM of THIS GPU delivers to the library's own streaming kernels in THIS process (misslap_measure_hbm:
    16 bytes per lane, 8 workgroups per CU, 1 GiB): a read-only pass and a copy -- the 'achievable' rates next to the
    8 TB/s of the data sheet (report both).  The roofline kernel only reads, so its fraction of the
    measured peak is quoted against the READ rate."""
    import ctypes as C
    from sslap_amd import _lib
    rd, cp = C.c_double(), C.c_double()
    _lib.check(_lib.load().misslap_measure_hbm(int(device), 1 << 30, 20, C.byref(rd), C.byref(cp)))
    return rd.value, cp.value


def run_concurrent(B, steps, make_solver, edges_per_solve, want_sha, digest):
    """B independent solves at a time on B handles / B HIP streams driven by B host threads (ctypes releases the GIL
    for the whole solve): what one GPU delivers when the application has many LAPs -- the reference's own harness
    solves them in a loop (benchmarking.py:84-142).  During 87 % of a C3 solve a single problem occupies ONE of the
    256 CUs (the tail kernels), so independent problems overlap almost freely."""
    import threading
    start = threading.Barrier(B + 1)
    shas, errs = [], []
    lock = threading.Lock()

    def worker():
        try:
            start.wait()
            for _ in range(steps):
                s = make_solver()
                sol = s.solve()
                d = digest(sol)
                with lock:
                    shas.append(d)
                del s
        except Exception as e:  # noqa: BLE001
            with lock:
                errs.append(repr(e))
    th = [threading.Thread(target=worker) for _ in range(B)]
    for t in th:
        t.start()
    start.wait()
    t0 = time.perf_counter()
    for t in th:
        t.join()
    wall = time.perf_counter() - t0
    if errs:
        raise SystemExit(f"concurrent solves failed: {errs[:2]}")
    n = B * steps
    return {"B": B, "solves": n, "wall_ms": round(1e3 * wall, 3), "ms_per_solve": round(1e3 * wall / n, 3),
            "aggregate_medges_s": round(edges_per_solve * n / wall / 1e6, 2),
            "all_sha256_equal_reference_run": all(d == want_sha for d in shas) and len(shas) == n,
            "hw_queues_env": os.environ.get("GPU_MAX_HW_QUEUES")}


def run_batch(B, group, steps, make_solver, edges_per_solve, want_sha, digest):
    """B independent solves in LOCKSTEP (misslap_solve_batch: the problems of a group share one HIP stream and every launch
    of the solve loop that several of them issue at the same point is one launch) -- the library's answer to "many LAPs on
    one GPU" where --concurrent (B host threads, B streams) stalls on the device's launch rate."""
    from sslap_amd import solve_batch
    walls, creates, infos, shas = [], [], [], []
    for _ in range(steps + 1):  # (one untimed warm-up batch: streams, fibers, the block cache)
        t0 = time.perf_counter()
        solvers = [make_solver() for _ in range(B)]
        t1 = time.perf_counter()
        sols, info = solve_batch(solvers, group)
        t2 = time.perf_counter()
        walls.append(t2 - t1)
        creates.append(t1 - t0)
        infos.append(info)
        shas = [digest(x) for x in sols]
        del solvers
    wall = sum(walls[1:]) / steps
    n = B
    return {"B": B, "group_size": group or 12, "groups": infos[-1]["groups"], "wall_ms": round(1e3 * wall, 3),
            "ms_per_solve": round(1e3 * wall / n, 3), "aggregate_medges_s": round(edges_per_solve * n / wall / 1e6, 2),
            "create_ms_for_all": round(1e3 * sum(creates[1:]) / steps, 3),
            "calls_recorded": infos[-1]["calls_recorded"], "launches_issued": infos[-1]["launches_issued"],
            "host_ms_summed_over_groups": infos[-1]["host_ms"],
            "all_sha256_equal_reference_run": all(d == want_sha for d in shas) and len(shas) == n}


def source_digest():
    """sha256 over the kernel sources the library is built from: ties a committed PMC measurement to the code it was
    taken on (the GPU box has no .git, so a commit hash alone could not be checked there)."""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, "sslap_amd", "csrc")
    for f in sorted(os.listdir(d)):
        if f.endswith((".hip", ".hpp")):
            h.update(f.encode())
            h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()


def gpu_tail_cand_edges(g):
    """edges of rows answered by candidate lines inside the tail kernel (not part of the grid launches)"""
    return int(g.get("tail_cand_edges", 0))


def _cpu_name():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", default="C3")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--cpu-rounds", type=int, default=100_000)
    ap.add_argument("--cpu-sample-only", action="store_true", help="cpu_baseline: the bounded sample only")
    ap.add_argument("--tail-threshold", type=int, default=None)
    ap.add_argument("--values", choices=("f32", "f64", "f32-as-f64"), default="f32",
                    help="f32: the BASELINE configs' fp32-exact costs (8 B/edge, the default and the metric's config); f64: the "
                         "same draws times (1 + 2^-30) -- arbitrary doubles, the reference's native value type: 12 B/edge; "
                         "f32-as-f64: the fp32-exact costs in the 12 B/edge layout (same assignment as the fixture)")
    ap.add_argument("--shuffle-rows", action="store_true",
                    help="random stored order inside every row (columns not ascending: the engine carries the stored index)")
    ap.add_argument("--concurrent", type=int, default=0,
                    help="N = 1 only: after the timed single-solve steps, B independent solves at a time from B host "
                         "threads (B handles, B streams); reported beside the single-solve `value`, never instead of it")
    ap.add_argument("--batch", type=int, default=0,
                    help="N = 1 only: after the timed single-solve steps, B copies of the workload solved in lockstep by "
                         "misslap_solve_batch; reported beside the single-solve `value`, never instead of it")
    ap.add_argument("--batch-group", type=int, default=0, help="problems per stream of --batch (0 = the library's default, 12)")
    ap.add_argument("--mode", choices=("sharded", "replicas"), default="sharded",
                    help="N > 1: 'sharded' = ONE problem, persons of the big rounds sharded over the ranks, RCCL "
                         "exchange (strong scaling; the default); 'replicas' = N independent problems, one per GPU, "
                         "no collective (weak scaling: how independent LAPs -- the reference's typical use -- scale)")
    args = ap.parse_args()

    backend = os.environ.get("MISSLAP_DIST_BACKEND", "nccl")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ and backend != "threads":
        launch_ranks(args.gpus, sys.argv[1:])  # does not return
    if args.batch > 1:
        os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")  # (a queue per group of the batched solve)
    if args.concurrent > 1:
        # hardware queues for the concurrent streams (the runtime's default is 4 per process: streams beyond that
        # share a queue and their kernels serialise -- C3, 16 at a time: 2.8x the single-solve throughput with 4 queues,
        # 4.8x with 16, 6.4x with 32); must be in the environment before the HIP runtime starts
        os.environ.setdefault("GPU_MAX_HW_QUEUES", str(min(max(2 * args.concurrent, 4), 32)))
    if backend == "threads" and args.gpus > 1:
        os.environ.setdefault("GPU_MAX_HW_QUEUES", str(min(max(2 * args.gpus, 4), 32)))  # a queue per rank thread

    import numpy as np
    import torch

    from sslap_amd import synth

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")

    if backend == "threads" and args.gpus > 1:
        # N rank THREADS on cuda:0 (one-GPU rehearsal of the N-rank line, e.g. N = 8: a one-GPU box admits only a few
        # PROCESSES on its card).  Same code per rank as under a launcher; the exchange goes through the library's custom
        # communicator (sslap_amd.dist.Comm.in_process), barriers and reductions through a ThreadGroup.
        import threading
        from sslap_amd import dist as mdist
        torch.cuda.set_device(0)
        loc, val = workload(args, synth)
        shared = (torch.from_numpy(loc).cuda(), torch.from_numpy(val).cuda(), int(loc.shape[0]))
        del loc, val
        torch.cuda.synchronize()
        group = mdist.ThreadGroup(args.gpus, timeout_s=float(os.environ.get("MISSLAP_BENCH_TIMEOUT_S", 3000)))
        rcs = [1] * args.gpus

        def rank_thread(r):
            try:
                torch.cuda.set_device(0)
                comm = None if args.mode == "replicas" else mdist.Comm.in_process(r, group)
                run_rank(args, r, args.gpus, 0, ThreadRanks(r, group, torch), comm, shared, "threads")
                rcs[r] = 0
            except BaseException as e:  # noqa: BLE001
                import traceback
                traceback.print_exc()
                group.abort()
                rcs[r] = e.code if isinstance(e, SystemExit) and isinstance(e.code, int) else 1
        th = [threading.Thread(target=rank_thread, args=(r,)) for r in range(args.gpus)]
        for t in th:
            t.start()
        for t in th:
            t.join()
        raise SystemExit(0 if all(rc == 0 for rc in rcs) else next(rc for rc in rcs if rc != 0) or 1)

    import torch.distributed as dist
    rank = int(os.environ.get("RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    # one rank per GPU; MISSLAP_DIST_BACKEND=gloo lets several ranks share one GPU (rehearsal of the
    # multi-rank path on a single-GPU box; RCCL needs a GPU per rank)
    local_rank = local_rank % max(torch.cuda.device_count(), 1) if backend != "nccl" else local_rank
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend, rank=rank, world_size=world)
    comm = None
    if world > 1 and args.mode != "replicas":
        from sslap_amd import dist as mdist
        # RCCL communicator of the library (the 128-byte id travels through torch.distributed); with
        # MISSLAP_DIST_BACKEND=gloo several ranks share one GPU and the exchange is staged through the host
        # MISSLAP_BENCH_COMM=torch: the exchange through torch.distributed's own all-reduces on the device buffers instead
        # (backend nccl: the RCCL inside torch) -- also what every rank falls back to when ANY rank cannot create the
        # library's communicator, so that a node on which ncclCommInitRank fails still yields a line (which says so)
        want = os.environ.get("MISSLAP_BENCH_COMM", "rccl" if backend == "nccl" else "staged")
        if want == "rccl":
            try:
                comm, why = mdist.Comm.from_torch_distributed(local_rank), None
            except Exception as e:  # noqa: BLE001
                comm, why = None, repr(e)
            ok = torch.tensor([0 if comm is None else 1], device="cuda")
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
            if int(ok.item()) == 0:
                print(f"bench.py rank {rank}: the library's RCCL communicator could not be created on every rank "
                      f"({why or 'another rank failed'}): falling back to torch.distributed's collectives", file=sys.stderr, flush=True)
                comm = None
                want = "torch"
        if want == "torch":
            comm = mdist.Comm.torch_collectives()
            comm.transport_note = ("one process per rank, torch.distributed all-reduces (backend %s) on the device buffers" % backend
                                   + ("" if os.environ.get("MISSLAP_BENCH_COMM") == "torch" else " -- FALLBACK: the library's own RCCL communicator failed"))
        elif want == "staged":
            comm = mdist.Comm.gloo_staged()
    run_rank(args, rank, world, local_rank, TorchRanks(rank, world, dist, torch), comm, None, backend)
    if world > 1:
        dist.destroy_process_group()


def run_rank(args, rank, world, local_rank, ranks, comm, shared, backend):
    """One rank's part of the benchmark; rank 0 prints the JSON line.  `ranks`: barrier / reductions / gather over the
    ranks (processes or threads); `shared`: device-resident inputs generated once for all rank threads, or None."""
    import numpy as np  # noqa: F401
    import torch

    from sslap_amd import AuctionSolver, synth

    replicas = world > 1 and args.mode == "replicas"
    barrier = ranks.barrier

    # synthetic workload, resident in HBM before anything is timed
    if shared is not None:
        d_loc, d_val, nnz = shared
    else:
        loc, val = workload(args, synth)
        nnz = int(loc.shape[0])
        d_loc = torch.from_numpy(loc).cuda()
        d_val = torch.from_numpy(val).cuda()
        del loc, val
    # (device-resident inputs ordered behind the stream that produced them: no device-wide wait per create)
    gpu_opts = dict(device=local_rank, profile=True, input_stream=torch.cuda.current_stream().cuda_stream)
    if args.tail_threshold is not None:
        gpu_opts["tail_threshold"] = args.tail_threshold
    if args.values == "f32-as-f64":
        gpu_opts["force_f64"] = True

    def one_step():
        if world == 1 or replicas:
            s = AuctionSolver.from_device_pointers(d_loc.data_ptr(), d_val.data_ptr(), nnz, problem="max",
                                                   max_iter=10**8, **gpu_opts)
            sol = s.solve()
            return s, sol
        s = AuctionSolver.from_device_pointers(d_loc.data_ptr(), d_val.data_ptr(), nnz, problem="max",
                                               max_iter=10**8, shard=(rank, world), **gpu_opts)
        sol = s.solve_sharded(comm)  # the exchange runs inside the library (misslap_solve_sharded)
        return s, sol

    for _ in range(args.warmup):
        one_step()
    barrier()
    t0 = time.perf_counter()
    runs = []
    s = None
    for _ in range(args.steps):
        # (a step ends with its handle destroyed -- the reference's solver object dies with the call, too; a handle that
        # outlived its step would also make the next create open a second HIP stream, 6 ms, inside the timed region)
        del s
        s, sol = one_step()
        runs.append((dict(s.meta), dict(s.gpu)))
    barrier()
    dt = time.perf_counter() - t0
    if replicas:
        dt = ranks.max_f(dt)
        edges_all = ranks.sum_i(sum(g["edges_scanned"] for _, g in runs))  # N independent solves: every rank's edges are unique work
        fs_all_edges, fs_max_ms = None, None
    elif world > 1:
        dt = ranks.max_f(dt)
        # a rank counts the bids of its own shard in the exchanged (sharded) rounds plus all bids of the
        # replicated rounds, which every rank repeats: unique work = sum over ranks of the sharded part + the
        # replicated part ONCE (redundant scans are not throughput)
        sh = sum(g["shard_edges"] for _, g in runs)
        repl = sum(g["edges_scanned"] for _, g in runs) - sh
        edges_all = ranks.sum_i(sh) + repl
        # bid-phase throughput of the full scans over all ranks: every rank scans its shard of the K = N rounds at
        # the same time, so the aggregate rate is (sum of the shards' edges) / (slowest rank's kernel time)
        fs_all_edges = ranks.sum_i(sum(g["fullscan_edges"] for _, g in runs))
        fs_max_ms = ranks.max_f(sum(g["fullscan_ms"] for _, g in runs))
    else:
        edges_all = sum(g["edges_scanned"] for _, g in runs)
        fs_all_edges, fs_max_ms = None, None

    # who took part: every rank's device, assignment hash and what its communicator says about itself
    from sslap_amd import _lib
    import ctypes as C
    name = C.create_string_buffer(128)
    uuid = C.create_string_buffer(40)
    cus, hbm = C.c_int32(), C.c_int64()
    _lib.load().misslap_device_info(local_rank, name, 128, C.byref(cus), C.byref(hbm))
    _lib.load().misslap_device_uuid(local_rank, uuid, 40)
    me = {"rank": rank, "device": name.value.decode(), "device_uuid": uuid.value.decode(), "local_rank": local_rank,
          "sol_sha256": synth.sol_digest(sol), "comm": comm.info() if comm is not None else None,
          "sharded_rounds_per_solve": runs[-1][1].get("sharded_rounds", 0)}
    rank_list = ranks.gather(me)
    if rank == 0:
        meta, gpu = runs[-1]
        bpe = gpu["bytes_per_edge"]
        bid_ms = sum(g["bid_ms"] for _, g in runs)
        bid_edges = sum(g["bid_edges"] for _, g in runs)
        bid_launches = sum(g["bid_launches"] for _, g in runs)
        til_ms = sum(g["tiled_ms"] for _, g in runs)
        til_edges = sum(g["tiled_edges"] for _, g in runs)
        til_launches = sum(g["tiled_launches"] for _, g in runs)
        fs_ms = sum(g["fullscan_ms"] for _, g in runs)
        fs_edges = sum(g["fullscan_edges"] for _, g in runs)
        fs_launches = sum(g["fullscan_launches"] for _, g in runs)
        tail_ms = sum(g["tail_ms"] for _, g in runs)
        tail_edges = sum(g["tail_edges"] for _, g in runs)
        tiled = bool(gpu.get("tiled_active")) and til_launches > 0
        # the roofline kernel: the one that performs the full CSR scans (K = N rounds)
        rk_name = "k_bid_tiled" if tiled else "k_bid"
        rk_ms, rk_edges, rk_launches = (til_ms, til_edges, til_launches) if tiled else (bid_ms, bid_edges, bid_launches)
        # The gather engine (configs without the tile-major copy) answers part of a launch's bids from candidate lines:
        # those rows are counted like the reference counts them but never read.  The roofline figures are formed from the
        # edges a launch actually STREAMED (misslap_meta.bid_edges_read / fullscan_edges_read); the reference-equivalent
        # figure goes next to them under its own key.  (The LDS-tiled engine reads every edge it counts: same number.)
        rk_edges_counted, fs_edges_counted = rk_edges, fs_edges
        if not tiled:
            rk_edges = sum(g.get("bid_edges_read", g["bid_edges"]) for _, g in runs)
            fs_edges = sum(g.get("fullscan_edges_read", g["fullscan_edges"]) for _, g in runs)
        achieved = rk_edges * bpe / (rk_ms * 1e-3) / 1e9 if rk_ms > 0 else 0.0
        fs_achieved = fs_edges * bpe / (fs_ms * 1e-3) / 1e9 if fs_ms > 0 else 0.0
        achieved_counted = rk_edges_counted * bpe / (rk_ms * 1e-3) / 1e9 if rk_ms > 0 else 0.0
        # HBM traffic of the roofline kernel from the committed PMC passes (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE on
        # this same command, summarised by tools/pmc_summary.py with the gfx950 corrections).  The file names the
        # kernel sources (sha256) and the commit it was measured on: any other code gets null, not a stale number.
        traffic, traffic_meta = None, {}
        import glob
        variant = args.config + {"f32": "", "f64": "_f64", "f32-as-f64": "_f32asf64"}[args.values] + ("_shuffled" if args.shuffle_rows else "")
        tpaths = sorted(glob.glob(os.path.join(ROOT, "profiles", f"r[0-9][0-9]_pmc_traffic_{variant}.json")))
        tpath = tpaths[-1] if tpaths else ""  # the newest round's measurement
        if os.path.exists(tpath):
            tj = json.load(open(tpath))
            traffic_meta = dict(traffic_commit=tj.get("commit"), traffic_source_sha256=tj.get("source_sha256"),
                                traffic_file=os.path.relpath(tpath, ROOT))
            # (the bid instance of the full-scan engine: template argument 10 of <..., MODE, kFmt> is MODE = 0; MODE 1 is the
            # eCE pass.  The gather engine: the instance whose launches the events of this run bracketed -- profile level 1
            # times the K = N launches only, and those are k_bid<E, PriceSource, kLines = 1> (lines in use: any line that
            # still answers is used, nothing is built) or <..., 0> (no lines); never the instance with the most bytes x
            # launches, which at C5 is the small-round instance kLines = 2)
            def is_bid_instance(k):
                if (rk_name + "<") not in k:
                    return False
                targs = [t.strip() for t in k[k.index("<") + 1:k.rindex(">")].split(",")]
                if rk_name != "k_bid_tiled":
                    return len(targs) == 3 and targs[1].endswith("PriceSource") and targs[2] in ("0", "1")
                return len(targs) >= 10 and targs[9] == "0"  # <threads, rows, batch, depth, cols, loaders, ABL, lanes, split, MODE, ...>
            tk = sorted(((k, v) for k, v in tj["kernels"].items() if is_bid_instance(k)),
                        key=lambda kv: -kv[1]["launches"])
            if tk and tj.get("source_sha256") == source_digest():
                # the timed launches may be TWO instances (the engine: the launches of a phase alternate their walking
                # direction, kRev; the gather kernel: a handle that takes its lines into use during the solve): bytes per
                # launch = the launch-weighted mean over them, like avg_launch_us
                n_l = sum(v["launches"] for _, v in tk)
                traffic = round(sum((v["read_avg"] + v["write_avg"]) * v["launches"] for _, v in tk) / max(n_l, 1))
                traffic_meta["traffic_kernel"] = " + ".join(k for k, _ in tk)
                traffic_meta["traffic_launches_measured"] = n_l
                traffic_meta["traffic_max_launch"] = round(max(v["read_max"] + v["write_max"] for _, v in tk))
            elif tj.get("source_sha256") == source_digest():
                # (small problems: the K = N launches are k_bid<..., 2>, the instance of every untimed mid round as well --
                # the per-kernel averages of the PMC pass cannot be attributed to the timed launches)
                traffic_meta["traffic_note"] = "the timed launches share their kernel instance with untimed rounds: not attributable"
            else:
                traffic_meta["traffic_stale"] = True
        # metric (ii) of SURVEY 8(d): throughput over ALL grid-kernel bid launches (full-scan engine + k_bid), from
        # one extra, untimed solve with every bid launch bracketed by HIP events (profile level 3 costs host time,
        # so it is kept out of the timed steps)
        grid_all = None
        if world == 1 or replicas:
            gpu_opts3 = dict(gpu_opts, profile=3)
            s3 = AuctionSolver.from_device_pointers(d_loc.data_ptr(), d_val.data_ptr(), nnz, problem="max",
                                                    max_iter=10**8, **gpu_opts3)
            s3.solve()
            g3 = s3.gpu
            ms3 = g3["bid_ms"] + g3["tiled_ms"]
            e3 = g3["bid_edges"] + g3["tiled_edges"]
            grid_all = {"launches": g3["bid_launches"] + g3["tiled_launches"], "ms": round(ms3, 3), "edges": e3,
                        "medges_s": round(e3 / (ms3 * 1e-3) / 1e6, 1) if ms3 else None,
                        "GBs_algorithmic": round(e3 * bpe / (ms3 * 1e-3) / 1e9, 1) if ms3 else None,
                        "edges_read": e3 - (g3["cand_edges"] - gpu_tail_cand_edges(g3)),
                        "note": "one extra untimed solve, every bid launch timed (profile 3); edges = reference-"
                                "equivalent row lengths of the bidders, edges_read = rows actually streamed; since round 4 a "
                                "launch of a round with K <= 2048 is the WHOLE round (k_round_fused: bids + resolve + "
                                "assign + compaction), so ms is not comparable with earlier rounds' bid-only figure"}
        # SURVEY 8(d), second solve figure: one more (untimed) step from HOST arrays -- the reference's `setup` timer
        # brackets the CSR build from host arrays (auction_.pyx:206-207, :265), here that includes the H2D copy of the
        # COO input (16 B per entry, pageable numpy memory)
        incl_h2d = None
        if world == 1 and args.config != "C5":
            loc_h, val_h = workload(args, synth)
            t_h = time.perf_counter()
            sh = AuctionSolver(loc_h, val_h, problem="max", max_iter=10**8, device=local_rank)
            sh.solve()
            wall_h = 1e3 * (time.perf_counter() - t_h)
            incl_h2d = {"solve_ms_incl_h2d": round(sh.gpu["setup_ms"] + sh.gpu["solve_ms"], 3),
                        "setup_ms_incl_h2d": round(sh.gpu["setup_ms"], 3), "wall_ms": round(wall_h, 3),
                        "h2d_bytes": int(loc_h.nbytes + val_h.nbytes),
                        "note": "create from host numpy arrays (H2D of the COO input + CSR build) + solve; never part of `value`"}
            del sh, loc_h, val_h
        read_peak, copy_peak = measured_hbm_peaks(local_rank)
        conc = None
        if args.concurrent > 1 and world == 1:
            opts_c = dict(gpu_opts, profile=False)
            # hipFree waits for EVERY stream of the device: a handle destroyed while fifteen other solves run stalls
            # behind their kernels.  The library parks freed blocks instead when its cache limits allow it (default:
            # small blocks only) -- with room for B handles no solve of the concurrent phase calls hipMalloc / hipFree
            cache_gb = 0 if os.environ.get("MISSLAP_BENCH_SMALL_CACHE") else 4 * args.concurrent
            if cache_gb:
                _lib.check(_lib.load().misslap_set_cache_limits(cache_gb << 30, 8 << 30, 4096))
            conc = run_concurrent(args.concurrent, max(1, args.steps),
                                  lambda: AuctionSolver.from_device_pointers(d_loc.data_ptr(), d_val.data_ptr(), nnz,
                                                                             problem="max", max_iter=10**8, **opts_c),
                                  gpu["edges_scanned"], synth.sol_digest(sol), synth.sol_digest)
            conc["block_cache_GiB"] = cache_gb
            conc["single_solve_ms_per_step"] = round(1e3 * dt / args.steps, 3)
            conc["throughput_vs_single"] = round(conc["aggregate_medges_s"] / (edges_all / dt / 1e6), 2)
        batch = None
        if args.batch > 1 and world == 1:
            opts_b = dict(gpu_opts, profile=False)
            cache_gb = 4 * args.batch
            _lib.check(_lib.load().misslap_set_cache_limits(cache_gb << 30, 8 << 30, 4096))
            batch = run_batch(args.batch, args.batch_group, max(1, args.steps),
                              lambda: AuctionSolver.from_device_pointers(d_loc.data_ptr(), d_val.data_ptr(), nnz,
                                                                         problem="max", max_iter=10**8, **opts_b),
                              gpu["edges_scanned"], synth.sol_digest(sol), synth.sol_digest)
            batch["single_solve_ms_per_step"] = round(1e3 * dt / args.steps, 3)
            batch["single_solve_ms"] = round(sum(g["solve_ms"] for _, g in runs) / len(runs), 3)
            # solves per second of the batch against ONE solve at a time (both without the CSR build)
            batch["throughput_vs_single_solve"] = round(batch["single_solve_ms"] / batch["ms_per_solve"], 2)
        out = {
            "metric": "Medges/s (bid-phase CSR nnz/s) + solve ms, N=200k d=0.1% sparse LAP",
            "value": round(edges_all / dt / 1e6, 2),
            "unit": "Medges/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * dt / args.steps, 3),
            "higher_is_better": True,
            "scaling": "weak" if replicas else "strong",
            "mode": ("replicas: one independent solve per GPU, no collective" if replicas else
                     "sharded: one problem, RCCL exchange in the big rounds" if world > 1 else "single GPU"),
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": f"{args.config} {int(s.num_rows)}x{int(s.num_cols)} sparse LAP "
                                   f"(BASELINE.json configs), one full auction_solve per step",
                       "n_rows": int(s.num_rows), "n_cols": int(s.num_cols), "nnz": nnz,
                       "problem": "max",
                       "values": {"f32": "fp32-exact, carried as f64 (prices/bids f64, eps f32)",
                                  "f64": "arbitrary doubles (fp32 draws x (1 + 2^-30)): int32 col + fp64 val",
                                  "f32-as-f64": "fp32-exact values in the 12 B/edge layout (int32 col + fp64 val)"}[args.values],
                       "rows": "stored order shuffled inside every row" if args.shuffle_rows else "columns ascending",
                       "tile_major_format": gpu.get("tiled_format") if gpu.get("tiled_active") else None,
                       "bytes_per_edge": bpe, "generator": "sslap_amd.synth seed=1"},
            "solve_ms": round(sum(g["solve_ms"] for _, g in runs) / len(runs), 3),
            "setup_ms": round(sum(g["setup_ms"] for _, g in runs) / len(runs), 3),
            "solve_incl_h2d": incl_h2d,
            "rounds": meta["its"], "eps_phases": meta["nreductions"] + 1,
            "grid_rounds": gpu["grid_rounds"], "tail_rounds": gpu["tail_rounds"],
            "edges_scanned_per_solve": gpu["edges_scanned"],
            # reference-equivalent count (sum of the bidders' row lengths, the oracle's number); rows that a
            # person's candidate line answered exactly are counted but not read:
            "edges_read_per_solve": gpu["edges_scanned"] - gpu["cand_edges"],
            # ... so this, not `value`, is what the memory system saw (never to be read as `value`'s bandwidth)
            "edges_read_medges_s": round((edges_all - sum(g["cand_edges"] for _, g in runs)) / dt / 1e6, 2) if world == 1 else None,
            "candidate_line_hit_rate": round(gpu["cand_hits"] / max(gpu["bids_made"], 1), 4),
            "sol_sha256": synth.sol_digest(sol), "obj_f64": gpu["obj_f64"],
            "complete_assignment": list(gpu["complete_assignment"]), "valid_assignment": gpu["valid_assignment"],
            "bid_phase": {
                "full_scan_kernel": rk_name,
                "fullscan_launches": fs_launches, "fullscan_avg_us": round(1e3 * fs_ms / max(fs_launches, 1), 2),
                "fullscan_medges_s": round(fs_edges / (fs_ms * 1e-3) / 1e6, 1) if fs_ms else None,
                "fullscan_medges_s_reference_equivalent": round(fs_edges_counted / (fs_ms * 1e-3) / 1e6, 1) if fs_ms else None,
                "fullscan_GBs": round(fs_achieved, 1), "fullscan_frac_of_hbm_peak": round(fs_achieved / HBM_PEAK_GBS, 4),
                # N > 1: the full scans of all ranks together (rank 0's own share is the line above)
                "fullscan_all_ranks_medges_s": (round(fs_all_edges / (fs_max_ms * 1e-3) / 1e6, 1)
                                                if fs_all_edges and fs_max_ms else None),
                "fullscan_all_ranks_frac_of_hbm_peak": (round(fs_all_edges * bpe / (fs_max_ms * 1e-3) / 1e9
                                                              / (HBM_PEAK_GBS * world), 4)
                                                        if fs_all_edges and fs_max_ms else None),
                "k_bid_tiled": {"launches": til_launches, "ms": round(til_ms, 3), "edges": til_edges,
                                "min_K": gpu.get("tiled_min_K")},
                # profile level 1 times only the FULL-SCAN launches of the gather kernel (configs where the tiled
                # layout does not apply); its ~3000 small launches per solve are not bracketed by events
                "k_bid_timed": {"launches": bid_launches, "ms": round(bid_ms, 3), "edges": bid_edges,
                                "medges_s": round(bid_edges / (bid_ms * 1e-3) / 1e6, 1) if bid_ms else None},
                "grid_all_launches": grid_all,
                "k_tail": {"ms_per_solve": round(tail_ms / len(runs), 3), "rounds_per_solve": gpu["tail_rounds"],
                           "us_per_round": round(1e3 * tail_ms / len(runs) / max(gpu["tail_rounds"], 1), 3),
                           "medges_s": round(tail_edges / (tail_ms * 1e-3) / 1e6, 1) if tail_ms else None},
            },
            "roofline": {
                "kernel": rk_name, "bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS,
                "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5),
                "launches": rk_launches, "avg_launch_us": round(1e3 * rk_ms / max(rk_launches, 1), 3),
                "algorithmic_bytes_per_edge": bpe,
                "algorithmic_bytes_per_launch": round(rk_edges * bpe / max(rk_launches, 1)),
                # the same achieved rate against what the library's own streaming kernels reach on this GPU in this
                # process: a read-only pass (the roofline kernel only reads: the fractions are against THIS) and a copy
                "peak_measured_read": round(read_peak, 1), "peak_measured_copy": round(copy_peak, 1),
                "frac_of_measured": round(achieved / read_peak, 5),
                "fullscan_frac_of_measured": round(fs_achieved / read_peak, 5),
                "timing": "HIP events handed to the launch (hipExtLaunchKernel): begin / end of the kernel itself",
                "traffic": traffic,
                "traffic_over_algorithmic": (round(traffic / max(rk_edges * bpe / max(rk_launches, 1), 1.0), 3)
                                             if traffic else None),
                "traffic_source": "rocprofv3 PMC passes (FETCH_SIZE x2 gfx950 correction calibrated on known-size "
                                  "kernels + WRITE_SIZE), bytes per launch" if traffic else None,
                **traffic_meta,
                # N ranks: the one figure of this path that is expected to scale -- the full scans (K = N launches) of all
                # ranks together, every rank scanning its shard at the same time: (sum of the shards' edges) / (slowest
                # rank's kernel time), against N x 8 TB/s -- and who took part.  DESIGN.md section 7 /
                # profiles/r06_scale_expectation.json say what to expect per N.
                "all_ranks": {
                    "n_ranks": world,
                    "fullscan_medges_s": (round(fs_all_edges / (fs_max_ms * 1e-3) / 1e6, 1) if fs_all_edges and fs_max_ms
                                          else round(fs_edges / (fs_ms * 1e-3) / 1e6, 1) if fs_ms else None),
                    "fullscan_frac_of_hbm_peak": (round(fs_all_edges * bpe / (fs_max_ms * 1e-3) / 1e9 / (HBM_PEAK_GBS * world), 4)
                                                  if fs_all_edges and fs_max_ms else round(fs_achieved / HBM_PEAK_GBS, 4)),
                    "sharded_rounds_per_solve": rank_list[0]["sharded_rounds_per_solve"],
                    "exchanges_per_solve": 2 * rank_list[0]["sharded_rounds_per_solve"],
                    "rccl_nranks": (rank_list[0]["comm"]["transport_ranks"]
                                    if rank_list[0]["comm"] and rank_list[0]["comm"]["kind"] == "rccl" else None),
                    "comm_kind": rank_list[0]["comm"]["kind"] if rank_list[0]["comm"] else None,
                    "comm_ranks_seen_by_every_rank": [r["comm"]["transport_ranks"] if r["comm"] else None for r in rank_list],
                    "distinct_gpus": len({r["device_uuid"] for r in rank_list}),
                    "mode": "replicas" if replicas else "sharded" if world > 1 else "single",
                },
                # (gather engine only) what the same launches come to when the rows that candidate lines answered are
                # counted as well -- the reference-equivalent count, NOT bytes moved; never to be read as a bandwidth
                "edges_counted_incl_rows_answered_from_lines": rk_edges_counted,
                "edges_read": rk_edges,
                "reference_equivalent_GBs": round(achieved_counted, 2),
                "counts_rows_answered_from_lines": False,
            },
            "device": name.value.decode(), "compute_units": int(cus.value),
            # N > 1: proof that N ranks took part and agree -- what the transport itself reports (ncclCommCount), every
            # rank's device and assignment hash, the exchanges a solve issued
            "ranks": rank_list,
            "rank_transport": (getattr(comm, "transport_note", None) or
                               ("one process per rank, RCCL" if backend == "nccl" else
                                "one process per rank on a shared GPU, exchange staged through the host (gloo)" if backend == "gloo"
                                else "one THREAD per rank on a shared GPU, exchange staged through the host")) if world > 1 else None,
            "rccl_nranks": (rank_list[0]["comm"]["transport_ranks"] if rank_list[0]["comm"] and rank_list[0]["comm"]["kind"] == "rccl" else None),
            "comm_kind": rank_list[0]["comm"]["kind"] if rank_list[0]["comm"] else None,
            "comm_ranks_seen_by_every_rank": [r["comm"]["transport_ranks"] if r["comm"] else None for r in rank_list],
            "distinct_gpus": len({r["device_uuid"] for r in rank_list}),
            "sharded_rounds_per_solve": rank_list[0]["sharded_rounds_per_solve"],
            "exchanges_per_solve": 2 * rank_list[0]["sharded_rounds_per_solve"],
            "sol_sha256_equal_on_all_ranks": len({r["sol_sha256"] for r in rank_list}) == 1,
            "concurrent": conc,
            "batch": batch,
        }
        if world == 1 and not args.no_cpu:
            whole = not args.cpu_sample_only and args.config != "C5"  # (C5: ~20 min of oracle time)
            out["cpu_baseline"] = cpu_baseline(args.config, args.cpu_rounds, whole, args)
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out))
        if not out["sol_sha256_equal_on_all_ranks"]:
            raise SystemExit("the ranks disagree on the assignment")


if __name__ == "__main__":
    main()
