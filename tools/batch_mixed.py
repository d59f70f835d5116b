#!/usr/bin/env python3
"""B DIFFERENT problems of one shape solved one after the other and as one batch (misslap_solve_batch).
usage: batch_mixed.py [n] [density] [B] [group]      e.g. 50000 0.005 32 16   (the C2 shape, 32 seeds)"""
import json
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from sslap_amd import AuctionSolver, solve_batch, synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 50000
dens = float(sys.argv[2]) if len(sys.argv) > 2 else 0.005
B = int(sys.argv[3]) if len(sys.argv) > 3 else 32
group = int(sys.argv[4]) if len(sys.argv) > 4 else 0
dev = []
for k in range(B):
    loc, val = synth.gen_sparse(n, n, dens, seed=500 + k)
    dev.append((torch.from_numpy(loc).cuda(), torch.from_numpy(val).cuda(), int(loc.shape[0])))
torch.cuda.synchronize()
mk = lambda d: AuctionSolver.from_device_pointers(d[0].data_ptr(), d[1].data_ptr(), d[2], problem="max", max_iter=10**8)  # noqa: E731
for d in dev[:2]:
    mk(d).solve()  # warm-up
t0 = time.perf_counter()
single = []
for d in dev:
    s = mk(d)
    single.append((s.solve(), s.gpu["solve_ms"], s.meta["its"], s.gpu["edges_scanned"]))
t_seq = time.perf_counter() - t0
solve_seq = sum(x[1] for x in single)
solvers = [mk(d) for d in dev]
sols, _ = solve_batch(solvers, group)  # warm-up of the batch path (streams, fibers)
solvers = [mk(d) for d in dev]
t0 = time.perf_counter()
sols, info = solve_batch(solvers, group)
t_batch = time.perf_counter() - t0
ok = all(np.array_equal(a, b[0]) for a, b in zip(sols, single))
print(json.dumps({"shape": [n, n], "density": dens, "B": B, "group_size": group or 12, "groups": info["groups"],
                  "sequential_wall_ms": round(1e3 * t_seq, 2), "sequential_solve_ms_sum": round(solve_seq, 2),
                  "rounds_min_max": [min(x[2] for x in single), max(x[2] for x in single)],
                  "batch_wall_ms": round(1e3 * t_batch, 2), "speedup_vs_sequential_solves": round(solve_seq / (1e3 * t_batch), 2),
                  "aggregate_medges_s": round(sum(x[3] for x in single) / t_batch / 1e6, 1),
                  "calls_recorded": info["calls_recorded"], "launches_issued": info["launches_issued"], "host_ms": info["host_ms"],
                  "all_assignments_equal_the_single_solves": bool(ok)}))
