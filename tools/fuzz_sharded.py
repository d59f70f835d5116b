#!/usr/bin/env python3
"""Randomized parity run of the SHARDED solve on one GPU: W rank threads (sslap_amd.dist.ThreadGroup / Comm.in_process:
the library's own loop and exchange calls, buffers reduced on the host), random instances and option mixes; every
rank's assignment, round count and prices against the oracle.  usage: fuzz_sharded.py [first_seed] [count]"""
import os
import sys
import threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden"))
import numpy as np
import cases
from oracle import oracle as orc
from sslap_amd import from_sparse, synth
from sslap_amd.dist import Comm, ThreadGroup, solve_sharded

first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
count = int(sys.argv[2]) if len(sys.argv) > 2 else 60
bad = 0
for seed in range(first, first + count):
    r = np.random.default_rng(9100 + seed)
    world = int(r.choice([2, 3, 4, 5, 8]))
    n = int(r.choice([40, 300, 900, 2500, 6000, 12000]))
    m = n if r.random() < 0.6 else int(n * r.uniform(1.02, 2.0))
    density = float(r.choice([3.0, 8.0, 30.0, 120.0])) / m
    ints = int(r.choice([0, 0, 3, 9]))
    prob = "max" if r.random() < 0.6 else "min"
    loc, val = synth.gen_sparse(n, m, density, seed=1900 + seed, integer_values=ints)
    if seed % 3 == 2:
        val = val + r.random(val.shape[0]) * 1e-7  # 12 B/edge layout
    if seed % 4 == 1:
        loc, val = synth.shuffle_within_rows(loc, val, seed)
    kw = dict(problem=prob, cardinality_check=False, max_iter=int(r.choice([10**8, 10**8, 10**8, 900, 37])),
              eps_start=float(r.choice([0.0, 0.0, 1.0])))
    gpu = dict(tail_threshold=[None, 0, 17, 200][seed % 4], tiled_min_k=[None, 1, -1][seed % 3], shard_min_k=[-1, None, -1][(seed // 3) % 3],
               rounds_per_sync=[None, 1, 4][(seed // 2) % 3], cand=[None, None, False][(seed // 3) % 3])
    if gpu["tiled_min_k"] == 1:
        gpu["engine"] = 1
        gpu["tiled_shape"] = [None, 8, 9][(seed // 3) % 3]
    gpu = {k: v for k, v in gpu.items() if v is not None}
    o = orc.from_sparse(loc, val.copy(), **kw)
    osol = o.solve()
    so = o.state()
    group = ThreadGroup(world, timeout_s=300.0)
    res, errs = {}, []

    def rank_main(rank):
        try:
            comm = Comm.in_process(rank, group)
            s = from_sparse(loc, val.copy(), shard=(rank, world), **kw, **gpu)
            sol = solve_sharded(s, comm)
            sg = s.state()
            res[rank] = (np.array_equal(sol, osol) and all(s.meta[k] == o.meta[k] for k in cases.META_KEYS)
                         and np.array_equal(sg["p"].view(np.uint64), so["p"].view(np.uint64)),
                         int(s.gpu.get("sharded_rounds", 0)))
        except Exception as e:  # noqa: BLE001
            errs.append((rank, repr(e)))
            group.abort()
    th = [threading.Thread(target=rank_main, args=(k,)) for k in range(world)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    ok = not errs and len(res) == world and all(v[0] for v in res.values())
    if not ok:
        bad += 1
        print("MISMATCH", seed, world, n, m, density * m, ints, kw, gpu, errs[:2], {k: v for k, v in res.items()}, flush=True)
    elif seed % 10 == 0:
        print("ok", seed, "W", world, n, m, "its", o.meta["its"], "sharded rounds", res[0][1], flush=True)
print(f"done {count} sharded solves, {bad} mismatches")
sys.exit(1 if bad else 0)
