#!/usr/bin/env python3
"""Randomized parity run of misslap_solve_batch on the GPU box: batches of random instances with random options -- of one
shape, or (every third batch) of one number of persons but different objects, entries per row and value layouts -- every
problem compared with its own single solve (assignment, meta, counters) and problem 0 with the oracle.
usage: fuzz_batch.py [first_seed] [count]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden"))
import numpy as np
import cases
from oracle import oracle as orc
from sslap_amd import from_sparse, solve_batch, synth

first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
count = int(sys.argv[2]) if len(sys.argv) > 2 else 50
bad = 0
for seed in range(first, first + count):
    r = np.random.default_rng(9100 + seed)
    n = int(r.choice([60, 300, 900, 2500, 6000, 12000]))
    m = n if r.random() < 0.6 else int(n * r.uniform(1.02, 2.2))
    density = float(r.choice([3.0, 8.0, 30.0, 120.0])) / m
    if seed % 9 == 8:
        n = m = int(r.choice([500, 1030]))
        density = 1.0  # dense rows: the long-row line builder, tail kernels relaunched on a budget
    ints = int(r.choice([0, 0, 3, 9]))
    prob = "max" if r.random() < 0.6 else "min"
    B = int(r.choice([2, 3, 5, 9, 17, 30]))
    group = int(r.choice([0, 1, 2, 4, 16, 32]))
    kw = dict(problem=prob, cardinality_check=False, max_iter=int(r.choice([10**8, 10**8, 10**8, 1500, 93])),
              eps_start=float(r.choice([0.0, 0.0, 1.0])))
    gpu = dict(tail_threshold=[None, 0, 5, 40, 300][seed % 5], tiled_min_k=[None, None, 1, -1][seed % 4],
               cand=[None, None, False][(seed // 3) % 3], force_f64=bool(seed % 5 == 2))
    if gpu["tiled_min_k"] == 1:
        gpu["engine"] = 1
        gpu["tiled_shape"] = [None, 8, 9][(seed // 4) % 3]
    gpu = {k: v for k, v in gpu.items() if v is not None}
    # every third batch is MIXED: the problems keep the number of persons (what a batch requires) but differ in objects,
    # entries per row and value layout -- merged launches then run on grids sized for another problem's objects / entries
    mixed = seed % 3 == 1 and density < 1.0
    probs, gpus = [], []
    for k in range(B):
        mk, dk, gk = m, density, dict(gpu)
        if mixed:
            mk = n if r.random() < 0.4 else int(n * r.uniform(1.0, 3.0))
            dk = float(r.choice([3.0, 8.0, 30.0, 120.0])) / mk
            if r.random() < 0.3:
                gk["force_f64"] = True
        loc, val = synth.gen_sparse(n, mk, dk, seed=4000 + 37 * seed + k, integer_values=ints)
        if (seed // 2) % 3 == 1:
            loc, val = synth.shuffle_within_rows(loc, val, seed + k)
        probs.append((loc, val))
        gpus.append(gk)
    singles = []
    for (loc, val), gk in zip(probs, gpus):
        s = from_sparse(loc, val.copy(), **kw, **gk)
        singles.append((s.solve(), dict(s.meta), dict(s.gpu)))
    solvers = [from_sparse(loc, val.copy(), **kw, **gk) for (loc, val), gk in zip(probs, gpus)]
    sols, info = solve_batch(solvers, group)
    ok = True
    why = []
    for k in range(B):
        sol1, meta1, gpu1 = singles[k]
        if not np.array_equal(sols[k], sol1):
            why.append((k, "sol"))
        why += [(k, key, solvers[k].meta[key], meta1[key]) for key in cases.META_KEYS if solvers[k].meta[key] != meta1[key]]
        # (counters that do not depend on WHEN the host saw which K: which rounds rebuild candidate lines does, so cand_hits is not compared)
        why += [(k, key, solvers[k].gpu[key], gpu1[key]) for key in ("obj_f64", "edges_scanned", "bids_made", "grid_rounds", "tail_rounds",
                                                                      "complete_assignment", "valid_assignment") if solvers[k].gpu[key] != gpu1[key]]
    ok = not why
    o = orc.from_sparse(probs[0][0], probs[0][1].copy(), **kw)
    ok = ok and np.array_equal(sols[0], o.solve())
    if not ok:
        bad += 1
        print("MISMATCH", seed, n, m, density, ints, B, group, kw, gpu, why[:6], flush=True)
    elif seed % 10 == 0:
        print("ok", seed, n, m, B, group, info["calls_recorded"], "->", info["launches_issued"], flush=True)
print("done", count, "batches,", bad, "mismatches", flush=True)
sys.exit(1 if bad else 0)
