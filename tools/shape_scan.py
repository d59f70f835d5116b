#!/usr/bin/env python3
"""Microseconds per auction round over a grid of shapes (GPU box): anomalies in this table are tuning thresholds on the
wrong side of a shape.  shape_scan.py dense|sparse"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from sslap_amd import from_matrix, from_sparse, synth
kind = sys.argv[1] if len(sys.argv) > 1 else "dense"
if kind == "dense":
    shapes = [(n, n) for n in (64, 128, 200, 256, 257, 320, 400, 600, 800, 1023, 1024, 1100, 1500, 2500, 4000)]
else:
    shapes = [(n, per) for n in (2000, 10000, 40000) for per in (8, 30, 100, 200, 256, 280, 400, 1000, 1100) if per < n]
for n, per in shapes:
    if kind == "dense":
        mat = np.float64(np.float32(np.random.RandomState(n).uniform(0, 100, (n, n))))
        mk = lambda: from_matrix(mat, problem="max", max_iter=10**8, cardinality_check=False)
    else:
        loc, val = synth.gen_sparse(n, n, per / n, seed=n + per)
        mk = lambda: from_sparse(loc, val.copy(), problem="max", max_iter=10**8, cardinality_check=False)
    best = None
    for _ in range(2):
        s = mk()
        s.solve()
        best = s.gpu["solve_ms"] if best is None else min(best, s.gpu["solve_ms"])
    g = s.gpu
    print(json.dumps(dict(n=n, per_row=per, solve_ms=round(best, 2), its=s.meta["its"], us_per_round=round(1e3 * best / max(s.meta["its"], 1), 2),
                          grid_rounds=g["grid_rounds"], tail_rounds=g["tail_rounds"], tail_modes={k: (v["rounds"], v["us_per_round"]) for k, v in g["tail_modes"].items()})), flush=True)
