#!/usr/bin/env python3
"""Solve time of dense random matrices (the reference's `mat=` shape: every row holds every object) by size, on the
GPU box.
usage: dense_sizes.py [n ...]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from sslap_amd import from_matrix
for n in [int(x) for x in sys.argv[1:]] or [1500, 3000, 8000]:
    mat = np.float64(np.float32(np.random.RandomState(n).uniform(0, 100, (n, n))))
    best = None
    for _ in range(2):
        s = from_matrix(mat, problem="max", max_iter=10**8, cardinality_check=False)
        s.solve()
        best = s.gpu["solve_ms"] if best is None else min(best, s.gpu["solve_ms"])
    print(json.dumps(dict(n=n, solve_ms=round(best, 2), its=s.meta["its"], obj=s.meta["obj"])), flush=True)
