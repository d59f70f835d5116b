#!/usr/bin/env python3
"""Solve time of sparse random problems by shape (n, edges per row) on the GPU box: sparse_shapes.py n:per_row ..."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sslap_amd import from_sparse, synth
for spec in sys.argv[1:] or ["20000:500"]:
    n, per = (int(x) for x in spec.split(":"))
    loc, val = synth.gen_sparse(n, n, per / n, seed=n + per)
    best = None
    for _ in range(2):
        s = from_sparse(loc, val.copy(), problem="max", max_iter=10**8, cardinality_check=False)
        s.solve()
        best = s.gpu["solve_ms"] if best is None else min(best, s.gpu["solve_ms"])
    print(json.dumps(dict(n=n, per_row=per, solve_ms=round(best, 2), its=s.meta["its"], tail_rounds=s.gpu["tail_rounds"],
                          lines_active=s.gpu["lines_active"])), flush=True)
