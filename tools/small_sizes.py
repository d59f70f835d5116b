#!/usr/bin/env python3
"""End-to-end time of auction_solve(mat) for small dense matrices (the reference's everyday use), on the GPU box.
usage: small_sizes.py [n ...]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from sslap_amd import auction_solve
for n in [int(x) for x in sys.argv[1:]] or [20, 100, 300, 1000]:
    mat = np.random.RandomState(n).uniform(0, 100, (n, n))
    for _ in range(3):
        auction_solve(mat.copy(), problem="max", cardinality_check=False)
    ts = []
    for _ in range(20):
        t0 = time.perf_counter()
        r = auction_solve(mat.copy(), problem="max", cardinality_check=False)
        ts.append(1e3 * (time.perf_counter() - t0))
    print(json.dumps(dict(n=n, wall_ms_median=round(sorted(ts)[len(ts) // 2], 3), wall_ms_min=round(min(ts), 3), its=r["meta"]["its"],
                          live=os.environ.get("MISSLAP_LIVE_STATUS", "1"))), flush=True)
