#!/usr/bin/env python3
"""Like shape_scan.py, over problem kind / rectangular / integer-valued variants: us per round and the regime split."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sslap_amd import from_sparse, synth
for n, m, per, ints, prob in [(20000, 20000, 50, 0, "max"), (20000, 20000, 50, 0, "min"), (20000, 20000, 50, 5, "max"), (20000, 20000, 50, 100, "max"),
                              (20000, 30000, 50, 0, "max"), (20000, 80000, 50, 0, "max"), (20000, 30000, 50, 0, "min"),
                              (100000, 100000, 20, 0, "max"), (100000, 100000, 20, 3, "max"), (100000, 400000, 20, 0, "max"),
                              (5000, 5000, 100, 0, "min"), (5000, 50000, 100, 0, "max")]:
    loc, val = synth.gen_sparse(n, m, per / m, seed=n + per + ints, integer_values=ints)
    best = None
    for _ in range(2):
        s = from_sparse(loc, val.copy(), problem=prob, max_iter=10**8, cardinality_check=False)
        s.solve()
        best = s.gpu["solve_ms"] if best is None else min(best, s.gpu["solve_ms"])
    g = s.gpu
    print(json.dumps(dict(n=n, m=m, per_row=per, ints=ints, prob=prob, solve_ms=round(best, 2), its=s.meta["its"],
                          us_per_round=round(1e3 * best / max(s.meta["its"], 1), 2), grid_rounds=g["grid_rounds"], tail_rounds=g["tail_rounds"],
                          tiled=g["tiled_active"], lines=g["lines_active"], bpe=g["bytes_per_edge"],
                          tail_modes={k: (v["rounds"], v["us_per_round"]) for k, v in g["tail_modes"].items()})), flush=True)
