#!/usr/bin/env python3
"""Tuning sweep on the GPU box: threshold of the full-scan engine (tiled_min_K, as a fraction of N) vs solve time.
usage: sweep_tiled_min_k.py <config> [fractions, comma separated] [reps]"""
import json, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sslap_amd import AuctionSolver, synth
cfg = sys.argv[1] if len(sys.argv) > 1 else "C4"
fracs = [float(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else (0.05, 0.1, 0.15, 0.2, 0.3)
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
loc, val = synth.gen_config(cfg)
nnz = loc.shape[0]
n = int(loc[:, 0].max()) + 1
dl, dv = torch.from_numpy(loc).cuda(), torch.from_numpy(val).cuda()
want = None
for f in fracs:
    best = None
    for rep in range(reps):
        s = AuctionSolver.from_device_pointers(dl.data_ptr(), dv.data_ptr(), nnz, problem="max", max_iter=10**8,
                                               tiled_min_k=max(1, int(f * n)), profile=1)
        sol = s.solve()
        d = synth.sol_digest(sol)
        want = want or d
        assert d == want, "the threshold changed the assignment"
        ms = s.gpu["solve_ms"]
        best = ms if best is None else min(best, ms)
    g = s.gpu
    print(json.dumps(dict(config=cfg, frac=f, tiled_min_K=int(f * n), solve_ms=round(best, 3), grid_rounds=g["grid_rounds"],
                          tail_rounds=g["tail_rounds"], sha=d[:8], engine_launches=g.get("tiled_launches"),
                          engine_us_per_launch=round(1e3 * g.get("tiled_ms", 0) / max(g.get("tiled_launches", 0), 1), 2),
                          engine_frac_of_8TBs=round(g.get("tiled_edges", 0) * 8 / max(g.get("tiled_ms", 0), 1e-9) / 1e6 / 8000, 4))), flush=True)
