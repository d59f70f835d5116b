#!/usr/bin/env python3
"""Tuning sweep on the GPU box: tail threshold / rounds per sync vs solve time (C3)."""
import json, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sslap_amd import AuctionSolver, synth
cfg = sys.argv[1] if len(sys.argv) > 1 else "C3"
thrs = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else (48, 64, 96, 128, 160, 192, 256, 384)
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 2
loc, val = synth.gen_config(cfg)
nnz = loc.shape[0]
dl, dv = torch.from_numpy(loc).cuda(), torch.from_numpy(val).cuda()
out = {}
for thr in thrs:
    for rps in (8,):
        best = None
        for rep in range(reps):
            s = AuctionSolver.from_device_pointers(dl.data_ptr(), dv.data_ptr(), nnz, problem="max", max_iter=10**8,
                                                   tail_threshold=thr, rounds_per_sync=rps)
            s.solve()
            ms = s.gpu["solve_ms"]
            best = ms if best is None else min(best, ms)
        out[f"thr{thr}_rps{rps}"] = dict(solve_ms=round(best, 1), grid_rounds=s.gpu["grid_rounds"], tail_rounds=s.gpu["tail_rounds"])
        print(thr, rps, out[f"thr{thr}_rps{rps}"], flush=True)
for rps in ((4, 8, 16, 32) if len(sys.argv) <= 2 else ()):
    s = AuctionSolver.from_device_pointers(dl.data_ptr(), dv.data_ptr(), nnz, problem="max", max_iter=10**8, rounds_per_sync=rps)
    s.solve()
    print("rps", rps, round(s.gpu["solve_ms"], 1), flush=True)
