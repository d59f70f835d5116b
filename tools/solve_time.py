#!/usr/bin/env python3
"""Wall time of full solves of a BASELINE config with a given set of GPU options (A/B of host-side knobs).
usage: solve_time.py [config] [repeats] key=value ...   e.g. profile=0 rounds_per_sync=8"""
import json
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sslap_amd import AuctionSolver, synth

cfg = sys.argv[1] if len(sys.argv) > 1 else "C3"
rep = int(sys.argv[2]) if len(sys.argv) > 2 else 3
opts = {k: int(v) for k, v in (kv.split("=") for kv in sys.argv[3:])}
loc, val = synth.gen_config(cfg)
dl, dv = torch.from_numpy(loc).cuda(), torch.from_numpy(val).cuda()
out = []
for _ in range(rep):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    s = AuctionSolver.from_device_pointers(dl.data_ptr(), dv.data_ptr(), int(loc.shape[0]), problem="max",
                                           max_iter=10**8, **opts)
    s.solve()
    out.append(round(1e3 * (time.perf_counter() - t0), 1))
print(json.dumps({"cfg": cfg, "opts": opts, "wall_ms": out, "solve_ms": round(s.gpu["solve_ms"], 1),
                  "its": s.meta["its"]}), flush=True)
