#!/usr/bin/env python3
"""Where a solve's time goes (GPU box): per-mode accounting of the tail kernel, candidate-line hit rate, wall time.
usage: tail_stats.py [config] [repeats] key=value ...   e.g. cand=0 tail_threshold=64 profile=1"""
import json
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sslap_amd import AuctionSolver, synth

cfg = sys.argv[1] if len(sys.argv) > 1 else "C3"
rep = int(sys.argv[2]) if len(sys.argv) > 2 else 2
opts = {k: int(v) for k, v in (kv.split("=") for kv in sys.argv[3:])}
opts.setdefault("profile", 1)
loc, val = synth.gen_config(cfg)
dl, dv = torch.from_numpy(loc).cuda(), torch.from_numpy(val).cuda()
wall = []
for _ in range(rep):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    s = AuctionSolver.from_device_pointers(dl.data_ptr(), dv.data_ptr(), int(loc.shape[0]), problem="max",
                                           max_iter=10**8, **opts)
    sol = s.solve()
    wall.append(round(1e3 * (time.perf_counter() - t0), 1))
g = s.gpu
print(json.dumps({
    "cfg": cfg, "opts": opts, "wall_ms": wall, "solve_ms": round(g["solve_ms"], 1), "its": s.meta["its"],
    "sol_sha256": synth.sol_digest(sol)[:16], "grid_rounds": g["grid_rounds"], "tail_rounds": g["tail_rounds"],
    "tail_ms": round(g.get("tail_ms", 0.0), 1), "tail_modes": g["tail_modes"], "tail_cand": g["tail_cand"], "tail_raw": g["tail_raw"],
    "bids": g["bids_made"], "cand_hits": g["cand_hits"], "hit_rate": round(g["cand_hits"] / max(g["bids_made"], 1), 3),
    "edges_scanned": g["edges_scanned"], "cand_edges": g["cand_edges"],
    "fullscan_us": round(1e3 * g.get("fullscan_ms", 0) / max(g.get("fullscan_launches", 0), 1), 1),
    "tiled": dict(launches=g.get("tiled_launches"), ms=round(g.get("tiled_ms", 0), 3), edges=g.get("tiled_edges"),
                  frac=round(g.get("tiled_edges", 0) * 8 / max(g.get("tiled_ms", 0), 1e-9) / 1e6 / 8000, 4)),
    "k_bid": dict(launches=g.get("bid_launches"), ms=round(g.get("bid_ms", 0), 3), edges=g.get("bid_edges")),
}), flush=True)
