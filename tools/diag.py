#!/usr/bin/env python3
"""Diagnostics on the GPU box: bid-kernel ablations (libmisslap_diag.so) and the tail kernel's accounting."""
import ctypes as C
import json
import sys
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sslap_amd import AuctionSolver, synth, _lib

cfg = sys.argv[1] if len(sys.argv) > 1 else "C3"
loc, val = synth.gen_config(cfg)
nnz = loc.shape[0]
dl, dv = torch.from_numpy(loc).cuda(), torch.from_numpy(val).cuda()
s = AuctionSolver.from_device_pointers(dl.data_ptr(), dv.data_ptr(), nnz, problem="max", max_iter=10**8)
out = {"cfg": cfg, "nnz": int(nnz)}
names = {0: "complete", 1: "no_gather", 2: "no_reduce", 3: "stream_only"}
if "--tiled-ablate" in sys.argv:
    s3 = AuctionSolver.from_device_pointers(dl.data_ptr(), dv.data_ptr(), nnz, problem="max", max_iter=10**8, tiled_shape=3)
    for mode, nm in ((10, "complete"), (11, "no_fill"), (12, "no_lookup_no_arith"), (15, "lookup_no_arith"), (16, "arith_no_lookup"),
                     (13, "no_edge_loads"), (14, "no_barriers"), (17, "half_fill"), (10, "complete2")):
        ms = C.c_float()
        _lib.check(_lib.load_diag().misslap_debug_time_bid(s3._h, mode, 20, C.byref(ms)))
        out["tiled_" + nm + "_us"] = round(ms.value * 1e3, 1)
    # the production shape (three loader wavefronts), back to back and COLD (1 GiB written between the launches: a scan's
    # 260 MB fit the MALL, back to back they never come from HBM)
    for cold, tag in ((0, "hot"), (0x100, "cold")):
        for mode, nm in ((20, "complete"), (21, "no_fill"), (22, "no_lookup_no_arith"), (27, "half_fill"), (20, "complete2")):
            ms = C.c_float()
            _lib.check(_lib.load_diag().misslap_debug_time_bid(s._h, mode | cold, 20, C.byref(ms)))
            out[f"shape0_{tag}_{nm}_us"] = round(ms.value * 1e3, 1)
        if tag == "hot":
            for mode, nm in ((10, "complete"), (17, "half_fill"), (12, "no_lookup_no_arith")):
                ms = C.c_float()
                _lib.check(_lib.load_diag().misslap_debug_time_bid(s3._h, mode | 0x100, 20, C.byref(ms)))
                out[f"shape3_cold_{nm}_us"] = round(ms.value * 1e3, 1)
    print(json.dumps(out, indent=1))
    sys.exit(0)
for mode in (0, 1, 2, 3, 0):
    ms = C.c_float()
    _lib.check(_lib.load_diag().misslap_debug_time_bid(s._h, mode, 20, C.byref(ms)))
    out["bid_" + names[mode] + "_us"] = round(ms.value * 1e3, 1)
    out["bid_" + names[mode] + "_GBs"] = round(nnz * 8 / (ms.value * 1e-3) / 1e9, 1)
# full-scan timing inside a real solve (HIP events), tiled vs gather kernel
shapes = ["1024x4x2x2h_L3", "1024x4x2x2h_L0", "1024x4x2x3big", "1024x4x2x2h_L1", "1024x8x4x1h_L1_g8", "1024x4x1x2h_L1", "1024x4x2x3h_L1", "1024x4x2x2h_L2"]
for tk, shape, eng, name in [(0, k, 1, "tiled_" + n) for k, n in enumerate(shapes)] + [(-1, 0, 0, "gather_only")]:
    st = AuctionSolver.from_device_pointers(dl.data_ptr(), dv.data_ptr(), nnz, problem="max", max_iter=10**8,
                                            profile=3, tiled_min_k=tk, tiled_shape=shape, engine=eng)
    st.solve()
    g = st.gpu
    out["solve_" + name] = dict(solve_ms=g["solve_ms"], setup_ms=g["setup_ms"], fullscan_us=1e3 * g["fullscan_ms"] / max(g["fullscan_launches"], 1),
                                fullscan_GBs=g["fullscan_edges"] * 8 / max(g["fullscan_ms"], 1e-9) / 1e6,
                                tiled_launches=g["tiled_launches"], tiled_ms=g["tiled_ms"], tiled_edges=g["tiled_edges"],
                                merge_ms=g["merge_ms"], merge_launches=g["merge_launches"],
                                bid_launches=g["bid_launches"], bid_ms=g["bid_ms"], bid_edges=g["bid_edges"], tail_ms=g["tail_ms"],
                                grid_rounds=g["grid_rounds"], its=st.meta["its"])
if "--tail" in sys.argv:
    # per-mode accounting of the tail kernel (always on) -- cycles per segment of a chain / team round come from the
    # diagnostic builds: hipcc ... -DMISSLAP_TAIL_STAMP (chain) or -DMISSLAP_TAIL_STAMP_TEAM, run through
    # MISSLAP_LIB=<that build> python tools/tail_stats.py (prints meta.tail_stats raw)
    s2 = AuctionSolver.from_device_pointers(dl.data_ptr(), dv.data_ptr(), nnz, problem="max", max_iter=10**8, profile=1)
    s2.solve()
    g = s2.gpu
    out["tail"] = dict(rounds=g["tail_rounds"], tail_ms=g["tail_ms"], us_per_round=1e3 * g["tail_ms"] / g["tail_rounds"],
                       modes=g["tail_modes"], lines=g["tail_cand"], bids=g["bids_made"], tail_edges=g["tail_edges"])
print(json.dumps(out, indent=1))
