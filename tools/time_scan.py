#!/usr/bin/env python3
"""Time the full-scan bid kernel of the library selected by MISSLAP_LIB (default: the in-tree build):
N launches of the product kernel on a fresh C3 state (misslap_debug_time_bid mode 10), plus optional modes.
usage: time_scan.py [config] [label] [modes...]      -> one JSON line"""
import ctypes as C
import json
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sslap_amd import AuctionSolver, synth, _lib

cfg = sys.argv[1] if len(sys.argv) > 1 else "C3"
label = sys.argv[2] if len(sys.argv) > 2 else os.environ.get("MISSLAP_LIB", "in-tree")
modes = [int(m) for m in sys.argv[3:]] or [10, 10, 10]
loc, val = synth.gen_config(cfg)
dl, dv = torch.from_numpy(loc).cuda(), torch.from_numpy(val).cuda()
s = AuctionSolver.from_device_pointers(dl.data_ptr(), dv.data_ptr(), int(loc.shape[0]), problem="max", max_iter=10**8,
                                       tiled_shape=int(os.environ.get("MISSLAP_TILED_SHAPE", 0)))  # ablation modes 11-16 need shape 3
out = []
for m in modes:
    ms = C.c_float()
    _lib.check(_lib.load_diag().misslap_debug_time_bid(s._h, m, 40, C.byref(ms)))
    out.append(round(ms.value * 1e3, 1))
print(json.dumps({"label": label, "cfg": cfg, "modes": modes, "us": out}), flush=True)
