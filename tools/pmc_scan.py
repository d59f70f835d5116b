#!/usr/bin/env python3
"""A few full-scan bid launches (K = N) of a BASELINE config and nothing else: the short program the
rocprofv3 --pmc passes on k_bid_tiled are collected on (max_iter = 1 => one bid round per solve).
usage: pmc_scan.py [config] [repeats]"""
import sys

import torch  # noqa: F401  (first: one HIP runtime per process)

from sslap_amd import AuctionSolver, synth

cfg = sys.argv[1] if len(sys.argv) > 1 else "C3"
rep = int(sys.argv[2]) if len(sys.argv) > 2 else 3
loc, val = synth.gen_config(cfg)
d_loc, d_val = torch.from_numpy(loc).cuda(), torch.from_numpy(val).cuda()
for _ in range(rep):
    s = AuctionSolver.from_device_pointers(d_loc.data_ptr(), d_val.data_ptr(), int(loc.shape[0]), problem="max",
                                           max_iter=1, device=0)
    s.solve()
    print(s.meta["its"], s.gpu["edges_scanned"], flush=True)
