#!/usr/bin/env python3
"""A few engine launches of a BASELINE config and nothing else: the short program the rocprofv3 --pmc passes on
k_bid_tiled are collected on (tools/pmc_passes.sh).  max_iter = 1: the K = N scan alone; max_iter = 2: the K = N scan
(forwards) and the first partial round of the phase behind it (backwards, kRev).
usage: pmc_scan.py [config] [repeats] [rounds] [shuffle]"""
import sys

import torch  # noqa: F401  (first: one HIP runtime per process)

from sslap_amd import AuctionSolver, synth

cfg = sys.argv[1] if len(sys.argv) > 1 else "C3"
rep = int(sys.argv[2]) if len(sys.argv) > 2 else 3
rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 1
loc, val = synth.gen_config(cfg)
if len(sys.argv) > 4 and sys.argv[4] == "shuffle":
    loc, val = synth.shuffle_within_rows(loc, val, 5)
d_loc, d_val = torch.from_numpy(loc).cuda(), torch.from_numpy(val).cuda()
for _ in range(rep):
    s = AuctionSolver.from_device_pointers(d_loc.data_ptr(), d_val.data_ptr(), int(loc.shape[0]), problem="max",
                                           max_iter=rounds, device=0)
    s.solve()
    print(s.meta["its"], s.gpu["edges_scanned"], s.gpu.get("tiled_format"), flush=True)
