#!/usr/bin/env python3
"""Analysis tooling (CPU, tools/sim model): reuse distance of bidders in the small rounds of a solve.
Answers: how many recently-bidding persons' rows must stay on chip for a given hit rate."""
import subprocess
import sys
import os
import numpy as np
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from sslap_amd import synth

cfg = sys.argv[1] if len(sys.argv) > 1 else "C2"
thr = int(sys.argv[2]) if len(sys.argv) > 2 else 256
# the trace comes from the analysis model tools/sim/auction_sim.c (checked against the golden fixture by run_sim.py)
subprocess.check_call([sys.executable, os.path.join(HERE, "sim", "run_sim.py"), cfg, "0", "1", "0", "0"],
                      env=dict(os.environ, SIM_TRACE=f"/tmp/sim_trace_{cfg}.bin", SIM_TRACE_THR=str(thr)))
loc, _ = synth.gen_config(cfg)
tr = np.fromfile(f"/tmp/sim_trace_{cfg}.bin", dtype=np.int32)
rounds = int((tr == -1).sum())
seq = tr[tr >= 0]
print(f"{cfg}: small rounds={rounds} bids in them={seq.size}")
# LRU stack distance via last-use timestamps + Fenwick tree
N = int(loc[:, 0].max()) + 1
last = np.full(N, -1, dtype=np.int64)
size = seq.size
tree = np.zeros(size + 1, dtype=np.int64)
def add(i, v):
    i += 1
    while i <= size:
        tree[i] += v
        i += i & -i
def prefix(i):
    r = 0
    while i > 0:
        r += tree[i]
        i -= i & -i
    return r
caps = [16, 64, 100, 256, 1024, 2500, 10000, 50000]
hits = {c: 0 for c in caps}
cold = 0
for t, p in enumerate(seq.tolist()):
    lp = last[p]
    if lp >= 0:
        d = prefix(t) - prefix(lp + 1)  # distinct persons since the last use
        for c in caps:
            if d < c:
                hits[c] += 1
        add(lp, -1)
    else:
        cold += 1
    add(t, 1)
    last[p] = t
print("cold (first touch in the trace):", cold, f"{cold/size:.3f}")
for c in caps:
    print(f"LRU capacity {c:6d} rows: hit rate {hits[c]/size:.3f}")
