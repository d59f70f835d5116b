#!/usr/bin/env python3
"""Analysis tooling (CPU, oracle): reuse distance of bidders in the small rounds of a solve.
Answers: how many recently-bidding persons' rows must stay on chip for a given hit rate."""
import ctypes as C
import sys
import os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import oracle as orc
from sslap_amd import synth

cfg = sys.argv[1] if len(sys.argv) > 1 else "C2"
thr = int(sys.argv[2]) if len(sys.argv) > 2 else 256
loc, val = synth.gen_config(cfg)
s = orc.from_sparse(loc, val, problem="max", max_iter=10**8, cardinality_check=False)
L = orc.lib()
L.oracle_set_trace.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int]
L.oracle_trace_len.argtypes = [C.c_void_p]
L.oracle_trace_len.restype = C.c_int64
buf = np.empty(60_000_000, dtype=np.int32)
L.oracle_set_trace(s._h, buf.ctypes.data, buf.size, thr)
s.solve()
n = L.oracle_trace_len(s._h)
tr = buf[:n]
rounds = int((tr == -1).sum())
seq = tr[tr >= 0]
print(f"{cfg}: its={s.meta['its']} small rounds={rounds} bids in them={seq.size}")
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
