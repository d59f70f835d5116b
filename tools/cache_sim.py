#!/usr/bin/env python3
"""Analysis (CPU, oracle): hit rate of an exact per-person top-C candidate cache on the bids of the small rounds.
A hit = the top-2 of the cached candidates at current prices provably equals the row's top-2 (prices only rise, so
every non-cached edge is bounded by the (C+1)-th value at build time).  usage: cache_sim.py [config] [C] [thr]"""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import oracle as orc
from sslap_amd import synth

cfg = sys.argv[1] if len(sys.argv) > 1 else "C3"
cc = int(sys.argv[2]) if len(sys.argv) > 2 else 8
thr = int(sys.argv[3]) if len(sys.argv) > 3 else 40
loc, val = synth.gen_config(cfg)
s = orc.from_sparse(loc, val, problem="max", max_iter=10**8, cardinality_check=False)
L = orc.lib()
L.oracle_set_cache_sim.argtypes = [C.c_void_p, C.c_int, C.c_int]
L.oracle_get_cache_sim.argtypes = [C.c_void_p] + [C.POINTER(C.c_int64)] * 3
L.oracle_set_cache_sim(s._h, cc, thr)
s.solve()
h, m, b = C.c_int64(), C.c_int64(), C.c_int64()
L.oracle_get_cache_sim(s._h, C.byref(h), C.byref(m), C.byref(b))
print(f"{cfg} C={cc} thr={thr}: its={s.meta['its']} hits={h.value} misses={m.value} "
      f"hit rate={h.value / max(h.value + m.value, 1):.3f} inconsistent={b.value}")
