#!/usr/bin/env python3
"""One instance beyond the BASELINE sizes (no fixture, no oracle run: too slow), checked through the properties of
tests/test_gpu_parity.py::test_fullsize_properties_without_fixture: permutation, chosen edges exist, eps-complementary
slackness at 1/N re-derived on the host from the prices, weak duality.  usage: scale_check.py [N] [edges_per_row]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sslap_amd import from_sparse, synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
per_row = float(sys.argv[2]) if len(sys.argv) > 2 else 80.0
t0 = time.time()
loc, val = synth.gen_sparse(n, n, per_row / n, seed=11)
print(f"generated {loc.shape[0]} entries in {time.time() - t0:.0f} s", flush=True)
s = from_sparse(loc, val.copy(), problem="max", max_iter=10**9, cardinality_check=False)
t0 = time.time()
sol = s.solve()
print(f"solved in {time.time() - t0:.2f} s: {s.meta['its']} rounds, {s.meta['nreductions']} eps reductions, "
      f"tail {s.gpu['tail_rounds']} rounds, line hit rate {s.gpu['cand_hits'] / max(1, s.gpu['bids_made']):.3f}", flush=True)
st = s.state()
assert len(np.unique(sol)) == n and s.meta["soln_found"] == 1 and st["K"] == 0
key = loc[:, 0].astype(np.int64) * n + loc[:, 1]
pick = np.searchsorted(key, np.arange(n, dtype=np.int64) * n + sol)
assert np.array_equal(key[pick], np.arange(n, dtype=np.int64) * n + sol)
obj = val[pick].sum()
assert abs(obj - s.gpu["obj_f64"]) <= 1e-9 * abs(obj)
p = st["p"]
v = val - p[loc[:, 1]]
rowmax = np.maximum.reduceat(v, np.searchsorted(loc[:, 0], np.arange(n)))
chosen = val[pick] - p[sol]
assert (chosen + 1.0 / n + 1e-7 >= rowmax).all()
assert obj >= p.sum() + rowmax.sum() - 1.0 - 1e-6 * abs(obj)
print("properties hold: permutation, edges exist, eps-CS at 1/N, weak duality; objective", obj)
