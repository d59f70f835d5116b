import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden"))
import numpy as np
import cases
from oracle import oracle as orc
from sslap_amd import from_sparse
spec = dict(kind="sparse", n=6000, m=40000, density=0.001)
loc, val = cases.synth_inputs(spec)
for shape in (9, 8, 0):
    for trial in range(3):
        bad = None
        for r in range(1, 16):
            o = orc.from_sparse(loc, val.copy(), problem="max", max_iter=r, cardinality_check=False); o.solve(); so = o.state()
            g = from_sparse(loc, val.copy(), problem="max", max_iter=r, cardinality_check=False, tail_threshold=0, tiled_min_k=1, engine=1, tiled_shape=shape)
            g.solve(); sg = g.state()
            same = sg["K"] == so["K"] and np.array_equal(sg["p"].view(np.uint64), so["p"].view(np.uint64)) and np.array_equal(sg["U"], so["U"])
            if not same:
                dp = np.nonzero(sg["p"].view(np.uint64) != so["p"].view(np.uint64))[0]
                bad = (r, sg["K"], so["K"], dp[:8].tolist(), sg["p"][dp[:4]].tolist(), so["p"][dp[:4]].tolist())
                break
        print("shape", shape, "trial", trial, "first bad round:", bad, flush=True)
