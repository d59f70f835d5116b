import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden"))
import numpy as np
from oracle import oracle as orc
from sslap_amd import from_sparse, synth, solve_batch

def run(n, m, dens, B, group, ints=0, thr=None, **kw):
    probs = [synth.gen_sparse(n, m, dens, seed=100 + k, integer_values=ints) for k in range(B)]
    t0 = time.perf_counter()
    singles = []
    for loc, val in probs:
        s = from_sparse(loc, val.copy(), problem="max", cardinality_check=False, max_iter=10**8, tail_threshold=thr, **kw)
        singles.append((s.solve(), dict(s.meta), s.gpu["obj_f64"], s.gpu["edges_scanned"]))
    t_single = time.perf_counter() - t0
    solvers = [from_sparse(loc, val.copy(), problem="max", cardinality_check=False, max_iter=10**8, tail_threshold=thr, **kw) for loc, val in probs]
    t0 = time.perf_counter()
    sols, info = solve_batch(solvers, group)
    t_batch = time.perf_counter() - t0
    ok = True
    for k in range(B):
        same = np.array_equal(sols[k], singles[k][0]) and solvers[k].meta["its"] == singles[k][1]["its"] and solvers[k].gpu["obj_f64"] == singles[k][2] and solvers[k].gpu["edges_scanned"] == singles[k][3]
        ok = ok and same
    ref = orc.auction_solve(loc=probs[0][0], val=probs[0][1].copy(), problem="max", cardinality_check=False, max_iter=10**8)
    ok0 = np.array_equal(sols[0], ref["sol"])
    print(f"n={n} m={m} B={B} group={group} thr={thr} kw={kw}: all equal to single solves: {ok}; problem 0 = oracle: {ok0}; "
          f"sequential {1e3*t_single:.1f} ms, batch {1e3*t_batch:.1f} ms ({t_single/t_batch:.1f}x); info {info}", flush=True)
    return ok and ok0

good = True
good &= run(300, 300, 0.05, 2, 0)
good &= run(300, 300, 0.05, 5, 0, ints=4)
good &= run(3000, 3000, 0.01, 8, 0)
good &= run(3000, 4000, 0.01, 8, 3, thr=0)
good &= run(6000, 40000, 0.001, 6, 0, tiled_min_k=1, engine=1)
good &= run(20000, 20000, 0.002, 24, 12)
good &= run(20000, 20000, 0.002, 24, 6)
good &= run(50000, 50000, 0.005, 24, 12)
good &= run(50000, 50000, 0.005, 24, 8)
print("ALL OK" if good else "MISMATCH")
