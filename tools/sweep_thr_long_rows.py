"""Tail threshold (default / 96 / 192) on long-row shapes: dense n x n and sparse rows of 300-1000 edges.
Run on the GPU box: python tools/sweep_thr_long_rows.py (nothing runs at import)."""
import json, os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from sslap_amd import from_matrix, from_sparse, synth


def run(mk, label):
    for thr in (None, 96, 192):
        best = None
        for _ in range(2):
            s = mk(thr); s.solve()
            best = s.gpu["solve_ms"] if best is None else min(best, s.gpu["solve_ms"])
        print(label, "thr", thr, "solve_ms", round(best, 2), "grid", s.gpu["grid_rounds"], "tail", s.gpu["tail_rounds"], flush=True)


def main():
    for n in (500, 1000):
        mat = np.float64(np.float32(np.random.RandomState(n).uniform(0, 100, (n, n))))
        run(lambda thr: from_matrix(mat, problem="max", max_iter=10**8, cardinality_check=False, **({} if thr is None else dict(tail_threshold=thr))), f"dense{n}")
    for n, per in ((10000, 1000), (20000, 500), (40000, 300), (5000, 400)):
        loc, val = synth.gen_sparse(n, n, per / n, seed=n + per)
        run(lambda thr: from_sparse(loc, val.copy(), problem="max", max_iter=10**8, cardinality_check=False, **({} if thr is None else dict(tail_threshold=thr))), f"sparse{n}:{per}")


if __name__ == "__main__":
    main()
