"""Maximum bipartite matching -- the feasibility guard of the front-end and the reference's public
`sslap.hopcroft_solve` (sslap/check_feasible.py:5-21 over feasibility_.pyx:227-283).

The matching itself runs in libmisslap.so (`misslap_hopcroft_karp`, host C++: the reference's Hopcroft-Karp with
the same phase / visiting order, so the pairing arrays are identical, without its N^2 queue and without
recursion); this module only assembles `loc` from the three input forms exactly like the reference does."""
import ctypes as C

import numpy as np

from . import _lib


def _solve(loc, n_rows, n_cols):
    loc = np.ascontiguousarray(loc, dtype=np.int32)
    left = np.empty(n_rows, dtype=np.int32)
    right = np.empty(n_cols, dtype=np.int32)
    size = C.c_int32()
    _lib.check(_lib.load().misslap_hopcroft_karp(loc.ctypes.data_as(C.c_void_p), int(loc.shape[0]), int(n_rows),
                                                 int(n_cols), C.byref(size), left.ctypes.data_as(C.c_void_p),
                                                 right.ctypes.data_as(C.c_void_p)))
    return dict(size=int(size.value), left_pairings=left, right_pairings=right)


def matching_gpu(loc, n_rows, n_cols, device=None):
    """Maximum matching on the GPU (misslap_matching_gpu: BFS-layered, one launch per layer).  Same cardinality as
    `hopcroft_solve` / the reference; the pairings are a maximum matching, not necessarily the reference's."""
    import os
    loc = np.ascontiguousarray(loc, dtype=np.int32)
    left = np.empty(n_rows, dtype=np.int32)
    right = np.empty(n_cols, dtype=np.int32)
    size, phases = C.c_int32(), C.c_int32()
    dev = int(os.environ.get("MISSLAP_DEVICE", 0)) if device is None else int(device)
    _lib.check(_lib.load().misslap_matching_gpu(loc.ctypes.data_as(C.c_void_p), int(loc.shape[0]), int(n_rows),
                                                int(n_cols), dev, C.byref(size), left.ctypes.data_as(C.c_void_p),
                                                right.ctypes.data_as(C.c_void_p), C.byref(phases)))
    return dict(size=int(size.value), left_pairings=left, right_pairings=right, phases=int(phases.value))


def cardinality(loc, n_rows, n_cols):
    """Size of a maximum matching of the graph `loc` (int[nnz, 2], rows ascending) -- what reaches the auction
    front-end (auction_.pyx:562-566, :608-612)."""
    return _solve(loc, n_rows, n_cols)["size"]


def hopcroft_solve(loc=None, mat=None, lookup=None):
    """Maximum matching of a bipartite graph with vertex sets I and J, given as ONE of

    loc: (E x 2) integer ndarray of edges (i, j), rows ascending;
    mat: 2D float ndarray, A_ij >= 0 marks an edge (negative: no edge);
    lookup: dict i -> list / ndarray of j.

    Returns dict(size, left_pairings int32[|I|], right_pairings int32[|J|]); -1 = unmatched.
    """
    n_none = (loc is None) + (mat is None) + (lookup is None)
    assert n_none == 2, "Exactly one of the arguments loc, mat, lookup must be provided."  # feasibility_.pyx:232
    if loc is not None:
        loc = np.asarray(loc)
        n, m = int(loc[:, 0].max()) + 1, int(loc[:, 1].max()) + 1  # :244-245
        return _solve(loc, n, m)
    if mat is not None:
        mat = np.asarray(mat)
        n, m = mat.shape
        r, c = np.nonzero(mat >= 0)  # row-major scan of :256-262
        return _solve(np.stack([r, c], axis=1), n, m)
    n = max(lookup) + 1  # :267
    m = max(map(max, lookup.values())) + 1
    edges = [(i, j) for i in lookup for j in lookup[i]]  # dict order, :273-277
    return _solve(np.asarray(edges, dtype=np.int64).reshape(-1, 2), n, m)
