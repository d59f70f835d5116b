"""ctypes wrapper of oracle/liboracle.so -- TEST INFRASTRUCTURE ONLY.

Mirrors the reference's Python front-end (sslap/auction_solve.py:6-55) and its
Cython adapters (sslap/auction_.pyx:528-617) on top of the plain-C restatement,
including the adapters' dimension quirks,
so that golden vectors captured from the reference can be replayed 1:1.
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


class OracleMeta(C.Structure):
    _fields_ = [
        ("start_eps", C.c_float), ("final_eps", C.c_float), ("target_eps", C.c_float),
        ("eCE", C.c_int), ("soln_found", C.c_int),
        ("its", C.c_int), ("nreductions", C.c_int), ("n_assigned", C.c_int), ("num_unassigned", C.c_int),
        ("obj_f32", C.c_float), ("obj_f64", C.c_double),
        ("edges_scanned", C.c_uint64), ("bids_made", C.c_uint64),
        ("t_bid", C.c_double), ("t_total", C.c_double),
        ("num_rows", C.c_int64), ("num_cols", C.c_int64),
    ]


def build(force=False):
    so = os.path.join(_HERE, "liboracle.so")
    src = os.path.join(_HERE, "auction_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B", "liboracle.so"])
    return so


def lib():
    global _LIB
    if _LIB is None:
        L = C.CDLL(build())
        L.oracle_create.restype = C.c_void_p
        L.oracle_create.argtypes = [C.c_int64, C.c_void_p, C.c_void_p, C.c_int, C.c_float, C.c_int64]
        L.oracle_solve.argtypes = [C.c_void_p]
        L.oracle_step.argtypes = [C.c_void_p]
        L.oracle_step.restype = C.c_int
        L.oracle_get_meta.argtypes = [C.c_void_p, C.POINTER(OracleMeta)]
        L.oracle_destroy.argtypes = [C.c_void_p]
        L.oracle_set_timing.argtypes = [C.c_void_p, C.c_int]
        L.oracle_set_assign_by_bidders.argtypes = [C.c_void_p, C.c_int]
        L.oracle_objective.argtypes = [C.c_void_p]
        L.oracle_objective.restype = C.c_double
        for name, typ in (("oracle_person_to_object", C.c_int), ("oracle_object_to_person", C.c_int),
                          ("oracle_prices", C.c_double), ("oracle_unassigned", C.c_int),
                          ("oracle_row_ptr", C.c_int)):
            f = getattr(L, name)
            f.argtypes = [C.c_void_p]
            f.restype = C.POINTER(typ)
        L.oracle_ece_arrays.restype = C.c_int
        L.oracle_ece_arrays.argtypes = [C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_float]
        L.oracle_dense_to_coo.restype = C.c_int64
        L.oracle_dense_to_coo.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p]
        _LIB = L
    return _LIB


class OracleSolver:
    """AuctionSolver restatement (auction_.pyx:164-523). `val` is mutated for 'min' like the reference."""

    def __init__(self, loc, val, problem="min", max_iter=1000000, eps_start=0.0):
        assert loc.dtype == np.int32 and loc.ndim == 2 and loc.shape[1] == 2
        if val.dtype != np.float64:
            raise ValueError("Buffer dtype mismatch, expected 'float_t' but got '%s'" % val.dtype.name)
        self._loc = np.ascontiguousarray(loc)
        self._val = val if val.flags.c_contiguous else np.ascontiguousarray(val)
        self._h = lib().oracle_create(self._loc.shape[0], self._loc.ctypes.data, self._val.ctypes.data,
                                      1 if problem == "max" else 0, float(np.float32(eps_start)), int(max_iter))
        m = self.raw_meta()
        self.N, self.M = m.num_rows, m.num_cols
        self.meta = {"start_eps": round(float(m.start_eps), 3)}  # auction_.pyx:264

    def __del__(self):
        if getattr(self, "_h", None):
            lib().oracle_destroy(self._h)
            self._h = None

    def raw_meta(self):
        m = OracleMeta()
        lib().oracle_get_meta(self._h, C.byref(m))
        return m

    def set_timing(self, on=True):
        lib().oracle_set_timing(self._h, int(on))

    def set_assign_by_bidders(self, on=True):
        """The assignment phase through the round's bidders, O(#bids), instead of the reference's O(M) walk over all
        objects (auction_.pyx:394): same result, used only for bench.py's `optimised` CPU figure."""
        lib().oracle_set_assign_by_bidders(self._h, int(on))

    def step(self):
        return lib().oracle_step(self._h)

    def state(self):
        """Snapshot (K, U[:K], prices, person_to_object, object_to_person, eps, its)."""
        m = self.raw_meta()
        L = lib()
        K = m.num_unassigned
        return dict(
            K=K,
            U=np.ctypeslib.as_array(L.oracle_unassigned(self._h), (self.N,))[:K].copy(),
            p=np.ctypeslib.as_array(L.oracle_prices(self._h), (self.M,)).copy(),
            p2o=np.ctypeslib.as_array(L.oracle_person_to_object(self._h), (self.N,)).copy(),
            o2p=np.ctypeslib.as_array(L.oracle_object_to_person(self._h), (self.M,)).copy(),
            eps=float(m.final_eps), its=m.its, nreductions=m.nreductions,
        )

    def solve(self):
        lib().oracle_solve(self._h)
        m = self.raw_meta()
        # auction_.pyx:297-304
        self.meta.update(
            eCE=int(m.eCE), its=int(m.its), nreductions=int(m.nreductions), soln_found=int(m.soln_found),
            n_assigned=int(m.n_assigned), obj=round(float(m.obj_f32), 3), final_eps=round(float(m.final_eps), 3),
            timer={"solve": "%.2fms" % (1000 * m.t_total)},
        )
        self.extra = dict(obj_f64=float(m.obj_f64), edges_scanned=int(m.edges_scanned), bids_made=int(m.bids_made),
                          t_bid=float(m.t_bid), t_total=float(m.t_total), final_eps_f32=float(m.final_eps),
                          start_eps_f32=float(m.start_eps))
        return np.ctypeslib.as_array(lib().oracle_person_to_object(self._h), (self.N,)).copy()


def ece_satisfied(loc, val, problem, prices, person_to_object, eps):
    """eCE_satisfied(eps) (auction_.pyx:443-485) for a state with everybody assigned, given as arrays: the caller's
    loc / val (val in the caller's sign; the solver works on -val for 'min', :236-237), prices, person_to_object."""
    loc = np.ascontiguousarray(loc, dtype=np.int32)
    n = int(loc[:, 0].max()) + 1
    row_ptr = np.searchsorted(loc[:, 0], np.arange(n + 1)).astype(np.int32)  # rows are sorted (cumulative_idxs, :33-48)
    cols = np.ascontiguousarray(loc[:, 1])
    v = np.ascontiguousarray(val if problem == "max" else -val, dtype=np.float64)
    p = np.ascontiguousarray(prices, dtype=np.float64)
    p2o = np.ascontiguousarray(person_to_object, dtype=np.int32)
    return bool(lib().oracle_ece_arrays(n, row_ptr.ctypes.data, cols.ctypes.data, v.ctypes.data, p.ctypes.data,
                                        p2o.ctypes.data, float(np.float32(eps))))


def from_matrix(mat, problem="min", eps_start=0.0, max_iter=1000000, fast=False, cardinality_check=True):
    """auction_.pyx:528-571 (_from_matrix); cardinality_check is not restated (out of scope)."""
    mat = np.ascontiguousarray(mat, dtype=np.float64)
    N, M = mat.shape
    loc = np.empty((N * M, 2), dtype=np.int32)
    val = np.empty(N * M, dtype=np.float64)
    ctr = lib().oracle_dense_to_coo(mat.ctypes.data, N, M, loc.ctypes.data, val.ctypes.data)
    if ctr < N:
        raise ValueError(f"Matrix is infeasible - Fewer than {N} valid values provided for {N} rows.")
    if fast:
        eps_start = np.float32(1.0 / N)  # :568-569
    return OracleSolver(loc[:ctr].copy(), val[:ctr].copy(), problem=problem, eps_start=eps_start, max_iter=max_iter)


def from_sparse(loc, val, problem="min", eps_start=0.0, max_iter=1000000, fast=False, size=None,
                cardinality_check=True):
    """auction_.pyx:575-617 (_from_sparse) including its N/M quirks (:591-595)."""
    if size is not None:
        M, N = size  # sic :592
    else:
        N = int(loc[:, 0].max())  # sic :594 (no +1)
        M = int(loc[:, 1].max())
    loc32 = loc.astype(np.int32)  # :601
    if loc.shape[0] < N:
        raise ValueError(f"Matrix is infeasible - Fewer than {N} valid values provided for {N} rows.")
    if fast:
        eps_start = np.float32(1.0 / N)  # :614-615
    return OracleSolver(loc32, val, problem=problem, eps_start=eps_start, max_iter=max_iter)


def auction_solve(mat=None, loc=None, val=None, coo_mat=None, problem="min", eps_start=0.0, max_iter=1000000,
                  fast=False, size=None, cardinality_check=True):
    """sslap/auction_solve.py:6-55"""
    kw = dict(problem=problem, eps_start=eps_start, max_iter=max_iter, fast=fast, cardinality_check=cardinality_check)
    if mat is not None:
        solver = from_matrix(mat, **kw)
    elif loc is not None and val is not None:
        solver = from_sparse(loc, val, size=size, **kw)
    elif coo_mat is not None:
        loc = np.stack([coo_mat.row, coo_mat.col], axis=-1)
        solver = from_sparse(loc, coo_mat.data, size=coo_mat.shape, **kw)
    else:
        raise ValueError("One of the following formats is expected as input to auction solve: "
                         "mat OR (loc & val) OR coo_mat.")
    sol = solver.solve()
    return dict(sol=sol, meta=solver.meta, extra=solver.extra)
