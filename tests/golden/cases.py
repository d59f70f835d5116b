"""Golden-vector case definitions, shared by make_golden.py (which runs the real
reference in the build container) and by the tests (which replay the same inputs
through the oracle / the HIP path).  Inputs are either regenerated from
sslap_amd.synth (counter-based, machine independent) or, for the reference's
own np.random-seeded demo inputs, stored in the fixture itself.
"""
import numpy as np

from sslap_amd import synth


def _shuffle_within_rows(loc, val, seed):
    """Permute the stored order of entries inside every row (rows stay sorted)."""
    return synth.shuffle_within_rows(loc, val, seed)


def _single_entry_rows(n, density, seed, n_single):
    """Square instance whose first n_single rows keep ONLY their guaranteed-matching edge
    (one-entry rows: w = -inf, +inf bids, inf prices -- auction_.pyx:344,:360)."""
    loc, val = synth.gen_sparse(n, n, density, seed=seed)
    keys = synth._stream(seed, 1, n)
    perm = np.argsort(keys, kind="stable")[:n]
    keep = (loc[:, 0] >= n_single) | (loc[:, 1] == perm[loc[:, 0]])
    return np.ascontiguousarray(loc[keep]), np.ascontiguousarray(val[keep])


def _with_duplicates(n, density, seed, integer_values, adjacent=False):
    """Instance where some (i, j) entries appear twice with different values (the second copy at the end of the row;
    adjacent=True: right behind the first one, which keeps the rows column-sorted)."""
    loc, val = synth.gen_sparse(n, n, density, seed=seed, integer_values=integer_values)
    pick = np.nonzero(synth._stream(seed, 78, loc.shape[0]) % np.uint64(7) == 0)[0]
    loc2 = np.concatenate([loc, loc[pick]], axis=0)
    val2 = np.concatenate([val, val[pick] + 1.0], axis=0)
    order = np.argsort(loc2[:, 0], kind="stable")
    if adjacent:
        order = np.lexsort((np.arange(loc2.shape[0]), loc2[:, 1], loc2[:, 0]))
    return np.ascontiguousarray(loc2[order]), np.ascontiguousarray(val2[order])


def _full_double(n, density, seed):
    """Values with 53 random mantissa bits (NOT fp32-representable): the 12 B/edge kernel path."""
    loc, _ = synth.gen_sparse(n, n, density, seed=seed)
    h = synth._stream(seed, 9, loc.shape[0])
    val = (h >> np.uint64(11)).astype(np.float64) * (100.0 / float(1 << 53))
    return loc, val


def _dense(n, m, seed, integer_values, common=0):
    """Every (i, j) present: rows of m edges (far longer than the 256 edges a wavefront keeps in registers and, at
    m = 1500, than the 1024 of four passes).  Values as in synth.gen_sparse.  common = c > 0: every person likes the
    same objects -- value = a column preference (weight c / 16) + personal noise, fp32-exact -- so that few objects
    are fought over for many rounds even when there are far more objects than persons."""
    ii, jj = np.meshgrid(np.arange(n, dtype=np.int32), np.arange(m, dtype=np.int32), indexing="ij")
    loc = np.ascontiguousarray(np.stack([ii.ravel(), jj.ravel()], axis=1))
    h = synth._stream(seed, 3, n * m)
    if integer_values:
        val = (1 + (h >> np.uint64(33)) % np.uint64(integer_values)).astype(np.float64)
    else:
        val = ((h >> np.uint64(40)).astype(np.float32) * np.float32(1.0 / (1 << 24)) * np.float32(100.0)).astype(np.float64)
    if common:
        hc = synth._stream(seed, 4, m)
        if integer_values:
            pref = ((hc >> np.uint64(33)) % np.uint64(integer_values * common)).astype(np.float64)
        else:  # multiples of 2^-10 below 100 c / 16: the sum with a 24-bit noise value of < 100 stays fp32-exact? no --
            # the sum is rounded to fp32 explicitly, which keeps the 8 B/edge layout
            pref = (hc >> np.uint64(44)).astype(np.float64) * (100.0 * common / 16.0 / float(1 << 20))
        val = (val + pref[jj.ravel()]).astype(np.float32).astype(np.float64)
    return loc, val


def synth_inputs(spec):
    """spec: dict(kind=..., ...) -> (loc, val)."""
    kind = spec["kind"]
    if kind == "dense":
        return _dense(spec["n"], spec["m"], spec.get("seed", 1), spec.get("ints", 0), spec.get("common", 0))
    if kind == "sparse":
        loc, val = synth.gen_sparse(spec["n"], spec["m"], spec["density"], seed=spec.get("seed", 1),
                                    integer_values=spec.get("ints", 0))
    elif kind == "shuffled":
        loc, val = synth.gen_sparse(spec["n"], spec["m"], spec["density"], seed=spec.get("seed", 1),
                                    integer_values=spec.get("ints", 0))
        loc, val = _shuffle_within_rows(loc, val, spec.get("seed", 1))
    elif kind == "single":
        loc, val = _single_entry_rows(spec["n"], spec["density"], spec.get("seed", 1), spec["n_single"])
    elif kind == "dups":
        loc, val = _with_duplicates(spec["n"], spec["density"], spec.get("seed", 1), spec["ints"], spec.get("adjacent", False))
    elif kind == "f64":
        loc, val = _full_double(spec["n"], spec["density"], spec.get("seed", 1))
    elif kind == "config":
        loc, val = synth.gen_config(spec["name"], seed=spec.get("seed", 1))
    else:
        raise KeyError(kind)
    return loc, val


def _sq(n, d, **kw):
    return dict(kind="sparse", n=n, m=n, density=d, **kw)


# name -> (input spec, auction_solve kwargs, entry point)
# entry: "locval" (loc=, val=), "locval_size" (… size=(n,m)), "mat" (dense, -1 = invalid), "coo"
SMALL_CASES = {}
for _n in (64, 257, 1000):
    for _prob in ("max", "min"):
        SMALL_CASES[f"sq{_n}_{_prob}"] = (_sq(_n, 0.05), dict(problem=_prob), "locval")
SMALL_CASES.update({
    "sq300_fast_sizeNone": (_sq(300, 0.05), dict(problem="max", fast=True), "locval"),
    "sq300_fast_size": (_sq(300, 0.05), dict(problem="max", fast=True), "locval_size"),
    "sq300_fast_mat": (_sq(300, 0.05), dict(problem="max", fast=True), "mat"),
    "sq300_mat_min": (_sq(300, 0.05), dict(problem="min"), "mat"),
    "sq300_coo_max": (_sq(300, 0.05), dict(problem="max"), "coo"),
    "sq300_eps1": (_sq(300, 0.05), dict(problem="max", eps_start=1.0), "locval"),
    "sq300_eps_tiny": (_sq(300, 0.05), dict(problem="min", eps_start=1e-4), "locval"),
    "sq300_maxiter10": (_sq(300, 0.05), dict(problem="max", max_iter=10), "locval"),
    "sq300_maxiter1": (_sq(300, 0.05), dict(problem="max", max_iter=1), "locval"),
    "sq300_maxiter0": (_sq(300, 0.05), dict(problem="max", max_iter=0), "locval"),
    "sq200_int3": (_sq(200, 0.05, ints=3), dict(problem="max"), "locval"),
    "sq200_int10": (_sq(200, 0.05, ints=10), dict(problem="max"), "locval"),
    "sq200_int100_min": (_sq(200, 0.05, ints=100), dict(problem="min"), "locval"),
    "sq500_int5_dense": (_sq(500, 0.2, ints=5), dict(problem="max"), "locval"),
    "sq200_shuffled_int4": (dict(kind="shuffled", n=200, m=200, density=0.08, ints=4), dict(problem="max"), "locval"),
    "sq400_shuffled": (dict(kind="shuffled", n=400, m=400, density=0.05), dict(problem="min"), "locval"),
    "sq200_single10": (dict(kind="single", n=200, density=0.05, n_single=10), dict(problem="max"), "locval"),
    "sq200_single10_min": (dict(kind="single", n=200, density=0.05, n_single=10), dict(problem="min"), "locval"),
    "sq150_dups_int6": (dict(kind="dups", n=150, density=0.06, ints=6), dict(problem="max"), "locval"),
    "sq300_f64": (dict(kind="f64", n=300, density=0.05), dict(problem="max"), "locval"),
    "sq300_f64_min": (dict(kind="f64", n=300, density=0.05), dict(problem="min"), "locval"),
    "rect100x150_max": (dict(kind="sparse", n=100, m=150, density=0.1), dict(problem="max"), "locval"),
    "rect100x150_min": (dict(kind="sparse", n=100, m=150, density=0.1), dict(problem="min"), "locval_size"),
    "rect1000x1500_max": (dict(kind="sparse", n=1000, m=1500, density=0.02), dict(problem="max"), "locval"),
    "rect1000x1500_min": (dict(kind="sparse", n=1000, m=1500, density=0.02), dict(problem="min"), "locval"),
    "sq2000_max": (_sq(2000, 0.01), dict(problem="max"), "locval"),
    "sq3000_int20": (_sq(3000, 0.01, ints=20), dict(problem="max"), "locval"),
})

# per-round trace: person_to_object after r = 1..TRACE_ROUNDS rounds (max_iter = r)
TRACE_CASES = {
    "trace_sq300": (_sq(300, 0.05), dict(problem="max")),
    "trace_sq200_int4": (_sq(200, 0.08, ints=4), dict(problem="max")),
    "trace_rect100x150": (dict(kind="sparse", n=100, m=150, density=0.1), dict(problem="min")),
}
TRACE_ROUNDS = 80

# long rows: dense inputs through the `mat=` / loc-val entries; full sol + meta, and 80-round traces
LONG_CASES = {
    "dense600_max_mat": (dict(kind="dense", n=600, m=600, seed=11), dict(problem="max"), "mat"),
    "dense600_int7_max": (dict(kind="dense", n=600, m=600, seed=12, ints=7), dict(problem="max"), "locval"),
    "dense400x700_min": (dict(kind="dense", n=400, m=700, seed=13), dict(problem="min"), "locval"),
    "dense1500_min_mat": (dict(kind="dense", n=1500, m=1500, seed=14), dict(problem="min"), "mat"),
    "dense1500_int50_max": (dict(kind="dense", n=1500, m=1500, seed=15, ints=50), dict(problem="max"), "locval"),
}
LONG_TRACE_CASES = {
    "trace_dense600": (dict(kind="dense", n=600, m=600, seed=11), dict(problem="max")),
    "trace_dense600_int7": (dict(kind="dense", n=600, m=600, seed=12, ints=7), dict(problem="max")),
    "trace_dense1500": (dict(kind="dense", n=1500, m=1500, seed=14), dict(problem="min")),
}

# rows beyond the regimes above: 8 193..16 384 edges per row (the 512-thread line builder,
# k_refresh_long<E, 512>) through the `mat=` entry, and rows of more than 16 384 edges (no lines at all, tail
# threshold 40).  Full sol + meta from the real reference, plus short traces (rounds per case).
XLONG_CASES = {
    "dense9000_max_mat": (dict(kind="dense", n=9000, m=9000, seed=21), dict(problem="max"), "mat"),
    "dense40x20000_max": (dict(kind="dense", n=40, m=20000, seed=22), dict(problem="max"), "locval"),
    "dense300x17000_common_max": (dict(kind="dense", n=300, m=17000, seed=24, common=64), dict(problem="max"), "locval"),
    "dense300x17000_int3_common_max": (dict(kind="dense", n=300, m=17000, seed=25, ints=3, common=8), dict(problem="max"), "locval"),
}
XLONG_TRACE_CASES = {
    "trace_dense9000": (dict(kind="dense", n=9000, m=9000, seed=21), dict(problem="max"), 24),
    "trace_dense300x17000_int3_common": (dict(kind="dense", n=300, m=17000, seed=25, ints=3, common=8), dict(problem="max"), 80),
}

# hash-only cases: the BASELINE.json configs (max_iter = 1e8)
LARGE_CASES = {
    "C1": (dict(kind="config", name="C1"), dict(problem="max", max_iter=10**8)),
    "C1_min": (dict(kind="config", name="C1"), dict(problem="min", max_iter=10**8)),
    "C2": (dict(kind="config", name="C2"), dict(problem="max", max_iter=10**8)),
    "C3": (dict(kind="config", name="C3"), dict(problem="max", max_iter=10**8)),
    "C4": (dict(kind="config", name="C4"), dict(problem="max", max_iter=10**8)),
    "C5": (dict(kind="config", name="C5"), dict(problem="max", max_iter=10**8)),
    # not a BASELINE config: dense 8000 x 8000, the shape of the reference's `mat=` entry (bench.py --config D1)
    "D1": (dict(kind="config", name="D1"), dict(problem="max", max_iter=10**8)),
}


def dense_from(loc, val, n, m):
    """Dense matrix for the `mat=` entry: -1 marks an invalid cell (auction_.pyx:549)."""
    mat = np.full((n, m), -1.0, dtype=np.float64)
    mat[loc[:, 0], loc[:, 1]] = val
    return mat


def call_kwargs(entry, loc, val, spec):
    """Build the keyword arguments of auction_solve for an entry point."""
    n = spec.get("n") or synth.CONFIGS[spec["name"]]["n_rows"]
    m = spec.get("m", n) if "n" in spec else synth.CONFIGS[spec["name"]]["n_cols"]
    if entry == "locval":
        return dict(loc=loc, val=val)
    if entry == "locval_size":
        return dict(loc=loc, val=val, size=(n, m))
    if entry == "mat":
        return dict(mat=dense_from(loc, val, n, m))
    if entry == "coo":
        from scipy.sparse import coo_matrix
        return dict(coo_mat=coo_matrix((val, (loc[:, 0], loc[:, 1])), shape=(n, m)))
    raise KeyError(entry)


META_KEYS = ("start_eps", "eCE", "its", "nreductions", "soln_found", "n_assigned", "obj", "final_eps")


# ---- maximum-matching fixtures (the feasibility guard: reference sslap.hopcroft_solve) ------------------
def matching_graph(spec):
    """spec -> int32 loc[nnz, 2] (rows ascending).  kind 'planted': synth.gen_sparse (has a perfect matching of
    the rows); 'thinned': the same graph with every edge whose (counter-based) hash is not 0 mod `keep_mod`
    removed, rows that lose all edges keep their first one (usually NO perfect matching left); 'narrow': all
    columns folded into the first `m_eff` ones (at most m_eff rows can be matched)."""
    loc, _ = synth.gen_sparse(spec["n"], spec["m"], spec["density"], seed=spec.get("seed", 1))
    kind = spec["kind"]
    if kind == "planted":
        return loc
    if kind == "thinned":
        h = synth._stream(spec.get("seed", 1), 91, loc.shape[0])
        keep = (h % np.uint64(spec["keep_mod"])) == 0
        first = np.r_[True, loc[1:, 0] != loc[:-1, 0]]
        has = np.zeros(spec["n"], bool)
        has[loc[keep, 0]] = True
        keep |= first & ~has[loc[:, 0]]
        return np.ascontiguousarray(loc[keep])
    if kind == "narrow":
        out = loc.copy()
        out[:, 1] %= spec["m_eff"]
        key = out[:, 0].astype(np.int64) * spec["m"] + out[:, 1]
        _, idx = np.unique(key, return_index=True)
        return np.ascontiguousarray(out[np.sort(idx)])
    raise KeyError(kind)


# name -> (graph spec, entry): entry 'loc' | 'mat' (dense float64, -1 = no edge) | 'lookup' (dict i -> [j])
MATCH_CASES = {
    "planted_60": (dict(kind="planted", n=60, m=60, density=0.08, seed=3), "loc"),
    "planted_400": (dict(kind="planted", n=400, m=400, density=0.01, seed=4), "loc"),
    "planted_rect": (dict(kind="planted", n=150, m=220, density=0.03, seed=5), "loc"),
    "thinned_300": (dict(kind="thinned", n=300, m=300, density=0.02, seed=6, keep_mod=3), "loc"),
    "thinned_1000": (dict(kind="thinned", n=1000, m=1000, density=0.004, seed=7, keep_mod=2), "loc"),
    "narrow_200": (dict(kind="narrow", n=200, m=200, density=0.05, seed=8, m_eff=120), "loc"),
    "dense_mat_40": (dict(kind="thinned", n=40, m=55, density=0.2, seed=9, keep_mod=2), "mat"),
    "lookup_80": (dict(kind="thinned", n=80, m=80, density=0.06, seed=10, keep_mod=2), "lookup"),
}


def matching_call(loc, spec, entry):
    """kwargs for hopcroft_solve(loc= | mat= | lookup=) from a graph."""
    if entry == "loc":
        return dict(loc=loc)
    if entry == "mat":
        mat = np.full((spec["n"], spec["m"]), -1.0)
        mat[loc[:, 0], loc[:, 1]] = 1.0 + (loc[:, 1] % 7)
        return dict(mat=mat)
    if entry == "lookup":
        lk = {}
        for i, j in loc.tolist():
            lk.setdefault(i, []).append(j)
        return dict(lookup=lk)
    raise KeyError(entry)
