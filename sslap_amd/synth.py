"""Deterministic synthetic sparse LAP instances (SURVEY.md §8(d) recipe).

The generator is counter-based (splitmix64 finaliser over integer counters, pure
numpy uint64 arithmetic) so the *same* (loc, val) arrays are produced in the
build container (where the golden fixtures are made with the real reference) and
on the GPU box (where bench.py / the gpu tests regenerate them).  It does not
use ``np.random`` streams, whose bit streams are a numpy implementation detail.

Instance recipe, restating ``BASELINE.json:configs`` / SURVEY.md §8(d):
  * ``k = max(1, round(density * M))`` column draws per row (with replacement,
    de-duplicated afterwards),
  * plus one edge ``(i, pi(i))`` of a random injection persons -> objects, which
    guarantees a perfect matching of persons (``cardinality_check`` can stay off),
  * columns ascending within a row, rows ascending: the input contract of the
    reference's ``cumulative_idxs`` (reference ``sslap/auction_.pyx:33-48``),
  * values uniform on [0, 100) with 24 random bits, rounded to fp32 and carried
    as float64 (the reference only accepts float64, ``auction_.pyx:202``).
"""
import hashlib

import numpy as np

_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)
_GOLD = np.uint64(0x9E3779B97F4A7C15)


def _mix_inplace(x, tmp):
    """splitmix64 finaliser, in place on a uint64 array (tmp: scratch of the same shape)."""
    np.right_shift(x, np.uint64(30), out=tmp); x ^= tmp
    x *= _M1
    np.right_shift(x, np.uint64(27), out=tmp); x ^= tmp
    x *= _M2
    np.right_shift(x, np.uint64(31), out=tmp); x ^= tmp
    return x


_CHUNK = 1 << 20


def _stream(seed, stream, n, offset=0, out=None, post=None):
    """n pseudo-random uint64: hash of (seed, stream id, counter).

    Works in cache-sized chunks with reused scratch (fresh 100-MB temporaries cost
    seconds of page faults).  ``post(chunk_u64) -> array`` optionally maps every
    chunk before it is stored into ``out`` (which then has post's dtype).
    """
    with np.errstate(over="ignore"):
        b = np.array([np.uint64(seed) * _GOLD + np.uint64(stream)], dtype=np.uint64)
        base = _mix_inplace(b, np.empty_like(b))[0]
        if out is None:
            out = np.empty(n, dtype=np.uint64)
        x = np.empty(min(n, _CHUNK), dtype=np.uint64)
        tmp = np.empty_like(x)
        for lo in range(0, n, _CHUNK):
            m = min(_CHUNK, n - lo)
            xs, ts = x[:m], tmp[:m]
            xs[:] = np.arange(offset + lo, offset + lo + m, dtype=np.uint64)
            xs *= _GOLD
            xs += base
            _mix_inplace(xs, ts)
            out[lo:lo + m] = xs if post is None else post(xs)
        return out


def gen_sparse(n_rows, n_cols, density, seed=1, integer_values=0):
    """Return (loc int32[nnz,2], val float64[nnz]) for an n_rows x n_cols instance.

    integer_values > 0 draws values from {1..integer_values} instead (heavy ties,
    exercising the in-row ">= keeps the last maximum" rule of auction_.pyx:351 and
    the "first bidder wins" rule of :379).
    """
    assert n_rows <= n_cols, "persons must not outnumber objects"
    N, M = int(n_rows), int(n_cols)
    k = max(1, int(round(density * M)))
    # random injection: first N entries of a hash-ordered permutation of objects
    keys = _stream(seed, 1, M)
    perm = np.argsort(keys, kind="stable")[:N].astype(np.int32)
    # k draws per row
    cols = np.empty((N, k + 1), dtype=np.int32)
    flat = np.empty(N * k, dtype=np.int32)
    _stream(seed, 2, N * k, out=flat, post=lambda h: h % np.uint64(M))
    cols[:, :k] = flat.reshape(N, k)
    del flat
    cols[:, k] = perm
    cols.sort(axis=1)
    keep = np.ones(cols.shape, dtype=bool)
    keep[:, 1:] = cols[:, 1:] != cols[:, :-1]
    counts = keep.sum(axis=1)
    nnz = int(counts.sum())
    loc = np.empty((nnz, 2), dtype=np.int32)
    loc[:, 0] = np.repeat(np.arange(N, dtype=np.int32), counts)
    loc[:, 1] = cols[keep]
    val = np.empty(nnz, dtype=np.float64)
    if integer_values:
        _stream(seed, 3, nnz, out=val,
                post=lambda h: 1 + (h >> np.uint64(33)) % np.uint64(integer_values))
    else:
        def to_val(h):
            u = (h >> np.uint64(40)).astype(np.float32) * np.float32(1.0 / (1 << 24))
            return u * np.float32(100.0)          # fp32 product, widened exactly on store
        _stream(seed, 3, nnz, out=val, post=to_val)
    return loc, val


def input_digest(loc, val):
    """sha256 over the raw bytes of (loc, val): pins the generator across machines."""
    h = hashlib.sha256()
    h.update(np.ascontiguousarray(loc, dtype=np.int32).tobytes())
    h.update(np.ascontiguousarray(val, dtype=np.float64).tobytes())
    return h.hexdigest()


def sol_digest(sol):
    return hashlib.sha256(np.ascontiguousarray(sol, dtype=np.int32).tobytes()).hexdigest()


# BASELINE.json configs (C1..C5)
CONFIGS = {
    "C1": dict(n_rows=5_000, n_cols=5_000, density=0.02),
    "C2": dict(n_rows=50_000, n_cols=50_000, density=0.005),
    "C3": dict(n_rows=200_000, n_cols=200_000, density=0.001),
    "C4": dict(n_rows=100_000, n_cols=150_000, density=0.002),
    "C5": dict(n_rows=1_000_000, n_cols=1_000_000, density=0.0001),
    # not a BASELINE config: the shape of the reference's `mat=` entry (every row holds every column)
    "D1": dict(n_rows=8_000, n_cols=8_000, density=1.0, dense=True),
}


def gen_dense(n_rows, n_cols, seed=1):
    """Every (i, j) present -- what the reference's `mat=` entry produces for a matrix without negative entries
    (auction_.pyx:546-557) -- with the same fp32-exact value stream as gen_sparse."""
    N, M = int(n_rows), int(n_cols)
    loc = np.empty((N * M, 2), dtype=np.int32)
    loc[:, 0] = np.repeat(np.arange(N, dtype=np.int32), M)
    loc[:, 1] = np.tile(np.arange(M, dtype=np.int32), N)
    val = np.empty(N * M, dtype=np.float64)

    def to_val(h):
        u = (h >> np.uint64(40)).astype(np.float32) * np.float32(1.0 / (1 << 24))
        return u * np.float32(100.0)
    _stream(seed, 3, N * M, out=val, post=to_val)
    return loc, val


def gen_config(name, seed=1):
    c = CONFIGS[name]
    if c.get("dense"):
        return gen_dense(c["n_rows"], c["n_cols"], seed=seed)
    return gen_sparse(c["n_rows"], c["n_cols"], c["density"], seed=seed)


def shuffle_within_rows(loc, val, seed):
    """Permute the stored order of the entries inside every row (rows stay ascending): legal input for the reference
    (cumulative_idxs, auction_.pyx:33-48, only needs the rows sorted), and its in-row tie rule follows the stored order
    (:351)."""
    n = loc.shape[0]
    key = _stream(seed, 77, n)
    order = np.lexsort((key, loc[:, 0]))
    return np.ascontiguousarray(loc[order]), np.ascontiguousarray(val[order])
