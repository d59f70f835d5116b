"""The feasibility guard: native Hopcroft-Karp in libmisslap.so against golden vectors captured
from the real reference's sslap.hopcroft_solve (cardinality AND both pairing arrays), against scipy on random
graphs, and the front-end's ValueError texts.  Host code only: runs without a GPU."""
import numpy as np
import pytest

import cases
from sslap_amd import from_sparse, hopcroft_solve, synth
from sslap_amd.check_feasible import cardinality


@pytest.mark.parametrize("name", sorted(cases.MATCH_CASES))
def test_matching_matches_reference(name, golden_matching, built_lib):
    man, arr = golden_matching
    spec, entry = cases.MATCH_CASES[name]
    loc = cases.matching_graph(spec)
    res = hopcroft_solve(**cases.matching_call(loc.astype(np.int32), spec, entry))
    assert res["size"] == man["cases"][name]["size"]
    assert res["left_pairings"].dtype == np.int32
    assert np.array_equal(res["left_pairings"], arr[name + "/left"])
    assert np.array_equal(res["right_pairings"], arr[name + "/right"])


def _valid_matching(loc, res):
    left, right = res["left_pairings"], res["right_pairings"]
    edges = set(map(tuple, loc.tolist()))
    m = [(i, int(j)) for i, j in enumerate(left) if j >= 0]
    assert len(m) == res["size"] == int((right >= 0).sum())
    assert all(e in edges for e in m)
    assert all(right[j] == i for i, j in m)
    assert len({j for _, j in m}) == len(m)


@pytest.mark.parametrize("seed", range(6))
def test_cardinality_equals_scipy_on_random_graphs(seed, built_lib):
    from scipy.sparse import csr_matrix
    from scipy.sparse.csgraph import maximum_bipartite_matching
    spec = dict(kind="thinned", n=450 + 37 * seed, m=500 + 50 * seed, density=0.006, seed=20 + seed, keep_mod=2)
    loc = cases.matching_graph(spec)
    n, m = spec["n"], spec["m"]
    res = hopcroft_solve(loc=loc)
    g = csr_matrix((np.ones(loc.shape[0], np.int8), (loc[:, 0], loc[:, 1])), shape=(n, m))
    assert res["size"] == int((maximum_bipartite_matching(g, perm_type="column") >= 0).sum())
    _valid_matching(loc, res)


def test_large_graph_needs_no_recursion_and_no_n_squared_queue(built_lib):
    """200 000 rows: the reference allocates an N^2-int queue (160 GB) and recurses one frame per path vertex."""
    loc, _ = synth.gen_sparse(200_000, 200_000, 0.00003, seed=5)  # ~7 edges per row, perfect matching planted
    assert cardinality(loc, 200_000, 200_000) == 200_000
    # a path graph forces ONE augmenting path through every vertex (depth 100 000)
    n = 100_000
    i = np.repeat(np.arange(n, dtype=np.int32), 2)
    j = np.stack([np.arange(n, dtype=np.int32), np.arange(1, n + 1, dtype=np.int32)], axis=1).reshape(-1)
    path = np.stack([i, j], axis=1)[:-1]  # row k -> columns k, k+1; the last row only k
    res = hopcroft_solve(loc=path)
    assert res["size"] == n


def test_rows_must_be_sorted_and_in_range(built_lib):
    loc = np.array([[1, 0], [0, 1]], dtype=np.int32)
    with pytest.raises(ValueError, match="ascending"):
        hopcroft_solve(loc=loc)
    with pytest.raises(AssertionError):
        hopcroft_solve()
    with pytest.raises(AssertionError):
        hopcroft_solve(loc=loc, mat=np.zeros((2, 2)))


def test_front_end_rejects_infeasible_input_with_the_reference_text(built_lib):
    """auction_.pyx:608-612: raised before any solver (or GPU) is touched."""
    spec = dict(kind="narrow", n=50, m=50, density=0.1, seed=3, m_eff=30)
    loc = cases.matching_graph(spec)
    val = np.ones(loc.shape[0])
    n_true = int(loc[:, 0].max()) + 1
    card = cardinality(loc, n_true, int(loc[:, 1].max()) + 1)
    assert card == 30
    with pytest.raises(ValueError, match=rf"Maximum matching possible only involves {card} out of {n_true} rows"):
        from_sparse(loc, val, problem="max", cardinality_check=True)


# ---- the GPU matcher (misslap_matching_gpu, csrc/kernels_matching.hpp): same cardinality, any maximum matching ------
@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(cases.MATCH_CASES))
def test_gpu_matching_cardinality_matches_reference(name, golden_matching, gpu_lib):
    from sslap_amd.check_feasible import matching_gpu
    man, _ = golden_matching
    spec, entry = cases.MATCH_CASES[name]
    loc = cases.matching_graph(spec)
    g = man["cases"][name]
    res = matching_gpu(loc, g["n_rows"], g["n_cols"])
    assert res["size"] == g["size"]  # the reference's cardinality (captured from sslap.hopcroft_solve)
    _valid_matching(loc, res)


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(6))
def test_gpu_matching_equals_scipy_on_random_graphs(seed, gpu_lib):
    from scipy.sparse import csr_matrix
    from scipy.sparse.csgraph import maximum_bipartite_matching
    from sslap_amd.check_feasible import matching_gpu
    spec = dict(kind="thinned", n=4500 + 370 * seed, m=5000 + 500 * seed, density=0.0006, seed=20 + seed, keep_mod=2)
    loc = cases.matching_graph(spec)
    n, m = spec["n"], spec["m"]
    res = matching_gpu(loc, n, m)
    g = csr_matrix((np.ones(loc.shape[0], np.int8), (loc[:, 0], loc[:, 1])), shape=(n, m))
    assert res["size"] == int((maximum_bipartite_matching(g, perm_type="column") >= 0).sum())
    assert res["size"] == cardinality(loc, n, m)  # ... and the host Hopcroft-Karp
    _valid_matching(loc, res)


@pytest.mark.gpu
def test_gpu_matching_at_200k_rows(gpu_lib):
    """200 000 rows against scipy -- a planted perfect matching, an infeasible thinned graph
    (many augmentation phases), a graph with empty rows, and a path graph (one long augmenting chain)."""
    from scipy.sparse import csr_matrix
    from scipy.sparse.csgraph import maximum_bipartite_matching
    from sslap_amd.check_feasible import matching_gpu
    n = 200_000
    loc, _ = synth.gen_sparse(n, n, 0.00003, seed=5)  # ~7 edges per row, perfect matching planted
    res = matching_gpu(loc, n, n)
    assert res["size"] == n and sorted(res["left_pairings"].tolist()) == list(range(n))
    thin = cases.matching_graph(dict(kind="thinned", n=n, m=n, density=0.00002, seed=9, keep_mod=2))
    thin = thin[thin[:, 0] % 17 != 3]  # some rows lose all their edges
    res = matching_gpu(thin, n, n)
    g = csr_matrix((np.ones(thin.shape[0], np.int8), (thin[:, 0], thin[:, 1])), shape=(n, n))
    want = int((maximum_bipartite_matching(g, perm_type="column") >= 0).sum())
    assert res["size"] == want and want < n
    left, right = res["left_pairings"], res["right_pairings"]
    m_rows = np.nonzero(left >= 0)[0]
    assert len(m_rows) == want and len(np.unique(left[m_rows])) == want and np.array_equal(right[left[m_rows]], m_rows)
    key = set((thin[:, 0].astype(np.int64) * n + thin[:, 1]).tolist())
    assert all(int(i) * n + int(left[i]) in key for i in m_rows[:: max(1, want // 5000)])
    # path graph: row k -> columns k, k + 1 (the last row only k), every row shifted so that greedy gets it wrong
    k = 20_000
    i = np.repeat(np.arange(k, dtype=np.int32), 2)
    j = np.stack([np.arange(1, k + 1, dtype=np.int32), np.arange(k, dtype=np.int32)], axis=1).reshape(-1)
    path = np.stack([i, j], axis=1)
    path = path[~((path[:, 0] == k - 1) & (path[:, 1] == k))]  # the last row keeps only column k - 1
    res = matching_gpu(path, k, k + 1)
    assert res["size"] == k


@pytest.mark.gpu
def test_front_end_uses_gpu_matcher_for_large_graphs(gpu_lib, monkeypatch):
    monkeypatch.setenv("MISSLAP_MATCHING_GPU_MIN_NNZ", "1")
    spec = dict(kind="narrow", n=50, m=50, density=0.1, seed=3, m_eff=30)
    loc = cases.matching_graph(spec)
    with pytest.raises(ValueError, match=r"Maximum matching possible only involves 30 out of 50 rows"):
        from_sparse(loc, np.ones(loc.shape[0]), problem="max", cardinality_check=True)
    loc, val = synth.gen_sparse(300, 300, 0.05, seed=2)
    from_sparse(loc, val, problem="max", cardinality_check=True).solve()


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(cases.MATCH_CASES))
def test_matching_on_a_solver_handle_matches_reference(name, golden_matching, gpu_lib):
    """misslap_matching_of: the matcher on the CSR a solver handle already holds in device memory (both edge layouts)
    gives the reference's cardinality -- the guard of the front-end without a host copy of the entries."""
    from sslap_amd.auction_solve import AuctionSolver
    manifest, arrays = golden_matching
    spec, _entry = cases.MATCH_CASES[name]
    loc = cases.matching_graph(spec)
    rows = np.unique(loc[:, 0])
    if rows.shape[0] != int(loc[:, 0].max()) + 1:
        pytest.skip("a row without entries: no solver handle exists for such input")
    want = manifest["cases"][name]["size"]
    for f64 in (False, True):
        s = AuctionSolver(loc.astype(np.int32), np.ones(loc.shape[0]), problem="max", force_f64=f64)
        assert s.matching_cardinality() == want, (name, f64)


@pytest.mark.gpu
def test_dense_front_end_guard_runs_on_the_handle(gpu_lib):
    """from_matrix(cardinality_check=True): feasible matrices pass, a matrix whose rows share too few valid columns
    raises the reference's text (auction_.pyx:566) -- the matching runs on the handle's device-resident CSR."""
    from sslap_amd import from_matrix
    r = np.random.default_rng(3)
    mat = r.random((300, 300)) * 10
    from_matrix(mat, problem="max", cardinality_check=True).solve()
    bad = np.full((60, 80), -1.0)
    bad[:, :25] = r.random((60, 25))  # 60 rows compete for 25 columns
    with pytest.raises(ValueError, match=r"Maximum matching possible only involves 25 out of 60 rows"):
        from_matrix(bad, problem="max", cardinality_check=True)


@pytest.mark.gpu
def test_gpu_matcher_falls_back_to_the_host_when_its_layer_budget_runs_out(gpu_lib, monkeypatch):
    """the GPU matcher augments one path per BFS tree and phase, so chain- / ladder-like graphs can need
    O(n) layers; the layers are budgeted and the host matcher finishes from the matching found so far.  A ladder
    (row k -> columns k, k + 1) whose greedy start leaves one long augmenting path, with the budget forced low."""
    from sslap_amd.check_feasible import matching_gpu
    n = 3000
    i = np.repeat(np.arange(n, dtype=np.int32), 2)
    j = np.stack([np.arange(1, n + 1, dtype=np.int32), np.arange(n, dtype=np.int32)], axis=1).reshape(-1)
    j[-2] = n - 1  # the last row only reaches column n - 1 (twice): one augmenting path through every vertex
    ladder = np.ascontiguousarray(np.stack([i, j], axis=1))
    for budget in ("4", None):
        if budget is None:
            monkeypatch.delenv("MISSLAP_MATCHING_MAX_LAYERS", raising=False)
        else:
            monkeypatch.setenv("MISSLAP_MATCHING_MAX_LAYERS", budget)
        res = matching_gpu(ladder, n, n + 1)
        assert res["size"] == cardinality(ladder, n, n + 1)
        _valid_matching(ladder, res)
    # the same through a solver handle's device-resident CSR (misslap_matching_of)
    monkeypatch.setenv("MISSLAP_MATCHING_MAX_LAYERS", "3")
    spec = dict(kind="thinned", n=900, m=900, density=0.01, seed=31, keep_mod=2)
    loc = cases.matching_graph(spec)
    from sslap_amd import AuctionSolver
    s = AuctionSolver(loc, np.ones(loc.shape[0]), problem="max")
    assert s.matching_cardinality() == cardinality(loc, 900, 900)
