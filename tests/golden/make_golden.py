#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by running the REAL reference.

Runs only in the build container (needs /root/reference and Cython):
  1. copies /root/reference to a scratch dir under /tmp and builds it there,
     unmodified (`python3 setup.py build_ext --inplace`, Cython 3.x => true
     division for `1/N`);
  2. imports it with the numpy aliases the 2021-era source expects
     (np.float / np.int were removed in numpy 1.24);
  3. runs every case of cases.py through `sslap.auction_solve` and stores the
     outputs (full arrays for small cases, sha256 + meta for the BASELINE
     configs), and
  4. replays the same inputs through the C oracle (oracle/) and REFUSES to write
     a fixture if the oracle differs in any bit -- this is what pins the oracle.

Nothing of the reference (source, bytecode, binaries) is written into the repo;
only inputs/outputs are.  Usage:
    python tests/golden/make_golden.py [small] [trace] [demo] [long] [xlong] [large] [D1] [C5] [matching]
"""
import json
import os
import shutil
import subprocess
import sys
import time
import warnings

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

import cases  # noqa: E402
from oracle import oracle as orc  # noqa: E402
from sslap_amd import synth  # noqa: E402

SCRATCH = "/tmp/sslap_ref_build"


def load_reference():
    ref = "/root/reference"
    if not os.path.isdir(ref):
        raise SystemExit("make_golden.py needs the reference checkout at /root/reference")
    marker = os.path.join(SCRATCH, "sslap")
    if not any(f.endswith(".so") for f in (os.listdir(marker) if os.path.isdir(marker) else [])):
        shutil.rmtree(SCRATCH, ignore_errors=True)
        shutil.copytree(ref, SCRATCH)
        subprocess.check_call("chmod -R u+w . && python3 setup.py build_ext --inplace > build.log 2>&1",
                              shell=True, cwd=SCRATCH)
    np.float = np.float64  # aliases expected by auction_.pyx:18,:25
    np.int = np.int_
    warnings.filterwarnings("ignore")
    sys.path.insert(0, SCRATCH)
    import Cython
    import sslap
    return sslap.auction_solve, dict(cython=Cython.__version__, numpy=np.__version__, sslap=sslap.__version__)


def meta_of(res):
    return {k: res["meta"][k] for k in cases.META_KEYS}


def check_same(name, ref, orc_res):
    if not np.array_equal(ref["sol"], orc_res["sol"]):
        raise SystemExit(f"[{name}] ORACLE MISMATCH in sol")
    for k in cases.META_KEYS:
        if ref["meta"][k] != orc_res["meta"][k]:
            raise SystemExit(f"[{name}] ORACLE MISMATCH in meta[{k}]: ref {ref['meta'][k]} oracle {orc_res['meta'][k]}")


def run_both(ref_solve, name, loc, val, kw, entry="locval", spec=None):
    """Run reference and oracle on private copies (the reference mutates val for 'min')."""
    a = cases.call_kwargs(entry, loc.copy(), val.copy(), spec)
    b = cases.call_kwargs(entry, loc.copy(), val.copy(), spec)
    t = time.time()
    ref = ref_solve(cardinality_check=False, **a, **kw)
    t_ref = time.time() - t
    t = time.time()
    o = orc.auction_solve(cardinality_check=False, **b, **kw)
    t_orc = time.time() - t
    check_same(name, ref, o)
    if "val" in a and not np.array_equal(a["val"], b["val"]):
        raise SystemExit(f"[{name}] in-place val mutation differs")
    mutated = bool("val" in a and not np.array_equal(a["val"], val))
    return ref, o, mutated, t_ref, t_orc


def do_small(ref_solve, versions):
    out = {}
    manifest = {"versions": versions, "cases": {}}
    for name, (spec, kw, entry) in cases.SMALL_CASES.items():
        loc, val = cases.synth_inputs(spec)
        ref, o, mutated, t_ref, t_orc = run_both(ref_solve, name, loc, val, kw, entry, spec)
        out[name + "/sol"] = ref["sol"].astype(np.int32)
        manifest["cases"][name] = dict(meta=meta_of(ref), input_sha256=synth.input_digest(loc, val),
                                       val_mutated=mutated, obj_f64=o["extra"]["obj_f64"],
                                       edges_scanned=o["extra"]["edges_scanned"])
        print(f"small {name}: its={ref['meta']['its']} nred={ref['meta']['nreductions']} "
              f"n_assigned={ref['meta']['n_assigned']} ok")
    np.savez_compressed(os.path.join(HERE, "small_cases.npz"), **out)
    json.dump(manifest, open(os.path.join(HERE, "small_cases.json"), "w"), indent=1, sort_keys=True)


def do_trace(ref_solve, versions):
    out = {}
    manifest = {"versions": versions, "rounds": cases.TRACE_ROUNDS, "cases": {}}
    for name, (spec, kw) in cases.TRACE_CASES.items():
        loc, val = cases.synth_inputs(spec)
        sols, its = [], []
        for r in range(1, cases.TRACE_ROUNDS + 1):
            ref, o, _, _, _ = run_both(ref_solve, f"{name}@{r}", loc, val, dict(kw, max_iter=r), "locval", spec)
            sols.append(ref["sol"].astype(np.int32))
            its.append(ref["meta"]["its"])
        out[name + "/p2o"] = np.stack(sols)
        manifest["cases"][name] = dict(its=its, input_sha256=synth.input_digest(loc, val))
        print(f"trace {name}: {len(sols)} rounds ok")
    np.savez_compressed(os.path.join(HERE, "trace_cases.npz"), **out)
    json.dump(manifest, open(os.path.join(HERE, "trace_cases.json"), "w"), indent=1, sort_keys=True)


def do_long(ref_solve, versions):
    """Long rows (> 256 and > 1024 edges): full solutions + meta, and 80-round traces, from the real reference."""
    out = {}
    manifest = {"versions": versions, "rounds": cases.TRACE_ROUNDS, "cases": {}, "traces": {}}
    for name, (spec, kw, entry) in cases.LONG_CASES.items():
        loc, val = cases.synth_inputs(spec)
        ref, o, mutated, t_ref, t_orc = run_both(ref_solve, name, loc, val, kw, entry, spec)
        out[name + "/sol"] = ref["sol"].astype(np.int32)
        manifest["cases"][name] = dict(meta=meta_of(ref), input_sha256=synth.input_digest(loc, val),
                                       val_mutated=mutated, obj_f64=o["extra"]["obj_f64"],
                                       edges_scanned=o["extra"]["edges_scanned"])
        print(f"long {name}: its={ref['meta']['its']} nred={ref['meta']['nreductions']} ok ({t_ref:.1f}s ref)")
    for name, (spec, kw) in cases.LONG_TRACE_CASES.items():
        loc, val = cases.synth_inputs(spec)
        sols, its = [], []
        for r in range(1, cases.TRACE_ROUNDS + 1):
            ref, o, _, _, _ = run_both(ref_solve, f"{name}@{r}", loc, val, dict(kw, max_iter=r), "locval", spec)
            sols.append(ref["sol"].astype(np.int32))
            its.append(ref["meta"]["its"])
        out[name + "/p2o"] = np.stack(sols)
        manifest["traces"][name] = dict(its=its, input_sha256=synth.input_digest(loc, val))
        print(f"long trace {name}: {len(sols)} rounds ok")
    np.savez_compressed(os.path.join(HERE, "long_cases.npz"), **out)
    json.dump(manifest, open(os.path.join(HERE, "long_cases.json"), "w"), indent=1, sort_keys=True)


def do_xlong(ref_solve, versions):
    """Rows of 9 000 and 20 000 edges: full solutions + meta and short traces."""
    out = {}
    manifest = {"versions": versions, "cases": {}, "traces": {}}
    for name, (spec, kw, entry) in cases.XLONG_CASES.items():
        loc, val = cases.synth_inputs(spec)
        ref, o, mutated, t_ref, t_orc = run_both(ref_solve, name, loc, val, kw, entry, spec)
        out[name + "/sol"] = ref["sol"].astype(np.int32)
        manifest["cases"][name] = dict(meta=meta_of(ref), input_sha256=synth.input_digest(loc, val),
                                       val_mutated=mutated, obj_f64=o["extra"]["obj_f64"],
                                       edges_scanned=o["extra"]["edges_scanned"],
                                       reference_wall_s=round(t_ref, 2), oracle_wall_s=round(t_orc, 2))
        print(f"xlong {name}: its={ref['meta']['its']} nred={ref['meta']['nreductions']} ok "
              f"({t_ref:.1f}s ref, {t_orc:.1f}s oracle)", flush=True)
    for name, (spec, kw, rounds) in cases.XLONG_TRACE_CASES.items():
        loc, val = cases.synth_inputs(spec)
        sols, its = [], []
        for r in range(1, rounds + 1):
            ref, o, _, _, _ = run_both(ref_solve, f"{name}@{r}", loc, val, dict(kw, max_iter=r), "locval", spec)
            sols.append(ref["sol"].astype(np.int32))
            its.append(ref["meta"]["its"])
        out[name + "/p2o"] = np.stack(sols)
        manifest["traces"][name] = dict(its=its, rounds=rounds, input_sha256=synth.input_digest(loc, val))
        print(f"xlong trace {name}: {len(sols)} rounds ok", flush=True)
    np.savez_compressed(os.path.join(HERE, "xlong_cases.npz"), **out)
    json.dump(manifest, open(os.path.join(HERE, "xlong_cases.json"), "w"), indent=1, sort_keys=True)


def do_demo(ref_solve, versions):
    """The reference's own seeded demo inputs (examples/test_auction.py:7-46) and a small instance
    of its benchmark recipe (benchmarking.py:17-45).  np.random streams are not portable, so the
    INPUT matrices are stored next to the outputs."""
    from scipy.sparse import coo_matrix
    out = {}
    manifest = {"versions": versions, "cases": {}}

    def record(name, call, inputs):
        ref = ref_solve(**call(), cardinality_check=False)
        o = orc.auction_solve(**call(), cardinality_check=False)
        check_same(name, ref, o)
        for k, v in inputs.items():
            out[f"{name}/{k}"] = v
        out[f"{name}/sol"] = ref["sol"].astype(np.int32)
        manifest["cases"][name] = dict(meta=meta_of(ref))
        print(f"demo {name}: sol={ref['sol'][:8]} meta={meta_of(ref)}")

    np.random.seed(1)
    mat = np.random.uniform(0, 10, (5, 5)).astype(np.float64)
    record("demo_dense_min", lambda: dict(mat=mat.copy(), problem="min"), dict(mat=mat))
    np.random.seed(2)
    mask = np.random.rand(5, 5) > 0.5
    m2 = mat.copy()
    m2[mask] = -1
    record("demo_sparse_max", lambda: dict(mat=m2.copy(), problem="max"), dict(mat=m2))
    m3 = mat.copy()
    m3[mask] = 0
    record("demo_coo_max", lambda: dict(coo_mat=coo_matrix(m3), problem="max"), dict(mat=m3))

    # benchmarking.py:29-45 recipe at a small size
    def improve(maskb):
        cnt = (~maskb).sum(axis=1)
        R, Cc = maskb.shape
        for r in np.nonzero(cnt == 0)[0]:
            maskb[r, np.random.randint(Cc)] = False
        cnt = (~maskb).sum(axis=0)
        for c in np.nonzero(cnt == 0)[0]:
            maskb[np.random.randint(R), c] = False
        return maskb
    for size, density, mode in ((120, 0.2, "float"), (150, 0.1, "int"), (60, 1.0, "float")):
        np.random.seed(1)
        if mode == "int":
            bm = np.random.randint(1, 100, (size, size)).astype(np.float64)
        else:
            bm = np.random.uniform(0., 100., size=(size, size)).astype(np.float64)
        np.random.seed(2)
        mk = improve(np.random.random(bm.shape) > density)
        bm[mk] = -1
        # the recipe does not guarantee feasibility; only keep feasible draws
        from scipy.sparse.csgraph import maximum_bipartite_matching
        from scipy.sparse import csr_matrix
        match = maximum_bipartite_matching(csr_matrix(bm >= 0), perm_type="column")
        if (match < 0).any():
            print(f"demo bench_{size}_{density}_{mode}: infeasible draw, skipped")
            continue
        record(f"bench_{size}_{mode}", lambda bm=bm: dict(mat=bm.copy(), problem="max"), dict(mat=bm))
    np.savez_compressed(os.path.join(HERE, "demo_cases.npz"), **out)
    json.dump(manifest, open(os.path.join(HERE, "demo_cases.json"), "w"), indent=1, sort_keys=True)


def do_large(ref_solve, versions, names):
    path = os.path.join(HERE, "large_cases.json")
    manifest = json.load(open(path)) if os.path.exists(path) else {"versions": versions, "cases": {}}
    for name in names:
        spec, kw = cases.LARGE_CASES[name]
        loc, val = cases.synth_inputs(spec)
        ref, o, _, t_ref, t_orc = run_both(ref_solve, name, loc, val, kw, "locval", spec)
        manifest["cases"][name] = dict(
            meta=meta_of(ref), sol_sha256=synth.sol_digest(ref["sol"]), input_sha256=synth.input_digest(loc, val),
            nnz=int(loc.shape[0]), obj_f64=o["extra"]["obj_f64"], edges_scanned=o["extra"]["edges_scanned"],
            bids_made=o["extra"]["bids_made"], final_eps_f32=o["extra"]["final_eps_f32"],
            reference_solve_timer=ref["meta"]["timer"]["solve"], reference_wall_s=round(t_ref, 2),
            oracle_wall_s=round(t_orc, 2),
            hardware="build container, 1 thread of an 8-vCPU Xeon @ 2.10 GHz")
        json.dump(manifest, open(path, "w"), indent=1, sort_keys=True)
        print(f"large {name}: {manifest['cases'][name]}")


def do_matching(versions):
    """sslap.hopcroft_solve (feasibility_.pyx:227-283) on the graphs of cases.MATCH_CASES: cardinality and
    both pairing arrays."""
    import sslap
    out = {}
    manifest = {"versions": versions, "cases": {}}
    for name, (spec, entry) in cases.MATCH_CASES.items():
        loc = cases.matching_graph(spec)
        res = sslap.hopcroft_solve(**cases.matching_call(loc.astype(np.int32), spec, entry))
        out[name + "/left"] = np.asarray(res["left_pairings"], dtype=np.int32)
        out[name + "/right"] = np.asarray(res["right_pairings"], dtype=np.int32)
        manifest["cases"][name] = dict(size=int(res["size"]), nnz=int(loc.shape[0]),
                                       n_rows=int(len(res["left_pairings"])), n_cols=int(len(res["right_pairings"])))
        print(f"matching {name}: size={res['size']} of {len(res['left_pairings'])} rows, nnz={loc.shape[0]}")
    np.savez_compressed(os.path.join(HERE, "matching_cases.npz"), **out)
    json.dump(manifest, open(os.path.join(HERE, "matching_cases.json"), "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    what = sys.argv[1:] or ["small", "trace", "demo", "long", "large", "matching"]
    ref_solve, versions = load_reference()
    if "small" in what:
        do_small(ref_solve, versions)
    if "trace" in what:
        do_trace(ref_solve, versions)
    if "demo" in what:
        do_demo(ref_solve, versions)
    if "long" in what:
        do_long(ref_solve, versions)
    if "large" in what:
        do_large(ref_solve, versions, ["C1", "C1_min", "C2", "C4", "C3"])
    if "xlong" in what:
        do_xlong(ref_solve, versions)
    if "D1" in what:
        do_large(ref_solve, versions, ["D1"])
    if "C5" in what:
        do_large(ref_solve, versions, ["C5"])
    if "matching" in what:
        do_matching(versions)
