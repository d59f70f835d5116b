"""The committed bench line (profiles/r01_bench.json, written by `python bench.py` on an MI355X) carries every
field of the driver's contract, the roofline / cpu_baseline objects, and numbers that are consistent with each
other.  Runs without a GPU: it checks the artefact, not the measurement."""
import json
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _line(name):
    with open(os.path.join(ROOT, "profiles", name)) as f:
        return json.loads(f.read().strip().splitlines()[-1])


def _latest_c3():
    """The newest committed bench line of the headline config (profiles/rNN_bench_C3.json)."""
    import glob
    names = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_bench_C3.json")))
    assert names, "no committed bench line"
    return os.path.basename(names[-1])


def test_bench_line_has_the_contract_fields():
    d = _line("r01_bench.json")
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["unit"] == "Medges/s" and d["higher_is_better"] is True and d["n_gpus"] == 1
    assert d["vs_baseline"] is None and d["data"] == "synthetic" and d["dtype"] == "f64"
    assert "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-4
    # algorithmic bytes per launch / average launch duration == achieved
    assert abs(r["algorithmic_bytes_per_launch"] / (r["avg_launch_us"] * 1e-6) / 1e9 - r["achieved"]) < 0.01 * r["achieved"]
    assert r["traffic"] is None or r["traffic"] >= 0.5 * r["algorithmic_bytes_per_launch"]
    c = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    assert c["kind"] in ("port", "reference") and c["cores"] >= 1
    # value = edges of the timed steps / wall time
    assert abs(d["value"] - d["edges_scanned_per_solve"] * d["steps"] / (d["ms_per_step"] * d["steps"] * 1e-3) / 1e6) \
        < 0.01 * d["value"]


def test_bench_line_matches_the_reference_fixture():
    d = _line("r01_bench.json")
    g = json.load(open(os.path.join(ROOT, "tests", "golden", "large_cases.json")))["cases"]["C3"]
    assert d["sol_sha256"] == g["sol_sha256"]
    assert d["rounds"] == g["meta"]["its"] and d["obj_f64"] == g["obj_f64"]
    assert d["edges_scanned_per_solve"] == g["edges_scanned"]


def test_latest_bench_line_has_the_contract_fields_and_this_rounds_additions():
    name = _latest_c3()
    d = _line(name)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline", "bid_phase", "sol_sha256"):
        assert k in d, k
    assert d["unit"] == "Medges/s" and d["n_gpus"] == 1 and d["dtype"] == "f64" and d["vs_baseline"] is None
    assert "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-4
    assert abs(r["algorithmic_bytes_per_launch"] / (r["avg_launch_us"] * 1e-6) / 1e9 - r["achieved"]) < 0.01 * r["achieved"]
    for k in ("traffic", "launches", "algorithmic_bytes_per_edge"):
        assert k in r, k
    bp = d["bid_phase"]
    for k in ("fullscan_avg_us", "fullscan_frac_of_hbm_peak", "grid_all_launches", "k_tail"):
        assert k in bp, k
    assert bp["grid_all_launches"]["launches"] >= r["launches"] / d["steps"]
    c = d["cpu_baseline"]
    assert c["kind"] in ("port", "reference") and c["cores"] >= 1 and "whole_solve_medges_s" in c
    assert abs(d["value"] - d["edges_scanned_per_solve"] * d["steps"] / (d["ms_per_step"] * d["steps"] * 1e-3) / 1e6) < 0.01 * d["value"]
    g = json.load(open(os.path.join(ROOT, "tests", "golden", "large_cases.json")))["cases"]["C3"]
    assert d["sol_sha256"] == g["sol_sha256"] and d["rounds"] == g["meta"]["its"] and d["obj_f64"] == g["obj_f64"]
    if name >= "r06":  # round 6: BASELINE.md section 3's second CPU figure, the N-rank figures inside the roofline block
        assert c["optimised_same_assignment"] is True and 0 < c["optimised_whole_solve_s"] <= c["whole_solve_s"]
        assert r["all_ranks"]["n_ranks"] == 1 and r["all_ranks"]["mode"] == "single"
    if name >= "r03":  # SURVEY 8(d) additions of round 3: the second solve figure and the second peak
        # (round 4: the library's own read-only streaming pass is the measured peak the read-only scan is held against)
        peak = r["peak_measured_read"] if name >= "r04" else r["peak_measured_copy"]
        assert peak > (5500.0 if name >= "r04" else 3000.0) and abs(r["frac_of_measured"] - r["achieved"] / peak) < 1e-3
        h = d["solve_incl_h2d"]
        assert h["solve_ms_incl_h2d"] > d["solve_ms"] and h["h2d_bytes"] == 16 * d["config"]["nnz"]
        assert d["complete_assignment"] == [True, True, True] and d["valid_assignment"] is True


def _template_args(kernel):
    return [t.strip() for t in kernel[kernel.index("<") + 1:kernel.rindex(">")].split(",")]


def test_every_committed_roofline_block_of_this_round_follows_from_its_pmc_file():
    """profiles/r06_bench_*.json: `traffic` (HBM bytes per launch from the PMC passes) must belong to the kernel
    instance(s) whose time is in `avg_launch_us` -- the engine's MODE 0 instances, or the gather kernel's K = N instances
    k_bid<E, PriceSource, 0 | 1> -- equal the launch-weighted mean of those instances in the committed PMC file, and never
    lie below the bytes the launches claim to have read."""
    import glob
    names = sorted(glob.glob(os.path.join(ROOT, "profiles", "r06_bench_*.json")))
    if not names:
        pytest.skip("no bench lines of this round committed yet")
    for path in names:
        d = _line(os.path.basename(path))
        r = d["roofline"]
        assert "all_ranks" in r and r["all_ranks"]["n_ranks"] == d["n_gpus"], path
        if r["traffic"] is None:
            assert r.get("traffic_note") or r.get("traffic_stale") or "traffic_file" not in r, path
            continue
        kernels = r["traffic_kernel"].split(" + ")
        for k in kernels:
            ta = _template_args(k)
            if r["kernel"] == "k_bid_tiled":
                assert "k_bid_tiled<" in k and ta[9] == "0", (path, k)  # MODE 0: the bid scans
            else:
                assert "k_bid<" in k and ta[1].endswith("PriceSource") and ta[2] in ("0", "1"), (path, k)
        tj = json.load(open(os.path.join(ROOT, r["traffic_file"])))
        sel = [tj["kernels"][k] for k in kernels]
        n_l = sum(v["launches"] for v in sel)
        want = sum((v["read_avg"] + v["write_avg"]) * v["launches"] for v in sel) / n_l
        assert abs(r["traffic"] - want) <= 1.0 + 1e-9 * want, path
        # (the engine's packed records are 6 / 10 bytes per edge against the 8 / 12 the metric counts: a K = N scan may read
        # as little as 0.75x the algorithmic bytes + segment table; anything below that cannot be the bytes of these launches)
        assert r["traffic"] >= 0.75 * r["algorithmic_bytes_per_launch"], (path, "traffic below the bytes the launches read")
        assert abs(r["traffic_over_algorithmic"] - r["traffic"] / r["algorithmic_bytes_per_launch"]) < 2e-3, path


def test_bench_gpus_n_without_a_launcher_spawns_the_ranks_before_touching_the_gpu():
    """`python bench.py --gpus 2` with no WORLD_SIZE: the parent starts two fresh ranks (RANK / WORLD_SIZE / MASTER_*
    set) and has not imported torch itself.  Without a GPU both ranks stop with the no-fallback message and the parent
    relays the failure."""
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["MISSLAP_BENCH_TRACE"] = "1"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0",
                        "--no-cpu", "--config", "C1", "--mode", "replicas"], env=env, capture_output=True, text=True, timeout=600)
    assert "spawning 2 ranks, torch_imported_in_parent=False" in r.stderr
    import torch
    if not torch.cuda.is_available():
        assert r.returncode != 0 and r.stderr.count("bench.py needs a GPU") == 2


@pytest.mark.gpu
@pytest.mark.parametrize("mode,cfg,transport", [("sharded", "C2", None), ("replicas", "C1", None), ("sharded", "C2", "torch"),
                                                ("sharded", "C2", "rccl")])
def test_bench_gpus_2_launches_itself_on_one_gpu_under_gloo(mode, cfg, transport, gpu_lib):
    """The driver's form of the N > 1 run -- `python bench.py --gpus 2`, no launcher -- rehearsed on a one-GPU box:
    MISSLAP_DIST_BACKEND=gloo lets the two ranks share cuda:0 (the exchange of the sharded mode is then staged through
    the host).  One JSON line, the reference's assignment, and the proof-of-participation fields: every rank's device
    and assignment hash, what the communicator reports about itself, the exchanges a solve issued.
    transport 'torch': MISSLAP_BENCH_COMM=torch -- the exchange through torch.distributed's own all-reduces on the device
    buffers; 'rccl': the library's RCCL communicator is ASKED for with both ranks on one GPU, which RCCL refuses -- the
    fallback a node with a broken ncclCommInitRank would take: every rank agrees to switch, the line says so."""
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MISSLAP_BENCH_COMM")}
    env.update(MISSLAP_DIST_BACKEND="gloo", MISSLAP_BENCH_TRACE="1")
    if transport:
        env.update(MISSLAP_BENCH_COMM=transport)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1",
                        "--no-cpu", "--config", cfg, "--mode", mode], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    assert "torch_imported_in_parent=False" in r.stderr
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    g = json.load(open(os.path.join(ROOT, "tests", "golden", "large_cases.json")))["cases"][cfg]
    assert d["n_gpus"] == 2 and d["sol_sha256"] == g["sol_sha256"] and d["rounds"] == g["meta"]["its"]
    assert d["scaling"] == ("weak" if mode == "replicas" else "strong")
    assert [x["rank"] for x in d["ranks"]] == [0, 1] and d["sol_sha256_equal_on_all_ranks"] is True
    assert all(x["sol_sha256"] == g["sol_sha256"] and len(x["device_uuid"]) == 32 for x in d["ranks"])
    assert d["distinct_gpus"] == 1  # (this rehearsal: both ranks on cuda:0; the driver's 8-GPU run must say 8)
    if mode == "sharded":
        assert d["comm_kind"] == "custom" and d["rccl_nranks"] is None  # gloo-staged here; RCCL reports ncclCommCount
        assert d["comm_ranks_seen_by_every_rank"] == [2, 2]
        assert d["sharded_rounds_per_solve"] > 0 and d["exchanges_per_solve"] == 2 * d["sharded_rounds_per_solve"]
        assert ("torch.distributed all-reduces" in d["rank_transport"]) == (transport is not None)
        assert ("FALLBACK" in d["rank_transport"]) == (transport == "rccl")
        assert ("falling back to torch.distributed" in r.stderr) == (transport == "rccl")
    else:
        assert d["comm_kind"] is None and d["exchanges_per_solve"] == 0


@pytest.mark.gpu
@pytest.mark.parametrize("backend,n,cfg", [("threads", 8, "C5"), ("gloo", 4, "C3")])
def test_bench_line_of_the_many_rank_run_is_well_formed_on_one_gpu(backend, n, cfg, gpu_lib):
    """BASELINE config 5 as stated -- C5 sharded over EIGHT ranks -- rehearsed on the one GPU a box has: the eight ranks
    are threads of one bench.py process (MISSLAP_DIST_BACKEND=threads; eight rank processes would exceed the handful a
    one-GPU box admits on its card), and C3 over four rank PROCESSES under gloo.  The line must be the N-rank line the
    driver will read on an 8-GPU node: one JSON object, the reference's assignment on every rank, every rank's
    communicator reporting N ranks, exchanges = 2 x sharded rounds."""
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(MISSLAP_DIST_BACKEND=backend)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", "1", "--warmup", "0",
                        "--no-cpu", "--config", cfg], env=env, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    g = json.load(open(os.path.join(ROOT, "tests", "golden", "large_cases.json")))["cases"][cfg]
    assert d["n_gpus"] == n and d["scaling"] == "strong" and d["sol_sha256"] == g["sol_sha256"] and d["rounds"] == g["meta"]["its"]
    assert [x["rank"] for x in d["ranks"]] == list(range(n)) and d["sol_sha256_equal_on_all_ranks"] is True
    assert all(x["sol_sha256"] == g["sol_sha256"] for x in d["ranks"])
    assert d["comm_kind"] == "custom" and d["comm_ranks_seen_by_every_rank"] == [n] * n
    assert d["sharded_rounds_per_solve"] > 0 and d["exchanges_per_solve"] == 2 * d["sharded_rounds_per_solve"]
    assert d["distinct_gpus"] == 1 and ("THREAD" in d["rank_transport"]) == (backend == "threads")
    assert d["value"] > 0 and d["roofline"]["frac"] <= 1.0 and d["cpu_baseline"] is None
    # the scaling-relevant figures sit inside the `roofline` block (what a SCALE_rNN record keeps of the line)
    ar = d["roofline"]["all_ranks"]
    assert ar["n_ranks"] == n and ar["mode"] == "sharded" and ar["comm_kind"] == "custom" and ar["rccl_nranks"] is None
    assert ar["comm_ranks_seen_by_every_rank"] == [n] * n and ar["distinct_gpus"] == 1
    assert ar["sharded_rounds_per_solve"] == d["sharded_rounds_per_solve"] and ar["exchanges_per_solve"] == 2 * ar["sharded_rounds_per_solve"]
    assert ar["fullscan_medges_s"] == d["bid_phase"]["fullscan_all_ranks_medges_s"] and ar["fullscan_medges_s"] > 0
    assert 0 < ar["fullscan_frac_of_hbm_peak"] <= 1.0
    # unique work: the ranks' shares of the sharded rounds + the replicated rounds once = the single-GPU edge count
    assert round(d["value"] * d["ms_per_step"] * 1e3) == pytest.approx(g["edges_scanned"], rel=1e-3)


@pytest.mark.gpu
def test_bench_batch_leg_reports_beside_the_single_solve_value(gpu_lib):
    """`python bench.py --batch B`: B copies of the workload solved in lockstep (misslap_solve_batch) AFTER the timed
    single-solve steps; `value` stays the single-solve figure, the batch is reported beside it with every assignment's
    sha256 checked against the single solve's (= the fixture's)."""
    import subprocess
    import sys
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "0", "--no-cpu", "--config", "C1",
                        "--batch", "12", "--batch-group", "4"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    g = json.load(open(os.path.join(ROOT, "tests", "golden", "large_cases.json")))["cases"]["C1"]
    b = d["batch"]
    assert d["sol_sha256"] == g["sol_sha256"] and b["all_sha256_equal_reference_run"] is True
    assert b["B"] == 12 and b["groups"] == 3 and 0 < b["launches_issued"] < b["calls_recorded"]
    assert b["throughput_vs_single_solve"] > 1.0 and d["value"] > 0 and d["n_gpus"] == 1
