"""The committed bench line (profiles/r01_bench.json, written by `python bench.py` on an MI355X) carries every
field of the driver's contract, the roofline / cpu_baseline objects, and numbers that are consistent with each
other.  Runs without a GPU: it checks the artefact, not the measurement."""
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _line(name):
    with open(os.path.join(ROOT, "profiles", name)) as f:
        return json.loads(f.read().strip().splitlines()[-1])


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
