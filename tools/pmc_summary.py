#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; one pass each) into per-kernel HBM traffic.

gfx950 corrections (MI355X_MICROARCH.md, HBM section): both counters are in KB; FETCH_SIZE reports half of
the bytes of a coalesced streaming read -- calibrated here on kernels with a known byte count
(k_ingest_vals reads nnz*8 B, k_build_edges_f32 reads nnz*16 B: both report exactly 1/2) -- so read bytes =
FETCH_SIZE * 1024 * 2; WRITE_SIZE is exact (k_build_edges_f32 writes nnz*8 B).
The output names the code it was measured on: sha256 of the kernel sources (bench.py::source_digest -- what
bench.py can check on the GPU box) and the git commit of the working tree at summary time.
usage: pmc_summary.py <fetch_counter_collection.csv> <write_counter_collection.csv> <out.json> [nnz]
"""
import collections
import csv
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import source_digest  # noqa: E402


def load(path):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        agg[r["Kernel_Name"].split("(")[0].replace("void ", "")].append(float(r["Counter_Value"]))
    return agg


fetch, write = load(sys.argv[1]), load(sys.argv[2])
nnz = int(sys.argv[4]) if len(sys.argv) > 4 else 40179959
try:
    commit = subprocess.check_output(["git", "-C", ROOT, "rev-parse", "HEAD"], text=True).strip()
    dirty = bool(subprocess.check_output(["git", "-C", ROOT, "status", "--porcelain", "sslap_amd/csrc"], text=True).strip())
except Exception:
    commit, dirty = None, None
out = {"units": "bytes per launch", "fetch_correction": 2.0, "kernels": {}, "source_sha256": source_digest(),
       "commit": commit, "commit_dirty_csrc": dirty}
for k in sorted(set(fetch) | set(write)):
    f, w = fetch.get(k, [0.0]), write.get(k, [0.0])
    out["kernels"][k] = dict(
        launches=len(f), read_avg=sum(f) / len(f) * 1024 * 2, read_max=max(f) * 1024 * 2,
        write_avg=sum(w) / len(w) * 1024, write_max=max(w) * 1024)
cal = out["kernels"]
out["calibration"] = {"k_ingest_vals_read_over_known": cal["misslap::k_ingest_vals"]["read_avg"] / (nnz * 8.0)}
if "misslap::k_build_edges_f32" in cal:  # 8 B/edge layout: reads 16 B, writes 8 B per entry
    out["calibration"]["k_build_edges_f32_read_over_known"] = cal["misslap::k_build_edges_f32"]["read_avg"] / (nnz * 16.0)
    out["calibration"]["k_build_edges_f32_write_over_known"] = cal["misslap::k_build_edges_f32"]["write_avg"] / (nnz * 8.0)
if "misslap::k_build_edges_f64" in cal:  # 12 B/edge layout: reads 16 B, writes 12 B per entry
    out["calibration"]["k_build_edges_f64_read_over_known"] = cal["misslap::k_build_edges_f64"]["read_avg"] / (nnz * 16.0)
    out["calibration"]["k_build_edges_f64_write_over_known"] = cal["misslap::k_build_edges_f64"]["write_avg"] / (nnz * 12.0)
json.dump(out, open(sys.argv[3], "w"), indent=1, sort_keys=True)
for k, v in out["kernels"].items():
    if v["read_avg"] + v["write_avg"] > 1e6:
        print(f"{k[:70]:70s} n={v['launches']:5d} read {v['read_avg']/1e6:9.1f} MB  write {v['write_avg']/1e6:8.1f} MB")
print(out["calibration"])
