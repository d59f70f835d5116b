#!/usr/bin/env python3
"""Analysis tooling (CPU): build and run tools/sim/auction_sim.c on a BASELINE config and check that the model's
`its` and sha256(sol) equal the golden fixture (so a drifting model is noticed).
usage: run_sim.py [config] [C] [policy] [build_thr] [use_thr]"""
import hashlib
import json
import os
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from sslap_amd import synth  # noqa: E402

cfg = sys.argv[1] if len(sys.argv) > 1 else "C3"
args = sys.argv[2:6] + ["15", "1", "64", "64"][len(sys.argv[2:6]):]
exe = "/tmp/auction_sim"
subprocess.check_call(["gcc", "-O2", "-ffp-contract=off", "-o", exe, os.path.join(HERE, "auction_sim.c"), "-lm"])
inp = f"/tmp/sim_{cfg}.bin"
if not os.path.exists(inp):
    loc, val = synth.gen_config(cfg)
    with open(inp, "wb") as f:
        f.write(np.int64(loc.shape[0]).tobytes())
        f.write(np.int32(1).tobytes())
        f.write(np.ascontiguousarray(loc, dtype=np.int32).tobytes())
        f.write(np.ascontiguousarray(val, dtype=np.float64).tobytes())
sol_path = f"/tmp/sim_{cfg}_{os.getpid()}.sol"
out = subprocess.check_output([exe, inp] + args + [sol_path]).decode()
res = json.loads(out)
sol = np.fromfile(sol_path, dtype=np.int32)
os.remove(sol_path)
g = json.load(open(os.path.join(ROOT, "tests", "golden", "large_cases.json")))["cases"].get(cfg)
if g:
    ok = res["its"] == g["meta"]["its"] and hashlib.sha256(sol.tobytes()).hexdigest() == g["sol_sha256"]
    res["matches_fixture"] = bool(ok)
modes = res.pop("modes")
for m in res.pop("lds_sim", []):
    print("  lds %-12s bids=%9d line_hit=%s record_hit=%s all_records_of_a_bid_hit=%s" % (
        m["mode"], m["bids"], m["line_hit"], m["record_hit"], m["all_records_of_a_bid_hit"]))
by_k = res.pop("rounds_by_K", None)
print(json.dumps(res))
for m in modes:
    print("  %-12s rounds_all=%8d counted=%8d bids=%9d hit=%.3f all-hit rounds=%.3f" % (
        m["mode"], m["rounds_all"], m["rounds"], m["bids"], m["hit_rate"], m["all_hit_rounds"]))
if by_k:
    print("  rounds with K = 1..32 bidders:", by_k)
