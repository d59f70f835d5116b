#!/usr/bin/env python3
"""Where a solve's wall time goes when it is not inside a kernel: from a rocprofv3 kernel trace of bench.py (one
stream), the idle gaps between consecutive kernels of the LAST solve in the trace, grouped by the pair of kernels
around them.

    rocprofv3 --kernel-trace -d D -o t --output-format csv -- python3 bench.py --config C4 --no-cpu ...
    python3 tools/trace_gaps.py D/t_kernel_trace.csv [solve_index]

solve_index: which solve of the trace (default 3: with --warmup 1 --steps 3 the last timed step; the solves behind it are
bench.py's profiled extra solve and the one from host arrays)."""
import csv
import json
import sys
from collections import defaultdict

which = int(sys.argv[2]) if len(sys.argv) > 2 else 3
first_name, last_name = "k_ingest_rows", "k_obj_sum"
rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "").replace("misslap::", "")))
rows.sort()
ends = [k for k, r in enumerate(rows) if last_name in r[2]]
which = min(which, len(ends) - 1)
hi = ends[which]
prev_end = ends[which - 1] if which > 0 else -1
lo = next(k for k in range(prev_end + 1, hi) if first_name in rows[k][2])
sel = rows[lo:hi + 1]
span = sel[-1][1] - sel[0][0]
busy = sum(e - s for s, e, _ in sel)
gaps = defaultdict(lambda: [0, 0])
for (s0, e0, n0), (s1, e1, n1) in zip(sel, sel[1:]):
    g = max(0, s1 - e0)
    k = f"{n0[:40]} -> {n1[:40]}"
    gaps[k][0] += g
    gaps[k][1] += 1
kern = defaultdict(lambda: [0, 0])
for s, e, n in sel:
    kern[n[:60]][0] += e - s
    kern[n[:60]][1] += 1
out = {"solve_index": which, "solves_in_trace": len(ends), "kernels": len(sel), "span_ms": round(span / 1e6, 3), "busy_ms": round(busy / 1e6, 3), "idle_ms": round((span - busy) / 1e6, 3),
       "top_gaps": [{"pair": k, "total_us": round(v[0] / 1e3, 1), "count": v[1], "avg_us": round(v[0] / v[1] / 1e3, 2)}
                    for k, v in sorted(gaps.items(), key=lambda kv: -kv[1][0])[:18]],
       "top_kernels": [{"kernel": k, "total_us": round(v[0] / 1e3, 1), "count": v[1], "avg_us": round(v[0] / v[1] / 1e3, 2)}
                       for k, v in sorted(kern.items(), key=lambda kv: -kv[1][0])[:18]]}
print(json.dumps(out, indent=1))
