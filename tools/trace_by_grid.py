#!/usr/bin/env python3
"""Duration of one kernel by launch size, from a rocprofv3 kernel trace: trace_by_grid.py <trace.csv> <kernel substring>"""
import csv, sys, json
from collections import defaultdict
b = defaultdict(list)
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        if sys.argv[2] in r["Kernel_Name"]:
            g = r.get("Grid_Size") or r.get("Grid_Size_X")
            w = r.get("Workgroup_Size") or r.get("Workgroup_Size_X")
            if g is None:
                sys.exit("columns: " + ", ".join(r.keys()))
            wgs = int(g) // max(int(w), 1)
            b[wgs].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
edges = [1, 2, 4, 8, 16, 32, 64, 128, 256, 512, 1024, 4096]
out = []
for lo, hi in zip(edges, edges[1:]):
    v = [x for k, xs in b.items() if lo <= k < hi for x in xs]
    if v:
        v.sort()
        out.append(dict(workgroups=f"{lo}..{hi - 1}", launches=len(v), mean_us=round(sum(v) / len(v), 2), median_us=round(v[len(v) // 2], 2),
                        p10_us=round(v[len(v) // 10], 2), p90_us=round(v[9 * len(v) // 10], 2), total_ms=round(sum(v) / 1e3, 2)))
print(json.dumps(out, indent=1))
