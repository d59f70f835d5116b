#!/usr/bin/env python3
"""Launch-boundary accounting from a rocprofv3 kernel trace (run on the GPU box).

    rocprofv3 --kernel-trace -d D -o t --output-format csv -- python3 tools/tail_stats.py C3 1
    python3 tools/trace_gaps.py D/*/t_kernel_trace.csv

Per kernel: launches, total and mean duration, and the mean idle time on the device before the launch (start minus
the latest end seen so far).  k_bid / k_round_fused launches are also bucketed by grid size: with one wavefront per
bidder the grid size is the host's upper bound of K, i.e. the buckets show what a round with so many bidders costs.
"""
import csv
import glob
import json
import sys
from collections import defaultdict


def main():
    paths = [p for a in sys.argv[1:] for p in glob.glob(a)]
    rows = []
    for p in paths:
        with open(p) as f:
            for r in csv.DictReader(f):
                rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"],
                             int(r.get("Grid_Size", r.get("Grid_Size_X", 0)) or 0),
                             int(r.get("Workgroup_Size", r.get("Workgroup_Size_X", 0)) or 0)))
    rows.sort()
    per = defaultdict(lambda: [0, 0, 0])
    durs = defaultdict(list)
    buckets = defaultdict(lambda: [0, 0, 0])
    last_end = None
    for s, e, name, grid, wg in rows:
        short = name.split("(")[0].split("<")[0].replace("misslap::", "").replace("void ", "")
        gap = 0 if last_end is None else max(0, s - last_end)
        if gap > 2_000_000:  # host pauses between solves are not launch boundaries
            gap = 0
        a = per[short]
        a[0] += 1
        a[1] += e - s
        a[2] += gap
        durs[short].append(e - s)
        if short in ("k_bid", "k_round_fused", "k_round_small") and wg:
            nb = grid // wg  # workgroups
            b = 1
            while b < nb:
                b *= 2
            c = buckets[(short, b)]
            c[0] += 1
            c[1] += e - s
            c[2] += gap
        last_end = e if last_end is None else max(last_end, e)
    out = {"kernels": {}, "by_workgroups": {}}
    for k, (n, d, g) in sorted(per.items(), key=lambda kv: -kv[1][1]):
        out["kernels"][k] = {"launches": n, "ms": round(d / 1e6, 3), "avg_us": round(d / n / 1e3, 2),
                             "gap_ms": round(g / 1e6, 3), "avg_gap_us": round(g / n / 1e3, 2),
                             "min_us": round(min(durs[k]) / 1e3, 2),
                             "p10_us": round(sorted(durs[k])[len(durs[k]) // 10] / 1e3, 2),
                             "median_us": round(sorted(durs[k])[len(durs[k]) // 2] / 1e3, 2)}
    for (k, b), (n, d, g) in sorted(buckets.items()):
        out["by_workgroups"][f"{k}<= {b}"] = {"launches": n, "avg_us": round(d / n / 1e3, 2),
                                              "avg_gap_us": round(g / n / 1e3, 2), "ms": round((d + g) / 1e6, 3)}
    json.dump(out, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main()
