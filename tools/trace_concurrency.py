#!/usr/bin/env python3
"""How many kernels run at a time: time-weighted histogram of the number of kernels in flight from a rocprofv3
kernel trace (run on the GPU box), overall and for the persistent tail kernels alone, plus per-queue busy time.

    rocprofv3 --kernel-trace -d D -o t --output-format csv -- python3 bench.py --concurrent 16 ...
    python3 tools/trace_concurrency.py D/*/t_kernel_trace.csv [t0_fraction t1_fraction]

Only the window [t0, t1] of the trace (fractions of its span; default: the last 45 %, where bench.py's concurrent
phase runs) is analysed."""
import csv
import glob
import json
import sys
from collections import defaultdict

paths = [p for a in sys.argv[1:2] for p in glob.glob(a)]
f0 = float(sys.argv[2]) if len(sys.argv) > 2 else 0.55
f1 = float(sys.argv[3]) if len(sys.argv) > 3 else 1.0
rows = []
for p in paths:
    with open(p) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "?")))
all_rows = list(rows)
lo, hi = min(r[0] for r in rows), max(r[1] for r in rows)
t0, t1 = lo + f0 * (hi - lo), lo + f1 * (hi - lo)
rows = [r for r in rows if r[1] > t0 and r[0] < t1]
ev = []
for s, e, name, q in rows:
    tail = "k_tail" in name
    ev.append((max(s, t0), 1, tail))
    ev.append((min(e, t1), -1, tail))
ev.sort()
hist, hist_tail = defaultdict(float), defaultdict(float)
n = nt = 0
prev = ev[0][0]
for t, d, tail in ev:
    hist[n] += t - prev
    hist_tail[nt] += t - prev
    prev = t
    n += d
    nt += d if tail else 0
span = t1 - t0
queues = defaultdict(float)
for s, e, name, q in rows:
    queues[q] += min(e, t1) - max(s, t0)
mean = lambda h: sum(k * v for k, v in h.items()) / span
out = {"window_ms": round(span / 1e6, 1), "kernels": len(rows), "queues_seen": len(queues),
       "mean_kernels_in_flight": round(mean(hist), 2), "mean_tail_kernels_in_flight": round(mean(hist_tail), 2),
       "idle_fraction": round(hist.get(0, 0.0) / span, 4),
       "in_flight_histogram": {str(k): round(v / span, 4) for k, v in sorted(hist.items())},
       "tail_in_flight_histogram": {str(k): round(v / span, 4) for k, v in sorted(hist_tail.items())},
       "queue_busy_fraction": {q: round(v / span, 3) for q, v in sorted(queues.items())}}
# per kernel name: mean duration while nothing else runs (before the first moment with 4 kernels in flight) and in the window
all_rows = sorted(all_rows)
ev_all = sorted([(s, 1) for s, e, *_ in all_rows] + [(e, -1) for s, e, *_ in all_rows])
n_all, split = 0, hi
for t, d in ev_all:
    n_all += d
    if n_all >= 4:
        split = t
        break
alone, crowd = defaultdict(list), defaultdict(list)
for s, e, name, q in all_rows:
    short = name.split("(")[0].replace("void ", "").replace("misslap::", "")
    if e <= split:
        alone[short].append(e - s)
    elif s >= t0:
        crowd[short].append(e - s)
per = {}
for k in sorted(crowd, key=lambda k: -sum(crowd[k]))[:14]:
    if alone.get(k):
        a, c = sum(alone[k]) / len(alone[k]) / 1e3, sum(crowd[k]) / len(crowd[k]) / 1e3
        per[k] = {"alone_us": round(a, 2), "concurrent_us": round(c, 2), "ratio": round(c / a, 2), "calls": len(crowd[k]),
                  "share_of_concurrent_kernel_time": round(sum(crowd[k]) / sum(sum(v) for v in crowd.values()), 3)}
out["per_kernel"] = per
print(json.dumps(out))
