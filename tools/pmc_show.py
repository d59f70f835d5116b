#!/usr/bin/env python3
"""Print the counters of rocprofv3 --pmc passes (tools/pmc_passes.sh) for kernels whose name contains a pattern, one block
per kernel INSTANCE (the forward and the backward walk of the engine are two instances).
usage: pmc_show.py <dir> [pattern]"""
import collections
import csv
import glob
import sys

pat = sys.argv[2] if len(sys.argv) > 2 else "k_bid_tiled"
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob(sys.argv[1] + "/*/*_counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        if pat in r["Kernel_Name"]:
            agg[r["Kernel_Name"].split("(")[0].replace("void misslap::", "")][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in sorted(agg.items()):
    print(f"== {k}")
    for c, v in sorted(cs.items()):
        print(f"   {c:45s} n={len(v)} avg={sum(v)/len(v):16.1f}")
