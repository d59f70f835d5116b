#!/usr/bin/env python3
"""Register / LDS / spill numbers of every kernel of the library, from hipcc -Rpass-analysis=kernel-resource-usage
(runs without a GPU).  usage: kernel_resources.py [substring ...] [-- extra hipcc flags]"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sslap_amd import build  # noqa: E402

args = sys.argv[1:]
extra = []
if "--" in args:
    k = args.index("--")
    args, extra = args[:k], args[k + 1:]
cmd = [build.hipcc()] + build.FLAGS + extra + ["-Rpass-analysis=kernel-resource-usage", os.path.join(build.CSRC, "misslap.hip"),
                                                "-o", "/tmp/_kernel_resources.so"]
txt = subprocess.run(cmd, cwd=build.CSRC, capture_output=True, text=True).stderr
KEYS = (("sgpr", r"TotalSGPRs"), ("vgpr", r"VGPRs"), ("scratch", r"ScratchSize \[bytes/lane\]"),
        ("occ", r"Occupancy \[waves/SIMD\]"), ("sgpr_spill", r"SGPRs Spill"), ("vgpr_spill", r"VGPRs Spill"),
        ("lds", r"LDS Size \[bytes/block\]"))
for b in re.split(r"remark: Function Name: ", txt)[1:]:
    name = subprocess.run(["c++filt", b.split()[0]], capture_output=True, text=True).stdout.strip()
    name = re.sub(r"^void misslap::", "", name).split("(")[0]
    if args and not any(a in name for a in args):
        continue
    vals = []
    for label, key in KEYS:
        m = re.search(key + r": (\d+)", b)
        vals.append(f"{label}={m.group(1) if m else '?'}")
    print(f"{name[:88]:88s} " + " ".join(vals))
