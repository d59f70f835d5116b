#!/usr/bin/env python3
"""Copy a profile round's outputs (tools/profile_round.sh, gpurun_out/<dir>) into profiles/rNN_*: the PMC traffic files get
the commit they were measured on (the GPU box has no .git) after their source digest has been checked against the tree.
usage: install_profiles.py <gpurun_out subdir> <rNN> [pmc|stats|bench ...]"""
import json, os, shutil, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import source_digest
src, rnd, what = os.path.join(ROOT, "gpurun_out", sys.argv[1]), sys.argv[2], sys.argv[3:] or ["pmc", "stats", "bench"]
sha = source_digest()
commit = subprocess.check_output(["git", "-C", ROOT, "rev-parse", "HEAD"], text=True).strip()
dirty = subprocess.run(["git", "-C", ROOT, "diff", "--quiet", "HEAD", "--", "sslap_amd/csrc"]).returncode != 0
for c in ("C1", "C2", "C3", "C4", "C5", "D1", "C2_f64", "C3_f64", "C4_f64", "C2_shuffled", "C3_f32asf64"):
    p = os.path.join(src, f"pmc_traffic_{c}.json")
    if "pmc" in what and os.path.exists(p):
        d = json.load(open(p))
        assert d["source_sha256"] == sha, (c, "measured on other kernel sources", d["source_sha256"], sha)
        d["commit"], d["commit_dirty_csrc"] = commit, dirty
        json.dump(d, open(os.path.join(ROOT, "profiles", f"{rnd}_pmc_traffic_{c}.json"), "w"), indent=1)
        shutil.copy(os.path.join(src, f"pmc_summary_{c}.txt"), os.path.join(ROOT, "profiles", f"{rnd}_pmc_summary_{c}.txt"))
    for kind, name in (("stats", f"kernel_stats_{c}.csv"), ("bench", f"bench_{c}.json")):
        p = os.path.join(src, name)
        if kind in what and os.path.exists(p):
            shutil.copy(p, os.path.join(ROOT, "profiles", f"{rnd}_{name}"))
print("installed", what, "from", src, "at", commit[:8], "sources", sha[:8], "dirty" if dirty else "clean")
