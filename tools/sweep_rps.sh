#!/bin/bash
# rounds_per_sync sweep (batch length of the small rounds between status reads): bash tools/sweep_rps.sh <outdir> <cfg>...
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/$1; shift; mkdir -p "$O"; cd "$R"
for cfg in "$@"; do for rps in 16 8 6 4 3 2; do
  timeout -k 10 300 python3 tools/solve_time.py $cfg 5 profile=0 rounds_per_sync=$rps 2>&1 | grep -v amdgpu.ids | tee -a "$O/sweep_rps.txt"
done; done
