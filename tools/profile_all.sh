#!/bin/bash
# How the profiles/rNN_* files of a round are made (GPU box; three gpurun calls A / B / C, each below 20 minutes):
#   bash tools/profile_all.sh A|B|C      -> gpurun_out/prof/...;  then  python tools/install_profiles.py prof rNN
# Every config: bench.py line, rocprofv3 --kernel-trace --stats, PMC passes (tools/profile_round.sh).
# bench.py looks the PMC traffic of its roofline kernel up in profiles/ (same kernel sources only), so the bench lines are
# taken once more after the install:  bash tools/bench_lines.sh; python tools/install_profiles.py prof rNN bench.
# tools/verify_gpu.sh: the whole -m gpu suite, plain and with every device block poisoned, + a sharded fuzz run.
set -e
case "$1" in
A) for c in C3 C2 C4; do bash tools/profile_round.sh prof $c bench stats pmc; done ;;
B) for c in C5 C1 D1; do bash tools/profile_round.sh prof $c bench stats pmc; done ;;
C) EXTRA="--values f64" TAG=_f64 bash tools/profile_round.sh prof C3 bench stats pmc
   EXTRA="--values f64 --no-cpu" TAG=_f64 bash tools/profile_round.sh prof C2 bench pmc
   EXTRA="--values f64 --no-cpu" TAG=_f64 bash tools/profile_round.sh prof C4 bench pmc
   EXTRA="--shuffle-rows --no-cpu" TAG=_shuffled bash tools/profile_round.sh prof C2 bench pmc
   EXTRA="--values f32-as-f64 --no-cpu" TAG=_f32asf64 bash tools/profile_round.sh prof C3 bench ;;
*) echo "usage: $0 A|B|C"; exit 2 ;;
esac
ls gpurun_out/prof
