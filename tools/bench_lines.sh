# the bench.py lines of every profiled config (after the PMC files of the same sources are installed), the default
# bench line, a sharded fuzz run -> gpurun_out/prof
set -e
for c in C3 C2 C4 C5 C1 D1; do bash tools/profile_round.sh prof $c bench; done
EXTRA="--values f64" TAG=_f64 bash tools/profile_round.sh prof C3 bench
EXTRA="--values f64 --no-cpu" TAG=_f64 bash tools/profile_round.sh prof C2 bench
EXTRA="--values f64 --no-cpu" TAG=_f64 bash tools/profile_round.sh prof C4 bench
EXTRA="--shuffle-rows --no-cpu" TAG=_shuffled bash tools/profile_round.sh prof C2 bench
cd $GRAFT_REPO_ROOT && timeout -k 10 300 python bench.py > gpurun_out/prof/default_bench.json 2> gpurun_out/prof/default_bench.err; tail -c 600 gpurun_out/prof/default_bench.json
timeout -k 10 600 python tools/fuzz_sharded.py 0 60 > gpurun_out/prof/fuzz_sharded_0_60.txt 2>&1 || { tail -30 gpurun_out/prof/fuzz_sharded_0_60.txt; exit 1; }
tail -3 gpurun_out/prof/fuzz_sharded_0_60.txt
