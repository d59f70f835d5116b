#!/bin/bash
# Round profiles on the GPU box (run through gpurun from the repo root):
#   [EXTRA="--values f64" TAG=_f64] bash tools/profile_round.sh <outdir under gpurun_out> <config> [bench|stats|pmc ...]
# EXTRA: further bench.py arguments (the workload variant), TAG: suffix of the output names for that variant.
# bench: bench.py JSON line; stats: rocprofv3 --kernel-trace --stats of the same command; pmc: two separate counter
# passes (FETCH_SIZE, WRITE_SIZE; kernel-trace only, as the pool requires) of one untimed-CPU solve.
set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$1; C=$2; shift 2
X=${EXTRA:-}; T=${TAG:-}
mkdir -p "$O"; cd /tmp && export TMPDIR=/tmp PYTHONPATH=$R
for what in "$@"; do
  case $what in
    bench) timeout -k 10 900 python3 "$R/bench.py" --config "$C" $X > "$O/bench_$C$T.json" 2> "$O/bench_$C$T.err" || { echo "bench $C$T failed"; tail -3 "$O/bench_$C$T.err"; exit 1; } ;;
    stats) timeout -k 10 600 rocprofv3 --kernel-trace --stats -d "$O/stats_$C$T" -o s --output-format csv -- python3 "$R/bench.py" --config "$C" $X --steps 2 --warmup 1 --no-cpu > "$O/bench_${C}${T}_under_rocprof.json" 2> "$O/stats_$C$T.err" || { echo "stats $C$T failed"; tail -3 "$O/stats_$C$T.err"; exit 1; }
           cp "$O/stats_$C$T"/*kernel_stats.csv "$O/kernel_stats_$C$T.csv"; rm -rf "$O/stats_$C$T" ;;
    pmc) for ctr in FETCH_SIZE WRITE_SIZE; do
           timeout -k 10 600 rocprofv3 --kernel-trace --pmc $ctr -d "$O/pmc_${C}${T}_$ctr" -o p --output-format csv -- python3 "$R/bench.py" --config "$C" $X --steps 1 --warmup 0 --no-cpu > "$O/pmc_${C}_$ctr.json" 2> "$O/pmc_${C}_$ctr.err" || { echo "pmc $ctr $C$T failed"; tail -3 "$O/pmc_${C}_$ctr.err"; exit 1; }
           cp "$O/pmc_${C}${T}_$ctr"/*counter_collection.csv "$O/pmc_${C}_$ctr.csv"
           rm -rf "$O/pmc_${C}${T}_$ctr"
         done
         # reduce on the box (the raw per-dispatch CSVs of a long solve exceed what gpurun copies back)
         python3 "$R/tools/pmc_summary.py" "$O/pmc_${C}_FETCH_SIZE.csv" "$O/pmc_${C}_WRITE_SIZE.csv" "$O/pmc_traffic_$C$T.json" \
                 $(python3 -c "import json; print(json.load(open('$O/pmc_${C}_FETCH_SIZE.json'))['config']['nnz'])") > "$O/pmc_summary_$C$T.txt" \
           && rm -f "$O/pmc_${C}_FETCH_SIZE.csv" "$O/pmc_${C}_WRITE_SIZE.csv" ;;
  esac
done
echo "profile_round $C $* done"
