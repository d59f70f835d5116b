#!/bin/bash
# A/B of library builds under rocprofv3 kernel statistics: bash tools/ab_stats.sh <outdir> <config> <lib name>...
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/$1; C=$2; shift 2; mkdir -p "$O"; cd /tmp; export TMPDIR=/tmp PYTHONPATH=$R
for n in "$@"; do
  MISSLAP_LIB=$R/build_ab/lib_$n.so timeout -k 10 300 rocprofv3 --kernel-trace --stats -d "$O/st_$n" -o s --output-format csv -- python3 $R/bench.py --config $C --no-cpu --steps 2 --warmup 1 > "$O/$n.json" 2> "$O/$n.err"; rc=$?
  cp "$O/st_$n"/*kernel_stats.csv "$O/kernel_stats_${C}_$n.csv" 2>/dev/null; rm -rf "$O/st_$n"
  echo "== $n rc=$rc"; grep -h "k_bid<\|k_bid_tiled\|k_round_small\|k_apply" "$O/kernel_stats_${C}_$n.csv" | cut -d, -f1-4 | sed 's/misslap:://g' | cut -c1-150
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "time limit hit"; exit 1; fi
done
