#!/bin/bash
# bench.py per launch shape of k_bid_tiled for one config: bash tools/ab_shapes_cfg.sh <outdir> <config> <shape>...   ("auto" = no override)
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/$1; C=$2; shift 2; mkdir -p "$O"
for n in "$@"; do
  if [ "$n" = auto ]; then unset MISSLAP_TILED_SHAPE; else export MISSLAP_TILED_SHAPE=$n; fi
  timeout -k 10 240 python3 $R/bench.py --no-cpu --steps 3 --config $C > "$O/${C}_shape$n.json" 2> "$O/${C}_shape$n.err"; rc=$?
  python3 - "$O/${C}_shape$n.json" "$C shape $n" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[2], 'ms/step', d['ms_per_step'], 'full', d['bid_phase']['fullscan_avg_us'], 'all', d['roofline']['avg_launch_us'], 'frac', d['roofline']['frac'], d['roofline']['kernel'], d['sol_sha256'][:8])
except Exception as e:
    print(sys.argv[2], 'ERR', e)
PY
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "time limit hit: stopping"; exit 1; fi
done
