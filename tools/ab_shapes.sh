#!/bin/bash
# bench.py (C3, 3 steps) per launch shape of k_bid_tiled: bash tools/ab_shapes.sh <outdir> <shape>...
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/$1; shift; mkdir -p "$O"
for n in "$@"; do
  MISSLAP_TILED_SHAPE=$n timeout -k 10 240 python3 $R/bench.py --no-cpu --steps 3 > "$O/shape$n.json" 2> "$O/shape$n.err"; rc=$?
  python3 - "$O/shape$n.json" "shape$n" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    f = d['bid_phase']['fullscan_avg_us']; a = d['roofline']['avg_launch_us']
    print(sys.argv[2], 'ms/step', d['ms_per_step'], 'full', f, 'all', a, 'partial %.1f' % ((17 * a - 10 * f) / 7), 'frac', d['roofline']['frac'], d['sol_sha256'][:8])
except Exception as e:
    print(sys.argv[2], 'ERR', e)
PY
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "time limit hit: stopping"; exit 1; fi
done
