#!/bin/bash
# A/B of library builds through bench.py (C3 or $CFG, 3 steps, no CPU leg): [CFG=C2] bash tools/ab_bench.sh <outdir> <lib name>...
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/$1; shift; mkdir -p "$O"
for n in "$@"; do
  MISSLAP_LIB=$R/build_ab/lib_$n.so timeout -k 10 240 python3 $R/bench.py --no-cpu --steps 3 --config ${CFG:-C3} > "$O/$n.json" 2> "$O/$n.err"; rc=$?
  python3 - "$O/$n.json" "$n" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    f = d['bid_phase']['fullscan_avg_us']; a = d['roofline']['avg_launch_us']
    print(sys.argv[2], 'ms/step', d['ms_per_step'], 'full', f, 'all', a, 'partial %.1f' % ((17 * a - 10 * f) / 7), 'frac', d['roofline']['frac'],
          'tail us/round', d['bid_phase']['k_tail']['us_per_round'], d['sol_sha256'][:8])
except Exception as e:
    print(sys.argv[2], 'ERR', e)
PY
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "time limit hit in $n: stopping"; exit 1; fi
done
