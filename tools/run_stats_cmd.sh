#!/bin/bash
# rocprofv3 --kernel-trace --stats of an arbitrary python tool on the GPU box, top kernels to <outdir>/<tag>_stats.txt:
#   bash tools/run_stats_cmd.sh <outdir> <tag> <script> [args...]
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/$1; T=$2; shift 2; mkdir -p "$O"; cd /tmp && export TMPDIR=/tmp PYTHONPATH=$R
rm -rf /tmp/st
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d /tmp/st -o s --output-format csv -- python3 "$R/$1" "${@:2}" > "$O/${T}_out.txt" 2> "$O/${T}_err.txt" || { tail -5 "$O/${T}_err.txt"; exit 1; }
python3 - /tmp/st/s_kernel_stats.csv > "$O/${T}_stats.txt" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(int(r['TotalDurationNs']) for r in rows)
print('total kernel ms', round(tot / 1e6, 3))
for r in rows[:22]:
    print(r['Name'].split('(')[0].replace('void ', '').replace('misslap::', '')[:64].ljust(64), r['Calls'].rjust(7), str(round(int(r['TotalDurationNs']) / 1e6, 3)).rjust(9), str(round(float(r['AverageNs']) / 1e3, 2)).rjust(8))
PY
cat "$O/${T}_out.txt" | tail -3; cat "$O/${T}_stats.txt"
