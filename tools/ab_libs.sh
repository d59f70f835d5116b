#!/bin/bash
# A/B of kernel variants on ONE GPU box: every library under build_ab/ named on the command line runs the same
# measurement script in a fresh process (MISSLAP_LIB selects the build).  A run that is killed at its time limit ends
# the whole sequence (nothing else is started on a GPU that may be unresponsive).
#   bash tools/ab_libs.sh <outdir under gpurun_out> <script + args, quoted> <lib name>...
# e.g. bash tools/ab_libs.sh ab1 "tools/tail_stats.py C3 3" base thr1 thr3
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/$1; CMD=$2; shift 2
mkdir -p "$O"; export PYTHONPATH=$R
for n in "$@"; do
  lib=$R/build_ab/lib_$n.so
  [ -f "$lib" ] || { echo "missing $lib"; continue; }
  MISSLAP_LIB=$lib timeout -k 10 240 python3 $R/$CMD > "$O/$n.json" 2> "$O/$n.err"
  rc=$?
  echo "$n rc=$rc $(tail -c 600 "$O/$n.json" | tr '\n' ' ')"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "time limit hit in $n: stopping"; exit 1; fi
done
echo "ab_libs done"
