#!/bin/bash
# dense sizes under variants of the long-row switches: bash tools/sweep_long_rows.sh <outdir> "<sizes>" <ENV=V,ENV=V | none>...
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/$1; SZ=$2; shift 2; mkdir -p "$O"; cd "$R"
for spec in "$@"; do
  ( if [ "$spec" != "none" ]; then OLDIFS=$IFS; IFS=,; for kv in $spec; do export "$kv"; done; IFS=$OLDIFS; fi
    echo "== $spec"; timeout -k 10 300 python3 tools/dense_sizes.py $SZ | cut -c1-60; timeout -k 10 300 python3 tools/small_sizes.py $SZ | cut -c1-75 ) 2>&1 | tee -a "$O/sweep_long_rows.txt"
done
