#!/bin/bash
# bench.py --concurrent B on the GPU box: bash tools/run_conc.sh <outdir> <config> <B[:ENV=V,...]>...
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/$1; C=$2; shift 2; mkdir -p "$O"; cd "$R"
for spec in "$@"; do
  B=${spec%%:*}; envs=""; [ "$spec" != "$B" ] && envs=${spec#*:}
  tag=${C}_B$B$(echo "$envs" | tr -c 'A-Za-z0-9=\n' '_')
  ( IFS=,; for kv in $envs; do export "$kv"; done
    timeout -k 10 500 python3 bench.py --config "$C" --no-cpu --steps 3 --warmup 1 --concurrent "$B" > "$O/conc_$tag.json" 2> "$O/conc_$tag.err" ); rc=$?
  python3 - "$O/conc_$tag.json" "$tag" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[2], 'single ms/step', d['ms_per_step'], 'value', d['value'], 'concurrent', d['concurrent'], 'peaks', d['roofline']['peak_measured_read'], d['roofline']['peak_measured_copy'])
except Exception as e:
    print(sys.argv[2], 'ERR', e)
PY
  if [ $rc -ne 0 ]; then echo "$spec failed ($rc)"; tail -5 "$O/conc_$tag.err"; exit 1; fi
done
