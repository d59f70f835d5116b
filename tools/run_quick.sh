#!/bin/bash
# One GPU-box call: the gpu test suite, then short bench lines of the named configs (no CPU leg), optionally under
# environment variants:  bash tools/run_quick.sh <outdir under gpurun_out> <pytest -k expr or "all" or "none"> <cfg[:ENV=V,...]>...
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/$1; K=$2; shift 2
mkdir -p "$O"; cd "$R"
if [ "$K" != "none" ]; then
  if [ "$K" = "all" ]; then KX=(); else KX=(-k "$K"); fi
  timeout -k 10 900 python3 -m pytest tests -m gpu -x -q "${KX[@]}" > "$O/pytest.log" 2>&1; rc=$?
  tail -5 "$O/pytest.log"
  if [ $rc -ne 0 ]; then echo "pytest failed ($rc)"; exit 1; fi
fi
for spec in "$@"; do
  cfg=${spec%%:*}; envs=""; [ "$spec" != "$cfg" ] && envs=${spec#*:}
  tag=$cfg$(echo "$envs" | tr -c 'A-Za-z0-9=\n' '_')
  ( IFS=,; for kv in $envs; do export "$kv"; done
    timeout -k 10 400 python3 bench.py --config "$cfg" --no-cpu --steps 3 --warmup 1 > "$O/bench_$tag.json" 2> "$O/bench_$tag.err" ); rc=$?
  python3 - "$O/bench_$tag.json" "$tag" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[2], 'ms/step', d['ms_per_step'], 'solve', d['solve_ms'], 'setup', d['setup_ms'], 'full', d['bid_phase']['fullscan_avg_us'],
          'all', d['roofline']['avg_launch_us'], 'frac', d['roofline']['frac'], 'tail us/round', d['bid_phase']['k_tail']['us_per_round'], d['sol_sha256'][:8])
except Exception as e:
    print(sys.argv[2], 'ERR', e)
PY
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "time limit hit in $spec: stopping"; exit 1; fi
done
echo "run_quick done"
