set -o pipefail
python -m pytest tests -m gpu -x -q > gpurun_out/r5_t3.log 2>&1; echo rc=$? >> gpurun_out/r5_t3.log; tail -5 gpurun_out/r5_t3.log
for v in "C3 --values f64" "C3 --values f32-as-f64" "C2 --values f64" "C4 --values f64" "C2 --shuffle-rows" "C5" "C3"; do
  set -- $v; cfg=$1; shift
  tag=$(echo "$v" | tr ' -' '__')
  timeout -k 10 400 python bench.py --config $cfg "$@" --steps 2 --warmup 1 --no-cpu > gpurun_out/r5_b_$tag.json 2> gpurun_out/r5_b_$tag.err || echo "bench $v failed"
  python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/r5_b_$tag.json").read().strip().splitlines()[-1])
    print("$v", "ms/step", d["ms_per_step"], "solve", d["solve_ms"], "fs_us", d["bid_phase"]["fullscan_avg_us"], "frac", d["roofline"]["frac"], "fs_frac", d["bid_phase"]["fullscan_frac_of_hbm_peak"], "fmt", d["config"].get("tile_major_format"), "bpe", d["config"]["bytes_per_edge"], d["sol_sha256"][:12])
except Exception as e: print("$v", "ERR", e)
PY
done
