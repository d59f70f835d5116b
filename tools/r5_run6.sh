set -o pipefail
echo start > gpurun_out/r5_t6.log
timeout -k 10 900 python -m pytest @tools/r5_rest_tests.txt -x -q -p no:cacheprovider >> gpurun_out/r5_t6.log 2>&1; echo rc=$? >> gpurun_out/r5_t6.log; tail -5 gpurun_out/r5_t6.log
for v in "C5" "C1" "D1"; do
  set -- $v; cfg=$1; shift
  tag=$(echo "$v" | tr ' -' '__')
  timeout -k 10 500 python bench.py --config $cfg "$@" --steps 2 --warmup 1 --no-cpu > gpurun_out/r5_b4_$tag.json 2> gpurun_out/r5_b4_$tag.err || echo "bench $v failed"
  python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/r5_b4_$tag.json").read().strip().splitlines()[-1])
    print("$v", "ms/step", d["ms_per_step"], "solve", d["solve_ms"], "fs_us", d["bid_phase"]["fullscan_avg_us"], "frac", d["roofline"]["frac"], "refeq", d["roofline"]["reference_equivalent_GBs"], d["sol_sha256"][:12])
except Exception as e: print("$v", "ERR", e)
PY
done
bash tools/r5_tail_stamps.sh run C3 > gpurun_out/r5_tail_stamps_C3.txt 2>&1; tail -8 gpurun_out/r5_tail_stamps_C3.txt
