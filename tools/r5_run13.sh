for cfg in "C2 64 64" "C2 64 32" "C2 64 16" "C2 64 12" "C2 128 64" "C3 32 32" "C1 64 64"; do
  set -- $cfg
  timeout -k 10 500 python bench.py --config $1 --steps 2 --warmup 1 --no-cpu --batch $2 --batch-group $3 > gpurun_out/r5_batch_$1_$2_$3.json 2> gpurun_out/r5_batch_$1_$2_$3.err || { echo "bench $cfg failed"; tail -5 gpurun_out/r5_batch_$1_$2_$3.err; }
  python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/r5_batch_$1_$2_$3.json").read().strip().splitlines()[-1])
    b=d["batch"]; print("$cfg", {k:b[k] for k in ("groups","wall_ms","ms_per_solve","launches_issued","calls_recorded","throughput_vs_single_solve","all_sha256_equal_reference_run","create_ms_for_all")})
except Exception as e: print("$cfg ERR", e)
PY
done
