mkdir -p gpurun_out/r5batch
GPU_MAX_HW_QUEUES=16 timeout -k 10 600 python tools/batch_mixed.py 50000 0.005 48 16 > gpurun_out/r5batch/mixed_C2shape_48.json 2> gpurun_out/r5batch/mixed.err; cat gpurun_out/r5batch/mixed_C2shape_48.json
GPU_MAX_HW_QUEUES=16 timeout -k 10 600 python tools/batch_mixed.py 5000 0.02 64 16 > gpurun_out/r5batch/mixed_C1shape_64.json 2>> gpurun_out/r5batch/mixed.err; cat gpurun_out/r5batch/mixed_C1shape_64.json
for cfg in "C2 64 16" "C2 128 32" "C2 16 16" "C1 64 16" "C3 32 16" "C4 32 16"; do
  set -- $cfg
  timeout -k 10 500 python bench.py --config $1 --steps 2 --warmup 1 --no-cpu --batch $2 --batch-group $3 > gpurun_out/r5batch/bench_$1_B$2_G$3.json 2> gpurun_out/r5batch/bench.err || { echo "bench $cfg failed"; tail -5 gpurun_out/r5batch/bench.err; }
  python - <<PY
import json
d=json.loads(open("gpurun_out/r5batch/bench_$1_B$2_G$3.json").read().strip().splitlines()[-1])
b=d["batch"]; print("$cfg", {k:b[k] for k in ("groups","wall_ms","ms_per_solve","aggregate_medges_s","throughput_vs_single_solve","all_sha256_equal_reference_run","launches_issued","calls_recorded")})
PY
done
