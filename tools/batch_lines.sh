mkdir -p gpurun_out/batch
for cfg in "C2 64 16" "C2 128 32" "C3 32 16" "C1 64 16"; do
  set -- $cfg
  GPU_MAX_HW_QUEUES=16 timeout -k 10 500 python bench.py --config $1 --steps 2 --warmup 1 --no-cpu --batch $2 --batch-group $3 > gpurun_out/batch/bench_$1_B$2_G$3.json 2> gpurun_out/batch/bench.err || { echo "bench $cfg failed"; tail -5 gpurun_out/batch/bench.err; }
  python - <<PY
import json
d=json.loads(open("gpurun_out/batch/bench_$1_B$2_G$3.json").read().strip().splitlines()[-1])
b=d["batch"]; b["config"]="$1"; print(json.dumps(b))
PY
done
