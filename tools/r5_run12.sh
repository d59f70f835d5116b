for cfg in "C2 64" "C2 24" "C1 64" "C3 16" "C4 32"; do
  set -- $cfg
  timeout -k 10 500 python bench.py --config $1 --steps 2 --warmup 1 --no-cpu --batch $2 > gpurun_out/r5_batch_$1_$2.json 2> gpurun_out/r5_batch_$1_$2.err || { echo "bench $cfg failed"; tail -5 gpurun_out/r5_batch_$1_$2.err; }
  python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/r5_batch_$1_$2.json").read().strip().splitlines()[-1])
    print("$cfg", d["batch"])
except Exception as e: print("$cfg ERR", e)
PY
done
