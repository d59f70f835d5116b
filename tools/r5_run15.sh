for st in 0 3000 8000; do for grp in 32 16; do
  MISSLAP_BATCH_STAGGER_US=$st timeout -k 10 500 python bench.py --config C2 --steps 2 --warmup 1 --no-cpu --batch 64 --batch-group $grp > gpurun_out/r5_bst.json 2> gpurun_out/r5_bst.err || { echo "failed"; tail -5 gpurun_out/r5_bst.err; }
  python - <<PY
import json
d=json.loads(open("gpurun_out/r5_bst.json").read().strip().splitlines()[-1])
b=d["batch"]; print("stagger $st group $grp", {k:b[k] for k in ("groups","wall_ms","ms_per_solve","throughput_vs_single_solve")})
PY
done; done
