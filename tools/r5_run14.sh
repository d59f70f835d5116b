for cfg in "C2 64 64" "C2 64 16"; do
  set -- $cfg
  timeout -k 10 500 python bench.py --config $1 --steps 2 --warmup 1 --no-cpu --batch $2 --batch-group $3 > gpurun_out/r5_batch_$1_$2_$3.json 2> gpurun_out/r5_batch_$1_$2_$3.err || { echo "bench $cfg failed"; tail -5 gpurun_out/r5_batch_$1_$2_$3.err; }
  python - <<PY
import json
d=json.loads(open("gpurun_out/r5_batch_$1_$2_$3.json").read().strip().splitlines()[-1])
b=d["batch"]; print("$cfg", {k:b[k] for k in ("groups","wall_ms","ms_per_solve","launches_issued","throughput_vs_single_solve","host_ms_summed_over_groups")})
PY
done
cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r5_batch_prof -o s --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --config C2 --steps 1 --warmup 0 --no-cpu --batch 64 --batch-group 64 > /dev/null 2>&1; head -30 $GRAFT_REPO_ROOT/gpurun_out/r5_batch_prof/*kernel_stats.csv | cut -c1-230
