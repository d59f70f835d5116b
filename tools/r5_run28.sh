mkdir -p gpurun_out/r5rev
timeout -k 10 600 python -X faulthandler -m pytest tests -m gpu -x -q -k "ece or final or config or golden or baseline or staged or tiled" > gpurun_out/r5rev/pytest2.txt 2>&1 || { tail -20 gpurun_out/r5rev/pytest2.txt; exit 1; }
tail -2 gpurun_out/r5rev/pytest2.txt
for cfg in C3 C2 C4; do
 for v in 0 1 0 1; do
  MISSLAP_TILED_ALTERNATE=$v timeout -k 10 300 python3 bench.py --no-cpu --steps 3 --config $cfg > gpurun_out/r5rev/bench_${cfg}_$v.json 2> gpurun_out/r5rev/bench_${cfg}_$v.err || { tail -3 gpurun_out/r5rev/bench_${cfg}_$v.err; exit 1; }
  python3 - gpurun_out/r5rev/bench_${cfg}_$v.json $cfg alternate=$v <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
f = d['bid_phase']['fullscan_avg_us']; a = d['roofline']['avg_launch_us']
print(sys.argv[2], sys.argv[3], 'ms/step', d['ms_per_step'], 'full', f, 'all', a, 'frac', d['roofline']['frac'], d['sol_sha256'][:8])
PY
 done
done
