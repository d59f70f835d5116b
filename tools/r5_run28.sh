mkdir -p gpurun_out/r5rev
timeout -k 10 600 python -X faulthandler -m pytest tests -m gpu -x -q -k "tiled or engine or repeated or staged or ece or fuzz or golden or config or baseline or batch or sharded" > gpurun_out/r5rev/pytest2.txt 2>&1 || { tail -30 gpurun_out/r5rev/pytest2.txt; exit 1; }
tail -2 gpurun_out/r5rev/pytest2.txt
for cfg in C3 C2 C4; do
 for v in 0 1 0 1; do
  MISSLAP_TILED_ALTERNATE=$v timeout -k 10 300 python3 bench.py --no-cpu --steps 3 --config $cfg > gpurun_out/r5rev/bench_${cfg}_$v.json 2> gpurun_out/r5rev/bench_${cfg}_$v.err || { tail -3 gpurun_out/r5rev/bench_${cfg}_$v.err; exit 1; }
  python3 - gpurun_out/r5rev/bench_${cfg}_$v.json $cfg alternate=$v <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
f = d['bid_phase']['fullscan_avg_us']; a = d['roofline']['avg_launch_us']; n = d['roofline']['launches']; nf = d['bid_phase']['fullscan_launches']
part = (n * a - nf * f) / max(n - nf, 1)
print(sys.argv[2], sys.argv[3], 'ms/step', d['ms_per_step'], 'full', f, 'all', a, 'partial %.1f' % part, 'frac', d['roofline']['frac'], d['sol_sha256'][:8])
PY
 done
done
