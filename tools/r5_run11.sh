timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -q -p no:cacheprovider -k "f32_filter or no_kernel_depends" 2>&1 | tail -3
for f in -1 0; do
  MISSLAP_F32_FILTER=$f timeout -k 10 500 python bench.py --config C5 --steps 2 --warmup 1 --no-cpu > gpurun_out/r5_c5_f$f.json 2> gpurun_out/r5_c5_f$f.err || echo fail
  python - <<PY
import json
d=json.loads(open("gpurun_out/r5_c5_f$f.json").read().strip().splitlines()[-1])
print("filter $f", "ms/step", d["ms_per_step"], "fs_us", d["bid_phase"]["fullscan_avg_us"], "k_bid_timed", d["bid_phase"]["k_bid_timed"], d["sol_sha256"][:12])
PY
done
