# diagnostic: epilogue / loop cycles of k_bid_tiled (wavefront 0 of every workgroup), partial rounds (lib_estamp3: -DMISSLAP_TILED_STAMP=3)
# and full scans (lib_estamp4: =4), for the launch shapes named on the command line (default 0)
mkdir -p gpurun_out/estamp
for v in 3 4; do for sh in ${@:-0}; do
  MISSLAP_LIB=$PWD/build_ab/lib_estamp$v.so timeout -k 10 200 python tools/tail_stats.py C3 1 tiled_shape=$sh > gpurun_out/estamp/s${v}_shape$sh.json 2> gpurun_out/estamp/s${v}_shape$sh.err || { echo fail; exit 1; }
  python3 - gpurun_out/estamp/s${v}_shape$sh.json $v $sh <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r = d['tail_raw']; w = r[5]
print({'3': 'partial', '4': 'full'}[sys.argv[2]], 'shape', sys.argv[3], 'wgs', w, 'per wavefront 0: loop %.0f ovf %.0f handoff %.0f merge %.0f finish %.0f cycles' % tuple(x / w for x in r[:5]), 'sum %.0f' % (sum(r[:5]) / w))
PY
done; done
