mkdir -p gpurun_out/r5pf
for v in pf1d pf2d; do
  MISSLAP_LIB=$GRAFT_REPO_ROOT/build_ab/lib_$v.so timeout -k 10 300 python tools/diag.py C3 --tiled-ablate > gpurun_out/r5pf/diag_$v.json 2> gpurun_out/r5pf/diag_$v.err || { tail -3 gpurun_out/r5pf/diag_$v.err; exit 1; }
  python3 - gpurun_out/r5pf/diag_$v.json $v <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print(sys.argv[2], {k: v for k, v in d.items() if k.startswith("shape0")})
PY
done
bash tools/ab_bench.sh r5pf/c3 pf1 pf2 pf1 pf2 || exit 1
CFG=C2 bash tools/ab_bench.sh r5pf/c2 pf1 pf2 || exit 1
CFG=C4 bash tools/ab_bench.sh r5pf/c4 pf1 pf2 || exit 1
