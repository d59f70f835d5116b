#!/bin/bash
# Stamped cycles of the SHIPPING tail instances (diagnostic builds: s_memtime around the segments of a round, wavefront 0):
#   bash tools/tail_stamps.sh build      (build container: build_ab/lib_stamp_{chain,duo,team,block}.so)
#   bash tools/tail_stamps.sh run [cfg]  (GPU box: one solve per build, cycles per round and segment)
R=$(cd "$(dirname "$0")/.." && pwd); cd "$R"
if [ "$1" = build ]; then
  bash tools/build_ab.sh stamp_chain "-DMISSLAP_TAIL_STAMP" stamp_duo "-DMISSLAP_TAIL_STAMP_DUO" stamp_team "-DMISSLAP_TAIL_STAMP_TEAM" stamp_block "-DMISSLAP_TAIL_STAMP_BLOCK"
  exit $?
fi
cfg=${2:-C3}; mkdir -p gpurun_out/stamps
for m in chain duo team block; do
  MISSLAP_LIB=$PWD/build_ab/lib_stamp_$m.so timeout -k 10 300 python tools/tail_stats.py $cfg 1 > gpurun_out/stamps/${cfg}_$m.json 2> gpurun_out/stamps/${cfg}_$m.err || { echo "stamp run $m failed"; tail -3 gpurun_out/stamps/${cfg}_$m.err; exit 1; }
  python3 - gpurun_out/stamps/${cfg}_$m.json $m $cfg <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r = d["tail_raw"]; m = sys.argv[2]
tm = d["tail_modes"]
names = {"chain": ["wait for the line", "record gather", "winner known + next line requested", "rest of the evaluation", "full scan of a missed person", "store / re-request"],
         "duo": ["evaluation up to the bid", "publish + wait for the next line + next gather issued", "barrier (= the other wavefront)", "exchange / patch of the records gathered ahead", "re-request", "-"],
         "team": ["wait for the line + record gather", "evaluation of my slot up to the bid", "publish", "barrier", "LDS reads landed", "dirty word / wait for the next line / stores / next gather issued / loop"],
         "block": ["lines landed", "records landed", "evaluation + barrier", "scan pass + barrier", "resolve / assign / compaction (wavefront 0)", "closing barrier"]}[m]
# rounds the stamped code ran: chain = K = 1 rounds are not counted separately by the kernel: use the simulator's histogram share
print(f"{sys.argv[3]} {m}: solve {d['solve_ms']} ms, sha {d['sol_sha256']}, tail modes {json.dumps(tm)}")
print("   cycles (wavefront 0, summed over the solve): " + "; ".join(f"{n}: {int(x)}" for n, x in zip(names, r) if n != "-"), "| total", int(sum(r)))
PY
done
