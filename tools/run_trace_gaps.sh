#!/bin/bash
# bash tools/run_trace_gaps.sh <outdir> <config>: kernel trace of a short bench run + tools/trace_gaps.py on its last solve
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/$1; mkdir -p "$O"; cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/tg
timeout -k 10 400 rocprofv3 --kernel-trace -d /tmp/tg -o t --output-format csv -- python3 "$R/bench.py" --config "$2" --no-cpu --steps 3 --warmup 1 > "$O/gaps_bench_$2.json" 2> "$O/gaps_bench_$2.err" || { tail -5 "$O/gaps_bench_$2.err"; exit 1; }
python3 "$R/tools/trace_gaps.py" /tmp/tg/t_kernel_trace.csv > "$O/gaps_$2.json" && cat "$O/gaps_$2.json"
