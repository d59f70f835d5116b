#!/bin/bash
# Kernel trace of bench.py --concurrent B on the GPU box + tools/trace_concurrency.py: bash tools/run_conc_trace.sh <outdir> <config> <B>
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/$1; mkdir -p "$O"; cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/tr
timeout -k 10 500 rocprofv3 --kernel-trace -d /tmp/tr -o t --output-format csv -- python3 "$R/bench.py" --config "$2" --no-cpu --steps 4 --warmup 1 --concurrent "$3" > "$O/conc_trace.json" 2> "$O/conc_trace.err" || { tail -5 "$O/conc_trace.err"; exit 1; }
python3 "$R/tools/trace_concurrency.py" "/tmp/tr/t_kernel_trace.csv" > "$O/conc_trace_summary_$2_B$3.json" && cat "$O/conc_trace_summary_$2_B$3.json"
