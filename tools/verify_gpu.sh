#!/bin/bash
# final verification on the GPU box: the -m gpu suite plain and poisoned (MISSLAP_DEBUG_POISON), a sharded fuzz run.
# Every step ends the script with its own exit status (a MISMATCH of a fuzz run is a failure of the script).
#   bash tools/verify_gpu.sh [outdir under gpurun_out]
O=gpurun_out/${1:-verify}; mkdir -p "$O"
timeout -k 10 900 python -X faulthandler -m pytest tests -m gpu -x -q > $O/pytest_gpu.txt 2>&1 || { tail -30 $O/pytest_gpu.txt; exit 1; }
tail -2 $O/pytest_gpu.txt
MISSLAP_DEBUG_POISON=255 timeout -k 10 900 python -X faulthandler -m pytest tests -m gpu -x -q > $O/pytest_gpu_poisoned.txt 2>&1 || { tail -30 $O/pytest_gpu_poisoned.txt; exit 1; }
tail -2 $O/pytest_gpu_poisoned.txt
timeout -k 10 400 python3 tools/fuzz_sharded.py 60 60 > $O/fuzz_sharded_60_120.txt 2>&1 || { tail -30 $O/fuzz_sharded_60_120.txt; exit 1; }
tail -2 $O/fuzz_sharded_60_120.txt
