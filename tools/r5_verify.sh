# final verification on the GPU box: the -m gpu suite plain and poisoned (MISSLAP_DEBUG_POISON), a sharded fuzz run
mkdir -p gpurun_out/r5final
timeout -k 10 900 python -X faulthandler -m pytest tests -m gpu -x -q > gpurun_out/r5final/pytest_gpu.txt 2>&1 || { tail -30 gpurun_out/r5final/pytest_gpu.txt; exit 1; }
tail -2 gpurun_out/r5final/pytest_gpu.txt
MISSLAP_DEBUG_POISON=255 timeout -k 10 900 python -X faulthandler -m pytest tests -m gpu -x -q > gpurun_out/r5final/pytest_gpu_poisoned.txt 2>&1 || { tail -30 gpurun_out/r5final/pytest_gpu_poisoned.txt; exit 1; }
tail -2 gpurun_out/r5final/pytest_gpu_poisoned.txt
timeout -k 10 400 python3 tools/fuzz_sharded.py 60 60 > gpurun_out/r5final/fuzz_sharded_60_120.txt 2>&1; tail -2 gpurun_out/r5final/fuzz_sharded_60_120.txt
