mkdir -p gpurun_out/r5final
timeout -k 10 600 python -X faulthandler -m pytest tests -m gpu -x -q > gpurun_out/r5final/pytest_gpu.txt 2>&1 || { tail -30 gpurun_out/r5final/pytest_gpu.txt; exit 1; }
tail -2 gpurun_out/r5final/pytest_gpu.txt
MISSLAP_DEBUG_POISON=255 timeout -k 10 300 python -X faulthandler -m pytest tests -m gpu -x -q -k "batch or poison or formats or c_client" > gpurun_out/r5final/pytest_gpu_poisoned.txt 2>&1 || { tail -30 gpurun_out/r5final/pytest_gpu_poisoned.txt; exit 1; }
tail -2 gpurun_out/r5final/pytest_gpu_poisoned.txt
timeout -k 10 500 python3 tools/fuzz_gpu.py 400 800 > gpurun_out/r5final/fuzz_400_800.txt 2>&1; tail -2 gpurun_out/r5final/fuzz_400_800.txt
