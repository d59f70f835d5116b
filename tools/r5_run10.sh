echo start > gpurun_out/r5_t10.log
timeout -k 10 1000 python -m pytest tests -m gpu -x -q -p no:cacheprovider >> gpurun_out/r5_t10.log 2>&1; echo rc=$? >> gpurun_out/r5_t10.log; tail -6 gpurun_out/r5_t10.log
echo start > gpurun_out/r5_t10p.log
MISSLAP_DEBUG_POISON=0xFF timeout -k 10 1000 python -m pytest tests/test_gpu_parity.py tests/test_matching.py -m gpu -x -q -p no:cacheprovider >> gpurun_out/r5_t10p.log 2>&1; echo rc=$? >> gpurun_out/r5_t10p.log; tail -6 gpurun_out/r5_t10p.log
