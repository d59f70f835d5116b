echo start > gpurun_out/r5_t16.log
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -q -x -p no:cacheprovider -k "batched" >> gpurun_out/r5_t16.log 2>&1; echo rc=$? >> gpurun_out/r5_t16.log; tail -15 gpurun_out/r5_t16.log
timeout -k 10 1000 python -m pytest tests -m gpu -x -q -p no:cacheprovider > gpurun_out/r5_t16b.log 2>&1; echo rc=$? >> gpurun_out/r5_t16b.log; tail -5 gpurun_out/r5_t16b.log
