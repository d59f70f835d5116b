set -o pipefail
echo start > gpurun_out/r5_t5.log
timeout -k 10 500 python -m pytest "tests/test_gpu_parity.py::test_hip_runtime_is_shared_in_either_import_order" -x -q >> gpurun_out/r5_t5.log 2>&1; echo rc=$? >> gpurun_out/r5_t5.log; tail -30 gpurun_out/r5_t5.log
