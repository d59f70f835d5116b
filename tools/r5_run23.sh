mkdir -p gpurun_out/r5p32
timeout -k 10 900 python -X faulthandler -m pytest tests -m gpu -x -q -k "fp32_tile" > gpurun_out/r5p32/pytest.txt 2>&1; rc=$?; tail -25 gpurun_out/r5p32/pytest.txt; exit $rc
