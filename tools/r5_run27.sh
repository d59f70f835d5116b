mkdir -p gpurun_out/r5rev gpurun_out/r5fuzz
timeout -k 10 300 python -X faulthandler -m pytest tests -m gpu -x -q -k "tail_launches_are_bounded or long_row or dense" > gpurun_out/r5rev/pytest.txt 2>&1; tail -3 gpurun_out/r5rev/pytest.txt
MISSLAP_LIB=sslap_amd/libmisslap_diag.so timeout -k 10 300 python tools/diag.py C3 --tiled-ablate > gpurun_out/r5rev/ablate.json 2> gpurun_out/r5rev/ablate.err; grep -E "shape0_(hot|cold)_complete" gpurun_out/r5rev/ablate.json
timeout -k 10 500 python3 tools/fuzz_gpu.py 800 400 > gpurun_out/r5fuzz/fuzz_gpu_800_1200.txt 2>&1; tail -1 gpurun_out/r5fuzz/fuzz_gpu_800_1200.txt
