mkdir -p gpurun_out/r5fuzz
timeout -k 10 500 python3 tools/fuzz_gpu.py 800 400 > gpurun_out/r5fuzz/fuzz_gpu_800_1200.txt 2>&1; tail -1 gpurun_out/r5fuzz/fuzz_gpu_800_1200.txt
timeout -k 10 300 python3 tools/fuzz_batch.py 400 150 > gpurun_out/r5fuzz/fuzz_batch_400_550.txt 2>&1; tail -1 gpurun_out/r5fuzz/fuzz_batch_400_550.txt
timeout -k 10 300 python3 tools/fuzz_sharded.py 120 120 > gpurun_out/r5fuzz/fuzz_sharded_120_240.txt 2>&1; tail -1 gpurun_out/r5fuzz/fuzz_sharded_120_240.txt
