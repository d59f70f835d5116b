mkdir -p gpurun_out/r5fuzz
for first in 1200 1600 2000; do
  timeout -k 10 700 python3 tools/fuzz_gpu.py $first 400 > gpurun_out/r5fuzz/fuzz_gpu_${first}.txt 2>&1; tail -1 gpurun_out/r5fuzz/fuzz_gpu_${first}.txt
done
timeout -k 10 400 python3 tools/fuzz_batch.py 400 200 > gpurun_out/r5fuzz/fuzz_batch_400_600.txt 2>&1; tail -1 gpurun_out/r5fuzz/fuzz_batch_400_600.txt
timeout -k 10 400 python3 tools/fuzz_sharded.py 120 180 > gpurun_out/r5fuzz/fuzz_sharded_120_300.txt 2>&1; tail -1 gpurun_out/r5fuzz/fuzz_sharded_120_300.txt
