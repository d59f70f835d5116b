mkdir -p gpurun_out/r5ab
for i in 1 2; do
  bash tools/ab_bench.sh r5ab/c3_$i poskey keycol || exit 1
done
CFG=C2 bash tools/ab_bench.sh r5ab/c2 poskey keycol || exit 1
CFG=C4 bash tools/ab_bench.sh r5ab/c4 poskey keycol || exit 1
timeout -k 10 600 python -X faulthandler -m pytest tests -m gpu -x -q -k "tiled or engine or formats or repeated or staged or ece or fuzz or golden or config" > gpurun_out/r5ab/pytest.txt 2>&1; tail -3 gpurun_out/r5ab/pytest.txt
