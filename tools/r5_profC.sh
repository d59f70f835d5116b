EXTRA="--values f64" TAG=_f64 bash tools/profile_round.sh r5prof C3 bench stats pmc || exit 1
EXTRA="--values f64 --no-cpu" TAG=_f64 bash tools/profile_round.sh r5prof C2 bench pmc || exit 1
EXTRA="--values f64 --no-cpu" TAG=_f64 bash tools/profile_round.sh r5prof C4 bench pmc || exit 1
EXTRA="--shuffle-rows --no-cpu" TAG=_shuffled bash tools/profile_round.sh r5prof C2 bench pmc || exit 1
EXTRA="--values f32-as-f64 --no-cpu" TAG=_f32asf64 bash tools/profile_round.sh r5prof C3 bench || exit 1
timeout -k 10 600 python3 tools/fuzz_gpu.py 0 400 > gpurun_out/r5prof/fuzz_0_400.txt 2>&1; tail -3 gpurun_out/r5prof/fuzz_0_400.txt
