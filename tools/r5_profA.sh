for c in C3 C2 C4; do bash tools/profile_round.sh r5prof $c bench stats pmc || exit 1; done
ls gpurun_out/r5prof
