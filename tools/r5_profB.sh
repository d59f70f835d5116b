for c in C5 C1 D1; do bash tools/profile_round.sh r5prof $c bench stats pmc || exit 1; done
