#!/bin/bash
# Counter evidence for the shipping full-scan engine (GPU box): C3 (4 lanes, format 0: forward K = N scan + backward
# partial round), C2 (16 lanes), C2 with shuffled rows (format 2) -> gpurun_out/<outdir>/pmc_counters_<cfg>.txt
#   bash tools/pmc_engine.sh <outdir under gpurun_out>
R=${GRAFT_REPO_ROOT:-$(pwd)}; D=gpurun_out/$1; mkdir -p "$R/$D"
GROUPS_="sq1 tcp1 tcc3 tcc1 tcc2"
PMC_CONFIG=C3 PMC_ROUNDS=2 bash "$R/tools/pmc_passes.sh" $D/C3 $GROUPS_ || exit 1
python3 "$R/tools/pmc_show.py" "$R/$D/C3" k_bid_tiled > "$R/$D/pmc_counters_C3_tiled.txt"
PMC_CONFIG=C2 PMC_ROUNDS=2 bash "$R/tools/pmc_passes.sh" $D/C2 $GROUPS_ || exit 1
python3 "$R/tools/pmc_show.py" "$R/$D/C2" k_bid_tiled > "$R/$D/pmc_counters_C2_tiled.txt"
PMC_CONFIG=C2 PMC_ROUNDS=2 PMC_SHUFFLE=shuffle bash "$R/tools/pmc_passes.sh" $D/C2s $GROUPS_ || exit 1
python3 "$R/tools/pmc_show.py" "$R/$D/C2s" k_bid_tiled > "$R/$D/pmc_counters_C2_shuffled_tiled.txt"
rm -rf "$R/$D/C3" "$R/$D/C2" "$R/$D/C2s"
tail -n +1 "$R/$D"/pmc_counters_*.txt | head -150
