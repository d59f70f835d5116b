#!/bin/bash
# rocprofv3 --pmc passes on tools/pmc_scan.py (full-scan bid launches only); one counter group per pass.
# usage (on the GPU box, from the repo root): [PMC_CONFIG=C5] [PMC_ROUNDS=2] [PMC_SHUFFLE=shuffle] bash tools/pmc_passes.sh <outdir> <group>...
set -u
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/$1; shift
mkdir -p "$O"; export PYTHONPATH=$R
declare -A G
G[tcp1]="TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum"
G[tcp2]="TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum"
G[ta]="TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_TOTAL_WAVEFRONTS_sum"
G[utcl1]="TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum TCP_UTCL1_STALL_INFLIGHT_MAX_sum"
G[tcc1]="TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_LEVEL_sum TCC_TAG_STALL_sum TCC_BUSY_sum"
G[tcc2]="TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum TCC_EA0_RDREQ_DRAM_sum"
G[sq1]="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
G[sq2]="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM"
G[tcc3]="TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum"
G[fetch]="FETCH_SIZE"
for g in "$@"; do
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc ${G[$g]} -d "$O/$g" -o "$g" --output-format csv -- python3 "$R/tools/pmc_scan.py" ${PMC_CONFIG:-C3} 2 ${PMC_ROUNDS:-1} ${PMC_SHUFFLE:-} > "$O/$g.log" 2>&1 || { echo "pass $g failed"; tail -5 "$O/$g.log"; exit 1; }
done
echo done
