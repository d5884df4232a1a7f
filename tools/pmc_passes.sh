#!/bin/bash
# Separate rocprofv3 --pmc passes over the bench workload (never combined with trace domains other than
# kernel-trace).  Usage (on the GPU box, from the repo root): bash tools/pmc_passes.sh <tag> [clips]
TAG=${1:-r01}
B=${2:-256}
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for P in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_VALU" \
         "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES"; do
  N=$(echo $P | cut -d' ' -f1)
  timeout 300 rocprofv3 --kernel-trace --pmc $P --output-format csv -d $R/gpurun_out/pmc_${TAG}_$N -o pmc -- python3 $R/tools/run_engine_once.py $B 2 > $R/gpurun_out/pmc_${TAG}_$N.log 2>&1
done
ls $R/gpurun_out/pmc_${TAG}_*/ | head -30
