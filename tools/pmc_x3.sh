#!/bin/bash
# PMC counters of the two-plane (x3) engine on the bench workload: bash tools/pmc_x3.sh [clips=256]  -> gpurun_out/x3_pmc.txt
B=${1:-256}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
for P in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_VALU" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES"; do
  N=$(echo $P | cut -d' ' -f1)
  rm -rf $O/pmc_x3_$N
  timeout 300 rocprofv3 --kernel-trace --pmc $P --output-format csv -d $O/pmc_x3_$N -o pmc -- python3 $R/bench.py --precision x3 --clips $B --steps 2 --warmup 1 --cpu-seconds 0 --no-parity --no-train-probe --no-hcqt > $O/pmc_x3_$N.log 2>&1
done
cd $R
python3 tools/pmc_summary.py $O/pmc_x3_* > $O/x3_pmc.txt 2>&1
rm -rf $O/pmc_x3_*
grep -A17 "^gemm_kernel<1, 1, 2>\|^conv3x3_kernel<2, 2" $O/x3_pmc.txt | head -80
