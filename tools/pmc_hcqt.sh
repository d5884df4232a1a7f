#!/bin/bash
# SQ counters of BASELINE config 3 (tools/bench_hcqt.py): bash tools/pmc_hcqt.sh [clips=512] [precision=bf16]  -> gpurun_out/hcqt_sq_pmc.txt
B=${1:-512}
PREC=${2:-bf16}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
for P in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_VALU" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES"; do
  N=$(echo $P | cut -d' ' -f1)
  rm -rf $O/pmc_hq_$N
  timeout 300 rocprofv3 --kernel-trace --pmc $P --output-format csv -d $O/pmc_hq_$N -o pmc -- python3 $R/tools/bench_hcqt.py $B $PREC > $O/pmc_hq_$N.log 2>&1
done
cd $R
python3 tools/pmc_summary.py $O/pmc_hq_* > $O/hcqt_sq_pmc.txt 2>&1
rm -rf $O/pmc_hq_*
grep -A17 "^cqt_basis_kernel\|^cqt_decimate2\|^conv3x3_gen_kernel\|convx12_kernel" $O/hcqt_sq_pmc.txt | head -140
