R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
for P in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_VALU" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES"; do
  N=$(echo $P | cut -d' ' -f1)
  rm -rf $O/pmc_of2_$N
  timeout 300 rocprofv3 --kernel-trace --pmc $P --output-format csv -d $O/pmc_of2_$N -o pmc -- python3 $R/tools/bench_of2.py 512 3 > $O/pmc_of2_$N.log 2>&1
done
cd $R
python3 tools/pmc_summary.py $O/pmc_of2_* > $O/r04s_of2_pmc.txt 2>&1
rm -rf $O/pmc_of2_*
grep -A18 "conv3x3_gen" $O/r04s_of2_pmc.txt | head -80
