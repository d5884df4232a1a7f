R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_VALU --output-format csv -d $R/gpurun_out/pmc_r01f_LDS -o pmc -- python3 $R/tools/run_engine_once.py 512 2 > $R/gpurun_out/pmc_r01f_LDS.log 2>&1
cd $R && python tools/pmc_summary.py gpurun_out/pmc_r01f_LDS | head -45
