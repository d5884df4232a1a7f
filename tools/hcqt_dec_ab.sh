# CQT tests + HCQT (BASELINE config 3) bench with the persistent decimator and with round 4's (AMTX_CQT_DECIM_V1=1) + rocprofv3 kernel stats -> gpurun_out/dec2_*
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_cqt.py -x -q -m gpu 2>&1 | tail -8 > gpurun_out/dec2_tests.txt
python tools/bench_hcqt.py 512 > gpurun_out/dec2_bench.txt 2>&1
AMTX_CQT_DECIM_V1=1 python tools/bench_hcqt.py 512 > gpurun_out/dec2_bench_v1.txt 2>&1
cd /tmp && export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out
rm -rf $O/dec2_stats
timeout 600 rocprofv3 --kernel-trace --stats -d $O/dec2_stats -o prof -- python3 $GRAFT_REPO_ROOT/tools/bench_hcqt.py 512 > $O/dec2_stats.log 2>&1
DB=$(ls $O/dec2_stats/*.db $O/dec2_stats/*/*.db 2>/dev/null | head -1)
cd $GRAFT_REPO_ROOT
python3 tools/rocpd_summary.py $DB > $O/dec2_kernel_stats.txt
rm -rf $O/dec2_stats
