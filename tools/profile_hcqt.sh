#!/bin/bash
# BASELINE config 3 evidence: rocprofv3 --kernel-trace --stats of tools/bench_hcqt.py + two PMC passes (FETCH_SIZE, WRITE_SIZE) of the same program
#   bash tools/profile_hcqt.sh r02g [clips]      -> gpurun_out/<tag>_hcqt_kernel_stats.txt, <tag>_hcqt_pmc.txt
TAG=${1:-r02x}
B=${2:-512}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf $O/${TAG}_hcqt_stats
timeout 600 rocprofv3 --kernel-trace --stats -d $O/${TAG}_hcqt_stats -o prof -- python3 $R/tools/bench_hcqt.py $B > $O/${TAG}_hcqt_stats.log 2>&1
DB=$(ls $O/${TAG}_hcqt_stats/*.db $O/${TAG}_hcqt_stats/*/*.db 2>/dev/null | head -1)
cd $R
{ grep "clips x" $O/${TAG}_hcqt_stats.log; [ -n "$DB" ] && python3 tools/rocpd_summary.py $DB; } > $O/${TAG}_hcqt_kernel_stats.txt
rm -rf $O/${TAG}_hcqt_stats
cd /tmp
for P in FETCH_SIZE WRITE_SIZE; do
  rm -rf $O/pmc_${TAG}_hcqt_$P
  timeout 300 rocprofv3 --kernel-trace --pmc $P --output-format csv -d $O/pmc_${TAG}_hcqt_$P -o pmc -- python3 $R/tools/bench_hcqt.py $B > $O/pmc_${TAG}_hcqt_$P.log 2>&1
done
cd $R
python3 tools/pmc_summary.py $O/pmc_${TAG}_hcqt_* > $O/${TAG}_hcqt_pmc.txt 2>&1
rm -rf $O/pmc_${TAG}_hcqt_*
head -30 $O/${TAG}_hcqt_kernel_stats.txt | cut -c1-150
