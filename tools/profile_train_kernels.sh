#!/bin/bash
# rocprofv3 --kernel-trace --stats of the training-step bench -> gpurun_out/<tag>_train_kernel_stats.txt (copy under profiles/ to keep)
#   bash tools/profile_train_kernels.sh r02a [extra bench.py flags]
TAG=${1:-r02x}; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf $O/${TAG}_train_stats
timeout 600 rocprofv3 --kernel-trace --stats -d $O/${TAG}_train_stats -o prof -- python3 $R/bench.py --mode train --cpu-seconds 0 --steps 10 --warmup 3 "$@" > $O/${TAG}_train_stats.log 2>&1
DB=$(ls $O/${TAG}_train_stats/*.db $O/${TAG}_train_stats/*/*.db 2>/dev/null | head -1)
cd $R
[ -n "$DB" ] && python3 tools/rocpd_summary.py $DB > $O/${TAG}_train_kernel_stats.txt
tail -1 $O/${TAG}_train_stats.log | cut -c1-160
head -45 $O/${TAG}_train_kernel_stats.txt | cut -c1-150
rm -rf $O/${TAG}_train_stats
