#!/bin/bash
# One GPU-box call that regenerates the evidence kept under profiles/ for a round tag:
#   bash tools/profile_round.sh r01f      (from the repo root, on the GPU box; writes gpurun_out/<tag>_*)
# 1. default bench.py line (with cpu_baseline)            -> <tag>_bench.json
# 2. rocprofv3 --kernel-trace --stats of the same command -> <tag>_bench_kernel_stats.txt
# 3. rocprofv3 --pmc passes (separate runs, kernel-trace only) -> <tag>_pmc_counters.txt
# 4. secondary benches (HCQT config 3, training step, latency, host-to-host transcription) -> <tag>_secondary_benches.txt
TAG=${1:-r01x}
CLIPS=${2:-1024}   # = bench.py's default --clips
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout 600 python3 bench.py > $O/${TAG}_bench.json 2> $O/${TAG}_bench.err
cd /tmp && export TMPDIR=/tmp
rm -rf $O/${TAG}_stats
timeout 600 rocprofv3 --kernel-trace --stats -d $O/${TAG}_stats -o prof -- python3 $R/bench.py --cpu-seconds 0 > $O/${TAG}_stats.log 2>&1
DB=$(ls $O/${TAG}_stats/*.db $O/${TAG}_stats/*/*.db 2>/dev/null | head -1)
cd $R
[ -n "$DB" ] && python3 tools/rocpd_summary.py $DB > $O/${TAG}_bench_kernel_stats.txt
bash tools/pmc_passes.sh $TAG $CLIPS > $O/${TAG}_pmc.log 2>&1
python3 tools/pmc_summary.py $O/pmc_${TAG}_* > $O/${TAG}_pmc_counters.txt
{
  echo "== tools/bench_hcqt.py 512 (BASELINE config 3)"; timeout 300 python3 tools/bench_hcqt.py 512 2>&1 | grep -v amdgpu.ids | tail -3
  echo "== tools/bench_of2.py 1024 (OnsetsFrames2 as shipped: model_complexity 3, offset head, HTK mel)"; timeout 300 python3 tools/bench_of2.py 1024 2>&1 | grep -v amdgpu.ids | tail -1
  echo "== tools/bench_of2.py 512 3 | 256 4 (model_complexity 3 at 512 clips, model_complexity 4)"; timeout 300 python3 tools/bench_of2.py 512 3 2>&1 | grep -v amdgpu.ids | tail -1; timeout 300 python3 tools/bench_of2.py 256 4 2>&1 | grep -v amdgpu.ids | tail -1
  echo "== tools/bench_train.py (BASELINE config 4, one GPU)"; timeout 300 python3 tools/bench_train.py 2>&1 | grep -v amdgpu.ids | tail -1
  echo "== tools/bench_train.py --of2 (OnsetsFrames2 as shipped, one GPU)"; timeout 300 python3 tools/watchdog_run.py 250 tools/bench_train.py --of2 --steps 5 --warmup 2 2>&1 | grep -v amdgpu.ids | tail -1
  echo "== tools/latency.py"; timeout 300 python3 tools/latency.py 2>&1 | grep -v amdgpu.ids | tail -6
  echo "== tools/bench_transcribe.py (BASELINE config 5 on one GPU, host to host)"; timeout 600 python3 tools/bench_transcribe.py 2>&1 | grep -v amdgpu.ids | tail -4
  echo "== tools/bench_transcribe.py --pcm16 (the same clips as 16-bit PCM)"; timeout 600 python3 tools/bench_transcribe.py 8192 256 --pcm16 2>&1 | grep -v amdgpu.ids | tail -1
  echo "== tools/bench_gemm.py / bench_gemm_vendor.py"; timeout 200 python3 tools/bench_gemm.py 2>&1 | grep "^M=" | head -3; timeout 200 python3 tools/bench_gemm_vendor.py 2>&1 | grep hipBLASLt
} > $O/${TAG}_secondary_benches.txt 2>&1
ls -la $O | grep $TAG
