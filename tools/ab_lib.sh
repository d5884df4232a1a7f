for i in 1 2 3; do
for L in new head; do
  if [ $L = head ]; then export AMTX_LIB_PATH=$PWD/tools/_dbg/libamtx_convxhead.so; else unset AMTX_LIB_PATH; fi
  python bench.py --precision x3 --clips 512 --steps 4 --warmup 2 --no-parity --cpu-seconds 0 --no-train-probe --no-hcqt 2>/dev/null | python -c "
import sys, json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$L conv2_pool', d['roofline']['kernel_ms_per_step'].get('conv2_pool'), 'conv3', d['roofline']['kernel_ms_per_step'].get('conv3_pool'), 'step', round(d['ms_per_step'],2))"
done; done
