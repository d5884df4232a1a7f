#!/usr/bin/env python3
"""Run a Python script with a watchdog: after N seconds every thread's traceback is dumped and the process exits.
    python tools/watchdog_run.py 90 tools/bench_train.py --steps 10
Used on the GPU box for anything that might hang (a hung command that leaves the GPU unresponsive costs a strike): it found the
intermittent two-stream hang of the training loop (stuck in torch.cuda.synchronize())."""
import faulthandler, runpy, sys
faulthandler.dump_traceback_later(int(sys.argv[1]), exit=True)
sys.argv = sys.argv[2:]
runpy.run_path(sys.argv[0], run_name='__main__')
