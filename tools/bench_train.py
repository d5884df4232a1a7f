#!/usr/bin/env python3
"""Training-step timing (BASELINE config 4 / metric ii) = `bench.py --mode train` (same flags, same JSON contract line; this file
is kept as the name earlier profiles refer to).

    python tools/bench_train.py [--gpus N] [--steps K] [--warmup W] [--clips 8] [--of2]
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench   # noqa: E402

if __name__ == '__main__':
    bench.main(['--mode', 'train'] + sys.argv[1:])
