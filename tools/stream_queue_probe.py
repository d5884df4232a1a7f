#!/usr/bin/env python3
"""Which freshly created HIP streams run BESIDE the caller's (null) stream, and which share its hardware queue?  Four probes per candidate stream
(time of the work on both streams at once / time on one): a chain of 40 small GEMMs, `torch.sort` x 4, `cumsum` x 10, and a chain of 20 short
spin kernels (`torch.cuda._sleep(20000)`) -- the probe amt_tools_amd/models.py `_pick_side_stream` uses: ~1.4 when the two overlap, ~1.9 when they
do not.  With `--force` the process first creates a one-rank RCCL communicator (bench.py --force-dist): its streams shift the creation order and
the first (and fifth) candidate land on the null stream's queue (GPU_MAX_HW_QUEUES = 4) -- the collision that cost the data-parallel training step
2 ms in rounds 4 - 5.  Usage: python tools/stream_queue_probe.py [--force]   (prints to stderr)"""
import os, sys, time, torch
sys.path.insert(0, os.getcwd())
import bench
force = '--force' in sys.argv
args = bench.parse(['--mode', 'train'] + (['--force-dist'] if force else []) + ['--cpu-seconds', '0'])
rank, world, device = bench.init_ranks(args)
dev = torch.device(device)
main = torch.cuda.current_stream(dev)
def P(*a): print(*a, file=sys.stderr)
a = torch.randn(256, 256, device=dev)
v = torch.randn(1 << 17, device=dev)
w = torch.randn(1 << 20, device=dev)
def work_mm():
    x = a
    for _ in range(40): x = x @ a * 0.01
def work_sort():
    for _ in range(4): v.sort()
def work_cumsum():
    for _ in range(10): w.cumsum(0)
def work_sleep20():
    for _ in range(20): torch.cuda._sleep(20000)
def test(work, s1, s2):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    with torch.cuda.stream(s1): work()
    torch.cuda.synchronize(); one = time.perf_counter() - t0
    t0 = time.perf_counter()
    with torch.cuda.stream(s1): work()
    with torch.cuda.stream(s2): work()
    torch.cuda.synchronize(); two = time.perf_counter() - t0
    return one * 1e3, two / one
streams = [torch.cuda.Stream(device=dev) for _ in range(6)]
for w_, name in ((work_mm, 'mm'), (work_sort, 'sort'), (work_cumsum, 'cumsum'), (work_sleep20, 'sleep20')):
    test(w_, main, streams[0]); test(w_, main, streams[0])
    res = [test(w_, main, s) for s in streams]
    P(name, 'one %.3f ms; ratio main+s_i:' % res[0][0], ' '.join('%.2f' % r[1] for r in res))
