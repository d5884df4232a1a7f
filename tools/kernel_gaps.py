#!/usr/bin/env python3
"""Inter-kernel gaps from a rocprofv3 --kernel-trace CSV: python tools/kernel_gaps.py <kernel_trace.csv>"""
import csv, sys, collections
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r['Start_Timestamp']))
gaps = collections.defaultdict(list)
prev = None
for r in rows:
    name = r['Kernel_Name'].replace('void ', '').replace('(anonymous namespace)::', '').split('(')[0].split('<')[0][:40]
    if prev is not None:
        gaps[(prev[0], name)].append((int(r['Start_Timestamp']) - prev[1]) / 1e3)
    prev = (name, int(r['End_Timestamp']))
tot = 0
for k, v in sorted(gaps.items(), key=lambda kv: -sum(kv[1])):
    if len(v) < 5:
        continue
    v2 = sorted(v)
    print(f'{k[0]:<42} -> {k[1]:<42} n={len(v):4d} median {v2[len(v)//2]:8.1f} us  mean {sum(v)/len(v):8.1f} us  max {v2[-1]:9.1f}')
