#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc CSV passes (tools/pmc_passes.sh) per kernel: mean counter value per launch.
Usage: python tools/pmc_summary.py gpurun_out/pmc_<tag>_*  > profiles/<name>.txt"""
import csv, glob, os, sys, collections

acc = collections.defaultdict(lambda: collections.defaultdict(list))
for d in sys.argv[1:]:
    for f in glob.glob(os.path.join(d, '*counter_collection.csv')):
        for row in csv.DictReader(open(f)):
            k = row['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0]
            acc[k][row['Counter_Name']].append(float(row['Counter_Value']))
counters = sorted({c for k in acc for c in acc[k]})
print('mean counter value per launch (rocprofv3 --pmc, separate passes)')
for k in sorted(acc, key=lambda k: -sum(acc[k].get('SQ_BUSY_CYCLES', [0]))):
    n = max(len(v) for v in acc[k].values())
    print(f'\n{k}  (launches seen: {n})')
    for c in counters:
        if c in acc[k]:
            v = acc[k][c]
            print(f'    {c:<28} {sum(v) / len(v):>18.1f}')
