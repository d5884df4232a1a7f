#!/usr/bin/env python3
"""profiles/pmc_traffic.json from a PMC summary (tools/pmc_summary.py output of tools/pmc_passes.sh: rocprofv3 --pmc FETCH_SIZE and
WRITE_SIZE in separate passes): HBM bytes per launch of EVERY kernel of one inference step, so that bench.py's whole-path figure is a real
sum.  FETCH_SIZE is doubled as MI355X_MICROARCH.md prescribes for gfx950 (128-byte requests tallied at 64 bytes); both counters are in KB.
Usage: python tools/pmc_traffic.py profiles/r03c_pmc_counters.txt r03c 1024 2 > profiles/pmc_traffic.json
       (last argument: the forward passes the profiled program ran = the divisor that turns "launches seen" into launches per step)"""
import json, re, sys

path, tag, clips, passes = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4])
# kernel name (as pmc_summary.py prints it) -> the bench.py stage(s) it runs; several stages may share a kernel
STAGES = {
    'convf_kernel': ['conv_stack'],
    'convf_kernel<false>': ['conv_stack'],       # strips of 60 frames (no ninth first-conv unit): what 625-frame clips run
    'convf_kernel<true>': ['conv_stack'],
    'conv3x3_kernel<2, 1, 0, 0, true, 1>': ['conv2_pool'],
    'conv3x3_kernel<4, 1, 0, 0, false, 0>': ['conv3_pool'],
    'gemm_glds_kernel<0, 256>': ['fc1_gemm'],
    'spec_power_ring_kernel<8, 4, 8, 20, 32>': ['spec_power'],
    'gemm_glds_kernel<1, 128>': ['rec_head_gemm', 'pitch_head_gemm', 'adj_head_gemm'],
    'gemm_skinny_kernel': ['rec_head_gemm', 'pitch_head_gemm', 'adj_head_gemm'],      # round 5: N <= 128
    'gemm_pp_kernel<0>': ['rec_xproj_gemm', 'adj_xproj_gemm'],
    'bilstm4_kernel<1, 0, 0, 2>': ['rec_bilstm', 'adj_bilstm'],
}
out = {'_source': f'profiles/pmc_traffic.json@{tag} (tools/pmc_traffic.py over {path}: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of '
                  f'tools/pmc_passes.sh, {clips} clips, every kernel of a step; FETCH_SIZE doubled per MI355X_MICROARCH.md)'}
cur, vals = None, {}
blocks = []
for line in open(path):
    m = re.match(r'^(\S.*?)\s+\(launches seen: (\d+)\)', line)
    if m:
        cur = {'kernel': m.group(1), 'seen': int(m.group(2))}
        blocks.append(cur)
        continue
    m = re.match(r'^\s+(FETCH_SIZE|WRITE_SIZE)\s+([0-9.]+)', line)
    if m and cur is not None:
        cur[m.group(1)] = float(m.group(2))
for b in blocks:
    if 'FETCH_SIZE' not in b or 'WRITE_SIZE' not in b:
        continue
    per_step = b['seen'] / passes
    key = b['kernel']
    out[key] = {'clips': clips, 'stages': STAGES.get(key, []), 'launches_per_step': per_step,
                'fetch_size_kb': b['FETCH_SIZE'], 'write_size_kb': b['WRITE_SIZE'],
                'hbm_bytes_corrected': (2 * b['FETCH_SIZE'] + b['WRITE_SIZE']) * 1024.0,
                'note': 'mean per launch'}
json.dump(out, sys.stdout, indent=1)
print()
