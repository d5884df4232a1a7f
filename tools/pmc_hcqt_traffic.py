#!/usr/bin/env python3
"""profiles/pmc_traffic_hcqt.json from a PMC summary of BASELINE config 3 (tools/pmc_hcqt.sh / tools/pmc_summary.py over rocprofv3 --pmc
FETCH_SIZE and WRITE_SIZE passes of tools/bench_hcqt.py): HBM bytes per front-end pass of the HCQT kernels, for bench.py's
`config.hcqt.roofline.frontend.traffic`.
FETCH_SIZE is taken as counted (factor 1) for these kernels: MI355X_MICROARCH.md's x 2 is calibrated for 16-byte-per-lane streaming reads only;
the decimator reads by `global_load_lds_dword`, and `cqt_scale16_kernel` is the calibration point -- it reads the fp32 power map exactly once,
clips x frames x 432 x 4 bytes = 553 MB at 512 clips, and the counter says 530 MB.
Usage: python tools/pmc_hcqt_traffic.py profiles/r05ze_hcqt_pmc.txt r05ze 512 10 > profiles/pmc_traffic_hcqt.json
       (last argument: front-end passes the profiled program ran = launches seen of cqt_basis_kernel)"""
import json, re, sys

path, tag, clips, passes = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4])
out = {'_source': f'profiles/pmc_traffic_hcqt.json@{tag} (tools/pmc_hcqt_traffic.py over {path}: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of '
                  f'tools/bench_hcqt.py {clips}; FETCH_SIZE as counted, calibrated on cqt_scale16_kernel\'s known input)', 'clips': clips}
cur, blocks = None, []
for line in open(path):
    m = re.match(r'^(\S.*?)\s+\(launches seen: (\d+)\)', line)
    if m:
        cur = {'kernel': m.group(1), 'seen': int(m.group(2))}
        blocks.append(cur)
        continue
    m = re.match(r'^\s+(FETCH_SIZE|WRITE_SIZE)\s+([0-9.]+)', line)
    if m and cur is not None:
        cur[m.group(1)] = float(m.group(2))
total = 0.0
for b in blocks:
    if not b['kernel'].startswith('cqt_') or 'FETCH_SIZE' not in b or 'WRITE_SIZE' not in b:
        continue
    per_pass = b['seen'] / passes
    nbytes = (b['FETCH_SIZE'] + b['WRITE_SIZE']) * 1024.0
    out[b['kernel']] = {'launches_per_pass': per_pass, 'fetch_size_kb': b['FETCH_SIZE'], 'write_size_kb': b['WRITE_SIZE'], 'hbm_bytes_per_launch': nbytes,
                        'hbm_bytes_per_pass': nbytes * per_pass}
    total += nbytes * per_pass
out['frontend_hbm_bytes_per_pass'] = total
json.dump(out, sys.stdout, indent=1)
print()
