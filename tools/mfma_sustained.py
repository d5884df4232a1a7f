#!/usr/bin/env python3
"""Build and run tools/mfma_sustained.hip on the GPU of this box; prints its JSON line with the summary bench.py reads
(`sustained_bf16_tflops_random` = the best sustained rate on random operands) and writes it to gpurun_out/mfma_sustained.json.
The committed copy is profiles/mfma_sustained.json."""
import json
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    exe = os.path.join(tempfile.gettempdir(), 'mfma_sustained')
    subprocess.run(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', os.path.join(ROOT, 'tools', 'mfma_sustained.hip'), '-o', exe], check=True)
    if '--build-only' in sys.argv:
        return
    out = subprocess.run([exe] + [a for a in sys.argv[1:] if not a.startswith('--')], check=True, stdout=subprocess.PIPE, text=True).stdout
    rec = json.loads(out.strip().splitlines()[-1])
    rnd = [r for r in rec['runs'] if r['data'] == 'random']
    zer = [r for r in rec['runs'] if r['data'] == 'zeros']
    rec['sustained_bf16_tflops_random'] = max(r['tflops'] for r in rnd)
    rec['sustained_bf16_tflops_zeros'] = max(r['tflops'] for r in zer)
    rec['nominal_dense_bf16_tflops'] = 2500.0
    rec['source'] = 'tools/mfma_sustained.py (register-resident v_mfma stream, ~0.3 s per launch, HIP events)'
    os.makedirs(os.path.join(ROOT, 'gpurun_out'), exist_ok=True)
    with open(os.path.join(ROOT, 'gpurun_out', 'mfma_sustained.json'), 'w') as f:
        json.dump(rec, f, indent=1)
    print(json.dumps(rec))


if __name__ == '__main__':
    main()
