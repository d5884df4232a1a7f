"""convg.hip keeps the weight chunks of a layer in LDS in one of four ways (all resident; two DMA buffers; one chunk per block; one buffer
refilled through registers).  The choice follows the LDS budget and must not change a bit of the result: the same model in child processes
with the A/B switches set (they are read once per process)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _digest(mc, precision, **env_extra):
    env = dict(os.environ)
    for k in ('AMTX_CONVG_NO_WDMA', 'AMTX_CONVG_NO_CSPLIT'):
        env.pop(k, None)
    env.update(env_extra)
    p = subprocess.run([sys.executable, os.path.join('tools', 'convg_mode_check.py'), str(mc), precision], cwd=ROOT, env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    lines = [l for l in p.stdout.splitlines() if l.startswith('{')]
    assert p.returncode == 0 and lines, p.stderr[-3000:]
    rec = json.loads(lines[-1])
    assert rec['finite'], rec
    return rec['sha256']


@pytest.mark.timeout(900)
@pytest.mark.parametrize('mc,precision,switch', [(4, 'bf16', 'AMTX_CONVG_NO_WDMA'), (5, 'bf16', 'AMTX_CONVG_NO_WDMA'), (4, 'f16', 'AMTX_CONVG_NO_WDMA'),
                                                 (3, 'x3', 'AMTX_CONVG_NO_CSPLIT'), (4, 'x3', 'AMTX_CONVG_NO_CSPLIT')])
def test_weight_chunk_modes_of_the_general_conv_kernel_return_the_same_bits(mc, precision, switch):
    assert _digest(mc, precision) == _digest(mc, precision, **{switch: '1'})
