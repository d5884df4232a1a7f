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


def _digest(mc, precision, shape=(), **env_extra):
    env = dict(os.environ)
    for k in ('AMTX_CONVG_NO_WDMA', 'AMTX_CONVG_NO_CSPLIT', 'AMTX_OF_OVERLAP'):
        env.pop(k, None)
    env.update(env_extra)
    p = subprocess.run([sys.executable, os.path.join('tools', 'convg_mode_check.py'), str(mc), precision] + [str(v) for v in shape], cwd=ROOT, env=env,
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


@pytest.mark.timeout(900)
@pytest.mark.parametrize('mc,precision,shape', [(2, 'bf16', (40, 625)), (2, 'x3', (8, 625)), (3, 'bf16', (8, 600)), (2, 'bf16', (3, 70))])
def test_pitch_head_on_the_side_stream_returns_the_bits_of_the_one_stream_engine(mc, precision, shape):
    """Round 6: from 4096 clip-frames on, the engine enqueues the folded pitch head's GEMM on a side stream beside the recurrent heads'
    recurrence (fork after the x-projection, join before the refinement stage: csrc/ofmodel.hip).  Same kernels, same arguments: every
    logit of OnsetsFrames2 (three LogisticBanks, offsets, refined frames) must equal the one-stream engine's (AMTX_OF_OVERLAP=0), bit for bit."""
    assert _digest(mc, precision, shape) == _digest(mc, precision, shape, AMTX_OF_OVERLAP='0')
