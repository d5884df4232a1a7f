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
    for k in ('AMTX_CONVG_NO_WDMA', 'AMTX_CONVG_NO_CSPLIT', 'AMTX_CONVF_NO_VIRT'):
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
@pytest.mark.parametrize('shape', [(40, 625), (130, 67), (3, 4001), (64, 200)])
def test_fused_conv_stack_over_the_virtual_row_stream_returns_the_bits_of_strips_per_clip(shape):
    """Round 6: `convf_kernel<false, true>` deals strips over ONE stream of B (T + 1) virtual rows (a zero row between clips) instead of
    ceil(T / 60) strips per clip -- 625-frame clips need 5 % fewer strips.  Every output keeps its accumulation order: all logits of
    OnsetsFrames2 (three heads through the fused stack) must equal the per-clip kernel's (AMTX_CONVF_NO_VIRT=1), bit for bit: clips that end
    inside a strip, a strip that starts on the padding row, T = 67 (the smallest frame count of the mode), whole tracks."""
    assert _digest(2, 'bf16', shape) == _digest(2, 'bf16', shape, AMTX_CONVF_NO_VIRT='1')
