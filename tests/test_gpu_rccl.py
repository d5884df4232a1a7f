"""RCCL on the one-GPU lease (BASELINE metric ii): a single-rank `nccl` process group with the flat gradient
all-reduce of amt_tools_amd.dp.DataParallelOptimizer forced on, running beside the persistent HIP BiLSTM autograd kernels for
300 steps of the config-4 training step (amt_tools/train.py:126-141), in a FRESH child process under a wall-clock limit: a hang
is a killed child and a failed test.  A one-rank sum returns the gradients' own bits, so the weights after the run must equal the
weights of the same steps without any collective, bit for bit."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _child(cmd, limit_s, clean_stdout=False):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0')
    env.pop('AMTX_DP_FORCE_COLLECTIVE', None)
    try:
        p = subprocess.run([sys.executable] + cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=limit_s)
    except subprocess.TimeoutExpired as e:          # subprocess.run has killed the child
        pytest.fail(f'child did not finish within {limit_s} s (hang?): {" ".join(cmd)}\n{(e.stderr or b"")[-2000:]}')
    lines = [l for l in p.stdout.splitlines() if l.startswith('{')]
    assert lines, f'no JSON line; rc {p.returncode}\nstdout: {p.stdout[-2000:]}\nstderr: {p.stderr[-4000:]}'
    if clean_stdout:      # the bench contract: ONE JSON line on stdout (RCCL's INFO-level version banner goes to stdout by itself: bench.py fences it off)
        out_lines = [l for l in p.stdout.splitlines() if l.strip()]
        assert len(out_lines) == 1 and out_lines[0].startswith('{'), p.stdout[-2000:]
    return p.returncode, json.loads(lines[-1]), p.stderr


@pytest.mark.timeout(900)
@pytest.mark.parametrize('of2', [False, True], ids=['OnsetsFrames_mc2', 'OnsetsFrames2_mc3'])
def test_single_rank_rccl_allreduce_300_steps_leaves_the_same_weights(of2):
    steps = 300
    rc, rec, err = _child([os.path.join('tools', 'rccl_single_rank.py'), '--steps', str(steps)] + (['--of2'] if of2 else []), 600)
    print(rec)
    assert rec['collectives'] == steps and rec['collectives_plain'] == 0, rec
    assert rec['deterministic_plain'], f'the training step itself is not run-to-run deterministic: {rec}'
    assert rec['identical'], f'weights differ between the RCCL run and the plain run: {rec}'
    assert rec['two_streams_identical_to_one_stream'], f'the two-stream step (pitch head on a side stream) differs from the one-stream step: {rec}'
    assert rc == 0, err[-2000:]


@pytest.mark.timeout(600)
def test_bench_train_line_with_a_forced_one_rank_nccl_group():
    """`python bench.py --mode train --gpus 1 --force-dist`: group created with device_id, barrier + MAX + gather over nccl, the
    all-reduce inside optimizer.step() and timed on its own -> `allreduce_ms_per_step` is a number."""
    rc, rec, err = _child(['bench.py', '--mode', 'train', '--gpus', '1', '--force-dist', '--steps', '20', '--warmup', '3', '--cpu-seconds', '0'], 400, clean_stdout=True)
    assert rc == 0, err[-2000:]
    cfg = rec['config']
    # RCCL's own account of the run travels in the line (which algorithm / protocol it picks for the 19.4 MB all-reduce)
    assert cfg['rccl'] is not None and any('version' in l for l in cfg['rccl']['init_lines']), cfg.get('rccl')
    # the two detector heads overlap also under a process group: the side stream is picked by a concurrency test (round 6: RCCL's streams had
    # pushed it onto the main stream's hardware queue, 11.4 instead of 9.5 ms); generous bound, boxes differ
    assert rec['ms_per_step'] < 10.8, rec['ms_per_step']
    assert rec['n_gpus'] == 1 and cfg['process_group'] == 'nccl'
    assert cfg['allreduce_ms_per_step'] is not None and cfg['allreduce_ms_per_step'] > 0
    assert cfg['collectives_per_step'] == 1.0, cfg
    assert 0 < rec['ms_per_step'] < 100


@pytest.mark.timeout(600)
def test_bench_infer_line_with_a_forced_one_rank_nccl_group():
    """The inference line's only collectives (timing barrier, MAX / gather of the elapsed times) over a one-rank nccl group."""
    rc, rec, err = _child(['bench.py', '--gpus', '1', '--force-dist', '--steps', '3', '--warmup', '1', '--clips', '64', '--cpu-seconds', '0',
                           '--no-parity', '--no-train-probe', '--no-hcqt'], 400, clean_stdout=True)
    assert rc == 0, err[-2000:]
    assert rec['n_gpus'] == 1 and rec['config']['process_group'] == 'nccl' and rec['value'] > 0
