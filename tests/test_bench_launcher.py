"""`python bench.py --gpus N` must really start N ranks (the flag was parsed and ignored).  CPU-only: the
`--dry-run` switch makes the ranks rendezvous over gloo and skip the GPU work, which leaves the launcher, the environment it
builds, the world-size check and the relay of rank 0's JSON line under test."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, 'bench.py')


def _run(args, env=None, timeout=180):
    e = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT')}
    e.update(env or {})
    return subprocess.run([sys.executable, BENCH] + args, env=e, capture_output=True, text=True, timeout=timeout)


def test_gpus_flag_starts_that_many_ranks_and_relays_one_json_line():
    p = _run(['--gpus', '2', '--dry-run'])
    assert p.returncode == 0, p.stderr
    lines = [l for l in p.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1
    rec = json.loads(lines[0])
    assert rec['n_gpus'] == 2 and rec['dry_run'] is True
    assert rec['ms_per_step'] == 2.0          # MAX over ranks of (rank + 1) ms: both ranks took part in the reduction


def test_world_size_mismatch_is_an_error_not_a_one_gpu_run():
    p = _run(['--gpus', '1', '--dry-run'], env={'RANK': '0', 'WORLD_SIZE': '2', 'LOCAL_RANK': '0'})
    assert p.returncode != 0
    assert 'WORLD_SIZE=2' in (p.stderr + p.stdout)


def test_failing_rank_fails_the_launcher():
    # an unknown flag makes every child exit with argparse's code 2 before any rendezvous
    p = _run(['--gpus', '2', '--dry-run', '--clips', 'not-a-number'])
    assert p.returncode != 0


def test_one_failing_rank_ends_the_run_at_once():
    """Only rank 1 fails (before the rendezvous): the launcher must terminate rank 0, which is waiting for its peer in
    init_process_group, and fail within seconds -- not after the store's rendezvous timeout."""
    import time
    t0 = time.time()
    p = _run(['--gpus', '2', '--dry-run', '--fail-rank', '1'], timeout=120)
    assert p.returncode != 0
    assert time.time() - t0 < 60
    assert 'a rank exited with an error' in p.stderr


def test_train_mode_shim_passes_through_the_same_launcher():
    p = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'bench_train.py'), '--gpus', '2', '--dry-run'],
                       env={k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK')},
                       capture_output=True, text=True, timeout=180)
    assert p.returncode == 0, p.stderr
    assert json.loads([l for l in p.stdout.splitlines() if l.startswith('{')][0])['n_gpus'] == 2


def test_force_dist_creates_a_one_rank_group():
    """`--force-dist` at --gpus 1: the process group exists and the line's collectives run through it (on the GPU box the same switch
    puts a one-rank RCCL communicator under test: tests/test_gpu_rccl.py)."""
    p = _run(['--gpus', '1', '--dry-run', '--force-dist'])
    assert p.returncode == 0, p.stderr
    rec = json.loads([l for l in p.stdout.splitlines() if l.startswith('{')][0])
    assert rec['n_gpus'] == 1 and rec['process_group'] == 'gloo' and rec['ms_per_step'] == 1.0
    p = _run(['--gpus', '1', '--dry-run'])
    assert json.loads([l for l in p.stdout.splitlines() if l.startswith('{')][0])['process_group'] is None


def test_eight_ranks_rendezvous_and_relay_one_line():
    """The driver's SCALE run is `--gpus 8`: eight fresh rank processes, one rendezvous, MAX over all eight, one relayed line."""
    p = _run(['--gpus', '8', '--dry-run'], timeout=400)
    assert p.returncode == 0, p.stderr
    lines = [l for l in p.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1
    rec = json.loads(lines[0])
    assert rec['n_gpus'] == 8 and rec['ms_per_step'] == 8.0 and rec['process_group'] == 'gloo'
