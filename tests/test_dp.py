"""Data-parallel training wrapper on CPU with the gloo backend, world_size 2 (the N>1 path of SURVEY 8(e)):
one averaged step over two ranks == one single-process step on the concatenated batch."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from amt_tools_amd import tools
from amt_tools_amd.dp import DataParallelOptimizer, broadcast_parameters, shard_indices, rank_log_dir
from amt_tools_amd.models import OnsetsFrames

DIM_IN, T = 16, 10


def _make_model(seed, freeze_bn=True):
    torch.manual_seed(seed)
    model = OnsetsFrames(DIM_IN, tools.PianoProfile(), 1, 2)
    for m in model.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
        if freeze_bn and isinstance(m, torch.nn.BatchNorm2d):   # the exact-equivalence test freezes the statistics; the policy test does not
            m.eval()
    return model


def _batch(idx):
    rng = np.random.default_rng(100 + idx)
    return {tools.KEY_FEATS: torch.from_numpy(rng.random((1, 1, DIM_IN, T)).astype(np.float32)),
            tools.KEY_MULTIPITCH: torch.from_numpy((rng.random((1, 88, T)) < 0.1).astype(np.float32)),
            tools.KEY_ONSETS: torch.from_numpy((rng.random((1, 88, T)) < 0.03).astype(np.float32))}


def _cat(batches):
    return {k: torch.cat([b[k] for b in batches]) for k in batches[0]}


def _train_mode(model):
    model.train()
    for m in model.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.eval()


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    torch.set_num_threads(1)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    model = _make_model(seed=rank)                 # deliberately different init per rank ...
    broadcast_parameters(model, src=0)             # ... made identical here
    _train_mode(model)
    opt = DataParallelOptimizer(model.parameters(), torch.optim.Adam, lr=1e-2)
    for it in range(2):
        mine = [_batch(4 * it + i) for i in shard_indices(4, rank, world)]
        opt.zero_grad()
        loss = model.run_on_batch(_cat(mine))[tools.KEY_LOSS][tools.KEY_LOSS_TOTAL]
        loss.backward()
        opt.step()
    if rank == 0:
        torch.save({k: v.clone() for k, v in model.state_dict().items()}, out)
    dist.barrier()
    dist.destroy_process_group()


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


@pytest.mark.timeout(300)
def test_two_rank_step_equals_single_process_step(tmp_path):
    out = str(tmp_path / 'rank0.pt')
    mp.spawn(_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    got = torch.load(out)
    torch.set_num_threads(2)
    model = _make_model(seed=0)
    _train_mode(model)
    opt = torch.optim.Adam(model.parameters(), lr=1e-2)
    for it in range(2):
        opt.zero_grad()
        model.run_on_batch(_cat([_batch(4 * it + i) for i in range(4)]))[tools.KEY_LOSS][tools.KEY_LOSS_TOTAL].backward()
        opt.step()
    for k, v in model.state_dict().items():
        if v.dtype.is_floating_point:
            assert torch.allclose(got[k], v, atol=2e-5, rtol=1e-4), k


def _bn_worker(rank, world, port, out, total=4, lr=1e-2):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    torch.set_num_threads(1)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    model = _make_model(seed=rank, freeze_bn=False)
    broadcast_parameters(model, src=0)
    model.train()                                   # BatchNorm in training mode: per-rank batch statistics
    opt = DataParallelOptimizer(model.parameters(), torch.optim.Adam, lr=lr, buffers=model.buffers())
    losses = []
    for it in range(2):
        mine = [_batch(total * it + i) for i in shard_indices(total, rank, world)]
        opt.zero_grad()
        loss = model.run_on_batch(_cat(mine))[tools.KEY_LOSS][tools.KEY_LOSS_TOTAL]
        loss.backward()
        opt.step()
        losses.append(float(loss.detach()))
    torch.save({'sd': {k: v.clone() for k, v in model.state_dict().items()}, 'losses': losses}, out + f'.{rank}')
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_batchnorm_policy_under_dp_per_rank_statistics_and_averaged_running_stats(tmp_path, capsys):
    """The stated BatchNorm policy of amt_tools_amd/dp.py with BatchNorm really in training mode: (1) after every step all ranks
    hold identical parameters AND identical running statistics (averaged in the gradient all-reduce); (2) what per-rank batch
    statistics cost against the single-process step on the whole batch (global statistics) is measured: the parameters after two
    Adam steps stay within a few percent of the single-process ones -- not bit-equal, which is why it is a documented policy."""
    out = str(tmp_path / 'bn')
    mp.spawn(_bn_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    r0, r1 = torch.load(out + '.0'), torch.load(out + '.1')
    for k, v in r0['sd'].items():
        if v.dtype.is_floating_point:
            assert torch.equal(v, r1['sd'][k]), k                          # parameters and running statistics agree bit for bit
    torch.set_num_threads(2)
    model = _make_model(seed=0, freeze_bn=False)
    model.train()
    opt = torch.optim.Adam(model.parameters(), lr=1e-2)
    ref_losses = []
    for it in range(2):
        opt.zero_grad()
        loss = model.run_on_batch(_cat([_batch(4 * it + i) for i in range(4)]))[tools.KEY_LOSS][tools.KEY_LOSS_TOTAL]
        loss.backward()
        opt.step()
        ref_losses.append(float(loss.detach()))
    dp_loss = [(a + b) / 2 for a, b in zip(r0['losses'], r1['losses'])]
    rel = []
    for k, v in model.state_dict().items():
        if v.dtype.is_floating_point and v.numel() > 1:
            rel.append(float((r0['sd'][k] - v).norm() / (v.norm() + 1e-12)))
    with capsys.disabled():
        print(f'\n[BatchNorm under DP] mean loss per step: 2 ranks x 2 clips (per-rank statistics) {dp_loss} vs one process x 4 clips '
              f'{ref_losses}; relative L2 distance of the tensors after 2 Adam steps: median {np.median(rel):.2e}, max {max(rel):.2e}')
    assert abs(dp_loss[0] - ref_losses[0]) / ref_losses[0] < 0.05          # first step: same weights, only the statistics differ
    assert np.median(rel) < 0.15                                          # Adam at lr 1e-2 on 320-value statistics amplifies it; see the printed numbers


@pytest.mark.timeout(600)
def test_batchnorm_policy_at_the_real_per_rank_batch_of_eight_clips(tmp_path, capsys):
    """The same policy at the batch the training configuration really uses (BASELINE config 4: 8 clips per GPU) and the reference's learning
    rate (6e-4, examples/papers/of_1.py): 2 ranks x 8 clips with per-rank statistics against one process on all 16.  VERDICT r02: the 2-clip
    drift (max 5e-1 at lr 1e-2) was a large number to wave through without this run."""
    out = str(tmp_path / 'bn8')
    mp.spawn(_bn_worker, args=(2, _free_port(), out, 16, 6e-4), nprocs=2, join=True)
    r0, r1 = torch.load(out + '.0'), torch.load(out + '.1')
    for k, v in r0['sd'].items():
        if v.dtype.is_floating_point:
            assert torch.equal(v, r1['sd'][k]), k
    torch.set_num_threads(2)
    model = _make_model(seed=0, freeze_bn=False)
    model.train()
    opt = torch.optim.Adam(model.parameters(), lr=6e-4)
    ref_losses = []
    for it in range(2):
        opt.zero_grad()
        loss = model.run_on_batch(_cat([_batch(16 * it + i) for i in range(16)]))[tools.KEY_LOSS][tools.KEY_LOSS_TOTAL]
        loss.backward()
        opt.step()
        ref_losses.append(float(loss.detach()))
    dp_loss = [(a + b) / 2 for a, b in zip(r0['losses'], r1['losses'])]
    rel, names = [], []
    for k, v in model.state_dict().items():
        if v.dtype.is_floating_point and v.numel() > 1:
            rel.append(float((r0['sd'][k] - v).norm() / (v.norm() + 1e-12)))
            names.append(k)
    worst = names[int(np.argmax(rel))]
    sdm = model.state_dict()
    big = [r for r, k in zip(rel, names) if float(sdm[k].norm()) > 0.1]      # tensors that are not a handful of Adam steps away from zero
    with capsys.disabled():
        print(f'\n[BatchNorm under DP, 2 ranks x 8 clips, lr 6e-4] mean loss per step {dp_loss} vs one process x 16 clips {ref_losses}; '
              f'relative L2 distance of the tensors after 2 Adam steps: median {np.median(rel):.2e}, max over tensors with |w| > 0.1 '
              f'{max(big):.2e}, max over all {max(rel):.2e} ({worst}, |w| = {float(sdm[worst].norm()):.1e}: a zero-initialised BatchNorm '
              f'bias two Adam steps old -- every entry is +-lr or +-2 lr, so an entry whose tiny gradient changes sign is a 100 % change)')
    for a, b in zip(dp_loss, ref_losses):
        assert abs(a - b) / b < 1e-4                                       # the loss itself agrees to ~1e-5 at both steps
    assert np.median(rel) < 5e-3 and max(big) < 0.05


def test_wrapper_survives_the_in_place_reinit_of_train_py():
    """amt_tools/train.py:111 re-runs the optimizer base-class __init__ in place on resume, then loads the state."""
    model = _make_model(0)
    _train_mode(model)
    opt = DataParallelOptimizer(model.parameters(), torch.optim.Adam, lr=6e-4)
    model.run_on_batch(_batch(0))[tools.KEY_LOSS][tools.KEY_LOSS_TOTAL].backward()
    opt.step()
    state = opt.state_dict()
    model2 = _make_model(0)
    model2.load_state_dict(model.state_dict())
    _train_mode(model2)
    super(type(opt), opt).__init__(model2.parameters(), opt.defaults)
    opt.load_state_dict(state)
    before = model2.onset_head[2].output_layer.bias.detach().clone()
    opt.zero_grad()
    model2.run_on_batch(_batch(1))[tools.KEY_LOSS][tools.KEY_LOSS_TOTAL].backward()
    opt.step()
    assert not torch.equal(before, model2.onset_head[2].output_layer.bias)
    assert opt.state_dict()['state'][0]['step'] == 2 or float(opt.state_dict()['state'][0]['step']) == 2.0


def test_shard_helpers():
    assert shard_indices(10, 1, 4) == [1, 5, 9]
    assert sorted(sum((shard_indices(10, r, 4) for r in range(4)), [])) == list(range(10))
    assert rank_log_dir('/x', 0) == '/x' and rank_log_dir('/x', 3) == '/x/.rank3'


def _forced_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    torch.set_num_threads(1)
    states = []
    for forced in (False, True):
        if forced:
            dist.init_process_group('gloo', rank=0, world_size=1)
        model = _make_model(seed=0, freeze_bn=False)
        model.train()
        opt = DataParallelOptimizer(model.parameters(), torch.optim.Adam, lr=1e-2, buffers=model.buffers(), force_collective=forced)
        torch.manual_seed(5)
        for it in range(3):
            opt.zero_grad()
            model.run_on_batch(_cat([_batch(2 * it), _batch(2 * it + 1)]))[tools.KEY_LOSS][tools.KEY_LOSS_TOTAL].backward()
            opt.step()
        states.append(({k: v.clone() for k, v in model.state_dict().items()}, opt.collectives_run))
    dist.destroy_process_group()
    torch.save(states, out)


@pytest.mark.timeout(300)
def test_forced_one_rank_collective_returns_the_same_bits(tmp_path):
    """`force_collective=True` in a one-rank group (what tests/test_gpu_rccl.py runs over RCCL on the GPU box): the flatten ->
    all-reduce -> unflatten path runs once per step and leaves bit-identical weights and running statistics."""
    out = str(tmp_path / 'forced.pt')
    mp.spawn(_forced_worker, args=(1, _free_port(), out), nprocs=1, join=True)
    (plain, n0), (forced, n1) = torch.load(out)
    assert n0 == 0 and n1 == 3
    for k, v in plain.items():
        assert torch.equal(v, forced[k]), k


def _flat_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    torch.set_num_threads(1)
    dist.init_process_group('gloo', rank=0, world_size=1)
    model = _make_model(seed=0, freeze_bn=False)
    model.to(memory_format=torch.channels_last)            # the GPU training layout of the conv weights (models.py change_device)
    model.train()
    opt = DataParallelOptimizer(model.parameters(), torch.optim.Adam, lr=1e-2, buffers=model.buffers(), force_collective=True)
    res = {}
    ptrs = []
    for it in range(3):
        opt.zero_grad(set_to_none=(it != 2))               # the last step keeps the gradients (zeroed in place, inside the flat buffer)
        if it == 2:
            res['kept_grads_are_zeroed_views'] = all(p.grad is not None and float(p.grad.abs().sum()) == 0.0 for p in model.parameters())
        model.run_on_batch(_cat([_batch(2 * it), _batch(2 * it + 1)]))[tools.KEY_LOSS][tools.KEY_LOSS_TOTAL].backward()
        opt.step()
        flat = opt._flat
        ptrs.append(flat.data_ptr())
        lo, hi = flat.data_ptr(), flat.data_ptr() + flat.numel() * 4
        res[f'grads_in_flat_{it}'] = all(lo <= p.grad.data_ptr() < hi for p in model.parameters())
        res[f'strides_match_{it}'] = all(p.grad.stride() == p.stride() for p in model.parameters())
    res['flat_allocated_once'] = len(set(ptrs)) == 1
    res['numel'] = int(opt._flat.numel())
    res['expected_numel'] = sum(p.numel() for p in model.parameters()) + sum(b.numel() for b in model.buffers() if b.dtype.is_floating_point)
    # a second exchange without a backward in between: gradients are already their views, nothing to copy, values unchanged (sum over one rank)
    before = [p.grad.clone() for p in model.parameters()]
    opt.allreduce_gradients()
    res['idempotent'] = all(torch.equal(a, p.grad) for a, p in zip(before, model.parameters()))
    res['collectives'] = opt.collectives_run
    dist.destroy_process_group()
    torch.save(res, out)


@pytest.mark.timeout(300)
def test_gradients_live_in_the_flat_buffer(tmp_path):
    """Round 6: ONE flat fp32 buffer allocated once; after the exchange every param.grad is a view into it with the
    parameter's own strides (channels-last conv weights included), zero_grad(set_to_none=False) zeroes it in place, and no copy back."""
    out = str(tmp_path / 'flat.pt')
    mp.spawn(_flat_worker, args=(1, _free_port(), out), nprocs=1, join=True)
    res = torch.load(out)
    assert res['flat_allocated_once'] and res['numel'] == res['expected_numel']
    assert res['kept_grads_are_zeroed_views'] and res['idempotent'] and res['collectives'] == 4
    for it in range(3):
        assert res[f'grads_in_flat_{it}'] and res[f'strides_match_{it}'], it
