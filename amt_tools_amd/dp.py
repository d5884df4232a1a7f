"""
Clip-level data parallelism for the training step (BASELINE config 4).

The reference has no distributed code at all (amt_tools/train.py:62-64 is a TODO) and `train()` must
stay unchanged, so the exchange step lives in an object `train()` already calls: the optimizer.

    one process per GPU (torch.distributed, backend 'nccl' = RCCL over xGMI; 'gloo' on CPU for tests)
    identical initial weights on every rank, each rank draws its own clips
    train.py:130  loss.backward()            -- local gradients
    train.py:137  optimizer.step()           -- DataParallelOptimizer.step():
                                                 grads -> persistent flat fp32 buffer (one multi-tensor copy)
                                                 -> ONE in-place all-reduce (sum; 4.85 M values = 19.4 MB
                                                 for OF1) -> / world -> param.grad = view of the buffer
                                                 -> inner optimizer step

The loss is already a batch mean (amt_tools/models/common.py:582), so averaging gradients over ranks is
the gradient of the mean over the global batch.

BatchNorm policy under DP (the reference has neither DP nor SyncBN, so there is nothing to mirror):
  * training-mode normalisation uses PER-RANK batch statistics (each rank's own clips x frames x bins --
    625 x 229 x clips values per channel, so the statistics of 8 clips per rank are already tight);
  * the running statistics (and any other floating-point buffer handed over as `buffers=`) are AVERAGED over
    ranks inside the same flat all-reduce as the gradients, so every rank holds the same eval-mode model and
    the checkpoint rank 0 writes describes all shards, not only its own;
  * `torch.nn.SyncBatchNorm.convert_sync_batchnorm(model)` keeps the state_dict keys if global batch
    statistics are wanted; the model then runs those layers as stock modules (their collective), not through
    the HIP BatchNorm passes.
tests/test_dp.py measures what the per-rank statistics cost against a single-process step on the whole batch.

`train()` re-initialises the optimizer in place on resume with
`super(type(optimizer), optimizer).__init__(model.parameters(), optimizer.defaults)` (train.py:111); the
wrapper therefore derives directly from `torch.optim.Optimizer` (so that call lands on the base class) and
re-binds the inner optimizer to its own `param_groups` / `state` at every step.
"""

import os

import torch
import torch.distributed as dist

__all__ = ['init_distributed', 'force_collective_default', 'DataParallelOptimizer', 'broadcast_parameters', 'shard_indices', 'rank_log_dir']


def force_collective_default():
    """`AMTX_DP_FORCE_COLLECTIVE=1`: run the process group and the flat all-reduce also at world size 1 (a one-rank RCCL communicator
    on a one-GPU box exercises group creation, RCCL's own stream beside the training kernels and the flatten / unflatten path)."""
    return os.environ.get('AMTX_DP_FORCE_COLLECTIVE', '0') not in ('', '0')


def init_distributed(backend=None, force=None):
    """Initialise torch.distributed from the torchrun environment (RANK / LOCAL_RANK / WORLD_SIZE /
    MASTER_*).  Returns (rank, world_size, device).  Single-process runs return (0, 1, device) untouched unless `force` (default:
    `AMTX_DP_FORCE_COLLECTIVE`) asks for a one-rank group."""
    rank = int(os.environ.get('RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    use_gpu = torch.cuda.is_available()
    device = torch.device(f'cuda:{local_rank}') if use_gpu else torch.device('cpu')
    if use_gpu:
        torch.cuda.set_device(device)
    if force is None:
        force = force_collective_default()
    if (world > 1 or force) and not dist.is_initialized():
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29500')
        backend = backend or ('nccl' if use_gpu else 'gloo')
        if backend == 'nccl':
            dist.init_process_group('nccl', rank=rank, world_size=world, device_id=device)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    return rank, world, device


def shard_indices(num_items, rank, world):
    """Round-robin shard of independent units (clips / tracks): item i belongs to rank i % world."""
    return list(range(rank, num_items, world))


def rank_log_dir(log_dir, rank):
    """Only rank 0 writes checkpoints/events into `log_dir`; other ranks get a throw-away sibling."""
    return log_dir if rank == 0 else os.path.join(log_dir, f'.rank{rank}')


def broadcast_parameters(model, src=0):
    """Make every rank start from rank `src`'s parameters and buffers."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return
    with torch.no_grad():
        for t in list(model.parameters()) + list(model.buffers()):
            dist.broadcast(t.data, src=src)


class DataParallelOptimizer(torch.optim.Optimizer):
    """Wraps an optimizer class; `step()` averages gradients over all ranks with one flat all-reduce first."""

    def __init__(self, params, optimizer_cls=torch.optim.Adam, process_group=None, buffers=None, force_collective=None,
                 **optimizer_kwargs):
        """`buffers`: optional iterable of tensors (e.g. `model.buffers()`) whose floating-point members -- BatchNorm running
        statistics -- are averaged over ranks in the same all-reduce as the gradients.
        `force_collective` (default: the `AMTX_DP_FORCE_COLLECTIVE` environment switch): run the flatten -> all-reduce -> unflatten
        path also in a one-rank group.  A sum over one rank divided by one returns the gradients' own bits, so a forced run must
        leave the same weights as an unforced one (tests/test_gpu_rccl.py) -- it exists to put RCCL under test on a one-GPU box."""
        params = list(params)
        inner = optimizer_cls(params, **optimizer_kwargs)
        self.__dict__['_inner'] = inner
        self.__dict__['_group'] = process_group
        self.__dict__['_flat'] = None
        self.__dict__['_force'] = force_collective_default() if force_collective is None else bool(force_collective)
        self.__dict__['collectives_run'] = 0
        self.__dict__['_buffers'] = [b for b in (buffers or []) if torch.is_tensor(b) and b.dtype.is_floating_point]
        super().__init__(params, dict(inner.defaults))

    def _bind(self):
        inner = self._inner
        inner.param_groups = self.param_groups
        inner.state = self.state
        return inner

    def _world(self):
        if dist.is_available() and dist.is_initialized():
            return dist.get_world_size(self._group)
        return 1

    def _layout(self, params, bufs):
        """The flat exchange buffer and the cached views into it, rebuilt only when the parameter set, a parameter's memory layout or the
        device changes (train.py:111 re-initialises param_groups with the SAME parameter objects, so a resume keeps the cache)."""
        key = (tuple(id(p) for p in params), tuple(p.stride() for p in params), tuple(id(b) for b in bufs), params[0].device)
        lay = self.__dict__.get('_lay')
        if lay is not None and lay['key'] == key:
            return lay
        numel = sum(p.numel() for p in params) + sum(b.numel() for b in bufs)
        flat = torch.zeros(numel, dtype=torch.float32, device=params[0].device)
        views, off = [], 0
        for p in params:
            n = p.numel()
            seg = flat[off:off + n]
            # a gradient keeps its parameter's memory layout (autograd's layout contract; the GPU training path holds conv weights channels-last):
            # the view has the parameter's own strides, so the optimizer's multi-tensor kernels see matching layouts and stay on their fast path
            dense = p.is_contiguous() or (p.dim() == 4 and p.is_contiguous(memory_format=torch.channels_last))
            views.append(seg.as_strided(p.shape, p.stride()) if dense else seg.view(p.shape))
            off += n
        n_grad = off
        bviews = []
        for b in bufs:
            n = b.numel()
            bviews.append(flat[off:off + n].view(b.shape))
            off += n
        lay = {'key': key, 'flat': flat, 'views': views, 'bviews': bviews, 'n_grad': n_grad}
        self.__dict__['_lay'] = lay
        self.__dict__['_flat'] = flat
        return lay

    @torch.no_grad()
    def allreduce_gradients(self):
        """Gradients (+ floating-point buffers) averaged over ranks with ONE in-place all-reduce of the flat buffer.  Round 6: the flat buffer
        and its per-parameter views are allocated once; after the exchange every `param.grad` IS its view (the optimizer reads the averaged
        gradients straight out of the flat buffer -- no copy back, no per-step view construction, no per-step zero fill).  `zero_grad()`
        (set_to_none, the torch default train.py:125 uses) drops those references and the next backward produces fresh gradients, which one
        multi-tensor copy moves into the buffer; a gradient that is still its view (a second exchange without a backward in between) is left
        where it is."""
        world = self._world()
        forced = self.__dict__.get('_force', False) and dist.is_available() and dist.is_initialized()
        if world == 1 and not forced:
            return
        params = [p for g in self.param_groups for p in g['params'] if p.requires_grad]
        if not params:
            return
        bufs = self.__dict__.get('_buffers', [])
        lay = self._layout(params, bufs)
        flat, views, bviews = lay['flat'], lay['views'], lay['bviews']
        src, dst = [], []
        for p, v in zip(params, views):
            g = p.grad
            if g is None:
                v.zero_()
            elif g.data_ptr() != v.data_ptr() or g.stride() != v.stride():
                src.append(g)
                dst.append(v)
        if src:
            torch._foreach_copy_(dst, src)
        if bufs:
            torch._foreach_copy_(bviews, [b.detach() for b in bufs])
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self._group)
        self.__dict__['collectives_run'] = self.__dict__.get('collectives_run', 0) + 1
        if world > 1:                 # x / 1 is exact anyway; skipping it keeps the one-rank run's kernel list short
            flat.div_(world)
        for p, v in zip(params, views):
            p.grad = v
        if bufs:
            torch._foreach_copy_([b.detach() for b in bufs], bviews)

    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        self.allreduce_gradients()
        self._bind().step()
        return loss

    def __getstate__(self):
        state = super().__getstate__()
        return state

    def __setstate__(self, state):
        super().__setstate__(state)
