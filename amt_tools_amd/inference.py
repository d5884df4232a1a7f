"""
Inference drivers.

* `run_offline(track_data, model, estimator=None)` -- same contract as amt_tools/inference.py:12-47: a dict of
  host arrays for ONE track -> float32 -> tensors -> batch of one -> `model.run_on_batch` -> host arrays ->
  optional estimator.  (Without the reference's four whole-dict deep copies.)
* `run_offline_batched(...)` -- new (BASELINE config 5): many equally long clips per launch, clips sharded
  over ranks round-robin (`i % world`), no collective on the data path; per-clip results equal `run_offline`'s.
"""

import numpy as np
import torch

from . import tools
from .dp import shard_indices

__all__ = ['run_offline', 'run_offline_batched']


def run_offline(track_data, model, estimator=None):
    track_id = tools.unpack_dict(track_data, tools.KEY_TRACK)
    track_data = tools.dict_to_dtype(track_data, dtype=tools.FLOAT32)
    track_data = tools.dict_unsqueeze(tools.dict_to_tensor(track_data))
    predictions = tools.dict_squeeze(tools.dict_to_array(model.run_on_batch(track_data)), dim=0)
    if estimator is not None:
        predictions.update(estimator.process_track(predictions, track_id))
    return predictions


def _frame_times(model, num_frames):
    """Frame grid of the model's on-device front-end: frames * hop_length / sample_rate in float64 (FeatureModule.get_times,
    amt_tools/features/common.py:232-258).  Without a front-end module there is nothing to derive the grid from."""
    for m in getattr(model, 'frontend', []):
        mod = getattr(m, 'module', None)
        if mod is not None and hasattr(mod, 'get_hop_length') and hasattr(mod, 'get_sample_rate'):
            return (np.arange(num_frames) * mod.get_hop_length()).astype(np.float64) / float(mod.get_sample_rate())
    raise ValueError('decode_notes=True needs `times` when the model has no front-end module to take hop_length / sample_rate from')


@torch.no_grad()
def run_offline_batched(clips, model, times=None, batch_size=256, rank=0, world=1, decode_notes=False, keep=None):
    """clips: (num_clips, N) float32 -- or int16 PCM -- array / CPU tensor of equally long clips (the model needs a front-end in
    `model.frontend`) or (num_clips, C, F, T) features.  Returns {clip index: predictions dict} for the clips
    this rank owns.  With decode_notes=True the note lists are decoded on the device (amtx_notes_decode).
    `keep`: keys of the model output to bring back to the host (default: every array; () = notes only).

    On a GPU model the loop is a three-stage pipeline: batch i+1 is uploaded on a copy stream (from pinned memory) while batch
    i's kernels run, and batch i-1's results are brought to the host / assembled into note arrays while the GPU is busy."""
    clips = torch.as_tensor(np.asarray(clips) if not torch.is_tensor(clips) else clips)
    mine = shard_indices(clips.shape[0], rank, world)
    key = tools.KEY_AUDIO if clips.dim() == 2 else tools.KEY_FEATS
    # 16-bit PCM hand-over (round 6): int16 clips travel over PCIe as they are -- half the bytes of the float32 array the reference's loader makes
    # of the same file (amt_tools/tools/io.py:80-82: librosa / soundfile return sample / 32768 for 16-bit files) -- and become exactly that
    # float32 array on the device: sample * 2^-15 is exact.  Host-to-host transcription is bound by the upload (2 KB per frame as float32)
    pcm16 = key == tools.KEY_AUDIO and clips.dtype == torch.int16
    wire = torch.int16 if pcm16 else torch.float32
    device = torch.device(f'cuda:{model.device}' if isinstance(model.device, int) else model.device)
    on_gpu = device.type == 'cuda' and torch.cuda.is_available()
    out = {}

    def finish(pending):
        idx, preds, handle = pending
        notes = handle.result() if handle is not None else None
        host = tools.dict_to_array({k: v for k, v in preds.items() if keep is None or k in keep})
        for j, i in enumerate(idx):
            out[i] = {k: v[j] for k, v in host.items() if isinstance(v, np.ndarray)}
            if notes is not None:
                out[i][tools.KEY_NOTES] = notes[j]

    staging = [None, None]            # pinned staging buffers (ring of two) + the event of the copy that last read them
    stage_count = [0]

    def stage(idx):
        lo, hi = int(idx[0]), int(idx[-1]) + 1
        contiguous = hi - lo == len(idx)
        if not on_gpu:
            sel = (clips[lo:hi] if contiguous else clips[torch.as_tensor(idx)]).float()
            return (sel * (1.0 / 32768.0) if pcm16 else sel), None
        with torch.cuda.stream(copy_stream):
            zero_copy = contiguous and clips.is_pinned() and clips.dtype == wire
            if zero_copy:
                src = clips[lo:hi]                                # zero-copy: the caller's pinned memory is the DMA source
            else:
                slot = stage_count[0] % 2
                stage_count[0] += 1
                if staging[slot] is None:
                    staging[slot] = [torch.empty((batch_size,) + tuple(clips.shape[1:]), dtype=wire).pin_memory(), None]
                buf, last = staging[slot]
                if last is not None:
                    last.synchronize()                            # the copy that read this buffer two batches ago is done
                src = buf[:len(idx)]
                if contiguous:
                    src.copy_(clips[lo:hi])
                else:
                    torch.index_select(clips.to(wire) if clips.dtype != wire else clips, 0, torch.as_tensor(idx), out=src)
            dev = src.to(device, non_blocking=True)
            if pcm16:
                dev = dev.to(torch.float32).mul_(1.0 / 32768.0)   # on the copy stream, behind the DMA: the model's stream sees float32 audio
            ev = torch.cuda.Event()
            ev.record(copy_stream)
            if not zero_copy:
                staging[slot][1] = ev
        return dev, (ev, src)

    copy_stream = torch.cuda.Stream(device) if on_gpu else None
    starts = list(range(0, len(mine), batch_size))
    nxt = stage(mine[starts[0]:starts[0] + batch_size]) if starts else None
    pending = None
    for n, s in enumerate(starts):
        idx = mine[s:s + batch_size]
        data, sync = nxt
        if sync is not None:
            main = torch.cuda.current_stream(device)
            main.wait_event(sync[0])
            # `data` was allocated from the copy stream's pool: tell the allocator that the main stream reads it, or the block
            # could be handed to the upload of batch n+2 (and overwritten by DMA) while this batch's first kernel still reads it
            data.record_stream(main)
        if n + 1 < len(starts):
            nxt = stage(mine[starts[n + 1]:starts[n + 1] + batch_size])
        preds = model.run_on_batch({key: data})
        handle = None
        if decode_notes:
            T = preds[tools.KEY_MULTIPITCH].shape[-1]
            t = times if times is not None else _frame_times(model, T)
            if preds[tools.KEY_MULTIPITCH].is_cuda:
                from .transcribe import decode_notes_batch_async
                handle = decode_notes_batch_async(preds[tools.KEY_ONSETS], preds[tools.KEY_MULTIPITCH], t, model.profile.low)
            else:
                from .transcribe import multi_pitch_to_notes

                class _Host(object):
                    def __init__(self, p):
                        self.p = p

                    def result(self):
                        mp, on = self.p[tools.KEY_MULTIPITCH].numpy(), self.p[tools.KEY_ONSETS].numpy()
                        return [multi_pitch_to_notes(mp[b], t if np.ndim(t) == 1 else t[b], model.profile.low, on[b]) for b in range(mp.shape[0])]
                handle = _Host(preds)
        if pending is not None:
            finish(pending)                                       # the GPU is busy with this batch meanwhile
        pending = (idx, preds, handle)
    if pending is not None:
        finish(pending)
    return out
