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


@torch.no_grad()
def run_offline_batched(clips, model, times=None, batch_size=256, rank=0, world=1, decode_notes=False):
    """clips: (num_clips, N) float32 array / CPU tensor of equally long clips (the model needs a front-end in
    `model.frontend`) or (num_clips, C, F, T) features.  Returns {clip index: predictions dict} for the clips
    this rank owns.  With decode_notes=True the note lists are decoded on the device (amtx_notes_decode)."""
    clips = torch.as_tensor(np.asarray(clips) if not torch.is_tensor(clips) else clips)
    mine = shard_indices(clips.shape[0], rank, world)
    key = tools.KEY_AUDIO if clips.dim() == 2 else tools.KEY_FEATS
    out = {}
    for s in range(0, len(mine), batch_size):
        idx = mine[s:s + batch_size]
        batch = {key: clips[idx].float()}
        preds = model.run_on_batch(batch)
        notes = None
        if decode_notes:
            from .transcribe import decode_notes_batch
            T = preds[tools.KEY_MULTIPITCH].shape[-1]
            t = times if times is not None else np.arange(T) * 512 / 22050.0
            notes = decode_notes_batch(preds[tools.KEY_ONSETS], preds[tools.KEY_MULTIPITCH], t, model.profile.low)
        host = tools.dict_to_array(preds)
        for j, i in enumerate(idx):
            out[i] = {k: v[j] for k, v in host.items() if isinstance(v, np.ndarray)}
            if notes is not None:
                out[i][tools.KEY_NOTES] = notes[j]
    return out
