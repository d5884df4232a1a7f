#!/usr/bin/env python3
"""
Reference-harness check (SURVEY section 4 "Integration").  BUILD CONTAINER ONLY: imports the REAL reference
from /root/reference under the import stubs of tools/gen_golden.py (Appendix C), never travels to the GPU box, CPU devices.

Drives the reference's UNMODIFIED callers with this package's objects and checks what comes out:

  1. amt_tools.train.train()            2 iterations, checkpoints, periodic validate(), on amt_tools_amd.models.OnsetsFrames +
                                         amt_tools_amd.dp.DataParallelOptimizer(Adam)            (train.py:19-191)
  2. the same train() resumed           torch.load(model) + in-place optimizer re-init           (train.py:72-113), with
                                         amt_tools_amd.tools.register_safe_globals() so torch >= 2.6's weights_only default passes
  3. amt_tools.inference.run_offline()  + ComboEstimator([NoteTranscriber, PitchListWrapper])    (inference.py:12-47)
  4. amt_tools.evaluate.validate()      with a stub dataset / evaluator                          (evaluate.py:52-101)
  5. the same objects through this package's own run_offline / NoteTranscriber: identical outputs

    python tools/check_reference_harness.py        -> prints one line per check and 'reference harness: OK'
"""
import os
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import gen_golden as gg   # noqa: E402  (installs the stubs, puts /root/reference and the repo root on sys.path, imports amt_tools)

import numpy as np        # noqa: E402
import torch              # noqa: E402

from amt_tools.train import train                                                     # noqa: E402  (the reference)
from amt_tools.evaluate import validate                                               # noqa: E402
from amt_tools.inference import run_offline as ref_run_offline                        # noqa: E402
from amt_tools.transcribe import ComboEstimator, NoteTranscriber as RefNoteTranscriber, PitchListWrapper as RefPitchListWrapper  # noqa: E402
import amt_tools.tools as rtools                                                      # noqa: E402

from amt_tools_amd import tools                                                       # noqa: E402
from amt_tools_amd.dp import DataParallelOptimizer                                    # noqa: E402
from amt_tools_amd.inference import run_offline                                       # noqa: E402
from amt_tools_amd.models import OnsetsFrames                                         # noqa: E402
from amt_tools_amd.transcribe import NoteTranscriber                                  # noqa: E402

DIM_IN, T, B = 40, 48, 2
HOP, SR = 512, 16000


def make_batches(n, seed):
    rng = np.random.default_rng(seed)
    out = []
    for _ in range(n):
        out.append({rtools.KEY_FEATS: torch.from_numpy(rng.random((B, 1, DIM_IN, T)).astype(np.float32)),
                    rtools.KEY_MULTIPITCH: torch.from_numpy((rng.random((B, 88, T)) < 0.05).astype(np.float32)),
                    rtools.KEY_ONSETS: torch.from_numpy((rng.random((B, 88, T)) < 0.01).astype(np.float32))})
    return out


class StubDataset(object):
    """What validate() needs of a TranscriptionDataset: `.tracks` and `.get_track_data(track_id)` (evaluate.py:77-80)."""

    def __init__(self, n, seed):
        rng = np.random.default_rng(seed)
        self.tracks = [f'track{i}' for i in range(n)]
        self._data = {t: {rtools.KEY_TRACK: t, rtools.KEY_FS: SR, rtools.KEY_FEATS: rng.random((1, DIM_IN, T)).astype(np.float32),
                          rtools.KEY_TIMES: np.arange(T) * HOP / float(SR),
                          rtools.KEY_MULTIPITCH: (rng.random((88, T)) < 0.05).astype(np.float32)} for t in self.tracks}

    def get_track_data(self, track_id):
        return dict(self._data[track_id])


class StubEvaluator(object):
    """Records what the reference hands to an Evaluator (evaluate.py:93, train.py:186)."""

    def __init__(self):
        self.seen, self.finalized = [], []

    def process_track(self, estimated, reference, track=None):
        assert set((rtools.KEY_ONSETS, rtools.KEY_MULTIPITCH, rtools.KEY_NOTES, rtools.KEY_PITCHLIST)) <= set(estimated.keys()), estimated.keys()
        assert estimated[rtools.KEY_MULTIPITCH].shape == reference[rtools.KEY_MULTIPITCH].shape == (88, T)
        self.seen.append(track)

    def average_results(self):
        return {'tracks': len(self.seen)}

    def finalize(self, writer, step):
        self.finalized.append(step)


def main():
    torch.manual_seed(0)
    torch.set_num_threads(4)
    profile = tools.PianoProfile()
    model = OnsetsFrames(DIM_IN, profile, 1, 2, device='cpu')
    model.change_device()
    optimizer = DataParallelOptimizer(model.parameters(), torch.optim.Adam, lr=6e-4)
    loader = make_batches(2, 1)
    val_set = StubDataset(2, 2)
    evaluator = StubEvaluator()
    estimator = ComboEstimator([RefNoteTranscriber(profile=profile), RefPitchListWrapper(profile=profile)])
    w0 = model.adjoin[1].output_layer.weight.detach().clone()

    with tempfile.TemporaryDirectory() as log_dir:
        # 1. fresh training with checkpoints + validation
        model = train(model, loader, optimizer, iterations=2, checkpoints=2, log_dir=log_dir, resume=False,
                      val_set=val_set, estimator=estimator, evaluator=evaluator)
        files = sorted(os.listdir(log_dir))
        assert [f for f in files if f.startswith('model-')] == ['model-1.pt', 'model-2.pt'], files
        assert [f for f in files if f.startswith('opt-state-')] == ['opt-state-1.pt', 'opt-state-2.pt'], files
        assert model.iter == 2 and evaluator.finalized == [1, 2] and len(evaluator.seen) == 4
        assert not torch.equal(w0, model.adjoin[1].output_layer.weight.detach())
        print(f'1. reference train(): 2 iterations x {len(loader)} batches, checkpoints {files}, validate() ran at {evaluator.finalized}: ok')

        # 2. resume (torch.load of the whole pickled module, optimizer re-initialised in place by the reference)
        tools.register_safe_globals()
        model2 = OnsetsFrames(DIM_IN, profile, 1, 2, device='cpu')
        opt2 = DataParallelOptimizer(model2.parameters(), torch.optim.Adam, lr=6e-4)
        model2 = train(model2, loader, opt2, iterations=3, checkpoints=0, log_dir=log_dir, resume=True)
        assert model2.iter == 3, model2.iter        # iter 2 came out of the checkpoint, one more iteration ran
        assert 'model-3.pt' in os.listdir(log_dir)
        assert len(opt2.state) > 0 and all('exp_avg' in s for s in opt2.state.values())
        print('2. reference train(resume=True): model + optimizer state restored from model-2.pt / opt-state-2.pt, iteration 3 ran: ok')

    # 3. reference run_offline + ComboEstimator on this package's model
    model.eval()
    track = val_set.get_track_data('track0')
    with torch.no_grad():
        pred_ref = ref_run_offline(dict(track), model, estimator)
        pred_own = run_offline(dict(track), model, NoteTranscriber(profile=profile))
    for k in (rtools.KEY_ONSETS, rtools.KEY_MULTIPITCH):
        assert pred_ref[k].shape == (88, T) and np.array_equal(pred_ref[k], pred_own[k]), k
    assert np.array_equal(pred_ref[rtools.KEY_NOTES], pred_own[tools.KEY_NOTES])
    times, pitch_list = pred_ref[rtools.KEY_PITCHLIST]
    assert len(pitch_list) == T
    print(f'3. reference run_offline() + ComboEstimator([NoteTranscriber, PitchListWrapper]): {len(pred_ref[rtools.KEY_NOTES])} notes, '
          f'same piano rolls and notes as amt_tools_amd.inference.run_offline + amt_tools_amd.transcribe.NoteTranscriber: ok')

    # 4. reference validate()
    ev = StubEvaluator()
    avg = validate(model, val_set, ev, estimator)
    assert avg == {'tracks': 2} and ev.seen == val_set.tracks
    print('4. reference validate(): 2 tracks through run_offline + estimators + evaluator: ok')
    print('reference harness: OK')


if __name__ == '__main__':
    main()
