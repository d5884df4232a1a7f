#!/usr/bin/env python3
"""Single-clip latency (the reference's run_offline pattern: one track per call): audio resident on the device -> piano rolls."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from amt_tools_amd import tools
from amt_tools_amd.synth import synth_clip
if '--of2' in sys.argv:      # OnsetsFrames2 as shipped (model_complexity 3, offset head, HTK mel)
    from amt_tools_amd.features import MelSpec
    from amt_tools_amd.models import OnsetsFrames2
    from amt_tools_amd.synth import synth_state_dict
    model = OnsetsFrames2(229, tools.PianoProfile(), 1, device='cuda:0', precision='bf16')
    model.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in synth_state_dict(0, dim_in=229, in_channels=1, model_complexity=3, offsets=True).items()})
    model.frontend = torch.nn.Sequential(MelSpec(sample_rate=22050, hop_length=512, n_mels=229, n_fft=2048, htk=True).frontend())
    model.change_device(); model.eval()
else:
    model, mel, sd = bench.build_model('cuda:0', 'bf16')
for B in (1, 2, 4, 8, 16, 32):
    audio = torch.from_numpy(np.stack([synth_clip(i) for i in range(B)])).cuda()
    with torch.no_grad():
        for _ in range(3): model.run_on_batch({tools.KEY_AUDIO: audio})
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            out = model.run_on_batch({tools.KEY_AUDIO: audio})
            torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 20
    print(f'{B} clip(s) x 625 frames per call: {dt * 1e3:.2f} ms per call, {B * 625 / dt / 1e6:.2f} M frames/s', flush=True)
