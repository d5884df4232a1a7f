import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from amt_tools_amd import tools
from amt_tools_amd.models import OnsetsFrames2
from amt_tools_amd.synth import synth_state_dict
sd = synth_state_dict(3, dim_in=229, in_channels=1, model_complexity=3, offsets=True)
ms = {}
for prec in ('x3', 'bf16'):
    m = OnsetsFrames2(229, tools.PianoProfile(), 1, device='cuda:0', precision=prec)
    m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()})
    m.change_device(); m.eval(); ms[prec] = m
rng = np.random.default_rng(0)
for (B, T) in [(1, 24), (1, 24), (2, 40), (4, 300), (1, 16), (1, 128)]:
    feats = torch.from_numpy(rng.random((B, 1, 229, T)).astype(np.float32)).cuda()
    with torch.no_grad():
        a = ms['x3'].engine_logits(feats); b = ms['bf16'].engine_logits(feats); b2 = ms['bf16'].engine_logits(feats)
    print(B, T, {k: round((a[k] - b[k]).abs().max().item(), 4) for k in a}, 'rerun diff', {k: (b[k] - b2[k]).abs().max().item() for k in b})
