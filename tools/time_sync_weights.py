import os, sys, time, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from amt_tools_amd import tools
from amt_tools_amd.models import OnsetsFrames, OnsetsFrames2
for cls, mc in ((OnsetsFrames, 2), (OnsetsFrames2, 3)):
    m = cls(229, tools.PianoProfile(), 1, mc, device='cuda:0')
    m.change_device(); m.eval()
    eng = m._get_engine(torch.device('cuda:0'))
    for i in range(3):
        with torch.no_grad():
            for p in m.parameters():
                p.add_(0.0)          # version bump
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        eng.sync_weights(m)
        torch.cuda.synchronize()
        print(cls.__name__, mc, f'sync_weights {1e3 * (time.perf_counter() - t0):.1f} ms')
