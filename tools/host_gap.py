import sys, os, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import bench
from amt_tools_amd import tools, _lib
from amt_tools_amd.synth import synth_clip
model, mel, sd = bench.build_model('cuda:0', 'bf16')
B = 512
base = np.stack([synth_clip(i) for i in range(4)])
audio = torch.from_numpy(base).cuda().repeat((B + 3) // 4, 1)[:B].contiguous()
def step():
    with torch.no_grad():
        return model.run_on_batch({tools.KEY_AUDIO: audio})
for _ in range(3): step()
torch.cuda.synchronize()
for prof in (0, 1):
    L = _lib.lib(); eng = model._get_engine(torch.device('cuda:0'))
    _lib.check(L.amtx_of_profile_enable(eng.handle, prof))
    for rep in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10): step()
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        print(f'prof={prof} enqueue {1e2*(t1-t0):.2f} ms/step, total {1e2*(t2-t0):.2f} ms/step', flush=True)
