import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import torch.nn.functional as F
from amt_tools_amd import _lib
L = _lib.lib()
def ref_fn(x, w, scale, shift):
    xx = x.permute(0, 3, 1, 2).double()
    y = F.conv2d(xx, (w * scale[:, None, None, None]).double(), padding=1) + shift.double()[None, :, None, None]
    return F.max_pool2d(F.relu(y), (1, 2)).permute(0, 2, 3, 1).float()
for (b, t, f, cin, cout) in [(1, 16, 32, 48, 48), (3, 5, 36, 48, 48), (1, 33, 18, 48, 96), (2, 40, 229, 48, 48)]:
    g = torch.Generator().manual_seed(1)
    x = torch.rand(b, t, f, cin, generator=g)
    w = torch.randn(cout, cin, 3, 3, generator=g) / (9.0 * cin) ** 0.5
    scale = torch.rand(cout, generator=g) + 0.5
    shift = torch.randn(cout, generator=g) * 0.1
    packed = np.zeros(L.amtx_conv3x3g_packed_elems(cin, cout, 1), dtype=np.uint16)
    _lib.check(L.amtx_conv3x3g_pack(_lib.ptr(w.numpy()), _lib.ptr(scale.numpy()), cin, cout, 1, _lib.ptr(packed)))
    wp = torch.from_numpy(packed.view(np.int16)).cuda()
    xd = x.cuda().bfloat16().contiguous()
    sd = shift.cuda()
    outs = []
    for rep in range(3):
        out = torch.full((b, t, f // 2, cout), -7.0, dtype=torch.bfloat16, device='cuda')
        _lib.check(L.amtx_conv3x3g_fwd(_lib.ptr(xd), 0, _lib.ptr(wp), 1, _lib.ptr(sd), _lib.ptr(out), b, t, f, cin, cout, _lib.current_stream()))
        torch.cuda.synchronize()
        outs.append(out.float().cpu())
    ref = ref_fn(xd.float().cpu(), w, scale, shift)
    e = (outs[0] - ref).abs()
    bad = (e > 0.05).nonzero()
    print((b, t, f, cin, cout), 'max err', e.max().item(), 'rerun', (outs[0] - outs[1]).abs().max().item(), (outs[1] - outs[2]).abs().max().item(), 'nbad', len(bad), 'of', e.numel())
    if len(bad):
        print('  bad t:', sorted(set(bad[:, 1].tolist()))[:20], ' fo:', sorted(set(bad[:, 2].tolist()))[:30], ' ch:', sorted(set(bad[:, 3].tolist()))[:48])
