import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (os.path.join(ROOT, 'stylegan3-editing_amd'), os.path.join(ROOT, 'tests'), ROOT):
    sys.path.insert(0, p)
import numpy as np, torch
from torch_utils.ops import modulated_conv as mc
torch.manual_seed(0)
n, ci, co, h, w = 1, int(sys.argv[1]) if len(sys.argv) > 1 else 64, 64, 30, 30
os.environ['SG3_F23_TN'] = '4'
x = torch.randn(n, ci, h, w, device='cuda'); wt = torch.randn(co, ci, 3, 3, device='cuda'); s = torch.ones(n, ci, device='cuda')
outs = {}
for mode in ('off', 'on'):
    mc.f23 = mode
    outs[mode] = mc.modulated_conv2d(x, wt, s, demodulate=False, padding=2, x_bound=8.0).float().cpu().numpy()
d = np.abs(outs['on'] - outs['off'])[0]
print('max err', d.max(), 'ref max', np.abs(outs['off']).max())
print('per channel block of 8:', np.round(d.reshape(8, 8, -1).max(axis=(1, 2)), 3))
print('per row:', np.round(d.max(axis=(0, 2)), 2))
print('per col:', np.round(d.max(axis=(0, 1)), 2))
# does the F23 output equal the reference computed with a subset of input channels?
for lo, hi in ((0, 16), (16, 32), (32, 48), (48, 64), (0, 32), (0, 48), (16, 64)):
    if hi > ci: continue
    xs = torch.zeros_like(x); xs[:, lo:hi] = x[:, lo:hi]
    mc.f23 = 'off'
    r = mc.modulated_conv2d(xs, wt, s, demodulate=False, padding=2, x_bound=8.0).float().cpu().numpy()
    print(f'channels {lo}:{hi} only -> diff to f23 output', np.abs(r - outs['on']).max())
