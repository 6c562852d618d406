"""Experiment: does a 128-byte-aligned output row pitch speed up the thin 1044^2-class convolutions and the filtered_lrelu that reads
them?  Same channels as L12 / L13 of T-1024, input widths 1044 (output rows of 1046 floats = 4184 B, every 128-B store segment
straddles two cache lines) vs 1054 (output rows of 1056 floats = 4224 B = 33 x 128)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, 'stylegan3-editing_amd'), os.path.join(ROOT, 'tests'), ROOT):
    sys.path.insert(0, p)
import numpy as np, torch  # noqa: E402
from torch_utils.ops import filtered_lrelu, modulated_conv  # noqa: E402
from oracle import oracle as O  # noqa: E402
dev = 'cuda:0'
fu = torch.from_numpy(O.design_lowpass_filter(12, 100.0, 200.0, 2048.0)).to(dev)
fd = torch.from_numpy(O.design_lowpass_filter(12, 100.0, 200.0, 2048.0)).to(dev)


def timeit(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


for ci, co in ((51, 32), (32, 32), (81, 51)):
    for w_in in (1044, 1054):
        x = (torch.randn(8, ci, w_in, w_in, device=dev) * 2).clamp(-256, 256)
        s = torch.randn(8, ci, device=dev) + 1
        wt = torch.randn(co, ci, 3, 3, device=dev)
        t_conv = timeit(lambda: modulated_conv.modulated_conv2d(x, wt, s, demodulate=True, padding=2, input_gain=torch.ones([], device=dev), x_bound=256.0))
        y = modulated_conv.modulated_conv2d(x, wt, s, demodulate=True, padding=2, input_gain=torch.ones([], device=dev), x_bound=256.0)
        b = torch.randn(co, device=dev)
        t_fl = timeit(lambda: filtered_lrelu.filtered_lrelu(y, fu=fu, fd=fd, b=b, up=2, down=2, padding=[9, 8, 9, 8], gain=np.sqrt(2), slope=0.2, clamp=256))
        px = (w_in + 2) ** 2
        print(f'{ci:3d}->{co:3d} in {w_in}: conv {t_conv:8.1f} us = {t_conv / px * 1e3:7.4f} ns/px   flrelu {t_fl:8.1f} us = {t_fl / px * 1e3:7.4f} ns/px', flush=True)
