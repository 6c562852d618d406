"""Experiment: row-pitch alignment around the thin 1044^2-class 3x3 convolutions (channels of L11 / L12 / L13 of T-1024, batch 8).
  out-aligned: output rows on a 128-byte pitch (align_rows) vs dense;  in-aligned: an input width whose rows are 128-byte aligned
  (1056) vs the network's 1044 (4176-byte rows), both with aligned output rows.  ns per output pixel."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, 'stylegan3-editing_amd'), os.path.join(ROOT, 'tests'), ROOT):
    sys.path.insert(0, p)
import torch  # noqa: E402
from torch_utils.ops import modulated_conv  # noqa: E402
dev = 'cuda:0'


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


with torch.no_grad():
    for ci, co in ((51, 32), (32, 32), (81, 51)):
        for w_in, align in ((1044, False), (1044, True), (1056, True)):
            x = (torch.randn(8, ci, w_in, w_in, device=dev) * 2).clamp(-256, 256)
            s = torch.randn(8, ci, device=dev) + 1
            wt = torch.randn(co, ci, 3, 3, device=dev)
            t = timeit(lambda: modulated_conv.modulated_conv2d(x, wt, s, demodulate=True, padding=2, input_gain=torch.ones([], device=dev), x_bound=256.0, align_rows=align))
            print(f'{ci:3d}->{co:3d} in {w_in} {"aligned out" if align else "dense out  "}: {t:8.1f} us = {t / (w_in + 2) ** 2 * 1e3:7.4f} ns/px', flush=True)
