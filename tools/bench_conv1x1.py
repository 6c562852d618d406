"""Micro-benchmark of the 1x1 modulated convolution at config-R shapes:  python tools/bench_conv1x1.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, 'stylegan3-editing_amd'), os.path.join(ROOT, 'tests'), ROOT):
    sys.path.insert(0, p)
import torch  # noqa: E402
from torch_utils.ops import modulated_conv  # noqa: E402
shapes = [(1024, 1024, 148), (1024, 645, 148), (645, 406, 276), (406, 256, 276), (256, 161, 532), (161, 102, 1044), (102, 64, 1044), (64, 64, 1044)]
n = 4
for ci, co, h in shapes:
    x = (torch.randn(n, ci, h, h, device='cuda') * 2).clamp(-256, 256); s = torch.randn(n, ci, device='cuda') + 1; w = torch.randn(co, ci, 1, 1, device='cuda')
    run = lambda: modulated_conv.modulated_conv2d(x, w, s, demodulate=True, padding=0, input_gain=torch.ones([], device='cuda'), x_bound=256.0)  # noqa: E731
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        run()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    print(f'{ci:5d} -> {co:5d} @ {h:4d}^2 x{n}: {ms * 1e3:8.1f} us  {2 * ci * co * h * h * n / ms / 1e9:7.1f} TFLOP/s  {(ci + co) * h * h * n * 4 / ms / 1e9:6.2f} TB/s')
