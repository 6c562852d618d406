"""Per-layer time of the PTI weight-gradient kernel at the T-1024 (or R-1024) layer shapes, batch 1:  python tools/bench_wgrad.py [T1024]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, 'stylegan3-editing_amd'), os.path.join(ROOT, 'tests'), ROOT):
    sys.path.insert(0, p)
import torch  # noqa: E402


def main():
    from models.stylegan3.networks_stylegan3 import Generator
    from synth_weights import CONFIGS
    from torch_utils.ops import modulated_conv
    cfg = sys.argv[1] if len(sys.argv) > 1 else 'T1024'
    G = Generator(**CONFIGS[cfg]).eval()
    tot = 0.0
    for name in G.synthesis.layer_names:
        L = getattr(G.synthesis, name)
        k, ins = L.conv_kernel, int(L.in_size[0])
        out = ins + k - 1
        x = torch.randn(1, L.in_channels, ins, ins, device='cuda')
        dy = torch.randn(1, L.out_channels, out, out, device='cuda')
        ax, ad = x.abs().max(), dy.abs().max()
        run = lambda: modulated_conv._weight_gradient(x, dy, k, k - 1, ax, ad)  # noqa: E731
        for _ in range(3):
            run()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for _ in range(10):
            run()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 100
        tot += us
        fl = 2 * L.in_channels * L.out_channels * k * k * out * out
        print(f'{name:22s} {L.in_channels:4d}->{L.out_channels:4d} @{ins:5d} k{k}: {us:8.1f} us  {fl / us * 1e-6:7.1f} TFLOP/s (kernel + partial-sum reduction)', flush=True)
    print(f'total {tot:.1f} us')


if __name__ == '__main__':
    main()
