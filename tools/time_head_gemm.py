"""Times sg3_head_gemm against torch.baddbmm (rocBLAS fp32) on the head shapes of the ReStyle encoders (16 heads, K = 4608 / 512,
N = 512) for M = images x pixels.  python tools/time_head_gemm.py  (GPU box; prints one line per shape, microseconds per launch)"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'stylegan3-editing_amd'))
from torch_utils.ops.head_gemm import PackedHeadWeights  # noqa: E402


def _time(fn, reps=200):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def main():
    dev = 'cuda:0'
    g, n = 16, 512
    for k in (4608, 512):
        w = torch.randn([g, k, n], device=dev) / k ** 0.5
        b = torch.randn([g, 1, n], device=dev)
        pw = PackedHeadWeights(w, b)
        for m in (8, 16, 32, 64, 128, 256, 512):
            a = torch.randn([g, m, k], device=dev)
            t_own = _time(lambda: pw.run(a))
            t_blas = _time(lambda: torch.baddbmm(b, a, w))
            mb = g * k * n * 4 / 1e6
            print(f'K {k:5d}  M {m:4d}  head_gemm {t_own:7.1f} us ({mb / t_own:6.2f} TB/s of weights)  baddbmm {t_blas:7.1f} us', flush=True)


if __name__ == '__main__':
    main()
