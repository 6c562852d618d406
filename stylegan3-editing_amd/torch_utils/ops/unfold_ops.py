"""Patch matrix of a 3x3, stride-2, padding-1 convolution on libsg3hip (csrc/sg3_unfold.hip: sg3_unfold3x3s2).

Replaces, for GPU inference, the im2col step of the GradualStyleBlock heads' convolutions (reference
models/setgan/encoder/encoders/map2style.py:8-25: Conv2d(3x3, stride 2, padding 1) + LeakyReLU per level) when those run as batched
GEMMs: one launch per level, reading the previous level's result where it lies (through strides) and applying its LeakyReLU."""
import ctypes

import torch

from .. import _sg3abi as abi


def unfold3x3s2(src, slope=1.0):
    """src: a float32 CUDA tensor VIEW of shape [G, N, C, IH, IW] (any strides: the data is read in place).
    Returns [G, N*OH*OW, 9*C] with OH = (IH+1)//2, OW = (IW+1)//2; column tap*C + c, tap = ky*3 + kx; every sample read goes
    through LeakyReLU(slope) (slope 1.0: none)."""
    if not (src.is_cuda and src.dtype == torch.float32 and src.ndim == 5):
        raise RuntimeError('unfold3x3s2: a 5-D float32 CUDA tensor [G,N,C,IH,IW] expected')
    g, n, c, ih, iw = (int(v) for v in src.shape)
    oh, ow = (ih + 1) // 2, (iw + 1) // 2
    dst = torch.empty([g, n * oh * ow, 9 * c], dtype=torch.float32, device=src.device)
    p = abi.UnfoldParams()
    p.src, p.dst = abi.ptr(src), abi.ptr(dst)
    p.srcStride = (ctypes.c_int64 * 5)(*[int(s) for s in src.stride()])
    p.G, p.N, p.C, p.IH, p.IW, p.slope = g, n, c, ih, iw, float(slope)
    with torch.cuda.device(src.device):
        abi.check(abi.load().sg3_unfold3x3s2(ctypes.byref(p), abi.stream_ptr(src.device)), 'sg3_unfold3x3s2')
    return dst
