"""Batched small-M GEMMs of the GradualStyleBlock heads on libsg3hip (csrc/sg3_head_gemm.hip: sg3_head_gemm_pack / sg3_head_gemm).

    out[g] = leaky_relu(a[g], slope) @ w[g] + bias[g]

For GPU inference this replaces the per-head Conv2d(3x3, stride 2, padding 1) of levels 2.. (as a GEMM over the patch matrix of
`unfold3x3s2`) and the closing EqualLinear (reference models/setgan/encoder/encoders/map2style.py:8-25).  Split-precision fp16 x 3
arithmetic, fp32-equivalent; operands outside the fp16 range raise plain_conv's range flag, on which the encoder repeats its forward
in 'fp32' (torch.baddbmm there: this module has no exact form of its own)."""
import ctypes

import torch

from .. import _sg3abi as abi
from . import plain_conv


class PackedHeadWeights:
    """w: [G, K, N] float32 CUDA (K % 16 == 0, N % 32 == 0), bias: [G, N] or None.  Packs once; `usable` is False when a weight
    lies outside the fp16 range (the caller keeps its fp32 path then)."""

    @staticmethod
    def supports(w):
        """True for a [G, K, N] float32 CUDA tensor with K % 16 == 0 and N % 32 == 0."""
        return bool(w.is_cuda and w.dtype == torch.float32 and w.ndim == 3 and int(w.shape[1]) % 16 == 0 and int(w.shape[2]) % 32 == 0)

    def __init__(self, w, bias=None):
        if not (w.is_cuda and w.dtype == torch.float32 and w.ndim == 3):
            raise RuntimeError('PackedHeadWeights: a [G,K,N] float32 CUDA tensor expected')
        g, k, n = (int(v) for v in w.shape)
        lib = abi.load()
        halfs = int(lib.sg3_head_gemm_packed_halfs(g, k, n))
        if halfs < 0:
            raise RuntimeError(f'PackedHeadWeights: unsupported shape K={k} (multiple of 16), N={n} (multiple of 32)')
        self.G, self.K, self.N = g, k, n
        self.packed = torch.empty([halfs], dtype=torch.float16, device=w.device)
        self.bias = None if bias is None else bias.reshape(g, n).to(torch.float32).contiguous()
        flag = torch.zeros([1], dtype=torch.int32, device=w.device)
        w = w.contiguous()
        with torch.cuda.device(w.device):
            abi.check(lib.sg3_head_gemm_pack(abi.ptr(w), abi.ptr(self.packed), g, k, n, abi.ptr(flag), abi.stream_ptr(w.device)),
                      'sg3_head_gemm_pack')
        self.usable = int(flag.item()) == 0

    def run(self, a, slope=1.0):
        """a: [G, M, K] float32 CUDA -> [G, M, N]."""
        if not (a.is_cuda and a.dtype == torch.float32 and a.ndim == 3 and int(a.shape[0]) == self.G and int(a.shape[2]) == self.K):
            raise RuntimeError(f'head_gemm: a [G={self.G}, M, K={self.K}] float32 CUDA tensor expected, got {tuple(a.shape)} {a.dtype}')
        if not self.usable:
            raise RuntimeError('head_gemm: these weights lie outside the fp16 range; the caller keeps its fp32 path')
        a = a.contiguous()
        m = int(a.shape[1])
        out = torch.empty([self.G, m, self.N], dtype=torch.float32, device=a.device)
        p = abi.HeadGemmParams()
        p.a, p.wPacked, p.bias, p.c = abi.ptr(a), abi.ptr(self.packed), (abi.ptr(self.bias) if self.bias is not None else None), abi.ptr(out)
        p.rangeFlag = abi.ptr(plain_conv._flag(a.device))
        p.G, p.M, p.K, p.N, p.slope = self.G, m, self.K, self.N, float(slope)
        with torch.cuda.device(a.device):
            abi.check(abi.load().sg3_head_gemm(ctypes.byref(p), abi.stream_ptr(a.device)), 'sg3_head_gemm')
        return out
