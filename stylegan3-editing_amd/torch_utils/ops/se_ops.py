"""Squeeze-and-excitation tail of the IR-SE residual units on libsg3hip (csrc/sg3_se.hip: sg3_se_residual).

Replaces, for GPU inference, `shortcut + res * sigmoid(fc2(relu(fc1(mean_hw(res)))))` of the reference's `SEModule` +
`bottleneck_IR_SE.forward` (models/setgan/encoder/encoders/helpers.py:57-73, :117-120): seven torch launches -> two."""
import ctypes

import torch

from .. import _sg3abi as abi


def se_residual(res, shortcut, fc1_weight, fc2_weight):
    """res [N,C,H,W] float32 contiguous (overwritten with the result and returned), shortcut [N,C,H,W] (any strides),
    fc1_weight [R,C,1,1] | [R,C], fc2_weight [C,R,1,1] | [C,R]."""
    if not (res.is_cuda and res.dtype == torch.float32 and res.is_contiguous() and shortcut.dtype == torch.float32 and shortcut.device == res.device):
        raise RuntimeError('se_residual: float32 CUDA tensors, res contiguous')
    n, c, h, w = (int(v) for v in res.shape)
    if tuple(shortcut.shape) != (n, c, h, w):
        raise RuntimeError(f'se_residual: shortcut {tuple(shortcut.shape)} vs res {tuple(res.shape)}')
    w1 = fc1_weight.detach().reshape(fc1_weight.shape[0], -1).to(torch.float32).contiguous()
    w2 = fc2_weight.detach().reshape(fc2_weight.shape[0], -1).to(torch.float32).contiguous()
    r = int(w1.shape[0])
    if tuple(w1.shape) != (r, c) or tuple(w2.shape) != (c, r):
        raise RuntimeError(f'se_residual: fc1 {tuple(w1.shape)}, fc2 {tuple(w2.shape)} for {c} channels')
    mean = torch.empty([n, c], dtype=torch.float32, device=res.device)
    p = abi.SeParams()
    p.res, p.shortcut, p.fc1, p.fc2, p.mean, p.out = abi.ptr(res), abi.ptr(shortcut), abi.ptr(w1), abi.ptr(w2), abi.ptr(mean), abi.ptr(res)
    p.scStride = abi.strides4(shortcut)
    p.N, p.C, p.H, p.W, p.R = n, c, h, w, r
    with torch.cuda.device(res.device):
        abi.check(abi.load().sg3_se_residual(ctypes.byref(p), abi.stream_ptr(res.device)), 'sg3_se_residual')
    return res
