"""Plain (non-modulated) convolution on the fp32 matrix cores, with the element-wise neighbours of the ReStyle
encoder's convolutions fused in: an eval-mode BatchNorm in front (per-channel affine applied to real pixels only, so
zero padding stays zero), a BatchNorm behind (folded into the packed weights and the bias), PReLU / leaky ReLU.

Replaces, for inference, the module calls Conv2d / BatchNorm2d / PReLU / LeakyReLU of the IR-SE50 backbone and the
GradualStyleBlock heads (reference models/setgan/encoder/encoders/helpers.py:98-120, restyle_psp_encoders.py:26-50,
map2style.py:15-24).  Inference only: no autograd (training the encoder is out of scope).
"""
import ctypes

import torch

from .. import _sg3abi as abi

ACT_NONE, ACT_PRELU, ACT_LRELU = 0, 1, 2

# Arithmetic of the convolutions (3x3 stride 1: the bulk of the backbone; 3x3 stride 2: the first unit of each stage and level 1
# of the style heads; 1x1: the projection shortcuts):
#   'f16x3' : fp16 hi/lo operand split, three fp16 MFMAs per K step, fp32 accumulation (fp32-equivalent, 5.3x the fp32 MFMA
#             rate).  Operands must stay inside the fp16 range; every launch checks that on the device and raises a flag
#             (`overflowed`), on which the caller repeats its forward with 'fp32'.
#   'fp32'  : v_mfma_f32_32x32x2_f32, exact products.
precision = 'f16x3'
_flags = {}


def _flag(device):
    key = (device.type, device.index)
    if key not in _flags:
        _flags[key] = torch.zeros([1], dtype=torch.int32, device=device)
    return _flags[key]


def reset_overflow(device):
    _flag(torch.device(device)).zero_()


def overflowed(device):
    """True when a split-precision launch since the last `reset_overflow` met an operand outside the fp16 range
    (synchronises with the device)."""
    return bool(int(_flag(torch.device(device)).item()))


class PackedConv:
    """Weights of one convolution in the implicit-GEMM operand layout, plus its fused epilogue / prologue vectors."""

    def __init__(self, weight, out_scale=None, bias=None, in_scale=None, in_shift=None, act=ACT_NONE, slope=None, stride=1, padding=0):
        assert weight.is_cuda and weight.dtype == torch.float32 and weight.ndim == 4 and weight.shape[2] == weight.shape[3]
        self.O, self.I, self.k = int(weight.shape[0]), int(weight.shape[1]), int(weight.shape[2])
        assert self.k in (1, 3) and stride in (1, 2)
        self.stride, self.padding, self.act = int(stride), int(padding), int(act)
        dev = weight.device
        self._w = weight.detach().contiguous()
        f32 = lambda t: None if t is None else t.detach().to(device=dev, dtype=torch.float32).contiguous()  # noqa: E731
        self.out_scale, self.bias, self.in_scale, self.in_shift, self.slope = f32(out_scale), f32(bias), f32(in_scale), f32(in_shift), f32(slope)
        self._packed = {}

    def packed(self, prec):
        """Packed operand image for one arithmetic form (built on first use)."""
        if prec not in self._packed:
            lib = abi.load()
            dev = self._w.device
            buf = torch.empty([int(lib.sg3_modconv_packed_floats(self.O, self.I, self.k, prec))], dtype=torch.float32, device=dev)
            with torch.cuda.device(dev):
                abi.check(lib.sg3_conv2d_pack(abi.ptr(self._w), abi.ptr(self.out_scale), abi.ptr(buf), self.O, self.I, self.k, prec, abi.stream_ptr(dev)), 'sg3_conv2d_pack')
            self._packed[prec] = buf
        return self._packed[prec]

    def __call__(self, x):
        return self.run(x)

    def run(self, x):
        assert x.is_cuda and x.dtype == torch.float32 and x.ndim == 4 and x.shape[1] == self.I
        x = x.contiguous()
        n, _, h, w = (int(v) for v in x.shape)
        oh = (h + 2 * self.padding - self.k) // self.stride + 1
        ow = (w + 2 * self.padding - self.k) // self.stride + 1
        out = torch.empty([n, self.O, oh, ow], dtype=torch.float32, device=x.device)
        split = precision == 'f16x3'
        prec = abi.SG3_CONV_F16X3 if split else abi.SG3_CONV_FP32
        p = abi.Conv2dParams()
        p.x, p.wPacked, p.out = abi.ptr(x), abi.ptr(self.packed(prec)), abi.ptr(out)
        p.precision, p.rangeFlag = prec, (abi.ptr(_flag(x.device)) if split else None)
        p.inScale, p.inShift, p.bias, p.slope = abi.ptr(self.in_scale), abi.ptr(self.in_shift), abi.ptr(self.bias), abi.ptr(self.slope)
        p.N, p.I, p.O, p.H, p.W = n, self.I, self.O, h, w
        p.k, p.stride, p.pad, p.act = self.k, self.stride, self.padding, self.act
        with torch.cuda.device(x.device):
            abi.check(abi.load().sg3_conv2d(ctypes.byref(p), abi.stream_ptr(x.device)), 'sg3_conv2d')
        return out


def bn_affine(bn):
    """Eval-mode BatchNorm2d as y = a * x + b."""
    a = bn.weight.detach() * torch.rsqrt(bn.running_var.detach() + bn.eps)
    b = bn.bias.detach() - bn.running_mean.detach() * a
    return a, b
