"""`conv2d` / `conv_transpose2d` entry points used by the StyleGAN3 graph
(operator API of reference torch_utils/ops/conv2d_gradfix.py:22-54).

In the reference the custom arbitrary-order-gradient path is off by default (`enabled = False`, :22) and is only
switched on by the out-of-scope SetGAN trainer, so on the inversion / editing path these functions are
pass-throughs to torch.nn.functional (:39, :44).  They are kept as pass-throughs here (MIOpen on ROCm) for every
caller that is not the synthesis hot path (e.g. the `ref` composites); the hot path's modulated convolution does not
come through here -- it runs the MFMA implicit-GEMM kernel of libsg3hip.so (torch_utils/ops/modulated_conv.py).
The `enabled` / `weight_gradients_disabled` switches and the `no_weight_gradients` context manager exist with the
reference's names and semantics: when `enabled` is set, convolutions go through an autograd.Function whose
backward is itself expressed with convolutions (so double-backward works) and honours `no_weight_gradients`.
"""
import contextlib

import torch

enabled = False                     # reference default (:22)
weight_gradients_disabled = False   # skip weight-gradient computation inside no_weight_gradients()


@contextlib.contextmanager
def no_weight_gradients(disable=True):
    global weight_gradients_disabled
    old = weight_gradients_disabled
    if disable:
        weight_gradients_disabled = True
    try:
        yield
    finally:
        weight_gradients_disabled = old


def conv2d(input, weight, bias=None, stride=1, padding=0, dilation=1, groups=1):  # pylint: disable=redefined-builtin
    if _should_use_custom_op(input):
        return _ConvNd.apply(input, weight, bias, False, _pair(stride), _pair(padding), (0, 0), _pair(dilation), groups)
    return torch.nn.functional.conv2d(input=input, weight=weight, bias=bias, stride=stride, padding=padding, dilation=dilation, groups=groups)


def conv_transpose2d(input, weight, bias=None, stride=1, padding=0, output_padding=0, groups=1, dilation=1):  # pylint: disable=redefined-builtin
    if _should_use_custom_op(input):
        return _ConvNd.apply(input, weight, bias, True, _pair(stride), _pair(padding), _pair(output_padding), _pair(dilation), groups)
    return torch.nn.functional.conv_transpose2d(input=input, weight=weight, bias=bias, stride=stride, padding=padding,
                                                output_padding=output_padding, groups=groups, dilation=dilation)


def _should_use_custom_op(input):  # pylint: disable=redefined-builtin
    assert isinstance(input, torch.Tensor)
    return bool(enabled) and input.device.type == 'cuda'


def _pair(v):
    return (int(v), int(v)) if isinstance(v, int) else tuple(int(a) for a in v)


class _ConvNd(torch.autograd.Function):
    """Convolution whose gradients are again convolutions built from differentiable torch ops, so any order of
    derivative is available; weight gradients can be suppressed with `no_weight_gradients()`."""

    @staticmethod
    def forward(ctx, x, w, b, transpose, stride, padding, output_padding, dilation, groups):  # pylint: disable=arguments-differ
        ctx.cfg = (transpose, stride, padding, output_padding, dilation, groups)
        ctx.save_for_backward(x, w)
        ctx.has_bias = b is not None
        if transpose:
            return torch.nn.functional.conv_transpose2d(x, w, b, stride, padding, output_padding, groups, dilation)
        return torch.nn.functional.conv2d(x, w, b, stride, padding, dilation, groups)

    @staticmethod
    def backward(ctx, dy):  # pylint: disable=arguments-differ
        x, w = ctx.saved_tensors
        transpose, stride, padding, output_padding, dilation, groups = ctx.cfg
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            if transpose:
                dx = conv2d(dy, w, None, stride, padding, dilation, groups)
            else:
                op = [x.shape[i + 2] - ((dy.shape[i + 2] - 1) * stride[i] - 2 * padding[i] + dilation[i] * (w.shape[i + 2] - 1) + 1) for i in range(2)]
                dx = conv_transpose2d(dy, w, None, stride, padding, tuple(op), groups, dilation)
        if ctx.needs_input_grad[1] and not weight_gradients_disabled:
            # under create_graph (this backward runs with grad mode on) the weight gradient must stay differentiable w.r.t. BOTH
            # dy and x: the reference's Conv2dGradWeight.backward returns grad_input as well (conv2d_gradfix.py:175-190), which an
            # R1 / path-length penalty on the weights needs
            higher_order = torch.is_grad_enabled()
            with torch.enable_grad():
                xd = x if (higher_order and x.requires_grad) else x.detach()
                wd = w.detach().requires_grad_(True)
                if transpose:
                    y = torch.nn.functional.conv_transpose2d(xd, wd, None, stride, padding, output_padding, groups, dilation)
                else:
                    y = torch.nn.functional.conv2d(xd, wd, None, stride, padding, dilation, groups)
                dw, = torch.autograd.grad(y, wd, dy, create_graph=higher_order and (dy.requires_grad or xd.requires_grad))
        if ctx.has_bias and ctx.needs_input_grad[2]:
            db = dy.sum([0, 2, 3])
        return dx, dw, db, None, None, None, None, None, None
