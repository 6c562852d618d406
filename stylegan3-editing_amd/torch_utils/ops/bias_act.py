"""Fused bias + activation + gain + clamp  (operator API of reference torch_utils/ops/bias_act.py:53-121).

`impl='cuda'` on a GPU tensor runs the HIP kernel in libsg3hip.so (csrc/sg3_bias_act.hip) through the
`bias_act_plugin` shim and supports first and second order gradients exactly like the reference's autograd classes
(reference :127-210).  CPU tensors, or `impl='ref'`, take the pure-PyTorch definition (reference `_bias_act_ref`,
:92-121) -- that is the reference's own behaviour for non-CUDA tensors and is what BASELINE config[0] runs; a GPU
tensor never falls back to it silently (a missing library raises).
"""
import os

import numpy as np
import torch

import dnnlib
from .. import custom_ops
from .. import misc

# name -> (python definition, default alpha, default gain, kernel index, which tensors backward needs, 2nd grad?)
activation_funcs = {
    'linear':   dnnlib.EasyDict(func=lambda x, **_: x,                                           def_alpha=0,   def_gain=1,          cuda_idx=1, ref='',  has_2nd_grad=False),
    'relu':     dnnlib.EasyDict(func=lambda x, **_: torch.nn.functional.relu(x),                 def_alpha=0,   def_gain=np.sqrt(2), cuda_idx=2, ref='y', has_2nd_grad=False),
    'lrelu':    dnnlib.EasyDict(func=lambda x, alpha, **_: torch.nn.functional.leaky_relu(x, alpha), def_alpha=0.2, def_gain=np.sqrt(2), cuda_idx=3, ref='y', has_2nd_grad=False),
    'tanh':     dnnlib.EasyDict(func=lambda x, **_: torch.tanh(x),                               def_alpha=0,   def_gain=1,          cuda_idx=4, ref='y', has_2nd_grad=True),
    'sigmoid':  dnnlib.EasyDict(func=lambda x, **_: torch.sigmoid(x),                            def_alpha=0,   def_gain=1,          cuda_idx=5, ref='y', has_2nd_grad=True),
    'elu':      dnnlib.EasyDict(func=lambda x, **_: torch.nn.functional.elu(x),                  def_alpha=0,   def_gain=1,          cuda_idx=6, ref='y', has_2nd_grad=True),
    'selu':     dnnlib.EasyDict(func=lambda x, **_: torch.nn.functional.selu(x),                 def_alpha=0,   def_gain=1,          cuda_idx=7, ref='y', has_2nd_grad=True),
    'softplus': dnnlib.EasyDict(func=lambda x, **_: torch.nn.functional.softplus(x),             def_alpha=0,   def_gain=1,          cuda_idx=8, ref='y', has_2nd_grad=True),
    'swish':    dnnlib.EasyDict(func=lambda x, **_: torch.sigmoid(x) * x,                        def_alpha=0,   def_gain=np.sqrt(2), cuda_idx=9, ref='x', has_2nd_grad=True),
}

_plugin = None
_null_tensor = torch.empty([0])


def _init():
    global _plugin
    if _plugin is None:
        _plugin = custom_ops.get_plugin(
            module_name='bias_act_plugin',
            sources=['sg3_bias_act.hip'],
            source_dir=os.path.join(os.path.dirname(__file__), '..', '..', 'csrc'))
    return True


def _resolve(act, alpha, gain, clamp):
    spec = activation_funcs[act]
    return (spec,
            float(spec.def_alpha if alpha is None else alpha),
            float(spec.def_gain if gain is None else gain),
            float(-1 if clamp is None else clamp))


def bias_act(x, b=None, dim=1, act='linear', alpha=None, gain=None, clamp=None, impl='cuda'):
    """y = clamp(act(x + b[dim]) * gain).  Same arguments and defaults as the reference op."""
    assert isinstance(x, torch.Tensor)
    assert impl in ['ref', 'cuda']
    if impl == 'cuda' and x.device.type == 'cuda' and _init():
        return _bias_act_cuda(dim=dim, act=act, alpha=alpha, gain=gain, clamp=clamp).apply(x, b)
    return _bias_act_ref(x=x, b=b, dim=dim, act=act, alpha=alpha, gain=gain, clamp=clamp)


@misc.profiled_function
def _bias_act_ref(x, b=None, dim=1, act='linear', alpha=None, gain=None, clamp=None):
    """Pure-PyTorch definition of the op (used for CPU tensors and as the autograd-anything reference)."""
    assert isinstance(x, torch.Tensor)
    assert clamp is None or clamp >= 0
    spec, alpha, gain, clamp = _resolve(act, alpha, gain, clamp)
    if b is not None:
        assert isinstance(b, torch.Tensor) and b.ndim == 1
        assert 0 <= dim < x.ndim
        assert b.shape[0] == x.shape[dim]
        view = [1] * x.ndim
        view[dim] = -1
        x = x + b.reshape(view)
    x = spec.func(x, alpha=alpha)
    if gain != 1:
        x = x * gain
    if clamp >= 0:
        x = x.clamp(-clamp, clamp)
    return x


_bias_act_cuda_cache = dict()


def _bias_act_cuda(dim=1, act='linear', alpha=None, gain=None, clamp=None):
    """autograd.Function pair (forward / gradient) around the HIP kernel, cached per parameter tuple."""
    assert clamp is None or clamp >= 0
    spec, alpha, gain, clamp = _resolve(act, alpha, gain, clamp)
    key = (dim, act, alpha, gain, clamp)
    if key in _bias_act_cuda_cache:
        return _bias_act_cuda_cache[key]
    keeps_x = 'x' in spec.ref or spec.has_2nd_grad
    trivial = act == 'linear' and gain == 1 and clamp < 0

    def fmt_of(t):
        return torch.channels_last if t.ndim > 2 and t.stride(1) == 1 else torch.contiguous_format

    class BiasActCuda(torch.autograd.Function):
        @staticmethod
        def forward(ctx, x, b):  # pylint: disable=arguments-differ
            ctx.memory_format = fmt_of(x)
            x = x.contiguous(memory_format=ctx.memory_format)
            b = b.contiguous() if b is not None else _null_tensor
            y = x
            if not trivial or b is not _null_tensor:
                y = _plugin.bias_act(x, b, _null_tensor, _null_tensor, _null_tensor, 0, dim, spec.cuda_idx, alpha, gain, clamp)
            ctx.save_for_backward(x if keeps_x else _null_tensor, b if keeps_x else _null_tensor,
                                  y if 'y' in spec.ref else _null_tensor)
            return y

        @staticmethod
        def backward(ctx, dy):  # pylint: disable=arguments-differ
            dy = dy.contiguous(memory_format=ctx.memory_format)
            x, b, y = ctx.saved_tensors
            dx = db = None
            if ctx.needs_input_grad[0] or ctx.needs_input_grad[1]:
                dx = dy if trivial else BiasActCudaGrad.apply(dy, x, b, y)
            if ctx.needs_input_grad[1]:
                db = dx.sum([i for i in range(dx.ndim) if i != dim])
            return dx, db

    class BiasActCudaGrad(torch.autograd.Function):
        @staticmethod
        def forward(ctx, dy, x, b, y):  # pylint: disable=arguments-differ
            ctx.memory_format = fmt_of(dy)
            dx = _plugin.bias_act(dy, b, x, y, _null_tensor, 1, dim, spec.cuda_idx, alpha, gain, clamp)
            ctx.save_for_backward(dy if spec.has_2nd_grad else _null_tensor, x, b, y)
            return dx

        @staticmethod
        def backward(ctx, d_dx):  # pylint: disable=arguments-differ
            d_dx = d_dx.contiguous(memory_format=ctx.memory_format)
            dy, x, b, y = ctx.saved_tensors
            d_dy = d_x = d_b = None
            if ctx.needs_input_grad[0]:
                d_dy = BiasActCudaGrad.apply(d_dx, x, b, y)
            if spec.has_2nd_grad and (ctx.needs_input_grad[1] or ctx.needs_input_grad[2]):
                d_x = _plugin.bias_act(d_dx, b, x, y, dy, 2, dim, spec.cuda_idx, alpha, gain, clamp)
            if spec.has_2nd_grad and ctx.needs_input_grad[2]:
                d_b = d_x.sum([i for i in range(d_x.ndim) if i != dim])
            return d_dy, d_x, d_b, None

    _bias_act_cuda_cache[key] = BiasActCuda
    return BiasActCuda
