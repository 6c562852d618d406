"""Fused bias + activation + gain + clamp  (operator API of reference torch_utils/ops/bias_act.py:53-121).

`impl='cuda'` on a GPU tensor runs the HIP kernel in libsg3hip.so (csrc/sg3_bias_act.hip) through the
`bias_act_plugin` shim, with first and second order gradients (what the reference's two autograd classes provide,
:127-210).  CPU tensors, or `impl='ref'`, take the PyTorch definition -- the reference's own behaviour for non-CUDA
tensors and what BASELINE config[0] runs; a GPU tensor never falls back to it silently (a missing library raises).

Layout of this module: one table of activations, one immutable `_Call` record per (dim, act, alpha, gain, clamp), and
two module-level autograd Functions that receive the record as a non-tensor argument.
"""
import math
import os
from collections import namedtuple

import torch
import torch.nn.functional as F

import dnnlib
from .. import custom_ops
from .. import misc

_R2 = math.sqrt(2.0)

# name: (kernel index, f(x, alpha), default alpha, default gain, tensor the gradient reads, has a second derivative)
_TABLE = {
    'linear':   (1, lambda x, a: x,                      0.0, 1.0, '',  False),
    'relu':     (2, lambda x, a: F.relu(x),              0.0, _R2, 'y', False),
    'lrelu':    (3, lambda x, a: F.leaky_relu(x, a),     0.2, _R2, 'y', False),
    'tanh':     (4, lambda x, a: torch.tanh(x),          0.0, 1.0, 'y', True),
    'sigmoid':  (5, lambda x, a: torch.sigmoid(x),       0.0, 1.0, 'y', True),
    'elu':      (6, lambda x, a: F.elu(x),               0.0, 1.0, 'y', True),
    'selu':     (7, lambda x, a: F.selu(x),              0.0, 1.0, 'y', True),
    'softplus': (8, lambda x, a: F.softplus(x),          0.0, 1.0, 'y', True),
    'swish':    (9, lambda x, a: torch.sigmoid(x) * x,   0.0, _R2, 'x', True),
}


def _entry(name):
    idx, fn, alpha, gain, reads, second = _TABLE[name]
    return dnnlib.EasyDict(func=lambda x, alpha=None, **_: fn(x, alpha), def_alpha=alpha, def_gain=gain,
                           cuda_idx=idx, ref=reads, has_2nd_grad=second)


# public table with the reference's field names (networks read `.def_gain` from it)
activation_funcs = {name: _entry(name) for name in _TABLE}

_Call = namedtuple('_Call', 'dim act idx alpha gain clamp reads second')   # clamp < 0: disabled


def _call_record(dim, act, alpha, gain, clamp):
    assert clamp is None or clamp >= 0
    idx, _, a0, g0, reads, second = _TABLE[act]
    return _Call(int(dim), act, idx, float(a0 if alpha is None else alpha), float(g0 if gain is None else gain),
                 float(-1 if clamp is None else clamp), reads, second)


_plugin = None
_NONE = torch.empty([0])


def _init():
    global _plugin
    if _plugin is None:
        _plugin = custom_ops.get_plugin(module_name='bias_act_plugin', sources=['sg3_bias_act.hip'],
                                        source_dir=os.path.join(os.path.dirname(__file__), '..', '..', 'csrc'))
    return True


def bias_act(x, b=None, dim=1, act='linear', alpha=None, gain=None, clamp=None, impl='cuda'):
    """y = clamp(act(x + b[dim]) * gain).  Same arguments and defaults as the reference op."""
    assert isinstance(x, torch.Tensor)
    assert impl in ['ref', 'cuda']
    call = _call_record(dim, act, alpha, gain, clamp)
    if impl == 'cuda' and x.device.type == 'cuda' and _init():
        return _Fused.apply(x, b, call)
    return _bias_act_ref(x, b, call=call)


@misc.profiled_function
def _bias_act_ref(x, b=None, dim=1, act='linear', alpha=None, gain=None, clamp=None, call=None):
    """PyTorch definition of the op (CPU tensors; differentiable to any order by autograd)."""
    assert isinstance(x, torch.Tensor)
    call = call or _call_record(dim, act, alpha, gain, clamp)
    if b is not None:
        assert isinstance(b, torch.Tensor) and b.ndim == 1
        assert 0 <= call.dim < x.ndim and b.shape[0] == x.shape[call.dim]
        x = x + b.reshape([-1 if i == call.dim else 1 for i in range(x.ndim)])
    y = _TABLE[call.act][1](x, call.alpha)
    y = y if call.gain == 1 else y * call.gain
    return y if call.clamp < 0 else y.clamp(-call.clamp, call.clamp)


def _layout(t):
    return torch.channels_last if (t.ndim > 2 and t.stride(1) == 1) else torch.contiguous_format


def _kernel(call, grad_order, x, b, xref, yref, dy):
    return _plugin.bias_act(x, b, xref, yref, dy, grad_order, call.dim, call.idx, call.alpha, call.gain, call.clamp)


def _over_all_but(t, dim):
    return t.sum([i for i in range(t.ndim) if i != dim])


class _Fused(torch.autograd.Function):
    """y = kernel(x, b); keeps whichever of x / y the activation's derivative is written in."""

    @staticmethod
    def forward(ctx, x, b, call):  # pylint: disable=arguments-differ
        ctx.call, ctx.layout = call, _layout(x)
        x = x.contiguous(memory_format=ctx.layout)
        b = _NONE if b is None else b.contiguous()
        ctx.identity = call.act == 'linear' and call.gain == 1 and call.clamp < 0
        y = x if (ctx.identity and b is _NONE) else _kernel(call, 0, x, b, _NONE, _NONE, _NONE)
        keep_x = 'x' in call.reads or call.second
        ctx.save_for_backward(x if keep_x else _NONE, b if keep_x else _NONE, y if 'y' in call.reads else _NONE)
        return y

    @staticmethod
    def backward(ctx, dy):  # pylint: disable=arguments-differ
        want_x, want_b = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        if not (want_x or want_b):
            return None, None, None
        dy = dy.contiguous(memory_format=ctx.layout)
        dx = dy if ctx.identity else _FusedGrad.apply(dy, *ctx.saved_tensors, ctx.call)
        return dx, (_over_all_but(dx, ctx.call.dim) if want_b else None), None


class _FusedGrad(torch.autograd.Function):
    """dx = dy * act'(.) * gain, masked by the clamp; differentiable once more for activations that need it."""

    @staticmethod
    def forward(ctx, dy, x, b, y, call):  # pylint: disable=arguments-differ
        ctx.call, ctx.layout = call, _layout(dy)
        ctx.save_for_backward(dy if call.second else _NONE, x, b, y)
        return _kernel(call, 1, dy, b, x, y, _NONE)

    @staticmethod
    def backward(ctx, ddx):  # pylint: disable=arguments-differ
        call = ctx.call
        ddx = ddx.contiguous(memory_format=ctx.layout)
        dy, x, b, y = ctx.saved_tensors
        g_dy = _FusedGrad.apply(ddx, x, b, y, call) if ctx.needs_input_grad[0] else None
        g_x = g_b = None
        if call.second and (ctx.needs_input_grad[1] or ctx.needs_input_grad[2]):
            g_x = _kernel(call, 2, ddx, b, x, y, dy)
            if ctx.needs_input_grad[2]:
                g_b = _over_all_but(g_x, call.dim)
        return g_dy, g_x, g_b, None, None
