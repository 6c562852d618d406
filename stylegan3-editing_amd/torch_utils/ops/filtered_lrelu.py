"""Filtered leaky ReLU: bias -> upsample FIR -> gain * lrelu -> clamp -> downsample FIR, as ONE fused op
(operator API of reference torch_utils/ops/filtered_lrelu.py:57-117).

GPU tensors with impl='cuda' run the fused streaming kernel of libsg3hip.so (csrc/sg3_filtered_lrelu.hip) via the
`filtered_lrelu_plugin` shim; configurations it has no kernel for take the generic composition
upfirdn2d -> filtered_lrelu_act_ -> upfirdn2d, still on HIP kernels (the reference does the same when its plugin
returns a negative code, :224-230).  Gradients w.r.t. x and b use the bit-packed sign tensor written by the forward
and the same op with up/down and the filters exchanged (reference :240-269).  CPU tensors / impl='ref' use the
composition of `bias_act` and `upfirdn2d` (reference `_filtered_lrelu_ref`, :122-154).

Layout of this module: a `_Setup` record (factors, padding, gain, slope, clamp, flip) per call and ONE module-level
autograd Function that takes the record plus an optional (signs, offset) triple; its backward first tries the fused
adjoint kernel that also delivers the bias gradient, and otherwise re-enters itself with the adjoint record.
"""
import math
import os
import warnings
from collections import namedtuple

import torch

from . import bias_act
from . import known_amax
from . import upfirdn2d
from ._resample_args import fir_extent, four_sided
from .. import custom_ops
from .. import misc

_plugin = None


def _init():
    global _plugin
    if _plugin is None:
        _plugin = custom_ops.get_plugin(module_name='filtered_lrelu_plugin',
                                        sources=['sg3_filtered_lrelu.hip', 'sg3_bias_act.hip'],
                                        source_dir=os.path.join(os.path.dirname(__file__), '..', '..', 'csrc'))
    return True


_Setup = namedtuple('_Setup', 'up down px0 px1 py0 py1 gain slope clamp flip')      # clamp: float, inf = none


def _setup(up, down, padding, gain, slope, clamp, flip_filter):
    assert isinstance(up, int) and up >= 1
    assert isinstance(down, int) and down >= 1
    assert gain == float(gain) and gain > 0
    assert slope == float(slope) and slope >= 0
    assert clamp is None or (clamp == float(clamp) and clamp >= 0)
    return _Setup(up, down, *four_sided(padding), float(gain), float(slope),
                  math.inf if clamp is None else float(clamp), bool(flip_filter))


def filtered_lrelu(x, fu=None, fd=None, b=None, up=1, down=1, padding=0, gain=math.sqrt(2), slope=0.2, clamp=None,
                   flip_filter=False, impl='cuda'):
    """For every channel of x [N,C,H,W]:
      1. add bias b[c];  2. insert up-1 zeros after each sample;  3. pad (negative = crop) by `padding`, given with
      respect to the upsampled image as int, [x, y] or [x0, x1, y0, y1];  4. convolve with `fu` (1-D separable,
      2-D, or None);  5.-7. multiply by up^2 * gain, leaky ReLU with `slope`, clamp to +-clamp;  8. convolve with
      `fd`;  9. keep every `down`-th sample.
    fu / fd are float32; b has x's dtype; flip_filter=True means correlation.  Returns [N,C,H',W'] in x's dtype."""
    assert isinstance(x, torch.Tensor)
    assert impl in ['ref', 'cuda']
    cfg = _setup(up, down, padding, gain, slope, clamp, flip_filter)
    if impl == 'cuda' and x.device.type == 'cuda' and _init():
        return _FusedFlrelu.apply(x, fu, fd, b, cfg, None, 0, 0)
    return _filtered_lrelu_ref(x, fu, fd, b, cfg=cfg)


def _out_extent(n_in, cfg, p0, p1, fu_n, fd_n):
    return (n_in * cfg.up + p0 + p1 - (fu_n - 1) - (fd_n - 1) + cfg.down - 1) // cfg.down


@misc.profiled_function
def _filtered_lrelu_ref(x, fu=None, fd=None, b=None, up=1, down=1, padding=0, gain=math.sqrt(2), slope=0.2, clamp=None,
                        flip_filter=False, cfg=None):
    """The op as four separate ops (materialises the whole upsampled image)."""
    cfg = cfg or _setup(up, down, padding, gain, slope, clamp, flip_filter)
    assert isinstance(x, torch.Tensor) and x.ndim == 4
    if b is not None:
        assert isinstance(b, torch.Tensor) and b.dtype == x.dtype
        misc.assert_shape(b, [x.shape[1]])
    (fu_w, fu_h), (fd_w, fd_h) = fir_extent(fu), fir_extent(fd)
    n, c, h, w = x.shape
    want = [n, c, _out_extent(h, cfg, cfg.py0, cfg.py1, fu_h, fd_h), _out_extent(w, cfg, cfg.px0, cfg.px1, fu_w, fd_w)]
    dtype = x.dtype

    y = bias_act.bias_act(x=x, b=b)
    y = upfirdn2d.upfirdn2d(x=y, f=fu, up=cfg.up, padding=[cfg.px0, cfg.px1, cfg.py0, cfg.py1], gain=cfg.up ** 2,
                            flip_filter=cfg.flip)
    y = bias_act.bias_act(x=y, act='lrelu', alpha=cfg.slope, gain=cfg.gain,
                          clamp=None if math.isinf(cfg.clamp) else cfg.clamp)
    y = upfirdn2d.upfirdn2d(x=y, f=fd, down=cfg.down, flip_filter=cfg.flip)

    misc.assert_shape(y, want)
    assert y.dtype == dtype
    return y


def _as_kernel_filter(f, factor, like):
    """None -> 1x1 ones; a separable single tap without resampling -> the equivalent 1x1 filter."""
    if f is None:
        return torch.ones([1, 1], dtype=torch.float32, device=like.device)
    assert 1 <= f.ndim <= 2
    if factor == 1 and f.ndim == 1 and f.shape[0] == 1:
        return f.square()[None]
    return f


def _warn_if_permuted(x):
    steps = [x.stride(i) for i in range(x.ndim) if x.size(i) > 1]
    if any(a < c for a, c in zip(steps, steps[1:])):
        warnings.warn('low-performance memory layout detected in filtered_lrelu input', RuntimeWarning)


def _adjoint(cfg, fu, fd, x_hw, y_hw, sx, sy):
    """Record, and sign offsets, of the op that maps dy to dx: (up, fu) <-> (down, fd), filters flipped, no clamp."""
    (xh, xw), (yh, yw) = x_hw, y_hw
    fu_w, fu_h = fu.shape[-1], fu.shape[0]
    fd_w, fd_h = fd.shape[-1], fd.shape[0]
    adj = _Setup(cfg.down, cfg.up,
                 (fu_w - 1) + (fd_w - 1) - cfg.px0, xw * cfg.up - yw * cfg.down + cfg.px0 - (cfg.up - 1),
                 (fu_h - 1) + (fd_h - 1) - cfg.py0, xh * cfg.up - yh * cfg.down + cfg.py0 - (cfg.up - 1),
                 cfg.gain * cfg.up ** 2 / cfg.down ** 2, cfg.slope, math.inf, not cfg.flip)
    return adj, sx - (fu_w - 1) + cfg.px0, sy - (fu_h - 1) + cfg.py0


_zero_biases = {}


def _zero_bias(c, dtype, device):
    """A [c] vector of zeros (the adjoint pass has no bias), made once per (c, dtype, device): kernels only read it."""
    key = (c, dtype, str(device))
    z = _zero_biases.get(key)
    if z is None:
        z = _zero_biases[key] = torch.zeros([c], dtype=dtype, device=device)
    return z


class _FusedFlrelu(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, fu, fd, b, cfg, si, sx, sy):  # pylint: disable=arguments-differ
        assert isinstance(x, torch.Tensor) and x.ndim == 4
        fu = _as_kernel_filter(fu, cfg.up, x)
        fd = _as_kernel_filter(fd, cfg.down, x)
        si = torch.empty([0]) if si is None else si
        b = torch.zeros([x.shape[1]], dtype=x.dtype, device=x.device) if b is None else b
        # signs are only worth writing when a gradient will be asked for
        write_signs = si.numel() == 0 and (x.requires_grad or b.requires_grad)
        _warn_if_permuted(x)
        pad = (cfg.px0, cfg.px1, cfg.py0, cfg.py1)

        # (no stream warning here: taps travel with each launch, so concurrent streams are safe -- the reference warns
        # at this point because its taps live in one global __constant__ buffer, :216-217)
        code = -1
        if x.dtype in (torch.float16, torch.float32):
            y, so, code = _plugin.filtered_lrelu(x, fu, fd, b, si, cfg.up, cfg.down, *pad, sx, sy,
                                                 cfg.gain, cfg.slope, cfg.clamp, cfg.flip, write_signs)
        if code < 0:
            # generic composition, all on HIP kernels; only the bit-packed signs are kept for backward
            y = x.add(b[:, None, None])
            y = upfirdn2d.upfirdn2d(x=y, f=fu, up=cfg.up, padding=list(pad), gain=cfg.up ** 2, flip_filter=cfg.flip)
            so = _plugin.filtered_lrelu_act_(y, si, sx, sy, cfg.gain, cfg.slope, cfg.clamp, write_signs)
            y = upfirdn2d.upfirdn2d(x=y, f=fd, down=cfg.down, flip_filter=cfg.flip)

        ctx.save_for_backward(fu, fd, si if si.numel() else so)
        ctx.cfg, ctx.sign_ofs = cfg, (sx, sy)
        ctx.x_hw, ctx.y_hw = tuple(x.shape[2:]), tuple(y.shape[2:])
        return y

    @staticmethod
    def backward(ctx, dy):  # pylint: disable=arguments-differ
        want_x, want_b = ctx.needs_input_grad[0], ctx.needs_input_grad[3]
        assert not any(ctx.needs_input_grad[i] for i in (1, 2, 4, 5, 6, 7))
        dx = db = None
        if want_x or want_b:
            fu, fd, signs = ctx.saved_tensors
            adj, sx, sy = _adjoint(ctx.cfg, fu, fd, ctx.x_hw, ctx.y_hw, *ctx.sign_ofs)
            if not torch.is_grad_enabled() and dy.dtype in (torch.float16, torch.float32):
                # first-order gradients: the fused adjoint kernel also accumulates the per-channel sum of dx, so the
                # bias gradient needs no second pass over dx
                zero_b = _zero_bias(int(dy.shape[1]), dy.dtype, dy.device)
                res = _plugin.filtered_lrelu(dy.contiguous(), fd, fu, zero_b, signs, adj.up, adj.down,
                                             adj.px0, adj.px1, adj.py0, adj.py1, sx, sy, adj.gain, adj.slope,
                                             adj.clamp, adj.flip, False, return_sum=True, return_amax=True)
                if res[2] == 0:                     # (y, signs, code, per-channel sums, max |y|)
                    dx, db = res[0], (res[3] if want_b else None)
                    if res[4] is not None:
                        known_amax.attach(dx, res[4])     # the convolution in front scales its gradient operands by it
            if dx is None:
                dx = _FusedFlrelu.apply(dy, fd, fu, None, adj, signs, sx, sy)
            if want_b and db is None:
                db = dx.sum([0, 2, 3])
        return dx, None, None, db, None, None, None, None
