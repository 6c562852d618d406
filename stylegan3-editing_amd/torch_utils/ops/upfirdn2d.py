"""Pad / upsample / FIR-filter / downsample for batches of 2-D images
(operator API of reference torch_utils/ops/upfirdn2d.py: setup_filter :71, upfirdn2d :119, filter2d :278,
upsample2d :314, downsample2d :353).

GPU tensors with impl='cuda' run the polyphase gather kernel of libsg3hip.so (csrc/sg3_upfirdn2d.hip) via the
`upfirdn2d_plugin` shim; the gradient is the same op with up/down exchanged and the filter flipped (reference
:240-265).  CPU tensors / impl='ref' use zero-stuffing followed by a depthwise convolution (the definition the
reference states in `_upfirdn2d_ref`, :168-212).  On the StyleGAN3 path this op is only the generic fallback behind
`filtered_lrelu`.

Layout of this module: a `_Geometry` record (factors, padding, flip, gain) built once per call, one module-level
autograd Function that takes it as a non-tensor argument, and the three convenience wrappers expressed through a
single `_same_size_padding` rule.
"""
import os
from collections import namedtuple

import torch

from . import conv2d_gradfix
from ._resample_args import fir_extent, four_sided, xy_factor
from .. import custom_ops
from .. import misc

_plugin = None


def _init():
    global _plugin
    if _plugin is None:
        _plugin = custom_ops.get_plugin(module_name='upfirdn2d_plugin', sources=['sg3_upfirdn2d.hip'],
                                        source_dir=os.path.join(os.path.dirname(__file__), '..', '..', 'csrc'))
    return True


# reference-compatible private names (pickled module source imports them)
_parse_scaling = xy_factor
_parse_padding = four_sided


def _get_filter_size(f):
    fw, fh = fir_extent(f)
    if f is not None:
        misc.assert_shape(f, [fh, fw][:f.ndim])
    return fw, fh


def setup_filter(f, device=torch.device('cpu'), normalize=True, flip_filter=False, gain=1, separable=None):
    """Taps given as list / array / tensor -> the float32 filter tensor `upfirdn2d` expects.

    A 1-D filter of 8 or more taps is kept separable unless `separable=False`; shorter ones are expanded to their outer
    product.  Then: optional normalisation to unit DC gain, optional flip, and `gain` spread evenly over the axes."""
    taps = torch.as_tensor(1 if f is None else f, dtype=torch.float32)
    assert taps.ndim <= 2 and taps.numel() > 0
    taps = taps.reshape(1) if taps.ndim == 0 else taps
    keep_1d = (taps.ndim == 1 and taps.numel() >= 8) if separable is None else bool(separable)
    if taps.ndim == 1 and not keep_1d:
        taps = torch.outer(taps, taps)
    assert taps.ndim == (1 if keep_1d else 2)
    if normalize:
        taps = taps / taps.sum()
    if flip_filter:
        taps = taps.flip(list(range(taps.ndim)))
    return (taps * gain ** (taps.ndim / 2)).to(device=device)


_Geometry = namedtuple('_Geometry', 'upx upy downx downy px0 px1 py0 py1 flip gain')


def _geometry(up, down, padding, flip_filter, gain):
    return _Geometry(*xy_factor(up), *xy_factor(down), *four_sided(padding), bool(flip_filter), gain)


def upfirdn2d(x, f, up=1, down=1, padding=0, flip_filter=False, gain=1, impl='cuda'):
    """Zero-insert upsample by `up`, pad (negative = crop), convolve with `f` (true convolution unless
    flip_filter), keep every `down`-th sample.  Arguments as in the reference op."""
    assert isinstance(x, torch.Tensor)
    assert impl in ['ref', 'cuda']
    g = _geometry(up, down, padding, flip_filter, gain)
    if impl == 'cuda' and x.device.type == 'cuda' and _init():
        return _Polyphase.apply(x, f, g)
    return _upfirdn2d_ref(x, f, geometry=g)


def _unit_filter(x):
    return torch.ones([1, 1], dtype=torch.float32, device=x.device)


def _pad_or_crop(x, g):
    x = torch.nn.functional.pad(x, [max(g.px0, 0), max(g.px1, 0), max(g.py0, 0), max(g.py1, 0)])
    h, w = x.shape[2:]
    return x[:, :, max(-g.py0, 0): h - max(-g.py1, 0), max(-g.px0, 0): w - max(-g.px1, 0)]


@misc.profiled_function
def _upfirdn2d_ref(x, f, up=1, down=1, padding=0, flip_filter=False, gain=1, geometry=None):
    """The op from standard PyTorch ops (slow: it convolves over the zero-stuffed image)."""
    g = geometry or _geometry(up, down, padding, flip_filter, gain)
    assert isinstance(x, torch.Tensor) and x.ndim == 4
    f = _unit_filter(x) if f is None else f
    assert isinstance(f, torch.Tensor) and f.ndim in [1, 2]
    assert f.dtype == torch.float32 and not f.requires_grad
    n, c, h, w = x.shape
    assert w * g.upx + g.px0 + g.px1 >= f.shape[-1] and h * g.upy + g.py0 + g.py1 >= f.shape[0]

    # each sample becomes the top-left corner of an upy x upx cell of zeros
    cells = torch.nn.functional.pad(x.reshape(n, c, h, 1, w, 1), [0, g.upx - 1, 0, 0, 0, g.upy - 1])
    x = _pad_or_crop(cells.reshape(n, c, h * g.upy, w * g.upx), g)

    # conv2d correlates, so a true convolution needs the taps reversed
    taps = (f * g.gain ** (f.ndim / 2)).to(x.dtype)
    taps = taps if g.flip else taps.flip(list(range(taps.ndim)))
    if taps.ndim == 2:
        x = conv2d_gradfix.conv2d(input=x, weight=taps.expand(c, 1, *taps.shape).contiguous(), groups=c)
    else:
        for shape in ((1, -1), (-1, 1)):                       # along x, then along y
            x = conv2d_gradfix.conv2d(input=x, weight=taps.reshape(1, 1, *shape).repeat(c, 1, 1, 1), groups=c)
    return x[:, :, ::g.downy, ::g.downx]


class _Polyphase(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, f, g):  # pylint: disable=arguments-differ
        assert isinstance(x, torch.Tensor) and x.ndim == 4
        f = _unit_filter(x) if f is None else f
        if f.ndim == 1 and f.shape[0] == 1:
            f = f.square().unsqueeze(0)                        # one separable tap is a 1x1 filter
        assert isinstance(f, torch.Tensor) and f.ndim in [1, 2]
        if f.ndim == 2:
            y = _plugin.upfirdn2d(x, f, g.upx, g.upy, g.downx, g.downy, g.px0, g.px1, g.py0, g.py1, g.flip, g.gain)
        else:                                                  # separable: a row pass, then a column pass
            y = _plugin.upfirdn2d(x, f.unsqueeze(0), g.upx, 1, g.downx, 1, g.px0, g.px1, 0, 0, g.flip, 1.0)
            y = _plugin.upfirdn2d(y, f.unsqueeze(1), 1, g.upy, 1, g.downy, 0, 0, g.py0, g.py1, g.flip, g.gain)
        ctx.save_for_backward(f)
        ctx.g, ctx.in_hw = g, tuple(x.shape[2:])
        return y

    @staticmethod
    def backward(ctx, dy):  # pylint: disable=arguments-differ
        assert not ctx.needs_input_grad[1]
        if not ctx.needs_input_grad[0]:
            return None, None, None
        (f,), g = ctx.saved_tensors, ctx.g
        (ih, iw), (oh, ow) = ctx.in_hw, dy.shape[2:]
        fw, fh = fir_extent(f)
        # adjoint: exchange up and down, flip the filter, pad so that the result has the input's size
        adj = _Geometry(g.downx, g.downy, g.upx, g.upy,
                        fw - g.px0 - 1, iw * g.upx - ow * g.downx + g.px0 - g.upx + 1,
                        fh - g.py0 - 1, ih * g.upy - oh * g.downy + g.py0 - g.upy + 1,
                        not g.flip, g.gain)
        return _Polyphase.apply(dy, f, adj), None, None


def _same_size_padding(f, padding, factor_x, factor_y, toward_lo):
    """User padding plus what keeps the output at `input * up / down`: the filter's extent beyond one resampling step,
    split between the two sides with the extra sample on the low side."""
    px0, px1, py0, py1 = four_sided(padding)
    fw, fh = fir_extent(f)
    ex, ey = fw - factor_x, fh - factor_y
    lo = (lambda e: (e + toward_lo) // 2)
    return [px0 + lo(ex), px1 + ex // 2, py0 + lo(ey), py1 + ey // 2]


def filter2d(x, f, padding=0, flip_filter=False, gain=1, impl='cuda'):
    """Filter with `f`; the output keeps the input size (user padding is added on top)."""
    p = _same_size_padding(f, padding, 1, 1, toward_lo=1)
    return upfirdn2d(x, f, padding=p, flip_filter=flip_filter, gain=gain, impl=impl)


def upsample2d(x, f, up=2, padding=0, flip_filter=False, gain=1, impl='cuda'):
    """Upsample by `up` with `f`; the output size is a multiple of the input size."""
    upx, upy = xy_factor(up)
    px0, px1, py0, py1 = four_sided(padding)
    fw, fh = fir_extent(f)
    p = [px0 + (fw + upx - 1) // 2, px1 + (fw - upx) // 2, py0 + (fh + upy - 1) // 2, py1 + (fh - upy) // 2]
    return upfirdn2d(x, f, up=up, padding=p, flip_filter=flip_filter, gain=gain * upx * upy, impl=impl)


def downsample2d(x, f, down=2, padding=0, flip_filter=False, gain=1, impl='cuda'):
    """Downsample by `down` with `f`; the output size is a fraction of the input size."""
    downx, downy = xy_factor(down)
    p = _same_size_padding(f, padding, downx, downy, toward_lo=1)
    return upfirdn2d(x, f, down=down, padding=p, flip_filter=flip_filter, gain=gain, impl=impl)
