"""Filtered leaky ReLU: bias -> upsample FIR -> gain * lrelu -> clamp -> downsample FIR, as ONE fused op
(operator API of reference torch_utils/ops/filtered_lrelu.py:57-117).

GPU tensors with impl='cuda' run the fused streaming kernel of libsg3hip.so (csrc/sg3_filtered_lrelu.hip) via the
`filtered_lrelu_plugin` shim; configurations it has no kernel for take the generic composition
upfirdn2d -> filtered_lrelu_act_ -> upfirdn2d, still on HIP kernels (same structure as the reference's rc<0
fallback, :224-230).  Gradients w.r.t. x and b use the bit-packed sign tensor written by the forward and the same
op with up/down and the filters swapped (reference :240-269).  CPU tensors / impl='ref' use the composite of
`bias_act` and `upfirdn2d` (reference `_filtered_lrelu_ref`, :122-154).
"""
import os
import warnings

import numpy as np
import torch

from . import bias_act
from . import upfirdn2d
from .. import custom_ops
from .. import misc

_plugin = None


def _init():
    global _plugin
    if _plugin is None:
        _plugin = custom_ops.get_plugin(
            module_name='filtered_lrelu_plugin',
            sources=['sg3_filtered_lrelu.hip', 'sg3_bias_act.hip'],
            source_dir=os.path.join(os.path.dirname(__file__), '..', '..', 'csrc'))
    return True


def _get_filter_size(f):
    if f is None:
        return 1, 1
    assert isinstance(f, torch.Tensor)
    assert 1 <= f.ndim <= 2
    return f.shape[-1], f.shape[0]  # width, height


def _parse_padding(padding):
    if isinstance(padding, int):
        padding = [padding, padding]
    assert isinstance(padding, (list, tuple))
    assert all(isinstance(x, (int, np.integer)) for x in padding)
    padding = [int(x) for x in padding]
    if len(padding) == 2:
        px, py = padding
        padding = [px, px, py, py]
    px0, px1, py0, py1 = padding
    return px0, px1, py0, py1


def _check_scalars(up, down, gain, slope, clamp):
    assert isinstance(up, int) and up >= 1
    assert isinstance(down, int) and down >= 1
    assert gain == float(gain) and gain > 0
    assert slope == float(slope) and slope >= 0
    assert clamp is None or (clamp == float(clamp) and clamp >= 0)


def filtered_lrelu(x, fu=None, fd=None, b=None, up=1, down=1, padding=0, gain=np.sqrt(2), slope=0.2, clamp=None,
                   flip_filter=False, impl='cuda'):
    """For every channel of x [N,C,H,W]:
      1. add bias b[c];  2. insert up-1 zeros after each sample;  3. pad (negative = crop) by `padding`, given with
      respect to the upsampled image as int, [x, y] or [x0, x1, y0, y1];  4. convolve with `fu` (1-D separable,
      2-D, or None);  5.-7. multiply by up^2 * gain, leaky ReLU with `slope`, clamp to +-clamp;  8. convolve with
      `fd`;  9. keep every `down`-th sample.
    fu / fd are float32; b has x's dtype; flip_filter=True means correlation.  Returns [N,C,H',W'] in x's dtype."""
    assert isinstance(x, torch.Tensor)
    assert impl in ['ref', 'cuda']
    if impl == 'cuda' and x.device.type == 'cuda' and _init():
        return _filtered_lrelu_cuda(up=up, down=down, padding=padding, gain=gain, slope=slope, clamp=clamp,
                                    flip_filter=flip_filter).apply(x, fu, fd, b, None, 0, 0)
    return _filtered_lrelu_ref(x, fu=fu, fd=fd, b=b, up=up, down=down, padding=padding, gain=gain, slope=slope,
                               clamp=clamp, flip_filter=flip_filter)


@misc.profiled_function
def _filtered_lrelu_ref(x, fu=None, fd=None, b=None, up=1, down=1, padding=0, gain=np.sqrt(2), slope=0.2, clamp=None,
                        flip_filter=False):
    """The op as a composition of bias_act and upfirdn2d (materialises the whole upsampled buffer)."""
    assert isinstance(x, torch.Tensor) and x.ndim == 4
    fu_w, fu_h = _get_filter_size(fu)
    fd_w, fd_h = _get_filter_size(fd)
    if b is not None:
        assert isinstance(b, torch.Tensor) and b.dtype == x.dtype
        misc.assert_shape(b, [x.shape[1]])
    _check_scalars(up, down, gain, slope, clamp)
    px0, px1, py0, py1 = _parse_padding(padding)
    n, c, in_h, in_w = x.shape
    in_dtype = x.dtype
    out_w = (in_w * up + (px0 + px1) - (fu_w - 1) - (fd_w - 1) + (down - 1)) // down
    out_h = (in_h * up + (py0 + py1) - (fu_h - 1) - (fd_h - 1) + (down - 1)) // down

    x = bias_act.bias_act(x=x, b=b)
    x = upfirdn2d.upfirdn2d(x=x, f=fu, up=up, padding=[px0, px1, py0, py1], gain=up ** 2, flip_filter=flip_filter)
    x = bias_act.bias_act(x=x, act='lrelu', alpha=slope, gain=gain, clamp=clamp)
    x = upfirdn2d.upfirdn2d(x=x, f=fd, down=down, flip_filter=flip_filter)

    misc.assert_shape(x, [n, c, out_h, out_w])
    assert x.dtype == in_dtype
    return x


_filtered_lrelu_cuda_cache = dict()


def _filtered_lrelu_cuda(up=1, down=1, padding=0, gain=np.sqrt(2), slope=0.2, clamp=None, flip_filter=False):
    """autograd.Function around the HIP plugin, cached per parameter tuple."""
    _check_scalars(up, down, gain, slope, clamp)
    px0, px1, py0, py1 = _parse_padding(padding)
    gain, slope = float(gain), float(slope)
    clamp = float(clamp if clamp is not None else 'inf')
    key = (up, down, px0, px1, py0, py1, gain, slope, clamp, flip_filter)
    if key in _filtered_lrelu_cuda_cache:
        return _filtered_lrelu_cuda_cache[key]

    class FilteredLReluCuda(torch.autograd.Function):
        @staticmethod
        def forward(ctx, x, fu, fd, b, si, sx, sy):  # pylint: disable=arguments-differ
            assert isinstance(x, torch.Tensor) and x.ndim == 4
            one = None
            if fu is None or fd is None:
                one = torch.ones([1, 1], dtype=torch.float32, device=x.device)
            fu = one if fu is None else fu
            fd = one if fd is None else fd
            assert 1 <= fu.ndim <= 2
            assert 1 <= fd.ndim <= 2
            # a separable single tap with no resampling is the same as a full 1x1 filter
            if up == 1 and fu.ndim == 1 and fu.shape[0] == 1:
                fu = fu.square()[None]
            if down == 1 and fd.ndim == 1 and fd.shape[0] == 1:
                fd = fd.square()[None]
            if si is None:
                si = torch.empty([0])
            if b is None:
                b = torch.zeros([x.shape[1]], dtype=x.dtype, device=x.device)
            # signs are only worth writing when a gradient will be asked for
            write_signs = (si.numel() == 0) and (x.requires_grad or b.requires_grad)

            strides = [x.stride(i) for i in range(x.ndim) if x.size(i) > 1]
            if any(a < c for a, c in zip(strides[:-1], strides[1:])):
                warnings.warn('low-performance memory layout detected in filtered_lrelu input', RuntimeWarning)

            # (no stream warning: taps travel with each launch, so concurrent streams are safe -- the reference warns
            # at this point because its taps live in one global __constant__ buffer, :216-217)
            if x.dtype in [torch.float16, torch.float32]:
                y, so, return_code = _plugin.filtered_lrelu(x, fu, fd, b, si, up, down, px0, px1, py0, py1, sx, sy,
                                                            gain, slope, clamp, flip_filter, write_signs)
            else:
                return_code = -1

            if return_code < 0:
                # generic composition, all on HIP kernels; keeps only the bit-packed signs for backward
                y = x.add(b.unsqueeze(-1).unsqueeze(-1))
                y = upfirdn2d.upfirdn2d(x=y, f=fu, up=up, padding=[px0, px1, py0, py1], gain=up ** 2, flip_filter=flip_filter)
                so = _plugin.filtered_lrelu_act_(y, si, sx, sy, gain, slope, clamp, write_signs)
                y = upfirdn2d.upfirdn2d(x=y, f=fd, down=down, flip_filter=flip_filter)

            ctx.save_for_backward(fu, fd, (si if si.numel() else so))
            ctx.x_shape = x.shape
            ctx.y_shape = y.shape
            ctx.s_ofs = sx, sy
            return y

        @staticmethod
        def backward(ctx, dy):  # pylint: disable=arguments-differ
            fu, fd, si = ctx.saved_tensors
            _, _, xh, xw = ctx.x_shape
            _, _, yh, yw = ctx.y_shape
            sx, sy = ctx.s_ofs
            dx = db = None
            for i in (1, 2, 4, 5, 6):
                assert not ctx.needs_input_grad[i]

            if ctx.needs_input_grad[0] or ctx.needs_input_grad[3]:
                # adjoint = same op with (up, fu) <-> (down, fd), flipped filters, no clamp, stored signs
                pp = [(fu.shape[-1] - 1) + (fd.shape[-1] - 1) - px0,
                      xw * up - yw * down + px0 - (up - 1),
                      (fu.shape[0] - 1) + (fd.shape[0] - 1) - py0,
                      xh * up - yh * down + py0 - (up - 1)]
                gg = gain * (up ** 2) / (down ** 2)
                sxb = sx - (fu.shape[-1] - 1) + px0
                syb = sy - (fu.shape[0] - 1) + py0
                if not torch.is_grad_enabled() and dy.dtype in (torch.float16, torch.float32):
                    # first-order gradients: the fused adjoint kernel also accumulates the per-channel sum of dx, so the bias
                    # gradient needs no second pass over dx
                    zb = torch.zeros([dy.shape[1]], dtype=dy.dtype, device=dy.device)
                    dxf, _, rc, db_f = _plugin.filtered_lrelu(dy.contiguous(), fd, fu, zb, si, down, up, pp[0], pp[1], pp[2], pp[3], sxb, syb,
                                                              gg, slope, float('inf'), not flip_filter, False, return_sum=ctx.needs_input_grad[3])
                    if rc == 0:
                        dx = dxf
                        if ctx.needs_input_grad[3] and db_f is not None:
                            db = db_f
                if dx is None:
                    dx = _filtered_lrelu_cuda(up=down, down=up, padding=pp, gain=gg, slope=slope, clamp=None,
                                              flip_filter=(not flip_filter)).apply(dy, fd, fu, None, si, sxb, syb)
            if ctx.needs_input_grad[3] and db is None:
                db = dx.sum([0, 2, 3])
            return dx, None, None, db, None, None, None

    _filtered_lrelu_cuda_cache[key] = FilteredLReluCuda
    return FilteredLReluCuda
