"""GPU parity tests proper: every operator through the C ABI (libsg3hip.so) on a real MI355X against the golden
vectors of the reference and against the CPU oracle on the same seeded inputs.

Tolerances (fp32 I/O, fp32 math; values are O(1)): 2e-5 max-abs for single ops; fp16 I/O: 4e-3 (one fp16 rounding
of in/out).  BASELINE's end-to-end requirement is 1e-4 max-abs on the final image (test_gpu_net.py)."""
import numpy as np
import pytest
import torch

from golden_cases import BIAS_ACT_CASES, FLRELU_CASES, FLRELU_GRAD_CASES, MODCONV_CASES, UPFIRDN_CASES, make_filter, rand
from helpers import golden, maxabs, oracle_design, product_design

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def T(a, dtype=None):
    if a is None:
        return None
    t = torch.from_numpy(np.ascontiguousarray(a)).to(DEV)
    return t if dtype is None else t.to(dtype)


def _flrelu(c, x, b, fu, fd, **kw):
    from torch_utils.ops import filtered_lrelu
    return filtered_lrelu.filtered_lrelu(x, fu=fu, fd=fd, b=b, up=c['up'], down=c['down'], padding=c['padding'], gain=c['gain'],
                                         slope=c['slope'], clamp=c['clamp'], flip_filter=c['flip'], **kw)


def test_library_loaded_and_launch_counter():
    from torch_utils import _sg3abi
    from torch_utils.ops import bias_act
    n0 = _sg3abi.launch_count
    bias_act.bias_act(torch.randn(4, 8, device=DEV), torch.randn(8, device=DEV), act='lrelu')
    assert _sg3abi.launch_count == n0 + 1 and _sg3abi.load().sg3_device_count() >= 1


@pytest.mark.parametrize('name', sorted(FLRELU_CASES))
def test_filtered_lrelu_fp32(name):
    c = FLRELU_CASES[name]
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        y = _flrelu(c, T(rand(11, *c['shape'])), T(rand(12, c['shape'][1])) if c['bias'] else None,
                    T(make_filter(c['fu'], product_design)), T(make_filter(c['fd'], product_design)))
    ref = golden('ops')['flrelu/' + name]
    assert tuple(y.shape) == ref.shape
    assert maxabs(y.cpu().numpy(), ref) <= 2e-5


@pytest.mark.parametrize('name', ['t_up2_dn2', 't_up4_dn2', 't_crit_l13', 't_torgb', 'r_up2_dnrad2'])
def test_filtered_lrelu_fp16(name):
    from oracle import oracle as O
    c = FLRELU_CASES[name]
    x16 = rand(11, *c['shape']).astype(np.float16); b16 = rand(12, c['shape'][1]).astype(np.float16)
    fu, fd = make_filter(c['fu'], oracle_design), make_filter(c['fd'], oracle_design)
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        y = _flrelu(c, T(x16), T(b16), T(fu), T(fd))
    assert y.dtype == torch.float16
    ref = O.filtered_lrelu(x16.astype(np.float32), fu, fd, b16.astype(np.float32), c['up'], c['down'], c['padding'], c['gain'], c['slope'], c['clamp'], c['flip'])
    assert maxabs(y.float().cpu().numpy(), ref) <= 4e-3


@pytest.mark.parametrize('shape,up,taps,pad', [
    ((1, 2, 150, 150), 2, 12, [9, 8, 9, 8]),          # 2 strips (out 148), L6-like
    ((2, 3, 86, 86), 4, 24, [-6, -9, -6, -9]),        # L5-like up4
    ((1, 1, 278, 278), 4, 24, [-6, -9, -6, -9]),      # 5 strips, multiple row chunks
    ((1, 2, 534, 300), 2, 12, [9, 8, 9, 8]),          # non-square, several chunks
    ((1, 1, 130, 1046), 2, 12, [-11, -12, -11, -12]), # full-width critically sampled rows (9 strips)
    ((3, 5, 38, 38), 2, 12, [9, 8, 9, 8]),
])
def test_filtered_lrelu_stream_kernel_sizes(shape, up, taps, pad):
    """The streaming kernel across strip / chunk boundaries, against the oracle."""
    from oracle import oracle as O
    fs = 64
    fu = O.design_lowpass_filter(taps, 4.0, 8.0, fs * up / 2)
    fd = O.design_lowpass_filter(12, 5.0, 9.0, fs)
    x = rand(3, *shape); b = rand(4, shape[1])
    c = dict(up=up, down=2, padding=pad, gain=float(np.sqrt(2)), slope=0.2, clamp=256, flip=False)
    y = _flrelu(c, T(x), T(b), T(fu), T(fd))
    ref = O.filtered_lrelu(x, fu, fd, b, up, 2, pad, c['gain'], 0.2, 256, False)
    assert tuple(y.shape) == ref.shape
    assert maxabs(y.cpu().numpy(), ref) <= 2e-5


@pytest.mark.parametrize('shape,up,taps,pad', [
    ((1, 2, 148, 148), 2, 12, [11, 10, 11, 10]),       # config-R up 2 (1x1 conv: no conv halo), 2 strips
    ((2, 3, 84, 84), 4, 24, [-2, -5, -2, -5]),          # config-R up 4
    ((1, 1, 276, 300), 4, 24, [-2, -5, -2, -5]),        # several strips and row chunks
    ((1, 2, 532, 200), 2, 12, [11, 10, 11, 10]),
])
def test_filtered_lrelu_radial_stream_kernel_sizes(shape, up, taps, pad):
    """Streaming kernel with the full 12x12 radial down filter (config R) across strip / chunk boundaries."""
    from oracle import oracle as O
    fs = 64
    fu = O.design_lowpass_filter(taps, 4.0, 8.0, fs * up / 2)
    fd = O.design_lowpass_filter(12, 5.0, 9.0, fs, radial=True)
    noise = 0.01 * np.random.RandomState(8).rand(12, 12).astype(np.float32)
    x = rand(3, *shape); b = rand(4, shape[1])
    # general taps (symmetry broken: exposes flips), then rows that read the same in both directions (what the radial design
    # produces; the kernel then adds the two samples under a shared tap first) with the vertical symmetry still broken
    for fd_case in ((fd + noise).astype(np.float32), (fd + noise + noise[:, ::-1]).astype(np.float32)):
        for flip in (False, True):
            c = dict(up=up, down=2, padding=pad, gain=float(np.sqrt(2)), slope=0.2, clamp=256, flip=flip)
            y = _flrelu(c, T(x), T(b), T(fu), T(fd_case))
            ref = O.filtered_lrelu(x, fu, fd_case, b, up, 2, pad, c['gain'], 0.2, 256, flip)
            assert tuple(y.shape) == ref.shape
            assert maxabs(y.cpu().numpy(), ref) <= 2e-5


def test_filtered_lrelu_strided_input_and_bias():
    """Element strides are honoured (reference filtered_lrelu.cpp:127-134): channel-sliced and row-padded views."""
    from oracle import oracle as O
    c = FLRELU_CASES['t_up2_dn2']
    fu, fd = make_filter(c['fu'], oracle_design), make_filter(c['fd'], oracle_design)
    big = rand(5, 2, 6, 40, 44)
    xv = T(big)[:, 1:4, 1:39, 3:41]                   # non-contiguous view, unit innermost stride
    bv = T(rand(6, 6))[::2]                           # strided bias
    y = _flrelu(c, xv, bv, T(fu), T(fd))
    ref = O.filtered_lrelu(np.ascontiguousarray(big[:, 1:4, 1:39, 3:41]), fu, fd, rand(6, 6)[::2].copy(), 2, 2, c['padding'], c['gain'], 0.2, 256, False)
    assert maxabs(y.cpu().numpy(), ref) <= 2e-5
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        ycl = _flrelu(c, xv.contiguous(memory_format=torch.channels_last), bv, T(fu), T(fd))   # generic path
    assert maxabs(ycl.cpu().numpy(), ref) <= 2e-5


@pytest.mark.parametrize('name', FLRELU_GRAD_CASES)
def test_filtered_lrelu_backward(name):
    """dx, db through the sign-tensor backward (forward writes 2-bit signs, backward = same op with up<->down)."""
    c = FLRELU_CASES[name]
    x = T(rand(11, *c['shape'])).requires_grad_(True)
    b = T(rand(12, c['shape'][1])).requires_grad_(True) if c['bias'] else None
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        y = _flrelu(c, x, b, T(make_filter(c['fu'], product_design)), T(make_filter(c['fd'], product_design)))
        (y * T(rand(13, *y.shape))).sum().backward()
    g = golden('grads')
    assert maxabs(y.detach().cpu().numpy(), golden('ops')['flrelu/' + name]) <= 2e-5
    assert maxabs(x.grad.cpu().numpy(), g[name + '/dx']) <= 5e-5
    if b is not None:
        assert maxabs(b.grad.cpu().numpy(), g[name + '/db']) <= 2e-3


@pytest.mark.parametrize('shape,up,taps,pad,clamp', [
    ((1, 2, 150, 150), 2, 12, [9, 8, 9, 8], 256),            # T up-2 layer: 2 strips forward; adjoint down 2
    ((2, 2, 86, 86), 4, 24, [-6, -9, -6, -9], 256),           # T up-4 layer: adjoint is up 2 / down 4, 2+ strips of 58
    ((1, 1, 278, 130), 2, 12, [9, 8, 9, 8], 0.7),             # several row chunks, tight clamp (code 2 frequent)
    ((1, 1, 120, 278), 4, 24, [-6, -9, -6, -9], None),        # no clamp, wide: 5 adjoint strips
])
def test_filtered_lrelu_fused_sign_kernels(shape, up, taps, pad, clamp):
    """Training forward (fused kernel writing the 2-bit sign tensor) and its adjoint (fused kernel reading it, incl. the
    up-2 / down-4 form) against autograd through the reference formulation on the CPU (`impl='ref'`)."""
    from oracle import oracle as O
    from torch_utils.ops import filtered_lrelu as fl
    fl._init()
    fs = 64
    fu = O.design_lowpass_filter(taps, 4.0, 8.0, fs * up / 2)
    fd = O.design_lowpass_filter(12, 5.0, 9.0, fs)
    xn, bn = rand(3, *shape), rand(4, shape[1])
    kw = dict(up=up, down=2, padding=pad, gain=float(np.sqrt(2)), slope=0.2, clamp=clamp, flip_filter=False)
    # the fused kernel accepts the sign-writing call (return code 0) and produces the sign tensor
    cl = float('inf') if clamp is None else clamp
    y0, so, rc = fl._plugin.filtered_lrelu(T(xn), T(fu), T(fd), T(bn), torch.empty(0), up, 2, *pad, 0, 0, kw['gain'], 0.2, cl, False, True)
    assert rc == 0 and so.dtype == torch.uint8 and so.numel() > 0
    xr = torch.from_numpy(xn).requires_grad_(True); br = torch.from_numpy(bn).requires_grad_(True)
    yr = fl.filtered_lrelu(xr, torch.from_numpy(fu), torch.from_numpy(fd), br, impl='ref', **kw)
    gy = rand(5, *yr.shape)
    (yr * torch.from_numpy(gy)).sum().backward()
    x = T(xn).requires_grad_(True); b = T(bn).requires_grad_(True)
    y = fl.filtered_lrelu(x, T(fu), T(fd), b, **kw)
    (y * T(gy)).sum().backward()
    assert maxabs(y0.cpu().numpy(), yr.detach().numpy()) <= 2e-5
    assert maxabs(y.detach().cpu().numpy(), yr.detach().numpy()) <= 2e-5
    assert maxabs(x.grad.cpu().numpy(), xr.grad.numpy()) <= 5e-5
    assert maxabs(b.grad.cpu().numpy(), br.grad.numpy()) <= 2e-3 * max(1.0, float(br.grad.abs().max()))
    # sign codes against the generic composition's (upfirdn2d + filtered_lrelu_act_): equal except where the upsampled
    # value is within rounding of zero / of the clamp
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        from torch_utils.ops import upfirdn2d
        u = upfirdn2d.upfirdn2d(T(xn) + T(bn)[None, :, None, None], T(fu), up=up, padding=pad, gain=up ** 2)
    so_ref = fl._plugin.filtered_lrelu_act_(u.clone(), torch.empty(0), 0, 0, kw['gain'], 0.2, cl, True)
    assert so.shape == so_ref.shape
    wact = u.shape[3] // 4                                   # whole bytes inside the active width
    diff = (so[:, :, :u.shape[2], :wact] != so_ref[:, :, :u.shape[2], :wact]).float().mean()
    assert float(diff) <= 1e-4, float(diff)


@pytest.mark.parametrize('shape,up,taps,pad,clamp', [
    ((1, 2, 84, 150), 2, 12, [11, 10, 11, 10], 1.5),            # config-R up-2 layer; adjoint: 12x12 up filter, down 2
    ((2, 2, 84, 84), 4, 24, [-2, -5, -2, -5], 256),             # config-R up-4 layer; adjoint: 12x12 up filter, down 4
    ((1, 1, 150, 276), 4, 24, [-2, -5, -2, -5], None),          # several strips / row chunks
])
@pytest.mark.parametrize('mirror', [False, True])
def test_filtered_lrelu_radial_training_kernels(shape, up, taps, pad, clamp, mirror):
    """Config-R training: the fused radial kernel writes the sign tensor and its adjoint (2-D 12x12 UP filter streamed from
    the scalar cache, sign read, separable down 2 / 4) runs fused too; output and gradients against autograd through the
    reference formulation on the CPU."""
    from oracle import oracle as O
    from torch_utils.ops import filtered_lrelu as fl
    from torch_utils import _sg3abi
    fl._init()
    fu = O.design_lowpass_filter(taps, 4.0, 8.0, 64.0 * up / 2)
    fd = O.design_lowpass_filter(12, 5.0, 9.0, 64.0, radial=True)
    noise = 0.01 * np.random.RandomState(8).rand(12, 12).astype(np.float32)
    # break the symmetry (exposes flips); `mirror` keeps each row palindromic, which selects the folding forward kernel
    fd = (fd + noise + (noise[:, ::-1] if mirror else 0)).astype(np.float32)
    xn, bn = rand(3, *shape), rand(4, shape[1])
    for flip in (False, True):
        kw = dict(up=up, down=2, padding=pad, gain=float(np.sqrt(2)), slope=0.2, clamp=clamp, flip_filter=flip)
        cl = float('inf') if clamp is None else clamp
        y0, so, rc = fl._plugin.filtered_lrelu(T(xn), T(fu), T(fd), T(bn), torch.empty(0), up, 2, *pad, 0, 0, kw['gain'], 0.2, cl, flip, True)
        assert rc == 0 and so.numel() > 0
        assert _sg3abi.load().sg3_filtered_lrelu_has_kernel(2, up, 12, 12, taps, 0) == 1      # the adjoint's shape
        xr = torch.from_numpy(xn).requires_grad_(True); br = torch.from_numpy(bn).requires_grad_(True)
        yr = fl.filtered_lrelu(xr, torch.from_numpy(fu), torch.from_numpy(fd), br, impl='ref', **kw)
        gy = rand(5, *yr.shape)
        (yr * torch.from_numpy(gy)).sum().backward()
        x = T(xn).requires_grad_(True); b = T(bn).requires_grad_(True)
        y = fl.filtered_lrelu(x, T(fu), T(fd), b, **kw)
        (y * T(gy)).sum().backward()
        assert maxabs(y0.cpu().numpy(), yr.detach().numpy()) <= 2e-5 and maxabs(y.detach().cpu().numpy(), yr.detach().numpy()) <= 2e-5
        assert maxabs(x.grad.cpu().numpy(), xr.grad.numpy()) <= 5e-5, flip
        assert maxabs(b.grad.cpu().numpy(), br.grad.numpy()) <= 2e-3 * max(1.0, float(br.grad.abs().max()))


@pytest.mark.parametrize('shape,up,taps,pad,radial', [
    ((1, 2, 150, 150), 2, 12, [9, 8, 9, 8], False),           # T up-2 (adjoint: up 2 / down 2)
    ((2, 2, 86, 86), 4, 24, [-6, -9, -6, -9], False),          # T up-4 (adjoint: up 2 / down 4)
    ((1, 2, 84, 150), 2, 12, [11, 10, 11, 10], True),          # R up-2, 12x12 down filter (adjoint: 12x12 up filter)
])
def test_filtered_lrelu_fused_sign_kernels_fp16_io(shape, up, taps, pad, radial):
    """fp16 activations through the sign-writing forward and the sign-reading adjoint (the reference's default mixed-precision
    PTI, filtered_lrelu.py:198-269 with x.dtype == float16): against the fp32 CPU formulation fed with the same fp16-rounded
    inputs; tolerance = one fp16 rounding of the result (values are O(1..4))."""
    from oracle import oracle as O
    from torch_utils.ops import filtered_lrelu as fl
    fl._init()
    fu = O.design_lowpass_filter(taps, 4.0, 8.0, 64.0 * up / 2)
    fd = O.design_lowpass_filter(12, 5.0, 9.0, 64.0, radial=radial)
    xh = torch.from_numpy(rand(3, *shape)).half()
    bh = torch.from_numpy(rand(4, shape[1])).half()
    kw = dict(up=up, down=2, padding=pad, gain=float(np.sqrt(2)), slope=0.2, clamp=256, flip_filter=False)
    y0, so, rc = fl._plugin.filtered_lrelu(xh.to(DEV), T(fu), T(fd), bh.to(DEV), torch.empty(0), up, 2, *pad, 0, 0, kw['gain'], 0.2, 256.0, False, True)
    assert rc == 0 and y0.dtype == torch.float16 and so.dtype == torch.uint8 and so.numel() > 0
    xr = xh.float().requires_grad_(True); br = bh.float().requires_grad_(True)
    yr = fl.filtered_lrelu(xr, torch.from_numpy(fu), torch.from_numpy(fd), br, impl='ref', **kw)
    gy = torch.from_numpy(rand(5, *yr.shape)).half()
    (yr * gy.float()).sum().backward()
    x = xh.to(DEV).requires_grad_(True); b = bh.to(DEV).requires_grad_(True)
    y = fl.filtered_lrelu(x, T(fu), T(fd), b, **kw)
    assert y.dtype == torch.float16
    (y.float() * gy.to(DEV).float()).sum().backward()
    scale = max(1.0, float(yr.detach().abs().max()))
    assert maxabs(y0.float().cpu().numpy(), yr.detach().numpy()) <= 1e-3 * scale
    assert maxabs(y.detach().float().cpu().numpy(), yr.detach().numpy()) <= 1e-3 * scale
    assert x.grad.dtype == torch.float16
    assert maxabs(x.grad.float().cpu().numpy(), xr.grad.numpy()) <= 1e-3 * max(1.0, float(xr.grad.abs().max()))
    assert maxabs(b.grad.float().cpu().numpy(), br.grad.numpy()) <= 4e-3 * max(1.0, float(br.grad.abs().max()))


def test_filtered_lrelu_act_signs_roundtrip():
    """filtered_lrelu_act_: written signs reproduce the activation derivative when read back."""
    from torch_utils.ops import filtered_lrelu
    filtered_lrelu._init()
    P = filtered_lrelu._plugin
    x0 = T(rand(7, 2, 3, 9, 21) * 2)
    x = x0.clone()
    so = P.filtered_lrelu_act_(x, torch.empty(0), 0, 0, 1.5, 0.2, 1.0, True)
    assert tuple(so.shape) == (2, 3, 9, 8) and so.dtype == torch.uint8
    v = x0 * 1.5
    ref = torch.where(v < 0, v * 0.2, v).clamp(-1, 1)
    assert maxabs(x.cpu().numpy(), ref.cpu().numpy()) <= 1e-6
    g = torch.ones_like(x0)
    P.filtered_lrelu_act_(g, so, 0, 0, 1.5, 0.2, 1.0, False)
    lin = torch.where(v < 0, torch.full_like(v, 0.2), torch.ones_like(v))
    dref = torch.where(lin * v.abs() > 1.0, torch.zeros_like(v), 1.5 * lin)
    assert maxabs(g.cpu().numpy(), dref.cpu().numpy()) <= 1e-6


@pytest.mark.parametrize('name', sorted(UPFIRDN_CASES))
def test_upfirdn2d(name):
    from torch_utils.ops import upfirdn2d
    c = UPFIRDN_CASES[name]
    x = T(rand(21, *c['shape'])).requires_grad_(True)
    y = upfirdn2d.upfirdn2d(x, T(make_filter(c['f'], product_design)), up=c['up'], down=c['down'], padding=c['padding'], flip_filter=c['flip'], gain=c['gain'])
    assert maxabs(y.detach().cpu().numpy(), golden('ops')['upfirdn/' + name]) <= 1e-5
    # adjoint test: <y, gy> gradient equals the ref path's gradient
    gy = T(rand(22, *y.shape))
    (y * gy).sum().backward()
    xr = torch.from_numpy(rand(21, *c['shape'])).requires_grad_(True)
    f = make_filter(c['f'], product_design)
    yr = upfirdn2d.upfirdn2d(xr, None if f is None else torch.from_numpy(f), up=c['up'], down=c['down'], padding=c['padding'], flip_filter=c['flip'], gain=c['gain'])
    (yr * gy.cpu()).sum().backward()
    assert maxabs(x.grad.cpu().numpy(), xr.grad.numpy()) <= 1e-5


@pytest.mark.parametrize('dtype', [torch.float32, torch.float16, torch.float64])
def test_upfirdn2d_dtypes(dtype):
    from torch_utils.ops import upfirdn2d
    c = UPFIRDN_CASES['sep_up2']
    x = torch.from_numpy(rand(21, *c['shape'])).to(dtype)
    f = torch.from_numpy(make_filter(c['f'], product_design))
    y = upfirdn2d.upfirdn2d(x.to(DEV), f.to(DEV), up=c['up'], down=c['down'], padding=c['padding'], gain=c['gain'])
    ref = upfirdn2d.upfirdn2d(x.double(), f, up=c['up'], down=c['down'], padding=c['padding'], gain=c['gain'])
    assert y.dtype == dtype
    assert maxabs(y.double().cpu().numpy(), ref.numpy()) <= {torch.float32: 1e-5, torch.float16: 4e-3, torch.float64: 1e-6}[dtype]


@pytest.mark.parametrize('name', sorted(BIAS_ACT_CASES))
def test_bias_act(name):
    from torch_utils.ops import bias_act
    c = BIAS_ACT_CASES[name]
    x = T(rand(31, *c['shape']) * 2); b = T(rand(32, c['shape'][c['dim']])) if c['bias'] else None
    y = bias_act.bias_act(x, b, dim=c['dim'], act=c['act'], alpha=c['alpha'], gain=c['gain'], clamp=c['clamp'])
    assert maxabs(y.cpu().numpy(), golden('ops')['bias_act/' + name]) <= 2e-6


@pytest.mark.parametrize('act', ['linear', 'relu', 'lrelu', 'tanh', 'sigmoid', 'elu', 'selu', 'softplus', 'swish'])
def test_bias_act_gradients(act):
    """First and second order gradients of the HIP op equal autograd of the pure-PyTorch definition (fp64)."""
    from torch_utils.ops import bias_act
    xn, bn = rand(33, 5, 6, 4).astype(np.float64), rand(34, 6).astype(np.float64)
    outs = []
    for dev, impl in ((DEV, 'cuda'), ('cpu', 'ref')):
        x = torch.from_numpy(xn).to(dev).requires_grad_(True); b = torch.from_numpy(bn).to(dev).requires_grad_(True)
        # clamp exactly representable in fp32 (the ABI carries it as float).  'linear' saves no output for backward in the
        # reference either (bias_act.py:23, ref=''), so its gradient ignores the clamp: test it unclamped.
        y = bias_act.bias_act(x, b, dim=1, act=act, clamp=(None if act == 'linear' else 1.25), impl=impl)
        gx, gb = torch.autograd.grad((y * y).sum(), [x, b], create_graph=True)
        ggx, = torch.autograd.grad((gx * gx).sum() + (gb * gb).sum(), [x])
        outs.append([t.detach().cpu().numpy() for t in (y, gx, gb, ggx)])
    for a, r in zip(*outs):
        assert maxabs(a, r) <= 1e-7 * max(1.0, float(np.abs(r).max()))   # gain (sqrt 2) crosses the ABI as fp32, as in the reference


@pytest.mark.parametrize('name', sorted(MODCONV_CASES))
def test_modulated_conv2d(name):
    from models.stylegan3.networks_stylegan3 import modulated_conv2d
    c = MODCONV_CASES[name]
    x = T(rand(41, c['n'], c['ci'], c['h'], c['w'])); w = T(rand(42, c['co'], c['ci'], c['k'], c['k'])); s = T(rand(43, c['n'], c['ci']) + 1)
    ig = None if c['input_gain'] is None else torch.tensor(c['input_gain'], device=DEV)
    y = modulated_conv2d(x, w, s, demodulate=c['demodulate'], padding=c['k'] - 1, input_gain=ig)
    ref = golden('ops')['modconv/' + name]
    assert tuple(y.shape) == ref.shape
    assert maxabs(y.cpu().numpy(), ref) <= 2e-5 * max(1.0, float(np.abs(ref).max()))


@pytest.mark.parametrize('n,ci,co,h,k', [(2, 323, 203, 22, 3), (1, 128, 81, 40, 3), (2, 51, 32, 70, 3), (1, 32, 3, 64, 1), (2, 512, 512, 12, 3),
                                         (2, 100, 161, 30, 1), (1, 64, 64, 33, 1)])
def test_modulated_conv2d_sizes(n, ci, co, h, k):
    """Odd channel counts / tile tails against the CPU oracle; also the [N,I] and [I] input-gain forms and padding 0."""
    from oracle import oracle as O
    from models.stylegan3.networks_stylegan3 import modulated_conv2d
    x = rand(51, n, ci, h, h + 3); w = rand(52, co, ci, k, k); s = rand(53, n, ci) + 1
    y = modulated_conv2d(T(x), T(w), T(s), demodulate=True, padding=k - 1, input_gain=torch.tensor(0.8, device=DEV))
    ref = O.modulated_conv2d(x, w, s, True, k - 1, 0.8)
    assert maxabs(y.cpu().numpy(), ref) <= 3e-5 * max(1.0, float(np.abs(ref).max()))
    gvec = rand(54, ci) * 0.1 + 1
    y2 = modulated_conv2d(T(x), T(w), T(s), demodulate=False, padding=0, input_gain=T(gvec))
    ref2 = O.modulated_conv2d(x * gvec[None, :, None, None], w, s, False, 0, None)
    assert maxabs(y2.cpu().numpy(), ref2) <= 3e-5 * max(1.0, float(np.abs(ref2).max()))


@pytest.mark.parametrize('k', [3, 1])
def test_modulated_conv2d_plane_beyond_descriptor_offsets(k):
    """Output planes so large that (padded O + 32) * plane bytes passes 2^31: the kernels leave the descriptor-store
    epilogue for 64-bit addresses.  Size-independent check: an interior window of the big result equals, bit for bit, the
    convolution of the matching input crop (which takes the descriptor path)."""
    from torch_utils.ops import modulated_conv as mc
    ci, co = 16, (32 if k == 3 else 64)
    side = 2960 if k == 3 else 2400                       # 64 * 2962^2 * 4 B = 2.25 GB;  (64 + 32) * 2400^2 * 4 B = 2.2 GB
    g = torch.Generator(device=DEV).manual_seed(5)
    x = (torch.rand([1, ci, side, side], device=DEV, generator=g) * 2 - 1) * 3
    w = T(rand(81, co, ci, k, k)); s = T(rand(82, 1, ci) + 1)
    kw = dict(demodulate=True, padding=k - 1, input_gain=torch.tensor(0.9, device=DEV), x_bound=4.0)
    big = mc.modulated_conv2d(x, w, s, **kw)
    assert tuple(big.shape) == (1, co, side + k - 1, side + k - 1)
    for (y0, x0) in ((0, 0), (1500, 1777), (side - 300, side - 300)):
        crop = x[:, :, y0:y0 + 300, x0:x0 + 300].contiguous()
        small = mc.modulated_conv2d(crop, w, s, **kw)
        m = k - 1                                          # rows / columns of `small` that see the crop's zero padding
        a = big[:, :, y0 + 2 * m:y0 + 300 - m + m, x0 + 2 * m:x0 + 300 - m + m]
        b = small[:, :, 2 * m:300, 2 * m:300]
        assert a.shape == b.shape and torch.equal(a, b), (k, y0, x0)
    del big, x
    torch.cuda.empty_cache()


@pytest.mark.parametrize('n,ci,co,h', [(2, 323, 203, 22), (1, 128, 81, 40), (2, 51, 32, 70), (2, 512, 512, 12), (1, 203, 128, 37), (3, 81, 51, 50),
                                      (4, 16, 832, 36),         # 520 eight-row tiles -> the ten-row tile (one round of 416)
                                      (8, 96, 100, 36), (2, 70, 512, 52), (1, 64, 64, 35), (8, 512, 512, 36)])   # narrow planes: flat pixel runs (2 - 4 per wave)
def test_modulated_conv2d_split_precision(n, ci, co, h):
    """fp16 hi/lo split on the fp16 matrix cores (x_bound given) is fp32-equivalent: compared with the fp64 result of
    the oracle, its error is of the same order as the exact-fp32 MFMA kernel's.  Also large styles (power-of-two
    rescale path) and a bound violated by nothing."""
    from oracle import oracle as O
    from torch_utils.ops import modulated_conv as mc
    x = np.clip(rand(71, n, ci, h, h + 3) * 40, -256, 256).astype(np.float32); w = rand(72, co, ci, 3, 3); s = rand(73, n, ci) + 1
    ref = O.modulated_conv2d(x.astype(np.float64), w.astype(np.float64), s.astype(np.float64), True, 2, 0.8)
    scale = max(1.0, float(np.abs(ref).max()))
    errs = {}
    for prec in ('f16x3', 'fp32'):
        mc.precision = prec
        try:
            y = mc.modulated_conv2d(T(x), T(w), T(s), demodulate=True, padding=2, input_gain=torch.tensor(0.8, device=DEV), x_bound=256.0)
        finally:
            mc.precision = 'f16x3'
        errs[prec] = maxabs(y.cpu().numpy(), ref) / scale
    print('relative max errors vs fp64:', errs)
    assert errs['fp32'] <= 5e-6 and errs['f16x3'] <= 5e-6, errs     # K up to 4608 fp32 accumulation
    # styles of magnitude 1e3: |x * s| would overflow fp16 without the per-sample power-of-two rescale
    s_big = (s * 1000).astype(np.float32)
    y = mc.modulated_conv2d(T(x), T(w), T(s_big), demodulate=False, padding=2, input_gain=None, x_bound=256.0)
    ref2 = O.modulated_conv2d(x.astype(np.float64), w.astype(np.float64), s_big.astype(np.float64), False, 2, None)
    assert bool(torch.isfinite(y).all())
    assert maxabs(y.cpu().numpy(), ref2) <= 3e-6 * float(np.abs(ref2).max())       # fp32 accumulation over K up to 4608 (2.1e-6 seen)


@pytest.mark.parametrize('n,ci,co,h', [(2, 645, 406, 20), (1, 1024, 1024, 12), (2, 161, 102, 70), (1, 64, 64, 33), (3, 102, 64, 50), (1, 17, 200, 9)])
def test_modulated_conv2d_split_precision_1x1(n, ci, co, h):
    """Config-R 1x1 kernels through the split-precision GEMM kernel (odd channel counts, tile tails, fp16 I/O)."""
    from oracle import oracle as O
    from torch_utils.ops import modulated_conv as mc
    x = np.clip(rand(81, n, ci, h, h + 3) * 40, -256, 256).astype(np.float32); w = rand(82, co, ci, 1, 1); s = rand(83, n, ci) + 1
    ref = O.modulated_conv2d(x.astype(np.float64), w.astype(np.float64), s.astype(np.float64), True, 0, 0.8)
    scale = max(1.0, float(np.abs(ref).max()))
    errs = {}
    for prec in ('f16x3', 'fp32'):
        mc.precision = prec
        try:
            y = mc.modulated_conv2d(T(x), T(w), T(s), demodulate=True, padding=0, input_gain=torch.tensor(0.8, device=DEV), x_bound=256.0)
        finally:
            mc.precision = 'f16x3'
        assert tuple(y.shape) == ref.shape
        errs[prec] = maxabs(y.cpu().numpy(), ref) / scale
    print('relative max errors vs fp64:', errs)
    assert errs['fp32'] <= 5e-6 and errs['f16x3'] <= 5e-6, errs
    s_big = (s * 1000).astype(np.float32)
    y = mc.modulated_conv2d(T(x), T(w), T(s_big), demodulate=False, padding=0, input_gain=None, x_bound=256.0)
    ref2 = O.modulated_conv2d(x.astype(np.float64), w.astype(np.float64), s_big.astype(np.float64), False, 0, None)
    assert bool(torch.isfinite(y).all())
    assert maxabs(y.cpu().numpy(), ref2) <= 2e-6 * float(np.abs(ref2).max())
    yh = mc.modulated_conv2d(T(x.astype(np.float16)), T(w), T(s), demodulate=True, padding=0, input_gain=None, x_bound=256.0)
    refh = O.modulated_conv2d(x.astype(np.float16).astype(np.float64), w.astype(np.float64), s.astype(np.float64), True, 0, None)
    assert yh.dtype == torch.float16 and maxabs(yh.float().cpu().numpy(), refh) <= 2e-3 * max(1.0, float(np.abs(refh).max()))


@pytest.mark.parametrize('n,ci,co,h,k', [(2, 203, 128, 37, 3), (1, 81, 51, 50, 3), (2, 645, 406, 20, 1), (1, 64, 64, 33, 1),
                                         (1, 64, 128, 131, 3), (1, 51, 32, 140, 3), (1, 81, 51, 127, 3), (1, 33, 70, 150, 3)])
def test_modulated_conv2d_fp16_form(n, ci, co, h, k):
    """fp16 tensors take the single-MFMA fp16 form (SG3_CONV_F16): equal to the fp64 result for operands rounded to
    fp16 up to fp16 rounding of (x * s), of the weights and of the output.  Outputs of 128 rows or more take the taller row
    stacks of that form (six rows per wave in the 64-channel tile, five in the 32-channel one, with and without the packed K tail)."""
    from oracle import oracle as O
    from torch_utils.ops import modulated_conv as mc
    x = np.clip(rand(91, n, ci, h, h + 3) * 20, -256, 256).astype(np.float16); w = rand(92, co, ci, k, k); s = rand(93, n, ci) + 1
    ref = O.modulated_conv2d(x.astype(np.float64), w.astype(np.float64), s.astype(np.float64), True, k - 1, 0.8)
    y = mc.modulated_conv2d(T(x), T(w), T(s), demodulate=True, padding=k - 1, input_gain=torch.tensor(0.8, device=DEV))
    assert y.dtype == torch.float16 and tuple(y.shape) == ref.shape
    err = np.abs(y.float().cpu().numpy() - ref)
    scale = float(np.abs(ref).max())
    assert err.max() <= 4e-3 * scale and err.mean() <= 4e-4 * scale
    mc.precision = 'fp32'
    try:
        y32 = mc.modulated_conv2d(T(x), T(w), T(s), demodulate=True, padding=k - 1, input_gain=torch.tensor(0.8, device=DEV))
    finally:
        mc.precision = 'f16x3'
    assert maxabs(y32.float().cpu().numpy(), ref) <= 1e-3 * scale                   # exact products: only the fp16 output rounding


@pytest.mark.parametrize('n,ci,co,h,w,k,pad', [(2, 70, 100, 21, 37, 3, 2), (1, 32, 32, 70, 45, 3, 2), (1, 130, 64, 40, 40, 3, 1), (2, 64, 3, 33, 50, 1, 0),
                                              (1, 200, 161, 30, 30, 1, 0), (1, 16, 16, 150, 150, 3, 2),
                                              (1, 51, 32, 60, 44, 3, 2), (1, 81, 51, 40, 36, 3, 2)])      # thin tiles: the waves share the taps of one / two blocks
def test_conv2d_wgrad_kernel(n, ci, co, h, w, k, pad):
    """Per-sample weight gradient kernel against the fp64 definition dW[n,o,i,ky,kx] = sum dy[n,o,y,x] xpad[n,i,y+ky,x+kx];
    operands of very different magnitude (activations O(100), gradients O(1e-6)) exercise the power-of-two scaling."""
    from torch_utils.ops import modulated_conv as mc
    x = (rand(101, n, ci, h, w) * 60).astype(np.float32)
    oh, ow = h + 2 * pad - k + 1, w + 2 * pad - k + 1
    dy = (rand(102, n, co, oh, ow) * 3e-6).astype(np.float32)
    got = mc._weight_gradient(T(x), T(dy), k, pad).cpu().numpy()
    xp = np.pad(x.astype(np.float64), ((0, 0), (0, 0), (pad, pad), (pad, pad)))
    ref = np.zeros((n, co, ci, k, k))
    for ky in range(k):
        for kx in range(k):
            ref[:, :, :, ky, kx] = np.einsum('noyx,niyx->noi', dy.astype(np.float64), xp[:, :, ky:ky + oh, kx:kx + ow])
    assert got.shape == ref.shape
    assert maxabs(got, ref) <= 3e-6 * float(np.abs(ref).max()), (maxabs(got, ref), float(np.abs(ref).max()))
    # fp16 tensors
    goth = mc._weight_gradient(T(x.astype(np.float16)), T((dy * 1e4).astype(np.float16)), k, pad).float().cpu().numpy()
    xh = np.pad(x.astype(np.float16).astype(np.float64), ((0, 0), (0, 0), (pad, pad), (pad, pad))); dh = (dy * 1e4).astype(np.float16).astype(np.float64)
    refh = np.zeros_like(ref)
    for ky in range(k):
        for kx in range(k):
            refh[:, :, :, ky, kx] = np.einsum('noyx,niyx->noi', dh, xh[:, :, ky:ky + oh, kx:kx + ow])
    assert maxabs(goth, refh) <= 3e-6 * float(np.abs(refh).max())


def test_modulated_conv2d_fp16_and_grad():
    from oracle import oracle as O
    from models.stylegan3.networks_stylegan3 import modulated_conv2d
    x = rand(61, 2, 24, 20, 20).astype(np.float16); w = rand(62, 16, 24, 3, 3); s = rand(63, 2, 24) + 1
    y = modulated_conv2d(T(x), T(w), T(s), demodulate=True, padding=2, input_gain=torch.tensor(1.1, device=DEV))
    assert y.dtype == torch.float16
    ref = O.modulated_conv2d(x.astype(np.float32), w, s, True, 2, 1.1)
    assert maxabs(y.float().cpu().numpy(), ref) <= 1e-2
    # gradients w.r.t. x, w, s equal those of the composite formulation
    from torch_utils.ops import modulated_conv as mc
    res = []
    for impl in ('cuda', 'ref'):
        xs = T(rand(64, 2, 6, 9, 9)).requires_grad_(True); ws = T(rand(65, 5, 6, 3, 3)).requires_grad_(True); ss = (T(rand(66, 2, 6)) + 1).requires_grad_(True)
        out = mc.modulated_conv2d(xs, ws, ss, demodulate=True, padding=2, input_gain=torch.tensor(0.9, device=DEV), impl=impl)
        (out * T(rand(67, *out.shape))).sum().backward()
        res.append([t.grad.cpu().numpy() for t in (xs, ws, ss)])
    for a, r in zip(*res):
        assert maxabs(a, r) <= 1e-4 * max(1.0, float(np.abs(r).max()))


def test_error_behaviour():
    """Argument errors surface as RuntimeError, like the reference's TORCH_CHECKs."""
    from torch_utils.ops import filtered_lrelu
    filtered_lrelu._init()
    P = filtered_lrelu._plugin
    x = torch.randn(1, 2, 8, 8, device=DEV); f = torch.ones(1, 1, device=DEV); b = torch.zeros(2, device=DEV)
    with pytest.raises(RuntimeError, match='float32'):
        P.filtered_lrelu(x, f.double(), f, b, torch.empty(0), 1, 1, 0, 0, 0, 0, 0, 0, 1.0, 0.2, 256.0, False, False)
    with pytest.raises(RuntimeError, match='same dtype'):
        P.filtered_lrelu(x, f, f, b.half(), torch.empty(0), 1, 1, 0, 0, 0, 0, 0, 0, 1.0, 0.2, 256.0, False, False)
    with pytest.raises(RuntimeError, match='rank 4'):
        P.filtered_lrelu(x[0], f, f, b, torch.empty(0), 1, 1, 0, 0, 0, 0, 0, 0, 1.0, 0.2, 256.0, False, False)
    y, so, rc = P.filtered_lrelu(x, torch.ones(3, 3, device=DEV), f, b, torch.empty(0), 1, 1, 1, 1, 1, 1, 0, 0, 1.0, 0.2, 256.0, False, False)
    assert rc == -1 and y.numel() == 0            # no fused kernel: the caller composes the generic path


def test_fourier_features_match_torch_ops():
    """The fused Fourier-feature kernel reproduces the reference's chain of torch ops (K = 2 matmul, + phase, * 2 pi, sin,
    * amplitude; networks_stylegan3.py:236-241) BIT FOR BIT, for both generator configs' input sizes and a batched transform."""
    from torch_utils.ops import fourier_features as ff
    for (n, c, size, sr, bw) in ((3, 64, 36, 16.0, 2.0), (2, 128, 36, 16.0, 2.0), (2, 32, 20, 8.0, 1.5)):
        g = torch.Generator(device=DEV).manual_seed(n * 100 + c)
        freqs = torch.randn([n, c, 2], device=DEV, generator=g) * bw
        phases = torch.rand([n, c], device=DEV, generator=g) - 0.5
        amps = torch.rand([n, c], device=DEV, generator=g)
        theta = torch.eye(2, 3, device=DEV)
        theta[0, 0] = 0.5 * size / sr
        theta[1, 1] = 0.5 * size / sr
        grid = torch.nn.functional.affine_grid(theta.unsqueeze(0), [1, 1, size, size], align_corners=False)
        x = (grid.unsqueeze(3) @ freqs.permute(0, 2, 1).unsqueeze(1).unsqueeze(2)).squeeze(3)
        ref = (torch.sin((x + phases.unsqueeze(1).unsqueeze(2)) * (np.pi * 2)) * amps.unsqueeze(1).unsqueeze(2)).permute(0, 3, 1, 2)
        got = ff.fourier_features(grid[0], freqs, phases, amps)
        assert got.shape == ref.shape
        assert torch.equal(got, ref.contiguous()), float((got - ref).abs().max())


@pytest.mark.parametrize('shape,up,taps,pad,radial', [
    ((1, 2, 150, 150), 2, 12, [9, 8, 9, 8], False),
    ((1, 2, 86, 86), 4, 24, [-6, -9, -6, -9], False),
    ((1, 2, 148, 148), 2, 12, [11, 10, 11, 10], True),
    ((1, 2, 84, 84), 4, 24, [-2, -5, -2, -5], True),
])
@pytest.mark.parametrize('clamp,slope', [(256.0, 0.2), (4.0, 0.2), (None, 0.2), (256.0, 0.0), (8.0, 1.0)])
def test_filtered_lrelu_nonlinearity_paths(shape, up, taps, pad, radial, clamp, slope):
    """lrelu + clamp inside the fused kernel on inputs with a quiet part and a band far beyond both clamp bounds (positive
    and negative), separable and radial down filters; no clamp, slope 0 (relu) and slope 1 as special values."""
    from oracle import oracle as O
    fs = 64
    fu = O.design_lowpass_filter(taps, 4.0, 8.0, fs * up / 2)
    fd = O.design_lowpass_filter(12, 5.0, 9.0, fs, radial=radial)
    x = rand(3, *shape); b = rand(4, shape[1])
    h = shape[2]
    x[:, :, h // 3: h // 2, :] *= 3000.0                 # far beyond clamp / slope for a band of rows ...
    x[:, :, :, : shape[3] // 4] *= 40.0                  # ... and moderately large in a band of columns
    c = dict(up=up, down=2, padding=pad, gain=float(np.sqrt(2)), slope=slope, clamp=clamp, flip=False)
    y = _flrelu(c, T(x), T(b), T(fu), T(fd))
    ref = O.filtered_lrelu(x, fu, fd, b, up, 2, pad, c['gain'], slope, clamp, False)
    assert tuple(y.shape) == ref.shape
    assert maxabs(y.cpu().numpy(), ref) <= 2e-5 * max(1.0, float(np.abs(ref).max()))


@pytest.mark.parametrize('shape,up,taps,pad,radial', [
    ((1, 3, 150, 150), 2, 12, [9, 8, 9, 8], False),            # two strips: the NaN sits in strip 0, strip 1 is untouched
    ((1, 3, 86, 86), 4, 24, [-6, -9, -6, -9], False),
    ((1, 3, 148, 148), 2, 12, [11, 10, 11, 10], True),
])
def test_filtered_lrelu_non_finite_input_stays_loud(shape, up, taps, pad, radial):
    """a NaN in the input must not come out of the plain forward as a finite +-clamp pixel (v_med3 drops NaNs).
    Contract of the kernel: from the row a wave stages a non-finite sample on, every output it writes in that strip is NaN --
    a superset of the NaN footprint of the reference formulation (`impl='ref'`: `if fabsf(v) > clamp: copysign`, NaN passes);
    rows above, other strips and other planes are bit-for-bit what they are without the NaN."""
    from oracle import oracle as O
    from torch_utils.ops import filtered_lrelu as fl
    fu = O.design_lowpass_filter(taps, 4.0, 8.0, 64.0 * up / 2)
    fd = O.design_lowpass_filter(12, 5.0, 9.0, 64.0, radial=radial)
    clean = rand(3, *shape); b = rand(4, shape[1])
    x = clean.copy()
    x[0, 0, 40, 30] = np.nan
    x[0, 1, 61, 17] = np.inf
    kw = dict(up=up, down=2, padding=pad, gain=float(np.sqrt(2)), slope=0.2, clamp=256, flip_filter=False)
    y = fl.filtered_lrelu(T(x), T(fu), T(fd), T(b), **kw).cpu().numpy()
    y_clean = fl.filtered_lrelu(T(clean), T(fu), T(fd), T(b), **kw).cpu().numpy()
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        ref = fl.filtered_lrelu(torch.from_numpy(x), torch.from_numpy(fu), torch.from_numpy(fd), torch.from_numpy(b), impl='ref', **kw).numpy()
    assert np.isfinite(y_clean).all() and np.isnan(ref[0, 0]).any() and np.isfinite(ref[0, 2]).all()
    assert np.isnan(y[np.isnan(ref)]).all()                                  # nothing the reference marks NaN comes out finite
    assert np.isnan(y[0, 1]).any()                                           # the infinity is flagged as well
    assert np.array_equal(y[0, 2], y_clean[0, 2])                            # clean plane untouched
    for plane in (0, 1):
        first = int(np.argmax(np.isnan(y[0, plane]).any(axis=1)))            # first output row with a NaN
        assert first > 0 and np.array_equal(y[0, plane, :first], y_clean[0, plane, :first])
        nan_cols = np.isnan(y[0, plane]).any(axis=0)
        assert not nan_cols.all()                                            # the other strip of the plane is untouched
        assert np.array_equal(y[0, plane][:, ~nan_cols], y_clean[0, plane][:, ~nan_cols])
    ok = np.isfinite(y) & np.isfinite(ref)
    assert float(np.abs(y[ok] - ref[ok]).max()) <= 2e-5 * max(1.0, float(np.abs(ref[ok]).max()))


@pytest.mark.parametrize('n,ci,co,h', [(2, 51, 32, 150), (1, 81, 51, 278), (1, 64, 70, 131)])
def test_modulated_conv2d_aligned_row_pitch(n, ci, co, h):
    """`align_rows`: the 3x3 split-precision kernel writes into a buffer whose rows start on 128-byte lines and returns the
    [..., :W'] view; values are bit-identical to the dense call, and filtered_lrelu consumes the strided view as it stands."""
    from oracle import oracle as O
    from torch_utils.ops import filtered_lrelu as fl, modulated_conv
    x = T(rand(41, n, ci, h, h)); w = T(rand(42, co, ci, 3, 3)); s = T(rand(43, n, ci) + 1.5)
    with torch.no_grad():
        dense = modulated_conv.modulated_conv2d(x, w, s, demodulate=True, padding=2, x_bound=8.0)
        view = modulated_conv.modulated_conv2d(x, w, s, demodulate=True, padding=2, x_bound=8.0, align_rows=True)
        assert tuple(view.shape) == tuple(dense.shape) and torch.equal(view, dense)
        assert view.stride(3) == 1 and view.stride(2) % 32 == 0 and view.stride(2) >= view.shape[3] and not view.is_contiguous()
        fu = T(O.design_lowpass_filter(12, 4.0, 8.0, 64.0)); fd = T(O.design_lowpass_filter(12, 5.0, 9.0, 64.0))
        b = T(rand(44, co))
        kw = dict(up=2, down=2, padding=[9, 8, 9, 8], gain=float(np.sqrt(2)), slope=0.2, clamp=256)
        assert torch.equal(fl.filtered_lrelu(view, fu, fd, b, **kw), fl.filtered_lrelu(dense, fu, fd, b, **kw))
    # with gradients recorded the request is ignored (the autograd graph keeps dense tensors)
    xg = x.clone().requires_grad_(True)
    assert modulated_conv.modulated_conv2d(xg, w, s, demodulate=True, padding=2, x_bound=8.0, align_rows=True).is_contiguous()


class _planes_per_wave:
    """Records, for every filtered_lrelu call inside the block, how many planes a wave of the streaming kernel works on."""

    def __enter__(self):
        from torch_utils import _hip_plugins
        _hip_plugins.planes_per_wave_log = self.log = []
        return self.log

    def __exit__(self, *exc):
        from torch_utils import _hip_plugins
        _hip_plugins.planes_per_wave_log = None


@pytest.mark.parametrize('shape,up,taps,pad,radial', [
    ((8, 6, 38, 38), 2, 12, [9, 8, 9, 8], False),          # T L0 / L1: 36 columns out
    ((2, 4, 38, 38), 4, 24, [-6, -9, -6, -9], False),      # T L2: up 4, 52 columns out
    ((1, 2, 54, 54), 2, 12, [9, 8, 9, 8], False),          # T L3: 52 columns out
    ((3, 2, 41, 56), 2, 12, [9, 8, 9, 8], False),          # 54 columns: the widest packed row; more rows than columns of lanes
    ((1, 4, 30, 39), 2, 12, [8, 8, 9, 8], False),          # odd output width (single stores), odd horizontal phase
    ((2, 2, 33, 27), 4, 24, [-5, -8, -6, -9], False),      # up 4 with the strip shifted by three upsampled columns
    ((2, 4, 36, 36), 2, 12, [11, 10, 11, 10], True),       # R: radial down filter, up 2
    ((1, 2, 36, 36), 4, 24, [-2, -5, -2, -5], True),       # R: radial, up 4
])
@pytest.mark.parametrize('dtype', [torch.float32, torch.float16])
def test_filtered_lrelu_two_planes_per_wave(shape, up, taps, pad, radial, dtype):
    """Narrow planes (output at most 54 columns): one wave works on two planes, 32 lanes each.  Same oracle, same tolerances as the
    one-plane form; the host query tells which form ran, so the case cannot pass on the other one."""
    from oracle import oracle as O
    fu = O.design_lowpass_filter(taps, 4.0, 8.0, 64.0 * up / 2)
    fd = O.design_lowpass_filter(12, 5.0, 9.0, 64.0, radial=radial)
    if radial:
        fd = (fd + 0.01 * np.random.RandomState(8).rand(12, 12)).astype(np.float32)      # symmetry broken: exposes flips
    x = rand(3, *shape); b = rand(4, shape[1])
    if dtype == torch.float16:
        x = x.astype(np.float16).astype(np.float32); b = b.astype(np.float16).astype(np.float32)
    for flip in ((False, True) if radial else (False,)):
        c = dict(up=up, down=2, padding=pad, gain=float(np.sqrt(2)), slope=0.2, clamp=256, flip=flip)
        with _planes_per_wave() as log:
            y = _flrelu(c, T(x, dtype), T(b, dtype), T(fu), T(fd))
        assert log == [2]
        ref = O.filtered_lrelu(x, fu, fd, b, up, 2, pad, c['gain'], 0.2, 256, flip)
        assert tuple(y.shape) == ref.shape and ref.shape[3] <= 54
        assert maxabs(y.float().cpu().numpy(), ref) <= (2e-5 if dtype == torch.float32 else 4e-3)


def test_filtered_lrelu_two_planes_per_wave_only_where_it_applies():
    """Wider rows, sliced (non-dense) planes and the training / adjoint forms keep one plane per wave; odd channel and plane
    counts pack.  All agree with the oracle."""
    from oracle import oracle as O
    fu = O.design_lowpass_filter(12, 4.0, 8.0, 64.0)
    fd = O.design_lowpass_filter(12, 5.0, 9.0, 64.0)
    c = dict(up=2, down=2, padding=[9, 8, 9, 8], gain=float(np.sqrt(2)), slope=0.2, clamp=256, flip=False)

    def check(xt, bt, xn, bn, expect):
        with _planes_per_wave() as log:
            y = _flrelu(c, xt, bt, T(fu), T(fd))
        assert log == [expect], (log, expect)
        assert maxabs(y.cpu().numpy(), O.filtered_lrelu(xn, fu, fd, bn, 2, 2, c['padding'], c['gain'], 0.2, 256, False)) <= 2e-5

    x = rand(3, 2, 4, 38, 38); b = rand(4, 4)
    check(T(x), T(b), x, b, 2)
    x3 = rand(5, 2, 3, 38, 38); b3 = rand(6, 3)
    check(T(x3), T(b3), x3, b3, 2)                                                   # odd C: one wave's planes straddle two images
    check(T(x3[:1]), T(b3), x3[:1], b3, 2)                                           # three planes: the last wave has a single one
    check(T(x)[:, 1:3], T(b)[1:3], np.ascontiguousarray(x[:, 1:3]), b[1:3].copy(), 1)   # channel slice: planes not dense over (n, c)
    big = rand(7, 2, 4, 40, 44)
    check(T(big)[:, :, 1:39, 3:41], T(b), np.ascontiguousarray(big[:, :, 1:39, 3:41]), b, 2)   # row-padded view: still dense over (n, c)
    xw = rand(8, 1, 2, 38, 60); bw = rand(9, 2)
    check(T(xw), T(bw), xw, bw, 1)                                                   # 58 columns out: too wide for 32 lanes
    xt = T(x).requires_grad_(True)                                                   # training forward writes signs: one plane
    with _planes_per_wave() as log:
        y = _flrelu(c, xt, T(b), T(fu), T(fd))
        y.sum().backward()
    assert log and all(v == 1 for v in log)
    # x alone requires a gradient above (the bias does not): same dx as when both do
    x2, b2 = T(x).requires_grad_(True), T(b).requires_grad_(True)
    _flrelu(c, x2, b2, T(fu), T(fd)).sum().backward()
    assert torch.equal(xt.grad, x2.grad) and b2.grad is not None


def test_filtered_lrelu_two_planes_per_wave_non_finite_input():
    """The NaN contract of the one-plane form, per plane: the plane that holds a NaN / infinity goes NaN from that row on, the
    plane sharing its wave is bit-for-bit what it is without it."""
    from oracle import oracle as O
    from torch_utils.ops import filtered_lrelu as fl
    fu = O.design_lowpass_filter(12, 4.0, 8.0, 64.0)
    fd = O.design_lowpass_filter(12, 5.0, 9.0, 64.0)
    clean = rand(3, 1, 6, 38, 54); b = rand(4, 6)
    x = clean.copy()
    x[0, 0, 20, 30] = np.nan              # first plane of wave 0 (its partner, plane 1, stays clean)
    x[0, 3, 25, 2] = np.inf               # second plane of wave 1 (plane 2 stays clean)
    x[0, 5, 10, 53] = np.nan              # last input column of the second plane of wave 2: staged by the third load of a row
    kw = dict(up=2, down=2, padding=[9, 8, 9, 8], gain=float(np.sqrt(2)), slope=0.2, clamp=256, flip_filter=False)
    with _planes_per_wave() as log:
        y = fl.filtered_lrelu(T(x), T(fu), T(fd), T(b), **kw).cpu().numpy()
        y_clean = fl.filtered_lrelu(T(clean), T(fu), T(fd), T(b), **kw).cpu().numpy()
    assert log == [2, 2]
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        ref = fl.filtered_lrelu(torch.from_numpy(x), torch.from_numpy(fu), torch.from_numpy(fd), torch.from_numpy(b), impl='ref', **kw).numpy()
    assert np.isfinite(y_clean).all()
    assert np.isnan(y[np.isnan(ref)]).all()
    for plane in (1, 2, 4):
        assert np.array_equal(y[0, plane], y_clean[0, plane])
    for plane in (0, 3, 5):
        first = int(np.argmax(np.isnan(y[0, plane]).any(axis=1)))
        assert first > 0 and np.isnan(y[0, plane, first:]).all() and np.array_equal(y[0, plane, :first], y_clean[0, plane, :first])


@pytest.mark.parametrize('shape,up,taps,pad,radial', [
    ((2, 2, 150, 150), 2, 12, [9, 8, 9, 8], False),        # T L6: 148 columns = 120 + 28
    ((1, 4, 86, 86), 4, 24, [-6, -9, -6, -9], False),      # T L5: up 4, 148 columns
    ((1, 2, 60, 278), 4, 24, [-6, -9, -6, -9], False),     # T L9-like rows: 532 columns = 4 x 120 + 52
    ((1, 2, 100, 150), 2, 12, [11, 10, 11, 10], True),     # R: radial down filter, up 2 (two launches)
    ((1, 3, 60, 84), 4, 24, [-2, -5, -2, -5], True),       # R L5: radial, up 4 (one launch), odd plane count
])
@pytest.mark.parametrize('dtype', [torch.float32, torch.float16])
def test_filtered_lrelu_full_strips_plus_packed_remainder(shape, up, taps, pad, radial, dtype):
    """Rows a little wider than a whole number of 120-column strips: the full strips run one plane per wave, the remainder strip
    (at most 54 columns) two planes per wave in a second launch.  Same oracle and tolerances."""
    from oracle import oracle as O
    fu = O.design_lowpass_filter(taps, 4.0, 8.0, 64.0 * up / 2)
    fd = O.design_lowpass_filter(12, 5.0, 9.0, 64.0, radial=radial)
    if radial:
        fd = (fd + 0.01 * np.random.RandomState(8).rand(12, 12)).astype(np.float32)
    x = rand(3, *shape); b = rand(4, shape[1])
    if dtype == torch.float16:
        x = x.astype(np.float16).astype(np.float32); b = b.astype(np.float16).astype(np.float32)
    c = dict(up=up, down=2, padding=pad, gain=float(np.sqrt(2)), slope=0.2, clamp=256, flip=False)
    with _planes_per_wave() as log:
        y = _flrelu(c, T(x, dtype), T(b, dtype), T(fu), T(fd))
    assert log == [3]
    ref = O.filtered_lrelu(x, fu, fd, b, up, 2, pad, c['gain'], 0.2, 256, False)
    assert tuple(y.shape) == ref.shape
    assert maxabs(y.float().cpu().numpy(), ref) <= (2e-5 if dtype == torch.float32 else 4e-3)
    # a NaN under the remainder strip: its plane goes NaN there from that row on; the full strip of the same plane and the plane
    # sharing the wave stay bit-for-bit clean
    if dtype == torch.float32 and not radial:
        xn = x.copy()
        xn[0, 1, shape[2] // 2, shape[3] - 3] = np.nan
        yn = _flrelu(c, T(xn), T(b), T(fu), T(fd)).cpu().numpy()
        yc = y.cpu().numpy()
        full = (yc.shape[3] - 1) // 120 * 120
        nan = np.isnan(yn)
        rows = nan[0, 1].any(axis=1)
        first = int(np.argmax(rows))
        assert first > 0 and nan[0, 1, first, full:].all()           # the whole width of the remainder strip, from the NaN's row on
        assert not nan[0, 1, :, :full].any() and not nan[0, 0].any() and not nan[1:].any()
        assert np.array_equal(yn[~nan], yc[~nan])


@pytest.mark.parametrize('n,co,ci,k,demod,gain', [(1, 64, 48, 3, True, 'scalar'), (3, 40, 70, 3, True, 'per_channel'), (2, 33, 17, 1, True, 'per_sample'),
                                                  (2, 3, 32, 1, False, 'scalar'), (1, 512, 512, 3, True, None), (4, 20, 24, 3, False, None)])
def test_modulation_backward_kernels_match_autograd(n, co, ci, k, demod, gain):
    """dL/dw, dL/ds of the effective-weight algebra (pre-normalisation, modulation, demodulation, input gain) in closed form
    (sg3_modulation_backward, four launches) == autograd through the reference formulation (`_effective_weights`)."""
    from torch_utils.ops import modulated_conv as mc
    g = torch.Generator(device=DEV).manual_seed(5)
    w = torch.randn([co, ci, k, k], device=DEV, generator=g, requires_grad=True)
    s = (torch.randn([n, ci], device=DEV, generator=g) + 1.0).requires_grad_(True)
    ig = {None: None, 'scalar': torch.tensor(0.7, device=DEV), 'per_channel': torch.rand([ci], device=DEV, generator=g) + 0.5,
          'per_sample': torch.rand([n, ci], device=DEV, generator=g) + 0.5}[gain]
    dw_eff = torch.randn([n, co, ci, k, k], device=DEV, generator=g)
    w_eff = mc._effective_weights(w, s, demod, ig, n)
    ref_w, ref_s = torch.autograd.grad(w_eff, [w, s], dw_eff)
    with torch.no_grad():
        got_w, got_s = mc._modulation_grads(dw_eff.clone(), w, s, ig, demod)
    assert got_w.shape == ref_w.shape and got_s.shape == ref_s.shape
    assert float((got_w - ref_w).abs().max()) <= 2e-5 * max(1e-3, float(ref_w.abs().max()))
    assert float((got_s - ref_s).abs().max()) <= 2e-5 * max(1e-3, float(ref_s.abs().max()))


@pytest.mark.parametrize('co,ci,k,norm', [(51, 81, 3, True), (512, 512, 3, True), (3, 32, 1, False), (7, 5, 3, False)])
def test_transposed_weights_kernel_matches_the_torch_ops(co, ci, k, norm):
    """sg3_modconv_transpose_weights: the data-gradient convolution's weights -- pre-normalised (networks_stylegan3.py:41-42), transposed,
    flipped -- in one launch, against the six torch ops it replaces (round 4: PTI step housekeeping)."""
    from torch_utils.ops import modulated_conv as mc
    w = T(rand(301, co, ci, k, k) * 1.7)
    ref = w * w.square().mean([1, 2, 3], keepdim=True).rsqrt() if norm else w
    ref = ref.flip([2, 3]).transpose(0, 1).contiguous()
    got = mc._transposed_weights(w, norm)
    assert tuple(got.shape) == (ci, co, k, k)
    assert maxabs(got.cpu().numpy(), ref.cpu().numpy()) <= 2e-6 * float(ref.abs().max())


@pytest.mark.parametrize('shape,up,taps,pad,radial', [
    ((1, 2, 150, 150), 2, 12, [9, 8, 9, 8], False),           # T up-2 (adjoint: up 2 / down 2)
    ((2, 3, 86, 86), 4, 24, [-6, -9, -6, -9], False),          # T up-4 (adjoint: up 2 / down 4, one column per lane)
    ((1, 2, 84, 150), 2, 12, [11, 10, 11, 10], True),          # R up-2 (adjoint: 12x12 up filter)
    ((1, 1, 278, 130), 2, 12, [9, 8, 9, 8], False),            # several row chunks
])
def test_filtered_lrelu_adjoint_hands_max_abs_dx_to_the_convolution_gradients(shape, up, taps, pad, radial):
    """The adjoint launch keeps max |dx| of what it stores (yAbsMaxPartial) and attaches it to dx (torch_utils/ops/known_amax.py):
    exactly dx.abs().max(); the lookup misses after an in-place change, on a view with another address, and when switched off."""
    from oracle import oracle as O
    from torch_utils.ops import filtered_lrelu as fl, known_amax
    fl._init()
    fu = O.design_lowpass_filter(taps, 4.0, 8.0, 64.0 * up / 2)
    fd = O.design_lowpass_filter(12, 5.0, 9.0, 64.0, radial=radial)
    x = T(rand(3, *shape)).requires_grad_(True); b = T(rand(4, shape[1])).requires_grad_(True)
    y = fl.filtered_lrelu(x, T(fu), T(fd), b, up=up, down=2, padding=pad, gain=float(np.sqrt(2)), slope=0.2, clamp=256, flip_filter=False)
    seen = {}

    def hook(g):                                                        # g: what the adjoint's backward returned for x
        seen['amax'] = known_amax.lookup(g)

    x.register_hook(hook)
    (y * T(rand(5, *y.shape)) * 37.0).sum().backward()
    assert seen['amax'] is not None, 'the adjoint kernel did not attach max |dx|'
    assert float(seen['amax']) == float(x.grad.abs().max())
    g = x.grad
    known_amax.attach(g, seen['amax'])
    assert known_amax.lookup(g) is seen['amax']
    assert known_amax.lookup(g[:, :, 1:]) is None                       # another tensor object
    saved, known_amax.enabled = known_amax.enabled, False
    try:
        assert known_amax.lookup(g) is None
    finally:
        known_amax.enabled = saved
    g.mul_(2.0)                                                         # an in-place change invalidates it
    assert known_amax.lookup(g) is None


def test_known_amax_reaches_the_modulated_convolution_backward_and_changes_nothing():
    """conv -> filtered_lrelu as SynthesisLayer chains them (networks_stylegan3.py:335-368): the convolution's backward finds the
    bound the adjoint launch left on dy (no reduction pass of its own), and every gradient is bit-identical to the run with the
    hand-off switched off."""
    from oracle import oracle as O
    from torch_utils.ops import filtered_lrelu as fl, known_amax
    from torch_utils.ops.modulated_conv import modulated_conv2d
    fl._init()
    fu = T(O.design_lowpass_filter(12, 4.0, 8.0, 64.0)); fd = T(O.design_lowpass_filter(12, 5.0, 9.0, 64.0))
    xn, wn, sn, bn = rand(1, 2, 48, 60, 60), rand(2, 64, 48, 3, 3), rand(3, 2, 48) + 1.5, rand(4, 64)

    def run():
        x = T(xn).requires_grad_(True); w = T(wn).requires_grad_(True); s = T(sn).requires_grad_(True); b = T(bn).requires_grad_(True)
        h = modulated_conv2d(x=x, w=w, s=s, padding=2, demodulate=True)
        y = fl.filtered_lrelu(h, fu, fd, b, up=2, down=2, padding=[9, 8, 9, 8], gain=float(np.sqrt(2)), slope=0.2, clamp=256)
        (y * y).sum().backward()
        return [t.grad.clone() for t in (x, w, s, b)]

    h0 = known_amax.hits
    with_it = run()
    assert known_amax.hits == h0 + 1, 'the convolution backward did not find the attached bound'
    saved, known_amax.enabled = known_amax.enabled, False
    try:
        without = run()
    finally:
        known_amax.enabled = saved
    for a, c in zip(with_it, without):
        assert torch.equal(a, c)


@pytest.mark.parametrize('n,ci,co,h,k', [(1, 512, 512, 36, 3), (2, 256, 130, 50, 3), (1, 512, 512, 52, 3), (1, 1024, 1024, 36, 1), (2, 645, 406, 40, 1)])
def test_modulated_conv2d_k_split_on_small_grids_matches_the_unsplit_kernel(n, ci, co, h, k):
    """Fewer tiles than CUs (batch 1 on the small maps, PTI): the direct 3x3 kernel and the 1x1 GEMM kernel split the input channels of a tile over up to four
    workgroups (sg3_modconv_params.splitScratch) and a second launch adds the partial sums.  Against the same call without scratch
    (one workgroup per tile walks all chunks) and against float64."""
    from torch_utils import _sg3abi
    from torch_utils.ops import modulated_conv as mc
    x, w, s = T(rand(1, n, ci, h, h)), T(rand(2, co, ci, k, k)), T(rand(3, n, ci) + 1.5)
    lib = _sg3abi.load()

    class _NoScratch:                                   # the library with the scratch query answering 0: the caller then passes none
        def __getattr__(self, name):
            if name == 'sg3_modconv_split_scratch_floats':
                return lambda p: 0
            return getattr(lib, name)

    with torch.no_grad():
        split = mc.modulated_conv2d(x=x, w=w, s=s, padding=k - 1, demodulate=True, x_bound=float(x.abs().max()) * 1.01)
        saved = _sg3abi._lib
        _sg3abi._lib = _NoScratch()
        try:
            plain = mc.modulated_conv2d(x=x, w=w, s=s, padding=k - 1, demodulate=True, x_bound=float(x.abs().max()) * 1.01)
        finally:
            _sg3abi._lib = saved
    ref = mc._composite(x.double(), w.double(), s.double(), True, k - 1, None)
    scale = float(ref.abs().max())
    assert float((split.double() - ref).abs().max()) <= 2e-6 * scale and float((plain.double() - ref).abs().max()) <= 3e-6 * scale
    assert float((split - plain).abs().max()) <= 4e-6 * scale          # two fp32 summation orders over K = 9 I
    if (n, ci, co, h) in ((1, 512, 512, 36), (1, 1024, 1024, 36)):      # L0 - L2 of a batch-1 PTI step (T: flat kernel, 96 tiles; R: 1x1 GEMM, 24 tiles): four splits
        assert not torch.equal(split, plain)                            # the split path did run
