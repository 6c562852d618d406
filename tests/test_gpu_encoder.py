"""GPU: the ReStyle encoder on libsg3hip's matrix-core convolution (BatchNorm / PReLU / leaky ReLU fused) against the
golden vectors built from the reference's residual units, and the fused conv2d op against the CPU oracle."""
import numpy as np
import pytest
import torch

from helpers import golden, maxabs
from test_encoder_cpu import _input, build_product_encoder

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


@pytest.mark.parametrize('n,ci,co,h,w,k,stride,pad,act', [
    (2, 6, 64, 40, 44, 3, 1, 1, 1), (1, 64, 64, 33, 31, 3, 2, 1, 0), (2, 64, 128, 20, 24, 1, 2, 0, 0), (1, 128, 256, 16, 16, 3, 1, 1, 1),
    (2, 512, 512, 16, 16, 3, 2, 1, 2), (3, 512, 512, 2, 2, 3, 2, 1, 2), (1, 100, 70, 19, 23, 3, 1, 1, 2), (1, 48, 8192, 9, 9, 3, 2, 1, 2),
    (2, 70, 96, 17, 19, 1, 1, 0, 1), (1, 256, 512, 32, 32, 1, 2, 0, 0), (1, 512, 1024, 16, 20 * 4, 3, 2, 1, 2)])
def test_conv2d_fused(n, ci, co, h, w, k, stride, pad, act):
    from oracle import oracle as O
    from torch_utils.ops.plain_conv import PackedConv
    r = np.random.RandomState(5)
    x = r.randn(n, ci, h, w).astype(np.float32); wt = (r.randn(co, ci, k, k) / np.sqrt(ci * k * k)).astype(np.float32)
    a_in, b_in = r.uniform(0.5, 1.5, ci).astype(np.float32), (0.2 * r.randn(ci)).astype(np.float32)
    a_out, bias = r.uniform(0.5, 1.5, co).astype(np.float32), (0.2 * r.randn(co)).astype(np.float32)
    slope = r.uniform(0.1, 0.4, co).astype(np.float32) if act == 1 else np.asarray([0.01], np.float32)
    T = lambda v: torch.from_numpy(v).to(DEV)  # noqa: E731
    conv = PackedConv(T(wt), out_scale=T(a_out), bias=T(bias), in_scale=T(a_in), in_shift=T(b_in), act=act, slope=T(slope), stride=stride, padding=pad)
    y = conv(T(x)).cpu().numpy()
    ref = O.conv2d(x * a_in[None, :, None, None] + b_in[None, :, None, None], wt * a_out[:, None, None, None], bias, stride, pad)
    if act:
        ref = np.where(ref >= 0, ref, ref * (slope[None, :, None, None] if act == 1 else slope[0]))
    assert y.shape == ref.shape
    assert maxabs(y, ref) <= 2e-5 * max(1.0, float(np.abs(ref).max()))


@pytest.mark.parametrize('name,ci,co,st', [('unit_same', 64, 64, 1), ('unit_down', 64, 64, 2), ('unit_proj', 64, 128, 2)])
def test_residual_units(name, ci, co, st):
    from models.setgan.encoder.encoders.helpers import bottleneck_IR_SE
    from synth_weights import synth_encoder_state_dict
    u = bottleneck_IR_SE(ci, co, st).eval()
    man = {('body.0.' + k): list(v.shape) for k, v in u.state_dict().items()}
    sd = synth_encoder_state_dict(man, seed=5)
    u.load_state_dict({k[len('body.0.'):]: torch.from_numpy(np.asarray(v)) for k, v in sd.items()})
    u = u.to(DEV)
    x = torch.from_numpy(np.random.RandomState(7).randn(2, ci, 20, 24).astype(np.float32)).to(DEV)
    with torch.no_grad():
        assert maxabs(u.forward_hip(x).cpu().numpy(), golden('encoder')[name + '/y']) <= 2e-5


def test_backbone_encoder():
    from torch_utils import _sg3abi
    g = golden('encoder')
    enc = build_product_encoder(DEV)
    x = torch.from_numpy(_input()).to(DEV)
    n0 = _sg3abi.launch_count
    with torch.no_grad():
        codes = enc(x)
        assert _sg3abi.launch_count - n0 >= 50 + 1 + 48, 'the fused HIP path did not run'
        ref_path = enc._forward_torch(x)                      # same modules through torch / MIOpen
    assert tuple(codes.shape) == (2, 16, 512)
    assert maxabs(codes.cpu().numpy(), g['codes']) <= 2e-4
    assert maxabs(codes.cpu().numpy(), ref_path.cpu().numpy()) <= 2e-3    # MIOpen's own fp32 path differs more than ours
    # packed weights follow parameter changes after invalidation
    with torch.no_grad():
        enc.input_layer[1].bias.add_(0.5)
        enc.invalidate_packed()
        assert maxabs(enc(x).cpu().numpy(), enc._forward_torch(x).cpu().numpy()) <= 2e-3
        assert maxabs(enc(x).cpu().numpy(), g['codes']) > 1e-3


def test_split_precision_range_guard():
    """The split-precision convolution flags operands outside the fp16 range on the device; the encoder then repeats its
    forward on the exact fp32 kernels, so huge activations still give the fp32 result."""
    from torch_utils.ops import plain_conv
    from torch_utils.ops.plain_conv import PackedConv
    from oracle import oracle as O
    from golden_cases import rand
    T = lambda v: torch.from_numpy(np.ascontiguousarray(v, dtype=np.float32)).to(DEV)  # noqa: E731
    w = rand(31, 32, 16, 3, 3) * 0.2
    conv = PackedConv(T(w), stride=1, padding=1)
    x = rand(32, 1, 16, 20, 40)
    plain_conv.reset_overflow(DEV)
    y = conv(T(x))
    assert not plain_conv.overflowed(DEV)
    assert maxabs(y.cpu().numpy(), O.conv2d(x, w, None, 1, 1)) <= 2e-5
    conv(T(x * 1e5))
    assert plain_conv.overflowed(DEV)                       # |x| ~ 4e5 > 65504
    plain_conv.reset_overflow(DEV)
    assert not plain_conv.overflowed(DEV)
    saved, plain_conv.precision = plain_conv.precision, 'fp32'
    try:
        y32 = conv(T(x * 1e5))
    finally:
        plain_conv.precision = saved
    ref = O.conv2d((x * 1e5).astype(np.float32), w, None, 1, 1)
    assert maxabs(y32.cpu().numpy(), ref) <= 2e-5 * float(np.abs(ref).max())
    # end to end: an encoder fed 1e6-scale inputs falls back and still matches its own fp32 run
    from models.setgan.encoder.encoders.restyle_psp_encoders import BackboneEncoder
    from synth_weights import synth_encoder_state_dict
    enc = BackboneEncoder(50, 'ir_se', 4)
    man = {k: list(v.shape) for k, v in enc.state_dict().items()}
    enc.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in synth_encoder_state_dict(man, seed=0).items()})
    enc = enc.eval().requires_grad_(False).to(DEV)
    xin = T(rand(33, 1, 6, 256, 256) * 1e6)
    with torch.no_grad():
        a = enc(xin)
        plain_conv.precision = 'fp32'
        try:
            b = enc(xin)
        finally:
            plain_conv.precision = saved
    assert bool(torch.isfinite(a).all()) and torch.equal(a, b)


def test_packed_weights_follow_parent_load_and_inplace_updates():
    """`net.load_state_dict(ckpt)` on the PARENT (pSp) recurses without calling the encoder's own load_state_dict, and
    in-place edits (optimizer.step / EMA copy_) pass no hook at all: the packed copies must follow both."""
    import types
    from models.setgan.encoder.psp3 import pSp
    from helpers import build_product_generator
    from synth_weights import synth_encoder_state_dict
    G = build_product_generator('Ttiny')
    opts = types.SimpleNamespace(encoder_type='BackboneEncoder', input_nc=6, checkpoint_path=None, n_iters_per_batch=1, resize_outputs=False)
    net = pSp(opts, decoder=G)
    man = {k: list(v.shape) for k, v in net.encoder.state_dict().items()}
    net.encoder.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in synth_encoder_state_dict(man, seed=0).items()})
    net = net.eval().requires_grad_(False).to(DEV)
    x = torch.from_numpy(_input()).to(DEV)
    with torch.no_grad():
        first = net.encoder(x).clone()
        other = {'encoder.' + k: torch.from_numpy(np.asarray(v)) for k, v in synth_encoder_state_dict(man, seed=3).items()}
        net.load_state_dict(other, strict=False)                       # parent-level load
        second = net.encoder(x).clone()
        assert maxabs(second.cpu().numpy(), net.encoder._forward_torch(x).cpu().numpy()) <= 2e-3
        assert maxabs(second.cpu().numpy(), first.cpu().numpy()) > 1e-2
        net.encoder.body[3].res_layer[4].running_mean.add_(0.3)        # in-place statistics update, no hook fires
        net.encoder.styles[2].linear.bias.mul_(2.0)
        third = net.encoder(x)
        assert maxabs(third.cpu().numpy(), net.encoder._forward_torch(x).cpu().numpy()) <= 2e-3
        assert maxabs(third.cpu().numpy(), second.cpu().numpy()) > 1e-3


def test_resnet34_encoder_hip_path():
    """ResNetBackboneEncoder / ResNetProgressiveBackboneEncoder on the matrix-core convolution (BasicBlocks: BN folded, ReLU in
    the epilogue, 1x1 stride-2 projections) against the oracle restatement and the plain PyTorch path of the same modules."""
    from oracle import oracle as O
    from test_encoder_cpu import build_resnet_encoder
    from torch_utils import _sg3abi
    enc, sd = build_resnet_encoder(4, device=DEV)
    x = _input()
    ref = O.resnet_backbone_encoder(sd, x, n_styles=4)
    n0 = _sg3abi.launch_count
    with torch.no_grad():
        codes = enc(torch.from_numpy(x).to(DEV))
        assert _sg3abi.launch_count - n0 >= 16 * 2 + 3, 'the fused HIP path did not run'
        torch_path = enc._forward_torch(torch.from_numpy(x).to(DEV))
    assert tuple(codes.shape) == (2, 4, 512)
    scale = max(1.0, float(np.abs(ref).max()))
    assert maxabs(codes.cpu().numpy(), ref) <= 2e-4 * scale
    assert maxabs(codes.cpu().numpy(), torch_path.cpu().numpy()) <= 2e-3 * scale
    prog, _ = build_resnet_encoder(4, device=DEV, progressive=True)
    with torch.no_grad():
        b = prog(torch.from_numpy(x).to(DEV))
    assert maxabs(b[:, 0].cpu().numpy(), codes[:, 0].cpu().numpy()) <= 1e-5 * scale
    assert maxabs(b[:, 3].cpu().numpy(), (codes[:, 0] + codes[:, 3]).cpu().numpy()) <= 1e-5 * scale


@pytest.mark.parametrize('n,c,h,w,strided', [(2, 64, 128, 128, False), (3, 128, 20, 24, True), (1, 512, 16, 16, False), (2, 256, 7, 9, True),
                                             (2, 70, 5, 6, False)])
def test_se_residual_kernels(n, c, h, w, strided):
    """shortcut + res * sigmoid(fc2 @ relu(fc1 @ mean_hw(res))) in two launches == the torch op chain of SEModule +
    bottleneck_IR_SE (dense and stride-2 subsampled shortcuts, plane sizes that are not multiples of 4, C not a multiple of 8)."""
    from torch_utils.ops import se_ops
    r = max(1, c // 16)
    g = np.random.RandomState(3)
    T = lambda v: torch.from_numpy(np.ascontiguousarray(v, dtype=np.float32)).to(DEV)  # noqa: E731
    res = T(g.randn(n, c, h, w))
    big = T(g.randn(n, c, 2 * h, 2 * w))
    shortcut = big[:, :, ::2, ::2] if strided else T(g.randn(n, c, h, w))
    fc1 = T(g.randn(r, c, 1, 1) / np.sqrt(c)); fc2 = T(g.randn(c, r, 1, 1) / np.sqrt(r))
    m = res.mean(dim=(2, 3))
    gate = torch.sigmoid(torch.relu(m @ fc1.flatten(1).t()) @ fc2.flatten(1).t())
    ref = shortcut + res * gate[:, :, None, None]
    buf = res.clone()
    out = se_ops.se_residual(buf, shortcut, fc1, fc2)
    assert out.data_ptr() == buf.data_ptr()
    assert float((out - ref).abs().max()) <= 2e-6 * max(1.0, float(ref.abs().max()))


@pytest.mark.parametrize('g,n,c,ih,iw,layout,slope', [
    (3, 2, 70, 8, 8, 'strip', 1.0), (16, 16, 512, 8, 8, 'strip', 1.0), (2, 3, 130, 5, 7, 'strip', 0.2),
    (4, 5, 96, 4, 4, 'rows', 0.01), (16, 16, 512, 2, 2, 'rows', 0.01), (2, 3, 33, 1, 1, 'rows', 0.3), (1, 2, 512, 3, 5, 'rows', 1.0)])
def test_unfold3x3s2_matches_torch_unfold(g, n, c, ih, iw, layout, slope):
    """sg3_unfold3x3s2 (the patch matrix of the style heads' stride-2 convolutions, map2style.py:8-25) against F.unfold of the
    activated input, for both source layouts the encoder hands it: the channels-first strip the level-1 convolution writes (images
    side by side with junk columns between them) and the pixel-major rows of a batched GEMM.  Bit-exact (pure data movement + one
    multiply)."""
    from torch_utils.ops.unfold_ops import unfold3x3s2
    r = np.random.RandomState(g * 100 + c)
    if layout == 'strip':
        period = iw + 3
        base = torch.from_numpy(r.randn(1, g * c, ih, period * n).astype(np.float32)).to(DEV)
        src = base[0].view(g, c, ih, n, period)[..., :iw].permute(0, 3, 1, 2, 4)
    else:
        base = torch.from_numpy(r.randn(g, n * ih * iw, c).astype(np.float32)).to(DEV)
        src = base.view(g, n, ih, iw, c).permute(0, 1, 4, 2, 3)
    assert tuple(src.shape) == (g, n, c, ih, iw)
    got = unfold3x3s2(src, slope)
    dense = torch.nn.functional.leaky_relu(src.contiguous(), slope).reshape(g * n, c, ih, iw)
    oh, ow = (ih + 1) // 2, (iw + 1) // 2
    ref = torch.nn.functional.unfold(dense, 3, padding=1, stride=2)                      # [g*n, c*9, oh*ow], rows c*9 + tap
    ref = ref.view(g, n, c, 9, oh * ow).permute(0, 1, 4, 3, 2).reshape(g, n * oh * ow, 9 * c)   # -> rows (n, pixel), columns tap*c + ch
    assert tuple(got.shape) == tuple(ref.shape)
    assert torch.equal(got, ref)
