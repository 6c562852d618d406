"""Transform-domain 3x3 modulated convolution (SG3_CONV_F16X3_F23, csrc/sg3_modconv_f23.hip: Winograd F(2,3) along x on the
split-precision matrix-core arithmetic) against the fp64 oracle and against the direct split-precision kernel.
Reference semantics: models/stylegan3/networks_stylegan3.py:24-63.

Tolerance: relative max error vs fp64 <= 5e-6 -- the bound test_gpu_ops.py holds the direct f16x3 and exact-fp32 kernels to."""
import os

import numpy as np
import pytest
import torch

from golden_cases import rand
from helpers import maxabs

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def T(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


class _f23:
    def __init__(self, mode, tn=None):
        self.mode, self.tn = mode, tn

    def __enter__(self):
        from torch_utils import _sg3abi as abi
        from torch_utils.ops import modulated_conv as mc
        self.prev = mc.f23
        mc.f23 = self.mode
        self.prev_tn = abi.load().sg3_modconv_f23_force_rows(self.tn or 0)
        return mc

    def __exit__(self, *exc):
        from torch_utils import _sg3abi as abi
        from torch_utils.ops import modulated_conv as mc
        mc.f23 = self.prev
        abi.load().sg3_modconv_f23_force_rows(self.prev_tn)


def _took_f23(mc, x, w, pad):
    from torch_utils import _sg3abi as abi
    n, ci, h, wd = x.shape
    return bool(abi.load().sg3_modconv_f23_supported(abi.SG3_F32, ci, int(w.shape[0]), h, wd, 3, pad, 0))


@pytest.mark.parametrize('tn', [4, 5, 7])
@pytest.mark.parametrize('n,ci,co,h,w,pad', [
    (2, 64, 64, 30, 30, 2),            # one M tile, one column tile
    (1, 323, 203, 22, 26, 2),          # odd channel counts: padded K chunk, an M block of pure padding
    (2, 81, 51, 40, 70, 2),            # O < 64: second M block inactive; three column tiles with a ragged last one
    (1, 512, 512, 20, 36, 2),          # 32 chunks
    (3, 17, 130, 35, 34, 0),           # pad 0 (the data-gradient form), two K chunks, rows ragged for every TN
    (1, 16, 96, 9, 10, 2),             # one chunk (no prefetch), tiny plane
])
def test_f23_matches_fp64_and_direct_kernel(n, ci, co, h, w, pad, tn):
    from oracle import oracle as O
    x = np.clip(rand(71, n, ci, h, w) * 40, -256, 256).astype(np.float32); wt = rand(72, co, ci, 3, 3); s = rand(73, n, ci) + 1
    ref = O.modulated_conv2d(x.astype(np.float64), wt.astype(np.float64), s.astype(np.float64), True, pad, 0.8)
    scale = max(1.0, float(np.abs(ref).max()))
    kw = dict(demodulate=True, padding=pad, input_gain=torch.tensor(0.8, device=DEV), x_bound=256.0)
    with _f23('on', tn) as mc:
        assert _took_f23(mc, x, wt, pad)
        y = mc.modulated_conv2d(T(x), T(wt), T(s), **kw)
    with _f23('off') as mc:
        yd = mc.modulated_conv2d(T(x), T(wt), T(s), **kw)
    assert tuple(y.shape) == ref.shape
    e23, ed = maxabs(y.cpu().numpy(), ref) / scale, maxabs(yd.cpu().numpy(), ref) / scale
    print(f'relative max error vs fp64: transform domain {e23:.2e}, direct {ed:.2e}')
    assert e23 <= 5e-6, (e23, ed)
    # large styles: the per-sample power-of-two rescale leaves one more bit of headroom for the transformed samples
    s_big = (s * 1000).astype(np.float32)
    with _f23('on', tn) as mc:
        y2 = mc.modulated_conv2d(T(x), T(wt), T(s_big), demodulate=False, padding=pad, input_gain=None, x_bound=256.0)
    ref2 = O.modulated_conv2d(x.astype(np.float64), wt.astype(np.float64), s_big.astype(np.float64), False, pad, None)
    assert bool(torch.isfinite(y2).all())
    assert maxabs(y2.cpu().numpy(), ref2) <= 5e-6 * float(np.abs(ref2).max())


def test_f23_saturated_input_stays_in_range():
    """Every sample at the bound with alternating signs: the transformed operands reach 2 x bound."""
    from oracle import oracle as O
    n, ci, co, h, w = 1, 32, 64, 12, 40
    x = np.full([n, ci, h, w], 256.0, dtype=np.float32); x[..., ::2] *= -1; x[:, ::3] *= -1
    wt = rand(5, co, ci, 3, 3); s = np.abs(rand(6, n, ci)) + 0.5
    with _f23('on') as mc:
        y = mc.modulated_conv2d(T(x), T(wt), T(s), demodulate=True, padding=2, x_bound=256.0)
    ref = O.modulated_conv2d(x.astype(np.float64), wt.astype(np.float64), s.astype(np.float64), True, 2, None)
    assert bool(torch.isfinite(y).all())
    assert maxabs(y.cpu().numpy(), ref) <= 5e-6 * float(np.abs(ref).max())


@pytest.mark.parametrize('n,ci,co,h', [(2, 64, 70, 150), (1, 128, 96, 278)])
def test_f23_aligned_row_pitch_and_batch_independence(n, ci, co, h):
    x = T(rand(41, n, ci, h, h)); w = T(rand(42, co, ci, 3, 3)); s = T(rand(43, n, ci) + 1.5)
    with _f23('on') as mc, torch.no_grad():
        dense = mc.modulated_conv2d(x, w, s, demodulate=True, padding=2, x_bound=8.0)
        view = mc.modulated_conv2d(x, w, s, demodulate=True, padding=2, x_bound=8.0, align_rows=True)
        assert view.stride(2) % 32 == 0 and torch.equal(view, dense)
        one = mc.modulated_conv2d(x[:1], w, s[:1], demodulate=False, padding=2, x_bound=8.0)
        both = mc.modulated_conv2d(x, w, s, demodulate=False, padding=2, x_bound=8.0)
        assert torch.equal(one[0], both[0])


def test_f23_unsupported_shapes_fall_back():
    """Odd widths / odd padding are refused by the query and run on the direct kernel."""
    from torch_utils import _sg3abi as abi
    lib = abi.load()
    assert lib.sg3_modconv_f23_supported(abi.SG3_F32, 64, 64, 30, 31, 3, 2, 0) == 0
    assert lib.sg3_modconv_f23_supported(abi.SG3_F32, 64, 64, 30, 30, 3, 1, 0) == 0
    assert lib.sg3_modconv_f23_supported(abi.SG3_F16, 64, 64, 30, 30, 3, 2, 0) == 1       # fp16 tensors: the SG3_CONV_F16_F23 form (round 4)
    assert lib.sg3_modconv_f23_supported(abi.SG3_F16, 64, 64, 30, 31, 3, 2, 0) == 0
    assert lib.sg3_modconv_f23_supported(abi.SG3_F32, 64, 64, 30, 30, 1, 0, 0) == 0
    assert lib.sg3_modconv_f23_supported(abi.SG3_F32, 64, 64, 30, 30, 3, 2, 0) == 1
    from oracle import oracle as O
    x = rand(1, 1, 64, 20, 31); w = rand(2, 64, 64, 3, 3); s = rand(3, 1, 64) + 1
    with _f23('on') as mc:
        y = mc.modulated_conv2d(T(x), T(w), T(s), demodulate=True, padding=2, x_bound=8.0)
    ref = O.modulated_conv2d(x, w, s, True, 2, None)
    assert maxabs(y.cpu().numpy(), ref) <= 3e-5 * max(1.0, float(np.abs(ref).max()))


def test_f23_repeated_with_shifting_allocations():
    """Regression for a hazard that only showed with cold, freshly allocated operands: the hand-issued loads of the kernel read
    scalar registers that hipcc had just reloaded from a spill (VALU-written SGPR -> vector-memory read needs 5 wait states; the
    requests now open with s_nop 4).  Many launches on new allocations, every tile height, against the direct kernel."""
    from torch_utils.ops import modulated_conv as mc
    g = torch.Generator(device=DEV).manual_seed(3)
    shapes = [(2, 64, 64, 30, 30, 2), (1, 323, 203, 22, 26, 2), (2, 81, 51, 40, 70, 2), (3, 17, 130, 35, 34, 0)]
    worst = 0.0
    for rep in range(12):
        for tn in (4, 5, 7):
            for (n, ci, co, h, w, pad) in shapes:
                x = torch.randn(n, ci, h, w, device=DEV, generator=g) * 3
                wt = torch.randn(co, ci, 3, 3, device=DEV, generator=g)
                s = torch.rand(n, ci, device=DEV, generator=g) + 0.5
                junk = torch.empty(1 << (20 + rep % 5), device=DEV)              # moves the next allocations
                with _f23('off') as m:
                    ref = m.modulated_conv2d(x, wt, s, demodulate=True, padding=pad, x_bound=16.0)
                with _f23('on', tn) as m:
                    got = m.modulated_conv2d(x, wt, s, demodulate=True, padding=pad, x_bound=16.0)
                worst = max(worst, float((got - ref).abs().max()) / float(ref.abs().max()))
                del junk
    assert worst <= 5e-6, worst


def test_f23_many_tiles_per_workgroup():
    """More tiles than CUs: every persistent workgroup walks several tiles (accumulators re-zeroed, LDS images reused after the
    output exchange); sizes whose tile count is not a multiple of 8 and of the CU count."""
    from oracle import oracle as O
    n, ci, co, h, w = 3, 32, 200, 118, 90
    x = np.clip(rand(5, n, ci, h, w) * 2, -8, 8).astype(np.float32); wt = rand(6, co, ci, 3, 3); s = rand(7, n, ci) + 1
    ref = O.modulated_conv2d(x.astype(np.float64), wt.astype(np.float64), s.astype(np.float64), True, 2, None)
    for tn in (4, 7):
        with _f23('on', tn) as mc:
            y = mc.modulated_conv2d(T(x), T(wt), T(s), demodulate=True, padding=2, x_bound=8.0)
        assert maxabs(y.cpu().numpy(), ref) <= 5e-6 * float(np.abs(ref).max()), tn


@pytest.mark.parametrize('ci,co,h', [(512, 512, 84), (323, 203, 60)])
def test_f23_data_gradient_at_layer_shapes_matches_fp64(ci, co, h):
    """the transform-domain kernel ALSO runs the data gradient of the PTI step (pad-0 form, transposed + flipped
    effective weights, per-sample scale vectors exchanged).  Here at real layer shapes -- 512 -> 512 with an 86^2 gradient (L5 of
    T-1024, N = 1) and 323 -> 203 (L8's channel counts: padded K chunk, an M block of pure padding) -- against the fp64 oracle:
    dx = J^T dy with J the reference's grouped convolution (networks_stylegan3.py:39-62), evaluated as the oracle's convolution of
    dy with the transposed, flipped effective weights, at the forward tolerance (5e-6 relative)."""
    from oracle import oracle as O
    x = np.clip(rand(91, 1, ci, h, h) * 40, -256, 256).astype(np.float32)
    wt = rand(92, co, ci, 3, 3); s = (rand(93, 1, ci) + 1).astype(np.float32)
    dy = (rand(94, 1, co, h + 2, h + 2) * 3).astype(np.float32)
    gain = 0.8
    with _f23('on') as mc:
        xt = T(x).requires_grad_(True)
        y = mc.modulated_conv2d(xt, T(wt), T(s), demodulate=True, padding=2, input_gain=torch.tensor(gain, device=DEV), x_bound=256.0)
        from torch_utils import _sg3abi as abi
        # the backward's convolution: I and O exchanged, the (h + 2)^2 gradient as its input, pad 0
        assert abi.load().sg3_modconv_f23_supported(abi.SG3_F32, co, ci, h + 2, h + 2, 3, 0, 0) == 1 and mc._f23_wanted(co, ci, h + 2, h + 2, 0)
        before = abi.launch_count
        (dx,) = torch.autograd.grad(y, xt, T(dy))
        assert abi.launch_count > before
    # fp64 effective weights of the single sample (reference :39-56), then the adjoint of the pad-2 correlation
    w64, s64 = wt.astype(np.float64), s.astype(np.float64)
    wn = w64 / np.sqrt(np.mean(np.square(w64), axis=(1, 2, 3), keepdims=True))
    sn = s64 / np.sqrt(np.mean(np.square(s64)))
    we = wn * sn[0][None, :, None, None]
    we = we / np.sqrt(np.sum(np.square(we), axis=(1, 2, 3), keepdims=True) + 1e-8) * gain                      # [O,I,3,3]
    wadj = np.ascontiguousarray(we[:, :, ::-1, ::-1].transpose(1, 0, 2, 3))                                         # [I,O,3,3]
    ref = O.modulated_conv2d(dy.astype(np.float64), wadj, np.ones([1, co]), False, 0, None)                        # [1,I,h,h]
    assert ref.shape == tuple(dx.shape)
    err = maxabs(dx.cpu().numpy(), ref) / max(1.0, float(np.abs(ref).max()))
    print(f'data gradient {co} -> {ci} @ {h + 2}^2, relative max error vs fp64: {err:.2e}')
    assert err <= 5e-6, err
    # and the forward of the same call, for completeness of the pair
    fref = O.modulated_conv2d(x.astype(np.float64), w64, s64, True, 2, gain)
    assert maxabs(y.detach().cpu().numpy(), fref) <= 5e-6 * max(1.0, float(np.abs(fref).max()))


@pytest.mark.parametrize('n,ci,co,h,w,pad', [
    (2, 64, 64, 30, 30, 2),
    (1, 323, 203, 22, 26, 2),          # padded K chunk, an M block of pure padding
    (2, 81, 51, 40, 70, 2),            # second M block partly inactive, ragged column tiles
    (1, 512, 512, 20, 36, 2),          # 32 chunks
    (3, 17, 130, 35, 34, 0),           # pad 0, two K chunks
])
@pytest.mark.parametrize('tn', [4, 5, 7])
def test_f23_fp16_form_matches_the_direct_fp16_kernel_and_the_oracles(n, ci, co, h, w, pad, tn):
    """SG3_CONV_F16_F23 (round 4): fp16 tensors -- the reference's `use_fp16` layers (networks_stylegan3.py:355-366, :61) -- in the
    transform domain: transformed inputs / weights rounded to fp16 once, one product per K step, fp32 accumulation, fp16 output.
    Yardsticks: the fp64 oracle (the exact result), the fp16-rounding oracle (`modulated_conv2d_fp16`: the reference's rounding points,
    unpinned) and the direct fp16 kernel, which rounds x * s and w instead of their transforms: the transform-domain form may be a
    small factor less accurate than the direct one (three rounded products are summed per output) but stays an fp16-grade result."""
    from oracle import oracle as O
    from torch_utils import _sg3abi as abi
    x = np.clip(rand(171, n, ci, h, w) * 40, -256, 256).astype(np.float16)
    wt = rand(172, co, ci, 3, 3); s = (rand(173, n, ci) + 1).astype(np.float32)
    ref64 = O.modulated_conv2d(x.astype(np.float64), wt.astype(np.float64), s.astype(np.float64), True, pad, 0.8)
    ref16 = O.modulated_conv2d_fp16(x, wt, s, True, pad, 0.8).astype(np.float64)
    scale = max(1.0, float(np.abs(ref64).max()))
    kw = dict(demodulate=True, padding=pad, input_gain=torch.tensor(0.8, device=DEV), x_bound=256.0)
    with _f23('on', tn) as mc:
        assert abi.load().sg3_modconv_f23_supported(abi.SG3_F16, ci, co, h, w, 3, pad, 0) == 1
        y = mc.modulated_conv2d(T(x), T(wt), T(s), **kw)
    with _f23('off') as mc:
        yd = mc.modulated_conv2d(T(x), T(wt), T(s), **kw)
    assert y.dtype == torch.float16 and tuple(y.shape) == ref64.shape
    y64, yd64 = y.float().cpu().numpy().astype(np.float64), yd.float().cpu().numpy().astype(np.float64)
    e23, ed, eo = maxabs(y64, ref64) / scale, maxabs(yd64, ref64) / scale, maxabs(ref16, ref64) / scale
    m23, md = float(np.abs(y64 - ref64).mean()) / scale, float(np.abs(yd64 - ref64).mean()) / scale
    print(f'vs fp64, relative to max |y|: transform-domain fp16 max {e23:.2e} mean {m23:.2e}; direct fp16 max {ed:.2e} mean {md:.2e}; fp16 oracle max {eo:.2e}')
    assert bool(torch.isfinite(y).all())
    assert e23 <= 4e-3 and m23 <= 4e-4, (e23, m23)
    assert e23 <= 3.0 * max(ed, eo) + 2.5e-4 and m23 <= 3.0 * md + 1e-5, (e23, ed, eo, m23, md)
