"""CPU: the product package (graph, filter design, schedule, API plumbing, `ref` operator path, pickling) against
the reference's golden vectors.  This is BASELINE config[0]'s plumbing case; GPU parity lives in test_gpu_*.py."""
import io
import pickle

import numpy as np
import pytest
import torch

from golden_cases import BIAS_ACT_CASES, FLRELU_CASES, FLRELU_GRAD_CASES, MODCONV_CASES, UPFIRDN_CASES, make_filter, rand
from helpers import build_product_generator, golden, manifest, maxabs, product_design
from synth_weights import CONFIGS, make_user_transform, synth_ws


def T(a):
    return None if a is None else torch.from_numpy(np.ascontiguousarray(a))


@pytest.mark.parametrize('cfg', ['T1024', 'R1024', 'R512', 'T256', 'Ttiny', 'Rtiny'])
def test_state_dict_manifest_and_filters(cfg):
    """Key names / shapes equal the reference's state_dict; designed taps and padding equal the reference's."""
    from models.stylegan3.networks_stylegan3 import Generator
    G = Generator(**CONFIGS[cfg])
    assert {k: list(v.shape) for k, v in G.state_dict().items()} == manifest()[cfg]
    assert sum(p.numel() for p in G.parameters()) == manifest()[cfg + '/num_params']
    gf = golden('filters')
    for name in G.synthesis.layer_names:
        layer = getattr(G.synthesis, name)
        key = f'{cfg}/{name}'
        assert layer.padding == gf[key + '/padding'].tolist()
        geom = [layer.in_channels, layer.out_channels, int(layer.in_size[0]), int(layer.out_size[0]), layer.up_factor, layer.down_factor,
                layer.up_taps, layer.down_taps, layer.conv_kernel, int(layer.use_fp16), int(layer.down_radial)]
        assert geom == gf[key + '/geom'].tolist()
        for buf, nm in ((layer.up_filter, 'up_filter'), (layer.down_filter, 'down_filter')):
            if buf is None:
                assert key + '/' + nm not in gf.files
            else:
                assert maxabs(buf.numpy(), gf[key + '/' + nm]) <= 1e-7


@pytest.mark.parametrize('cfg', ['Ttiny', 'Rtiny'])
def test_tiny_network_ref_path(cfg):
    g = golden('net_tiny')
    G = build_product_generator(cfg)
    ws = T(synth_ws(2, G.num_ws, G.w_dim, seed=1))
    assert maxabs(G.synthesis(ws, noise_mode='const', force_fp32=True).numpy(), g[cfg + '/img']) <= 1e-6
    assert maxabs(G.synthesis.input(ws[:, 0]).numpy(), g[cfg + '/input']) <= 1e-6
    G.synthesis.input.transform = T(make_user_transform())
    assert maxabs(G.synthesis(ws, noise_mode='const', force_fp32=True).numpy(), g[cfg + '/img_tr']) <= 1e-6
    trb = np.stack([make_user_transform((0.1, -0.05), 15.0), make_user_transform((-0.2, 0.07), -30.0)])
    G.synthesis.input.transform = T(trb)            # callers assign a batched [B,3,3] (reference psp3.py:64-65)
    assert maxabs(G.synthesis(ws, noise_mode='const', force_fp32=True).numpy(), g[cfg + '/img_trb']) <= 1e-6
    assert maxabs(G.synthesis.input(ws[:, 0]).numpy(), g[cfg + '/input_trb']) <= 1e-6
    G.synthesis.input.transform = torch.eye(3)
    all_s = G.synthesis.W2S(ws)
    assert set(all_s) == {'input'} | set(G.synthesis.layer_names)
    for k, v in all_s.items():
        assert maxabs(v.numpy(), g[f'{cfg}/w2s/{k}']) <= 1e-6
    assert maxabs(G.synthesis(None, all_s=all_s, noise_mode='const', force_fp32=True).numpy(), g[cfg + '/img_alls']) <= 1e-6
    z = T(np.random.RandomState(5).randn(3, G.z_dim).astype(np.float32))
    assert maxabs(G.mapping(z, None).numpy(), g[cfg + '/mapping_psi1']) <= 1e-5
    assert maxabs(G.mapping(z, None, truncation_psi=0.7).numpy(), g[cfg + '/mapping_psi07']) <= 1e-5
    assert maxabs(G.mapping(z, None, truncation_psi=0.5, truncation_cutoff=8).numpy(), g[cfg + '/mapping_psi05_cut8']) <= 1e-5
    assert maxabs(G(z[:1], None, truncation_psi=0.7, noise_mode='const', force_fp32=True).numpy(), g[cfg + '/gen_psi07']) <= 1e-6


@pytest.mark.parametrize('name', sorted(FLRELU_CASES))
def test_filtered_lrelu_ref_path(name):
    from torch_utils.ops import filtered_lrelu
    c = FLRELU_CASES[name]
    x = T(rand(11, *c['shape'])); b = T(rand(12, c['shape'][1])) if c['bias'] else None
    fu, fd = T(make_filter(c['fu'], product_design)), T(make_filter(c['fd'], product_design))
    y = filtered_lrelu.filtered_lrelu(x, fu=fu, fd=fd, b=b, up=c['up'], down=c['down'], padding=c['padding'], gain=c['gain'],
                                      slope=c['slope'], clamp=c['clamp'], flip_filter=c['flip'])
    assert maxabs(y.numpy(), golden('ops')['flrelu/' + name]) <= 1e-6


@pytest.mark.parametrize('name', FLRELU_GRAD_CASES)
def test_filtered_lrelu_ref_grad(name):
    from torch_utils.ops import filtered_lrelu
    c = FLRELU_CASES[name]
    x = T(rand(11, *c['shape'])).requires_grad_(True)
    b = T(rand(12, c['shape'][1])).requires_grad_(True) if c['bias'] else None
    fu, fd = T(make_filter(c['fu'], product_design)), T(make_filter(c['fd'], product_design))
    y = filtered_lrelu.filtered_lrelu(x, fu=fu, fd=fd, b=b, up=c['up'], down=c['down'], padding=c['padding'], gain=c['gain'],
                                      slope=c['slope'], clamp=c['clamp'], flip_filter=c['flip'])
    (y * T(rand(13, *y.shape))).sum().backward()
    g = golden('grads')
    assert maxabs(x.grad.numpy(), g[name + '/dx']) <= 1e-5
    if b is not None:
        assert maxabs(b.grad.numpy(), g[name + '/db']) <= 1e-3


@pytest.mark.parametrize('name', sorted(UPFIRDN_CASES))
def test_upfirdn2d_ref_path(name):
    from torch_utils.ops import upfirdn2d
    c = UPFIRDN_CASES[name]
    y = upfirdn2d.upfirdn2d(T(rand(21, *c['shape'])), T(make_filter(c['f'], product_design)), up=c['up'], down=c['down'],
                            padding=c['padding'], flip_filter=c['flip'], gain=c['gain'])
    assert maxabs(y.numpy(), golden('ops')['upfirdn/' + name]) <= 1e-6


@pytest.mark.parametrize('name', sorted(BIAS_ACT_CASES))
def test_bias_act_ref_path(name):
    from torch_utils.ops import bias_act
    c = BIAS_ACT_CASES[name]
    x = T(rand(31, *c['shape']) * 2); b = T(rand(32, c['shape'][c['dim']])) if c['bias'] else None
    y = bias_act.bias_act(x, b, dim=c['dim'], act=c['act'], alpha=c['alpha'], gain=c['gain'], clamp=c['clamp'])
    assert maxabs(y.numpy(), golden('ops')['bias_act/' + name]) <= 1e-6


@pytest.mark.parametrize('name', sorted(MODCONV_CASES))
def test_modulated_conv2d_ref_path(name):
    from models.stylegan3.networks_stylegan3 import modulated_conv2d
    c = MODCONV_CASES[name]
    x = T(rand(41, c['n'], c['ci'], c['h'], c['w'])); w = T(rand(42, c['co'], c['ci'], c['k'], c['k'])); s = T(rand(43, c['n'], c['ci']) + 1)
    ig = None if c['input_gain'] is None else torch.tensor(c['input_gain'])
    y = modulated_conv2d(x, w, s, demodulate=c['demodulate'], padding=c['k'] - 1, input_gain=ig)
    assert maxabs(y.numpy(), golden('ops')['modconv/' + name]) <= 1e-5


def test_upfirdn2d_helpers():
    """setup_filter / filter2d / upsample2d / downsample2d keep the reference's conventions."""
    from torch_utils.ops import upfirdn2d
    f = upfirdn2d.setup_filter([1, 3, 3, 1])
    assert f.shape == (4, 4) and abs(float(f.sum()) - 1) < 1e-6
    f8 = upfirdn2d.setup_filter(list(range(1, 9)))
    assert f8.ndim == 1
    x = torch.randn(1, 2, 8, 8)
    assert upfirdn2d.filter2d(x, f).shape == (1, 2, 8, 8)
    assert upfirdn2d.upsample2d(x, f, up=2).shape == (1, 2, 16, 16)
    assert upfirdn2d.downsample2d(x, f, down=2).shape == (1, 2, 4, 4)
    # DC gain is preserved by up / down sampling of a constant image (interior)
    c = torch.ones(1, 1, 16, 16)
    assert abs(float(upfirdn2d.upsample2d(c, f, up=2)[0, 0, 8:24, 8:24].mean()) - 1) < 1e-5


def test_persistence_pickle_roundtrip():
    """A pickled generator carries its module source and unpickles through torch_utils.persistence
    (the mechanism official .pkl checkpoints rely on); outputs are unchanged."""
    from torch_utils import persistence
    G = build_product_generator('Ttiny')
    ws = T(synth_ws(1, G.num_ws, G.w_dim, seed=1))
    ref = G.synthesis(ws, noise_mode='const', force_fp32=True)
    assert persistence.is_persistent(G) and G.init_kwargs['img_resolution'] == 64
    buf = io.BytesIO()
    pickle.dump(dict(G_ema=G), buf)
    raw = buf.getvalue()
    assert b'_reconstruct_persistent_obj' in raw and b'class SynthesisLayer' in raw
    G2 = pickle.loads(raw)['G_ema']
    assert type(G2).__name__ == 'Generator' and G2 is not G
    assert torch.equal(G2.synthesis(ws, noise_mode='const', force_fp32=True), ref)


def test_misc_and_dnnlib():
    import dnnlib
    from torch_utils import misc
    d = dnnlib.EasyDict(a=1)
    d.b = 2
    assert d['b'] == 2 and d.a == 1
    with pytest.raises(AttributeError):
        _ = d.c
    misc.assert_shape(torch.zeros(2, 3), [2, None])
    with pytest.raises(AssertionError):
        misc.assert_shape(torch.zeros(2, 3), [3, None])
    a, b = torch.nn.Linear(3, 2), torch.nn.Linear(3, 2)
    misc.copy_params_and_buffers(a, b, require_all=True)
    assert torch.equal(a.weight, b.weight)
    assert dnnlib.util.get_obj_by_name('torch.nn.Linear') is torch.nn.Linear


def test_sg3generator_wrapper(tmp_path):
    """SG3Generator builds config-T / config-R, loads a .pt state_dict (dropping a mismatched transform) and a .pkl."""
    from models.stylegan3 import model as m
    from models.stylegan3.networks_stylegan3 import Generator
    G = Generator(z_dim=512, c_dim=0, w_dim=512, img_resolution=256, img_channels=3, channel_base=2048, channel_max=16)
    sd = G.state_dict()
    sd['synthesis.input.transform'] = torch.eye(3).repeat(2, 1, 1)      # batched transform saved by a caller
    torch.save(sd, tmp_path / 'g.pt')
    m.CONFIG_T.update(channel_base=2048, channel_max=16)
    try:
        w = m.SG3Generator(checkpoint_path=tmp_path / 'g.pt', res=256, config='landscape')
    finally:
        m.CONFIG_T.update(channel_base=32768, channel_max=512)
    assert torch.equal(w.decoder.synthesis.L0_36_16.weight, G.synthesis.L0_36_16.weight)
    with open(tmp_path / 'g.pkl', 'wb') as f:
        pickle.dump(dict(G_ema=G), f)
    w2 = m.SG3Generator(checkpoint_path=str(tmp_path / 'g.pkl'), device='cpu')
    assert torch.equal(w2.decoder.synthesis.L0_36_16.weight, G.synthesis.L0_36_16.weight)


@pytest.mark.parametrize('transpose', [False, True])
def test_conv2d_gradfix_custom_op_second_order(transpose):
    """with `enabled` set the convolutions go through _ConvNd, whose weight gradient must stay differentiable w.r.t.
    x as well as dy (reference conv2d_gradfix.py:175-190).  gradgradcheck in fp64 against numerical second derivatives."""
    from torch_utils.ops.conv2d_gradfix import _ConvNd
    r = torch.Generator().manual_seed(3)
    x = torch.randn(2, 3, 6, 7, generator=r, dtype=torch.float64, requires_grad=True)
    w = torch.randn((3, 4, 3, 3) if transpose else (4, 3, 3, 3), generator=r, dtype=torch.float64, requires_grad=True)
    b = torch.randn(4, generator=r, dtype=torch.float64, requires_grad=True)
    fn = lambda x_, w_, b_: _ConvNd.apply(x_, w_, b_, transpose, (2, 2) if transpose else (1, 1), (1, 1), (1, 1) if transpose else (0, 0), (1, 1), 1)  # noqa: E731
    want = (torch.nn.functional.conv_transpose2d(x, w, b, 2, 1, 1) if transpose else torch.nn.functional.conv2d(x, w, b, 1, 1))
    assert torch.allclose(fn(x, w, b), want)
    assert torch.autograd.gradcheck(fn, (x, w, b))
    assert torch.autograd.gradgradcheck(fn, (x, w, b))
