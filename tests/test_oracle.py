"""CPU: the oracle (oracle/sg3_oracle.c + oracle/oracle.py) against golden vectors produced by the reference."""
import numpy as np
import pytest

from golden_cases import BIAS_ACT_CASES, FLRELU_CASES, MODCONV_CASES, UPFIRDN_CASES, make_filter, rand
from helpers import build_oracle_generator, golden, manifest, maxabs, oracle_design
from oracle import oracle as O
from synth_weights import CONFIGS, make_user_transform, synth_ws


@pytest.mark.parametrize('name', sorted(FLRELU_CASES))
def test_filtered_lrelu(name):
    c = FLRELU_CASES[name]
    x = rand(11, *c['shape'])
    b = rand(12, c['shape'][1]) if c['bias'] else None
    fu, fd = make_filter(c['fu'], oracle_design), make_filter(c['fd'], oracle_design)
    y = O.filtered_lrelu(x, fu, fd, b, c['up'], c['down'], c['padding'], c['gain'], c['slope'], c['clamp'], c['flip'])
    ref = golden('ops')['flrelu/' + name]
    assert y.shape == ref.shape
    assert maxabs(y, ref) <= 2e-6        # fp32 rounding-order differences only
    y64 = O.filtered_lrelu(x.astype(np.float64), fu, fd, None if b is None else b.astype(np.float64), c['up'], c['down'],
                           c['padding'], c['gain'], c['slope'], c['clamp'], c['flip'])
    assert maxabs(y64, golden('ops')['flrelu64/' + name]) <= 1e-12


@pytest.mark.parametrize('name', sorted(UPFIRDN_CASES))
def test_upfirdn2d(name):
    c = UPFIRDN_CASES[name]
    y = O.upfirdn2d(rand(21, *c['shape']), make_filter(c['f'], oracle_design), c['up'], c['down'], c['padding'], c['flip'], c['gain'])
    ref = golden('ops')['upfirdn/' + name]
    assert y.shape == ref.shape and maxabs(y, ref) <= 1e-6


@pytest.mark.parametrize('name', sorted(BIAS_ACT_CASES))
def test_bias_act(name):
    c = BIAS_ACT_CASES[name]
    x = rand(31, *c['shape']) * 2
    b = rand(32, c['shape'][c['dim']]) if c['bias'] else None
    y = O.bias_act(x, b, c['dim'], c['act'], c['alpha'], c['gain'], c['clamp'])
    assert maxabs(y, golden('ops')['bias_act/' + name]) <= 2e-6


@pytest.mark.parametrize('name', sorted(MODCONV_CASES))
def test_modulated_conv2d(name):
    c = MODCONV_CASES[name]
    x = rand(41, c['n'], c['ci'], c['h'], c['w']); w = rand(42, c['co'], c['ci'], c['k'], c['k']); s = rand(43, c['n'], c['ci']) + 1
    y = O.modulated_conv2d(x, w, s, c['demodulate'], c['k'] - 1, c['input_gain'])
    ref = golden('ops')['modconv/' + name]
    assert y.shape == ref.shape and maxabs(y, ref) <= 2e-5 * max(1.0, float(np.abs(ref).max()))


@pytest.mark.parametrize('cfg', ['T1024', 'T256', 'R1024', 'R512', 'Ttiny', 'Rtiny'])
def test_filter_design_and_schedule(cfg):
    """Kaiser / radial taps, padding and per-layer geometry of every layer of every config."""
    gf = golden('filters')
    sched = O.layer_schedule(**CONFIGS[cfg])
    for layer in sched['layers']:
        key = f"{cfg}/{layer['name']}"
        geom = gf[key + '/geom']
        assert [layer['in_channels'], layer['out_channels'], layer['in_size'], layer['out_size'], layer['up'], layer['down'],
                layer['up_taps'], layer['down_taps'], layer['conv_kernel'], int(layer['use_fp16']), int(layer['down_radial'])] == geom.tolist()
        assert layer['padding'] == gf[key + '/padding'].tolist()
        fu = O.design_lowpass_filter(**layer['up_filter_args'])
        fd = O.design_lowpass_filter(**layer['down_filter_args'])
        for f, nm in ((fu, 'up_filter'), (fd, 'down_filter')):
            if f is None:
                assert key + '/' + nm not in gf.files
            else:
                assert maxabs(f, gf[key + '/' + nm]) <= 2e-7, (key, nm)


@pytest.mark.parametrize('cfg', ['Ttiny', 'Rtiny'])
def test_tiny_network(cfg):
    g = golden('net_tiny')
    sd, sched = build_oracle_generator(cfg)
    ws = synth_ws(2, sched['num_ws'], sched['w_dim'], seed=1)
    img, feats = O.synthesis(sd, sched, ws=ws, return_layers=True)
    assert maxabs(img, g[cfg + '/img']) <= 2e-5
    assert maxabs(O.synthesis_input(sd, sched, w=ws[:, 0]), g[cfg + '/input']) <= 2e-5
    for layer, f in zip(sched['layers'], feats):
        assert maxabs(f[:, :2, :24, :24], g[f"{cfg}/feat/{layer['name']}"]) <= 1e-4
    # user transforms ([3,3] and batched [B,3,3])
    assert maxabs(O.synthesis(sd, sched, ws=ws, transform=make_user_transform()), g[cfg + '/img_tr']) <= 2e-5
    trb = np.stack([make_user_transform((0.1, -0.05), 15.0), make_user_transform((-0.2, 0.07), -30.0)])
    assert maxabs(O.synthesis(sd, sched, ws=ws, transform=trb), g[cfg + '/img_trb']) <= 2e-5
    # StyleSpace path
    all_s = O.w2s(sd, sched, ws)
    for k, v in all_s.items():
        assert maxabs(v, g[f'{cfg}/w2s/{k}']) <= 1e-5
    assert maxabs(O.synthesis(sd, sched, all_s=all_s), g[cfg + '/img_alls']) <= 2e-5
    # mapping
    z = np.random.RandomState(5).randn(3, CONFIGS[cfg]['z_dim']).astype(np.float32)
    assert maxabs(O.mapping(sd, z, sched['num_ws']), g[cfg + '/mapping_psi1']) <= 1e-4
    assert maxabs(O.mapping(sd, z, sched['num_ws'], truncation_psi=0.7), g[cfg + '/mapping_psi07']) <= 1e-4
    assert maxabs(O.mapping(sd, z, sched['num_ws'], truncation_psi=0.5, truncation_cutoff=8), g[cfg + '/mapping_psi05_cut8']) <= 1e-4


def test_t256_image():
    """BASELINE config[0]: 256x256 config-T single image on CPU."""
    g = golden('net_t256')
    sd, sched = build_oracle_generator('T256')
    ws = synth_ws(1, sched['num_ws'], sched['w_dim'], seed=1)
    img, feats = O.synthesis(sd, sched, ws=ws, return_layers=True)
    assert img.shape == (1, 3, 256, 256)
    assert maxabs(img, g['T256/img']) <= 1e-4
    stats = np.asarray([[f.mean(), f.std(), np.abs(f).max()] for f in feats])
    assert np.abs(stats - g['T256/stats']).max() <= 1e-3


def test_manifest_counts():
    man = manifest()
    assert man['T1024/num_params'] == 21785091 or man['T1024/num_params'] > 2.1e7
    assert len(man['T1024']) == 114


def test_mixed_fp16_forward_rounds_where_the_reference_does():
    """`oracle.synthesis(mixed_fp16=True)` (the yardstick of the GPU mixed-precision tests; no reference fixture exists for it: the
    reference's fp16 path needs a GPU): every use_fp16 layer hands on fp16-representable activations, its convolution equals the fp32
    convolution of the fp16-rounded operands, and the image stays within fp16 rounding of the pinned fp32 forward."""
    sd, sched = build_oracle_generator('Tmini')
    assert all(l['use_fp16'] for l in sched['layers'])
    ws = synth_ws(1, sched['num_ws'], sched['w_dim'], seed=2)
    img32, feats32 = O.synthesis(sd, sched, ws=ws, return_layers=True)
    img16, feats16 = O.synthesis(sd, sched, ws=ws, mixed_fp16=True, return_layers=True)
    for f in feats16[1:]:                                                   # layer outputs (the input features stay fp32)
        assert np.array_equal(f, f.astype(np.float16).astype(np.float32))
    assert 0 < maxabs(img16, img32) <= 2e-3 * max(1.0, float(np.abs(img32).max()))
    # the convolution alone: fp16 operands, fp32 accumulation, fp16 result
    r = np.random.RandomState(3)
    x = (r.randn(2, 12, 9, 9) * 3).astype(np.float32); w = r.randn(8, 12, 3, 3).astype(np.float32); s = (r.randn(2, 12) + 1).astype(np.float32)
    y = O.modulated_conv2d_fp16(x, w, s, demodulate=True, padding=2, input_gain=0.7)
    assert np.array_equal(y, y.astype(np.float16).astype(np.float32))
    ref = O.modulated_conv2d(x.astype(np.float16).astype(np.float32), w, s, demodulate=True, padding=2, input_gain=0.7)
    assert maxabs(y, ref) <= 4e-3 * float(np.abs(ref).max())               # weights rounded to fp16 after modulation + output rounding
