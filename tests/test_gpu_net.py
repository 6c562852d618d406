"""GPU: whole synthesis / generator forwards through the HIP kernels against the reference's golden images.
BASELINE target: max-abs <= 1e-4 on the final image with fp32 (force_fp32) execution."""
import numpy as np
import pytest
import torch

from helpers import build_product_generator, golden, maxabs
from synth_weights import make_user_transform, synth_ws

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def T(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


@pytest.mark.parametrize('cfg', ['Ttiny', 'Rtiny'])
def test_tiny_network(cfg):
    g = golden('net_tiny')
    G = build_product_generator(cfg, device=DEV)
    ws = T(synth_ws(2, G.num_ws, G.w_dim, seed=1))
    import warnings
    with warnings.catch_warnings(), torch.no_grad():
        warnings.simplefilter('ignore')
        feats = []
        hooks = [getattr(G.synthesis, n).register_forward_hook(lambda m, i, o: feats.append(o)) for n in G.synthesis.layer_names]
        img = G.synthesis(ws, noise_mode='const', force_fp32=True)
        for h in hooks:
            h.remove()
        for n, f in zip(G.synthesis.layer_names, feats):
            assert maxabs(f[:, :2, :24, :24].cpu().numpy(), g[f'{cfg}/feat/{n}']) <= 1e-4, n
        assert maxabs(img.cpu().numpy(), g[cfg + '/img']) <= 1e-4
        G.synthesis.input.transform = T(make_user_transform())
        assert maxabs(G.synthesis(ws, noise_mode='const', force_fp32=True).cpu().numpy(), g[cfg + '/img_tr']) <= 1e-4
        trb = np.stack([make_user_transform((0.1, -0.05), 15.0), make_user_transform((-0.2, 0.07), -30.0)])
        G.synthesis.input.transform = T(trb)
        assert maxabs(G.synthesis(ws, noise_mode='const', force_fp32=True).cpu().numpy(), g[cfg + '/img_trb']) <= 1e-4
        G.synthesis.input.transform = torch.eye(3, device=DEV)
        all_s = G.synthesis.W2S(ws)
        for k, v in all_s.items():
            assert maxabs(v.cpu().numpy(), g[f'{cfg}/w2s/{k}']) <= 1e-5
        assert maxabs(G.synthesis(None, all_s=all_s, noise_mode='const', force_fp32=True).cpu().numpy(), g[cfg + '/img_alls']) <= 1e-4
        z = T(np.random.RandomState(5).randn(3, G.z_dim).astype(np.float32))
        assert maxabs(G.mapping(z, None, truncation_psi=0.7).cpu().numpy(), g[cfg + '/mapping_psi07']) <= 1e-4
        assert maxabs(G(z[:1], None, truncation_psi=0.7, noise_mode='const', force_fp32=True).cpu().numpy(), g[cfg + '/gen_psi07']) <= 1e-4


def test_t256_image():
    g = golden('net_t256')
    G = build_product_generator('T256', device=DEV)
    ws = T(synth_ws(1, G.num_ws, G.w_dim, seed=1))
    with torch.no_grad():
        img = G.synthesis(ws, noise_mode='const', force_fp32=True)
    assert maxabs(img.cpu().numpy(), g['T256/img']) <= 1e-4


def test_t1024_batch_against_reference_samples():
    """FFHQ-1024 config-T (BASELINE config[1] geometry): per-layer corner/centre samples, strided image subsample,
    three full rows and global statistics of the reference; plus batch consistency (sample 0 of a batch of 2)."""
    g = golden('net_t1024_stats')
    G = build_product_generator('T1024', device=DEV)
    ws1 = synth_ws(1, G.num_ws, G.w_dim, seed=1)
    ws = T(np.concatenate([ws1, synth_ws(1, G.num_ws, G.w_dim, seed=2)]))
    feats = []
    hooks = [getattr(G.synthesis, n).register_forward_hook(lambda m, i, o: feats.append(o)) for n in G.synthesis.layer_names]
    with torch.no_grad():
        img = G.synthesis(ws, noise_mode='const', force_fp32=True)
    for h in hooks:
        h.remove()
    for n, f in zip(G.synthesis.layer_names, feats):
        assert maxabs(f[:1, :2, :32, :32].cpu().numpy(), g[f'T1024/corner/{n}']) <= 2e-4, n
        assert maxabs(f[:1, -1:, f.shape[2] // 2, :].cpu().numpy(), g[f'T1024/center/{n}']) <= 2e-4, n
    stats = np.asarray([[f[:1].mean().item(), f[:1].std().item(), f[:1].abs().max().item()] for f in feats])
    assert np.abs(stats - g['T1024/stats']).max() <= 2e-3
    assert maxabs(img[:1, :, ::16, ::16].cpu().numpy(), g['T1024/img_sub']) <= 1e-4
    assert maxabs(img[:1, :, [0, 511, 1023], :].cpu().numpy(), g['T1024/img_rows']) <= 1e-4
    assert abs(img[:1].mean().item() - g['T1024/img_stats'][0]) <= 1e-5


def test_synthesis_properties_full_size():
    """Size-independent properties at full 1024 size: determinism, batch-order equivariance, and translation
    equivariance of the alias-free generator under an integer-pixel input translation (interior crop)."""
    G = build_product_generator('T1024', device=DEV)
    ws = T(synth_ws(2, G.num_ws, G.w_dim, seed=3))
    with torch.no_grad():
        a = G.synthesis(ws, noise_mode='const', force_fp32=True)
        b = G.synthesis(ws.flip(0), noise_mode='const', force_fp32=True).flip(0)
        c = G.synthesis(ws, noise_mode='const', force_fp32=True)
    assert torch.equal(a, c)
    assert maxabs(a.cpu().numpy(), b.cpu().numpy()) <= 1e-5


def test_mixed_precision_against_fp16_oracle():
    """the reference's default mixed-precision execution (use_fp16 layers, networks_stylegan3.py:355-366) on the
    HIP path against an ORACLE forward with the reference's rounding points -- activations, per-sample convolution weights, bias
    and layer outputs rounded to fp16, fp32 accumulation (oracle.synthesis(mixed_fp16=True)) -- instead of against the product's
    own fp32 image.  The two fp16 executions round at different places (the HIP kernels modulate the activations, the reference
    the weights), so they agree to a few fp16 ulps of the O(1..50) activations, not bit for bit.  T-256 (BASELINE configs[0]
    architecture, 10 of 15 layers in fp16)."""
    from helpers import build_oracle_generator
    from oracle import oracle as O
    G = build_product_generator('T256', device=DEV)
    flags = [getattr(G.synthesis, n).use_fp16 for n in G.synthesis.layer_names]
    sd, sched = build_oracle_generator('T256')
    assert flags == [l['use_fp16'] for l in sched['layers']] and sum(flags) >= 8
    ws = synth_ws(1, G.num_ws, G.w_dim, seed=1)
    with torch.no_grad():
        mixed = G.synthesis(T(ws), noise_mode='const').cpu().numpy()
    ref16 = O.synthesis(sd, sched, ws=ws, mixed_fp16=True)
    ref32 = O.synthesis(sd, sched, ws=ws)
    d16, d32, dref = np.abs(mixed - ref16), np.abs(mixed - ref32), np.abs(ref16 - ref32)
    print(f'HIP mixed vs fp16 oracle: max {d16.max():.3e} mean {d16.mean():.3e};  vs fp32 oracle: max {d32.max():.3e} mean {d32.mean():.3e};  '
          f'fp16 oracle vs fp32 oracle: max {dref.max():.3e} mean {dref.mean():.3e}')
    # both fp16 executions sit at the same distance from the fp32 image, and closer to each other than twice that distance
    # measured on MI355X: max 4.9e-4, mean 7.4e-5 (image range +-1); the fp16 oracle itself sits 3.2e-4 / 5.6e-5 from the fp32 one
    assert d16.max() <= 2e-3 and d16.mean() <= 3e-4, (d16.max(), d16.mean())
    assert d32.mean() <= 2.0 * dref.mean() + 1e-4


def test_mixed_precision_default_path():
    """The reference's default execution on a GPU at the headline size: layers flagged use_fp16 run with fp16 activations (fp16
    MFMA convolution with fp32 accumulation, fp16 I/O in filtered_lrelu with fp32 arithmetic).  No golden exists (the
    reference's CPU path is always fp32): the yardstick is the oracle forward with the reference's fp16 rounding points
    (oracle.synthesis(mixed_fp16=True), see test_mixed_precision_against_fp16_oracle), with the product's own fp32 image beside it."""
    from helpers import build_oracle_generator
    from oracle import oracle as O
    G = build_product_generator('T1024', device=DEV)
    assert [getattr(G.synthesis, n).use_fp16 for n in G.synthesis.layer_names] == [False] * 5 + [True] * 10
    ws = T(synth_ws(1, G.num_ws, G.w_dim, seed=1))
    feats = []
    hooks = [getattr(G.synthesis, n).register_forward_hook(lambda m, i, o: feats.append(o.dtype)) for n in G.synthesis.layer_names]
    with torch.no_grad():
        a = G.synthesis(ws, noise_mode='const')
        for h in hooks:
            h.remove()
        b = G.synthesis(ws, noise_mode='const', force_fp32=True)
    assert feats == [torch.float32] * 5 + [torch.float16] * 10 and a.dtype == torch.float32
    assert bool(torch.isfinite(a).all())
    d = (a - b).abs()
    sd, sched = build_oracle_generator('T1024')
    ref16 = O.synthesis(sd, sched, ws=ws.cpu().numpy(), mixed_fp16=True)
    d16 = np.abs(a.cpu().numpy() - ref16)
    print(f'mixed vs own fp32: max {float(d.max()):.3e} mean {float(d.mean()):.3e};  mixed vs fp16 oracle: max {d16.max():.3e} mean {d16.mean():.3e}')
    # the same bounds as the T-256 test (the 1e-2 / 1e-3 of round 3 were 20 x what was measured)
    assert d16.max() <= 2e-3 and d16.mean() <= 3e-4, (d16.max(), d16.mean())
    assert float(d.max()) <= 5e-3 and float(d.mean()) <= 5e-4


def test_graphed_synthesis_replay_equals_eager():
    """hipGraph replay (sg3_runtime.GraphedSynthesis: what bench.py times) returns the eager result bit for bit, for the
    W path, for new inputs copied into the static buffers, and for the StyleSpace path."""
    from sg3_runtime import GraphedSynthesis
    G = build_product_generator('T256', device=DEV)
    ws = T(synth_ws(2, G.num_ws, G.w_dim, seed=1))
    ws2 = T(synth_ws(2, G.num_ws, G.w_dim, seed=2))
    g = GraphedSynthesis(G, 2)
    with torch.no_grad():
        e1 = G.synthesis(ws, noise_mode='const', force_fp32=True)
        e2 = G.synthesis(ws2, noise_mode='const', force_fp32=True)
        assert torch.equal(g(ws).clone(), e1)
        assert torch.equal(g(ws2).clone(), e2)
        assert maxabs(e1[:1].cpu().numpy(), golden('net_t256')['T256/img']) <= 1e-4
        all_s = G.synthesis.W2S(ws)
        gs = GraphedSynthesis(G, 2, all_s_template=all_s)
        assert torch.equal(gs(all_s=all_s).clone(), G.synthesis(None, all_s=all_s, noise_mode='const', force_fp32=True))


@pytest.mark.parametrize('cfg', ['R512', 'R1024'])
def test_config_r_full_size_against_reference_samples(cfg):
    """Config R at full size (1x1 split-precision GEMM, radial streaming filtered_lrelu): per-layer corner / centre
    samples, strided image subsample, three full rows and global statistics of the reference's forward."""
    g = golden('net_r_stats')
    G = build_product_generator(cfg, device=DEV)
    ws = T(synth_ws(1, G.num_ws, G.w_dim, seed=1))
    feats = []
    hooks = [getattr(G.synthesis, n).register_forward_hook(lambda m, i, o: feats.append(o)) for n in G.synthesis.layer_names]
    with torch.no_grad():
        img = G.synthesis(ws, noise_mode='const', force_fp32=True)
    for h in hooks:
        h.remove()
    res = G.img_resolution
    for n, f in zip(G.synthesis.layer_names, feats):
        assert maxabs(f[:1, :2, :32, :32].cpu().numpy(), g[f'{cfg}/corner/{n}']) <= 2e-4, n
        assert maxabs(f[:1, -1:, f.shape[2] // 2, :].cpu().numpy(), g[f'{cfg}/center/{n}']) <= 2e-4, n
    stats = np.asarray([[f[:1].mean().item(), f[:1].std().item(), f[:1].abs().max().item()] for f in feats])
    assert np.abs(stats - g[f'{cfg}/stats']).max() <= 2e-3
    assert maxabs(img[:1, :, ::16, ::16].cpu().numpy(), g[f'{cfg}/img_sub']) <= 1e-4
    assert maxabs(img[:1, :, [0, res // 2 - 1, res - 1], :].cpu().numpy(), g[f'{cfg}/img_rows']) <= 1e-4
    assert abs(img[:1].mean().item() - g[f'{cfg}/img_stats'][0]) <= 1e-5


@pytest.mark.gpu
def test_batched_prep_matches_per_layer_prep():
    """Inference batches the weight / style preparation of all convolutions into two launches (prepare_batch), evaluates all
    affine layers in one (affine_batch) and the input's transform algebra in one (input_transform); with the switch off every
    layer prepares for itself with torch ops.  Same convolution kernels on the same prepared operands: a convolution with a
    batch-prepared entry equals the self-preparing call bit for bit, cached packed weights included; the images agree to
    rounding (the affine dot products are summed in another order).  A prepared entry used for another call shape is refused."""
    import torch
    from synth_weights import synth_ws
    from torch_utils.ops import modulated_conv as mc
    G = build_product_generator('Ttiny', device='cuda:0')
    ws = torch.from_numpy(synth_ws(2, G.num_ws, G.w_dim, 3)).cuda()
    with torch.no_grad():
        a = G.synthesis(ws, noise_mode='const', force_fp32=True)
        a16 = G.synthesis(ws, noise_mode='const')
        s = G.synthesis.W2S(ws)
        a_s = G.synthesis(None, all_s=s, noise_mode='const', force_fp32=True)
        G.synthesis.batch_prep = False
        try:
            b = G.synthesis(ws, noise_mode='const', force_fp32=True)
            b16 = G.synthesis(ws, noise_mode='const')
        finally:
            del G.synthesis.batch_prep
    for x, y, tol in ((a, b, 5e-6), (a_s, b, 5e-6), (a16, b16, 2e-3)):
        assert float((x - y).abs().max()) <= tol * max(1.0, float(y.abs().max()))
    with torch.no_grad():
        for j in (0, 3, len(G.synthesis.layers()) - 1):
            layer = G.synthesis.layers()[j]
            spec = layer.conv_spec(s[G.synthesis.layer_names[j]], 2, True)
            x = torch.randn(2, layer.in_channels, int(layer.in_size[1]), int(layer.in_size[0]), device='cuda:0').clamp(-3, 3)
            kw = dict(padding=spec['padding'], input_gain=spec['input_gain'], x_bound=3.0, demodulate=spec['demodulate'])
            spec = dict(spec, x_bound=3.0)
            own = mc.modulated_conv2d(x, layer.weight, spec['s'], **kw)
            first = mc.modulated_conv2d(x, layer.weight, spec['s'], prepared=mc.prepare_batch([spec])[0], **kw)
            again = mc.prepare_batch([spec])[0]                      # second time: the packed weights come from the cache
            assert again.params.reuseWeights == 1
            assert torch.equal(own, first) and torch.equal(own, mc.modulated_conv2d(x, layer.weight, spec['s'], prepared=again, **kw))
        # a changed weight is packed again (optimiser steps, copy_, load_state_dict bump the version)
        layer.weight.mul_(1.5)
        changed = mc.prepare_batch([spec])[0]
        assert changed.params.reuseWeights == 0
        assert torch.equal(mc.modulated_conv2d(x, layer.weight, spec['s'], prepared=changed, **kw), mc.modulated_conv2d(x, layer.weight, spec['s'], **kw))
    layer = G.synthesis.layers()[0]
    spec = layer.conv_spec(s[G.synthesis.layer_names[0]], 2, True)
    pr = mc.prepare_batch([spec])[0]
    x = torch.randn(2, layer.in_channels, int(layer.in_size[1]) + 2, int(layer.in_size[0]), device='cuda:0')
    with pytest.raises(RuntimeError, match='prepared for'):
        mc.modulated_conv2d(x, layer.weight, spec['s'], padding=spec['padding'], input_gain=spec['input_gain'], x_bound=1e3, prepared=pr)


@pytest.mark.gpu
@pytest.mark.parametrize('cfg', ['Ttiny', 'Rtiny', 'T1024'])
def test_affine_batch_and_input_transform_match_the_torch_ops(cfg):
    """All affine layers in one launch == each layer's own addmm; the input's transform algebra in one launch == the torch op
    chain of SynthesisInput.forward ([3,3] and per-sample [N,3,3] user transforms, t normalised inside or before).  Changing a
    parameter in place is seen (the packs are keyed on parameter versions)."""
    import torch
    from synth_weights import make_user_transform, synth_ws
    from torch_utils.ops import fourier_features as ff
    G = build_product_generator(cfg, device='cuda:0')
    syn = G.synthesis
    layers = syn.layers()
    n = 3
    ws = torch.from_numpy(synth_ws(n, G.num_ws, G.w_dim, 8)).cuda()
    fcs = [syn.input.affine] + [layer.affine for layer in layers]
    post = [1.0] + [layer.style_gain() for layer in layers]

    def check():
        outs = syn._affine_pack()(ws, fcs, range(len(fcs)), post)
        assert len(outs) == len(fcs) and all(o.is_contiguous() for o in outs)
        refs = [syn.input.affine(ws[:, 0])] + [layer.styles_from_w(ws[:, j + 1]) for j, layer in enumerate(layers)]
        for o, r in zip(outs, refs):
            assert o.shape == r.shape and float((o - r).abs().max()) <= 2e-6 * max(1.0, float(r.abs().max()))
        return outs
    with torch.no_grad():
        outs = check()
        layers[1].affine.weight.mul_(0.5); layers[2].affine.bias.add_(0.25)
        outs2 = check()
        assert not torch.equal(outs[2], outs2[2]) and not torch.equal(outs[3], outs2[3])
        # non-contiguous latents (a slice of a wider tensor) through the strides
        wide = torch.randn(n, G.num_ws + 2, G.w_dim, device='cuda:0')
        got = syn._affine_pack()(wide[:, 1:-1], fcs, range(len(fcs)), post)
        assert float((got[4] - layers[3].styles_from_w(wide[:, 5])).abs().max()) <= 2e-6 * float(got[4].abs().max())

        inp = syn.input
        t_raw = outs2[0]
        users = [torch.eye(3, device='cuda:0'), torch.from_numpy(make_user_transform((0.1, -0.05), 15.0)).float().cuda(),
                 torch.from_numpy(np.stack([make_user_transform((0.1 * i, -0.05), 15.0 * i) for i in range(n)])).float().cuda()]
        for user in users:
            t = t_raw / t_raw[:, :2].norm(dim=1, keepdim=True)
            rot = torch.eye(3, device='cuda:0').repeat(n, 1, 1)
            rot[:, 0, 0], rot[:, 0, 1], rot[:, 1, 0], rot[:, 1, 1] = t[:, 0], -t[:, 1], t[:, 1], t[:, 0]
            trans = torch.eye(3, device='cuda:0').repeat(n, 1, 1)
            trans[:, 0, 2], trans[:, 1, 2] = -t[:, 2], -t[:, 3]
            m = rot @ trans @ user
            f = inp.freqs.unsqueeze(0)
            ph = inp.phases.unsqueeze(0) + (f @ m[:, :2, 2:]).squeeze(2)
            f = f @ m[:, :2, :2]
            am = (1 - (f.norm(dim=2) - inp.bandwidth) / (inp.sampling_rate / 2 - inp.bandwidth)).clamp(0, 1)
            for tt, normalise in ((t_raw, True), (t, False)):
                gf, gp, ga = ff.input_transform(tt, user, inp.freqs, inp.phases, inp.bandwidth, inp.sampling_rate, normalise)
                scale = max(1.0, float(f.abs().max()))
                assert float((gf - f).abs().max()) <= 2e-6 * scale and float((gp - ph).abs().max()) <= 2e-6 * scale
                assert float((ga - am).abs().max()) <= 2e-6


def test_torgb_epilogue_fused_and_refused():
    """The ToRGB convolution fuses bias + clamp (+ scale) into its stores: identical to conv -> filtered_lrelu(up = down = 1,
    gain = slope = 1) -> scale.  Other convolution shapes refuse the epilogue (C ABI and Python wrapper)."""
    from torch_utils.ops import filtered_lrelu as fl
    from torch_utils.ops import modulated_conv as mc
    g = torch.Generator(device=DEV).manual_seed(11)
    x = torch.randn([2, 32, 64, 64], device=DEV, generator=g) * 50
    w = torch.randn([3, 32, 1, 1], device=DEV, generator=g)
    s = torch.randn([2, 32], device=DEV, generator=g) + 1
    b = torch.randn([3], device=DEV, generator=g) * 10
    with torch.no_grad():
        plain = mc.modulated_conv2d(x, w, s, demodulate=False, padding=0, input_gain=torch.tensor(0.7, device=DEV))
        want = fl.filtered_lrelu(plain, b=b, up=1, down=1, padding=0, gain=1, slope=1, clamp=40.0) * 0.25
        got = mc.modulated_conv2d(x, w, s, demodulate=False, padding=0, input_gain=torch.tensor(0.7, device=DEV), epilogue=(b, 40.0, 0.25))
        assert torch.equal(got, want) and float(want.abs().max()) == 10.0          # the clamp binds somewhere
        w3 = torch.randn([8, 32, 3, 3], device=DEV, generator=g)
        with pytest.raises(RuntimeError, match='ToRGB'):
            mc.modulated_conv2d(x, w3, s, padding=2, epilogue=(torch.zeros(8, device=DEV), 40.0, 1.0))


@pytest.mark.parametrize('dtype', [torch.float32, torch.float16])
def test_torgb_epilogue_keeps_nan(dtype):
    """the fused ToRGB output stage must let a NaN through its clamp, as the reference's filtered_lrelu does
    (filtered_lrelu.cu:412-419): checked against the CPU composite (torch_utils ref path), NaN positions and finite values."""
    from torch_utils.ops import filtered_lrelu as fl
    from torch_utils.ops import modulated_conv as mc
    g = torch.Generator(device=DEV).manual_seed(12)
    x = (torch.randn([1, 16, 32, 32], device=DEV, generator=g) * 30).to(dtype)
    x[0, 3, 5, 7] = float('nan'); x[0, 0, 20, 1] = float('nan')
    w = torch.randn([3, 16, 1, 1], device=DEV, generator=g)
    s = torch.randn([1, 16], device=DEV, generator=g) + 1
    b = torch.randn([3], device=DEV, generator=g)
    with torch.no_grad():
        got = mc.modulated_conv2d(x, w, s, demodulate=False, padding=0, epilogue=(b, 20.0, 0.5))
        plain = mc._composite(x.float().cpu(), w.cpu(), s.cpu(), False, 0, None)
        want = fl.filtered_lrelu(plain, b=b.cpu(), up=1, down=1, padding=0, gain=1, slope=1, clamp=20.0, impl='ref') * 0.5
    got = got.float().cpu()
    nan = torch.isnan(want)
    assert int(nan.sum()) == 6 and torch.equal(torch.isnan(got), nan)           # two pixels x three output channels
    assert maxabs(got[~nan].numpy(), want[~nan].numpy()) <= (2e-5 if dtype == torch.float32 else 2e-2) * 20


def test_t1024_batch8_headline_workload():
    """BASELINE configs[1] exactly as bench.py runs it (batch 8, force_fp32): sample 0 against the reference's golden samples /
    statistics, sample 7 against its own batch-1 forward, eagerly and through the captured hipGraph."""
    from sg3_runtime import GraphedSynthesis
    g = golden('net_t1024_stats')
    G = build_product_generator('T1024', device=DEV)
    ws_np = np.concatenate([synth_ws(1, G.num_ws, G.w_dim, seed=1)] + [synth_ws(1, G.num_ws, G.w_dim, seed=20 + i) for i in range(7)])
    ws = T(ws_np)
    with torch.no_grad():
        img = G.synthesis(ws, noise_mode='const', force_fp32=True)
        one = G.synthesis(ws[7:8], noise_mode='const', force_fp32=True)
    assert tuple(img.shape) == (8, 3, 1024, 1024)
    assert maxabs(img[:1, :, ::16, ::16].cpu().numpy(), g['T1024/img_sub']) <= 1e-4
    assert maxabs(img[:1, :, [0, 511, 1023], :].cpu().numpy(), g['T1024/img_rows']) <= 1e-4
    assert abs(img[:1].mean().item() - g['T1024/img_stats'][0]) <= 1e-5
    assert maxabs(img[7:8].cpu().numpy(), one.cpu().numpy()) <= 1e-5
    replay = GraphedSynthesis(G, 8)(ws)
    assert maxabs(replay.cpu().numpy(), img.cpu().numpy()) <= 1e-6


def test_graph_replay_follows_transform_rebinding_and_refuses_stale_weights():
    """callers rebind `synthesis.input.transform` to fresh tensors (pSp.forward, PTI); a replay must render under the
    CURRENT transform, not read the freed tensor captured earlier, and must refuse to replay after the weights changed."""
    import gc
    from sg3_runtime import GraphedSynthesis
    G = build_product_generator('Ttiny', device=DEV)
    ws = T(synth_ws(2, G.num_ws, G.w_dim, seed=1))
    graphed = GraphedSynthesis(G, 2)
    for tr in (make_user_transform(), np.stack([make_user_transform((0.1, -0.05), 15.0), make_user_transform((-0.2, 0.07), -30.0)]), np.eye(3, dtype=np.float32)):
        G.synthesis.input.transform = T(tr)                    # a fresh tensor, as the callers do
        junk = [torch.randn(3, 3, device=DEV) for _ in range(64)]     # recycle the allocator blocks the old transforms lived in
        del junk; gc.collect()
        got = graphed(ws).clone()
        G.synthesis.input.transform = T(tr)
        with torch.no_grad():
            want = G.synthesis(ws, noise_mode='const', force_fp32=True)
        assert maxabs(got.cpu().numpy(), want.cpu().numpy()) <= 1e-6
    got = graphed(ws, transform=T(make_user_transform((0.3, 0.1), 45.0))).clone()
    G.synthesis.input.transform = T(make_user_transform((0.3, 0.1), 45.0))
    with torch.no_grad():
        assert maxabs(got.cpu().numpy(), G.synthesis(ws, noise_mode='const', force_fp32=True).cpu().numpy()) <= 1e-6
    with torch.no_grad():
        G.synthesis.L3_36_12.bias.add_(0.5)                    # tuning step
    with pytest.raises(RuntimeError, match='changed since capture'):
        graphed(ws)


def test_graph_leaves_the_users_transform_on_the_module():
    """the graph's static [batch,3,3] transform buffer shadows `synthesis.input.transform` only during capture.  Afterwards
    the module carries the caller's tensor again: eager calls at another batch size work, `state_dict()` holds a loadable [3,3]
    transform, and a [1,3,3] transform broadcasts over the batch eagerly and through the graph alike."""
    from sg3_runtime import GraphedSynthesis
    G = build_product_generator('Ttiny', device=DEV)
    user = T(make_user_transform((0.1, -0.05), 15.0))
    G.synthesis.input.transform = user
    graphed = GraphedSynthesis(G, 2)
    assert G.synthesis.input.transform is user
    assert tuple(G.state_dict()['synthesis.input.transform'].shape) == (3, 3)
    build_product_generator('Ttiny').load_state_dict(G.state_dict())            # loads into a fresh generator
    ws3 = T(synth_ws(3, G.num_ws, G.w_dim, seed=2))
    with torch.no_grad():
        img3 = G.synthesis(ws3, noise_mode='const', force_fp32=True)            # another batch size, eager
        assert tuple(img3.shape)[0] == 3
        got = graphed(ws3[:2]).clone()
        assert G.synthesis.input.transform is user
        assert maxabs(got.cpu().numpy(), img3[:2].cpu().numpy()) <= 1e-6
        # [1,3,3]: broadcast over the batch (reference: rot @ trans @ self.transform)
        G.synthesis.input.transform = user.unsqueeze(0)
        one = G.synthesis(ws3, noise_mode='const', force_fp32=True)
        assert maxabs(one.cpu().numpy(), img3.cpu().numpy()) <= 1e-6
        assert maxabs(graphed(ws3[:2]).cpu().numpy(), img3[:2].cpu().numpy()) <= 1e-6


def test_packed_weight_cache_follows_the_weights_lifetime():
    """the inference cache of packed convolution weights drops an entry when its weight tensor is collected."""
    import gc
    from torch_utils.ops import modulated_conv as mc
    mc.clear_weight_cache()
    G = build_product_generator('Ttiny', device=DEV)
    ws = T(synth_ws(1, G.num_ws, G.w_dim, seed=1))
    with torch.no_grad():
        G.synthesis(ws, noise_mode='const', force_fp32=True)
    assert len(mc._packed_weights) > 0
    del G
    gc.collect()
    assert len(mc._packed_weights) == 0


def test_graph_replay_survives_cache_turnover():
    """Inference caches (packed convolution weights, affine pack, input constants, gains) are filled during the warm-up and read by
    the captured kernels; clearing or rebuilding them afterwards must not pull memory from under the graph."""
    import gc
    from sg3_runtime import GraphedSynthesis
    from models.stylegan3 import networks_stylegan3 as ns
    from torch_utils.ops import modulated_conv as mc
    G = build_product_generator('Ttiny', device=DEV)
    ws = T(synth_ws(2, G.num_ws, G.w_dim, seed=1))
    graphed = GraphedSynthesis(G, 2)
    with torch.no_grad():
        want = G.synthesis(ws, noise_mode='const', force_fp32=True).clone()
    assert maxabs(graphed(ws).cpu().numpy(), want.cpu().numpy()) <= 1e-6
    mc.clear_weight_cache()
    for m in G.synthesis.modules():
        ns._derived.pop(m, None)
    gc.collect()
    junk = [torch.randn(1 << 16, device=DEV) for _ in range(256)]          # recycle whatever the caches would have freed
    other = build_product_generator('Rtiny', device=DEV)                    # and fill the caches with another network's tensors
    with torch.no_grad():
        other.synthesis(T(synth_ws(3, other.num_ws, other.w_dim, seed=2)), noise_mode='const', force_fp32=True)
    del junk
    assert maxabs(graphed(ws).cpu().numpy(), want.cpu().numpy()) <= 1e-6


def test_hooked_modules_keep_their_own_forward():
    """GPU inference merges the affine layers and the input's algebra into two kernels that bypass those modules' forward; a
    module somebody registered a hook on keeps its own forward (the hook fires and sees the module's real output), and the image
    is the same to rounding."""
    G = build_product_generator('Ttiny', device=DEV)
    ws = T(synth_ws(2, G.num_ws, G.w_dim, seed=4))
    syn = G.synthesis
    with torch.no_grad():
        plain = syn(ws, noise_mode='const', force_fp32=True)
        seen = {}
        layer = syn.layers()[2]
        hooks = [layer.affine.register_forward_hook(lambda m, i, o: seen.__setitem__('styles', o.clone())),
                 syn.input.register_forward_hook(lambda m, i, o: seen.__setitem__('features', o.clone()))]
        hooked = syn(ws, noise_mode='const', force_fp32=True)
        for h in hooks:
            h.remove()
        again = syn(ws, noise_mode='const', force_fp32=True)
    assert set(seen) == {'styles', 'features'}
    assert float((seen['styles'] - layer.affine(ws[:, 3])).abs().max()) == 0.0
    assert tuple(seen['features'].shape) == (2, syn.input.channels, int(syn.input.size[1]), int(syn.input.size[0]))
    assert float((hooked - plain).abs().max()) <= 5e-6 * max(1.0, float(plain.abs().max()))
    assert torch.equal(again, plain)
