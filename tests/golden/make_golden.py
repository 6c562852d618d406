"""Generate the golden fixtures by running the REFERENCE's own pure-PyTorch `ref` path on CPU.

Run in the build container only (the reference is mounted read-only at /root/reference and never
travels to the GPU box):

    python tests/golden/make_golden.py [--only ops|filters|net|big]

Outputs (committed): tests/golden/{filters,ops,grads,net_tiny,net_t256,net_t1024_stats}.npz, manifest.json
Nothing from the reference is copied: the fixtures are inputs-by-seed and reference OUTPUTS only.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
TESTS = os.path.dirname(HERE)
sys.path.insert(0, TESTS)
sys.path.insert(0, '/root/reference')

import torch  # noqa: E402

from golden_cases import (BIAS_ACT_CASES, FLRELU_CASES, FLRELU_GRAD_CASES, MODCONV_CASES, UPFIRDN_CASES,  # noqa: E402
                          make_filter, rand)
from synth_weights import CONFIGS, make_user_transform, synth_state_dict, synth_ws  # noqa: E402

from models.stylegan3.networks_stylegan3 import Generator, SynthesisLayer, modulated_conv2d  # noqa: E402
from torch_utils.ops import bias_act, filtered_lrelu, upfirdn2d  # noqa: E402

torch.set_grad_enabled(False)


def ref_design(numtaps, cutoff, width, fs, radial=False):
    f = SynthesisLayer.design_lowpass_filter(numtaps=numtaps, cutoff=cutoff, width=width, fs=fs, radial=radial)
    return None if f is None else f.numpy()


def T(a):
    return None if a is None else torch.from_numpy(np.ascontiguousarray(a))


def gen_filters():
    out = {}
    for cfg in ['T1024', 'T256', 'R1024', 'R512', 'Ttiny', 'Rtiny']:
        kw = dict(CONFIGS[cfg]); kw.pop('magnitude_ema_beta', None)
        G = Generator(**CONFIGS[cfg])
        for name in G.synthesis.layer_names:
            layer = getattr(G.synthesis, name)
            if layer.up_filter is not None:
                out[f'{cfg}/{name}/up_filter'] = layer.up_filter.numpy()
            if layer.down_filter is not None:
                out[f'{cfg}/{name}/down_filter'] = layer.down_filter.numpy()
            out[f'{cfg}/{name}/padding'] = np.asarray(layer.padding, np.int32)
            out[f'{cfg}/{name}/geom'] = np.asarray([layer.in_channels, layer.out_channels, int(layer.in_size[0]), int(layer.out_size[0]),
                                                     layer.up_factor, layer.down_factor, layer.up_taps, layer.down_taps,
                                                     layer.conv_kernel, int(layer.use_fp16), int(layer.down_radial)], np.int32)
        del G
    np.savez_compressed(os.path.join(HERE, 'filters.npz'), **out)
    print('filters.npz', len(out))


def gen_manifest():
    man = {}
    for cfg in CONFIGS:
        G = Generator(**CONFIGS[cfg])
        man[cfg] = {k: list(v.shape) for k, v in G.state_dict().items()}
        man[cfg + '/num_params'] = int(sum(p.numel() for p in G.parameters()))
        del G
    with open(os.path.join(HERE, 'manifest.json'), 'w') as f:
        json.dump(man, f, indent=0, sort_keys=True)
    print('manifest.json')


def gen_ops():
    out = {}
    for name, c in FLRELU_CASES.items():
        x = rand(11, *c['shape'])
        b = rand(12, c['shape'][1]) if c['bias'] else None
        fu, fd = make_filter(c['fu'], ref_design), make_filter(c['fd'], ref_design)
        y = filtered_lrelu.filtered_lrelu(T(x), fu=T(fu), fd=T(fd), b=T(b), up=c['up'], down=c['down'], padding=c['padding'],
                                          gain=c['gain'], slope=c['slope'], clamp=c['clamp'], flip_filter=c['flip'], impl='ref')
        out[f'flrelu/{name}'] = y.numpy()
        # also the fp64 result, as a high-precision anchor
        y64 = filtered_lrelu.filtered_lrelu(T(x).double(), fu=T(fu), fd=T(fd), b=None if b is None else T(b).double(), up=c['up'], down=c['down'],
                                            padding=c['padding'], gain=c['gain'], slope=c['slope'], clamp=c['clamp'], flip_filter=c['flip'], impl='ref')
        out[f'flrelu64/{name}'] = y64.numpy()
    for name, c in UPFIRDN_CASES.items():
        x = rand(21, *c['shape'])
        f = make_filter(c['f'], ref_design)
        y = upfirdn2d.upfirdn2d(T(x), T(f), up=c['up'], down=c['down'], padding=c['padding'], flip_filter=c['flip'], gain=c['gain'], impl='ref')
        out[f'upfirdn/{name}'] = y.numpy()
    for name, c in BIAS_ACT_CASES.items():
        x = rand(31, *c['shape']) * 2
        b = rand(32, c['shape'][c['dim']]) if c['bias'] else None
        y = bias_act.bias_act(T(x), T(b), dim=c['dim'], act=c['act'], alpha=c['alpha'], gain=c['gain'], clamp=c['clamp'], impl='ref')
        out[f'bias_act/{name}'] = y.numpy()
    for name, c in MODCONV_CASES.items():
        x = rand(41, c['n'], c['ci'], c['h'], c['w'])
        w = rand(42, c['co'], c['ci'], c['k'], c['k'])
        s = rand(43, c['n'], c['ci']) + 1.0
        ig = None if c['input_gain'] is None else torch.tensor(c['input_gain'], dtype=torch.float32)
        y = modulated_conv2d(T(x), T(w), T(s), demodulate=c['demodulate'], padding=c['k'] - 1, input_gain=ig)
        out[f'modconv/{name}'] = y.numpy()
    np.savez_compressed(os.path.join(HERE, 'ops.npz'), **out)
    print('ops.npz', len(out))


def gen_grads():
    out = {}
    torch.set_grad_enabled(True)
    for name in FLRELU_GRAD_CASES:
        c = FLRELU_CASES[name]
        x = T(rand(11, *c['shape'])).requires_grad_(True)
        b = T(rand(12, c['shape'][1])).requires_grad_(True) if c['bias'] else None
        fu, fd = make_filter(c['fu'], ref_design), make_filter(c['fd'], ref_design)
        y = filtered_lrelu.filtered_lrelu(x, fu=T(fu), fd=T(fd), b=b, up=c['up'], down=c['down'], padding=c['padding'],
                                          gain=c['gain'], slope=c['slope'], clamp=c['clamp'], flip_filter=c['flip'], impl='ref')
        gy = T(rand(13, *y.shape))
        (y * gy).sum().backward()
        out[f'{name}/dx'] = x.grad.numpy()
        if b is not None:
            out[f'{name}/db'] = b.grad.numpy()
    torch.set_grad_enabled(False)
    np.savez_compressed(os.path.join(HERE, 'grads.npz'), **out)
    print('grads.npz', len(out))


def load_synth(G, cfg, seed=0):
    man = {k: list(v.shape) for k, v in G.state_dict().items()}
    sd = synth_state_dict(man, seed=seed, input_bandwidth=float(G.synthesis.input.bandwidth))
    missing, unexpected = G.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
    assert not unexpected and all(k.endswith('_filter') for k in missing), (missing, unexpected)
    return G.eval().requires_grad_(False)


def layer_feats(G, ws, **kw):
    feats = []
    hooks = [getattr(G.synthesis, n).register_forward_hook(lambda m, i, o: feats.append(o)) for n in G.synthesis.layer_names]
    img = G.synthesis(ws, noise_mode='const', force_fp32=True, **kw)
    for h in hooks:
        h.remove()
    return img, feats


def gen_net_tiny():
    out = {}
    for cfg in ['Ttiny', 'Rtiny']:
        G = load_synth(Generator(**CONFIGS[cfg]), cfg)
        ws = T(synth_ws(2, G.num_ws, G.w_dim, seed=1))
        img, feats = layer_feats(G, ws)
        out[f'{cfg}/img'] = img.numpy()
        for n, f in zip(G.synthesis.layer_names, feats):
            out[f'{cfg}/feat/{n}'] = f[:, :2, :24, :24].numpy()          # corner incl. the padded border region
            out[f'{cfg}/featstat/{n}'] = np.asarray([f.mean().item(), f.std().item(), f.abs().max().item()], np.float64)
        out[f'{cfg}/input'] = G.synthesis.input(ws[:, 0]).numpy()
        # user transform: [3,3] and batched [B,3,3] (callers assign a batched tensor: reference psp3.py:64-65)
        tr = make_user_transform()
        G.synthesis.input.transform = T(tr)
        out[f'{cfg}/img_tr'] = G.synthesis(ws, noise_mode='const', force_fp32=True).numpy()
        trb = np.stack([make_user_transform((0.1, -0.05), 15.0), make_user_transform((-0.2, 0.07), -30.0)])
        G.synthesis.input.transform = T(trb)
        out[f'{cfg}/img_trb'] = G.synthesis(ws, noise_mode='const', force_fp32=True).numpy()
        out[f'{cfg}/input_trb'] = G.synthesis.input(ws[:, 0]).numpy()
        G.synthesis.input.transform = torch.eye(3)
        # StyleSpace path
        all_s = G.synthesis.W2S(ws)
        for k, v in all_s.items():
            out[f'{cfg}/w2s/{k}'] = v.numpy()
        out[f'{cfg}/img_alls'] = G.synthesis(None, all_s=all_s, noise_mode='const', force_fp32=True).numpy()
        # mapping + full generator with truncation
        z = T(np.random.RandomState(5).randn(3, G.z_dim).astype(np.float32))
        out[f'{cfg}/mapping_psi1'] = G.mapping(z, None).numpy()
        out[f'{cfg}/mapping_psi07'] = G.mapping(z, None, truncation_psi=0.7).numpy()
        out[f'{cfg}/mapping_psi05_cut8'] = G.mapping(z, None, truncation_psi=0.5, truncation_cutoff=8).numpy()
        out[f'{cfg}/gen_psi07'] = G(z[:1], None, truncation_psi=0.7, noise_mode='const', force_fp32=True).numpy()
    np.savez_compressed(os.path.join(HERE, 'net_tiny.npz'), **out)
    print('net_tiny.npz', len(out))


def gen_big():
    out = {}
    t0 = time.time()
    G = load_synth(Generator(**CONFIGS['T256']), 'T256')
    ws = T(synth_ws(1, G.num_ws, G.w_dim, seed=1))
    img, feats = layer_feats(G, ws)
    out['T256/img'] = img.numpy()
    out['T256/stats'] = np.asarray([[f.mean().item(), f.std().item(), f.abs().max().item()] for f in feats], np.float64)
    print('T256 done', time.time() - t0)
    np.savez_compressed(os.path.join(HERE, 'net_t256.npz'), **out)
    del G
    out = {}
    for cfg in ['T1024']:
        G = load_synth(Generator(**CONFIGS[cfg]), cfg)
        ws = T(synth_ws(1, G.num_ws, G.w_dim, seed=1))
        img, feats = layer_feats(G, ws)
        out[f'{cfg}/stats'] = np.asarray([[f.mean().item(), f.std().item(), f.abs().max().item()] for f in feats], np.float64)
        out[f'{cfg}/img_stats'] = np.asarray([img.mean().item(), img.std().item(), img.abs().max().item()], np.float64)
        # 64x64 strided subsample + two full rows of the final image, and a 32x32 corner of every layer's channel 0
        out[f'{cfg}/img_sub'] = img[:, :, ::16, ::16].numpy()
        out[f'{cfg}/img_rows'] = img[:, :, [0, 511, 1023], :].numpy()
        for n, f in zip(G.synthesis.layer_names, feats):
            out[f'{cfg}/corner/{n}'] = f[:, :2, :32, :32].numpy()
            out[f'{cfg}/center/{n}'] = f[:, -1:, f.shape[2] // 2, :].numpy()
        print(cfg, 'done', time.time() - t0)
        del G
    np.savez_compressed(os.path.join(HERE, 'net_t1024_stats.npz'), **out)


def gen_big_r():
    """Full-size config-R forwards (the reference's default generator branch, models/stylegan3/model.py:42-54): R-512
    (BASELINE configs[4]) and R-1024 (the inversion / PTI decoder), batch 1, samples and statistics only."""
    out = {}
    t0 = time.time()
    for cfg in ['R512', 'R1024']:
        G = load_synth(Generator(**CONFIGS[cfg]), cfg)
        ws = T(synth_ws(1, G.num_ws, G.w_dim, seed=1))
        img, feats = layer_feats(G, ws)
        res = G.img_resolution
        out[f'{cfg}/stats'] = np.asarray([[f.mean().item(), f.std().item(), f.abs().max().item()] for f in feats], np.float64)
        out[f'{cfg}/img_stats'] = np.asarray([img.mean().item(), img.std().item(), img.abs().max().item()], np.float64)
        out[f'{cfg}/img_sub'] = img[:, :, ::16, ::16].numpy()
        out[f'{cfg}/img_rows'] = img[:, :, [0, res // 2 - 1, res - 1], :].numpy()
        for n, f in zip(G.synthesis.layer_names, feats):
            out[f'{cfg}/corner/{n}'] = f[:, :2, :32, :32].numpy()
            out[f'{cfg}/center/{n}'] = f[:, -1:, f.shape[2] // 2, :].numpy()
        print(cfg, 'done', time.time() - t0)
        del G
    np.savez_compressed(os.path.join(HERE, 'net_r_stats.npz'), **out)


if __name__ == '__main__':
    ap = argparse.ArgumentParser()
    ap.add_argument('--only', default=None)
    a = ap.parse_args()
    steps = dict(filters=gen_filters, manifest=gen_manifest, ops=gen_ops, grads=gen_grads, net=gen_net_tiny, big=gen_big, big_r=gen_big_r)
    for k, fn in steps.items():
        if a.only is None or a.only == k:
            fn()
