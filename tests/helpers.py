"""Shared test helpers: golden loading, synthetic generators, product / oracle construction."""
import json
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = os.path.join(HERE, 'golden')
_cache = {}


def golden(name):
    if name not in _cache:
        _cache[name] = np.load(os.path.join(GOLD, name + '.npz'))
    return _cache[name]


def manifest():
    if 'manifest' not in _cache:
        with open(os.path.join(GOLD, 'manifest.json')) as f:
            _cache['manifest'] = json.load(f)
    return _cache['manifest']


def oracle_design(numtaps, cutoff, width, fs, radial=False):
    from oracle import oracle as O
    return O.design_lowpass_filter(numtaps, cutoff, width, fs, radial)


def product_design(numtaps, cutoff, width, fs, radial=False):
    from models.stylegan3.networks_stylegan3 import SynthesisLayer
    f = SynthesisLayer.design_lowpass_filter(numtaps=numtaps, cutoff=cutoff, width=width, fs=fs, radial=radial)
    return None if f is None else f.numpy()


def build_product_generator(cfg, seed=0, device='cpu'):
    """Product Generator with the deterministic synthetic weights (filters designed by the product itself)."""
    import torch
    from models.stylegan3.networks_stylegan3 import Generator
    from synth_weights import CONFIGS, synth_state_dict
    G = Generator(**CONFIGS[cfg]).eval().requires_grad_(False)
    man = {k: list(v.shape) for k, v in G.state_dict().items()}
    sd = synth_state_dict(man, seed=seed, input_bandwidth=float(G.synthesis.input.bandwidth))
    missing, unexpected = G.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
    assert not unexpected and all(k.endswith('_filter') for k in missing)
    return G.to(device)


def build_oracle_generator(cfg, seed=0):
    """(state dict incl. oracle-designed filters, schedule) for the numpy/C oracle."""
    from oracle import oracle as O
    from synth_weights import CONFIGS, synth_state_dict
    sched = O.layer_schedule(**CONFIGS[cfg])
    if cfg in manifest():
        man = {k: v for k, v in manifest()[cfg].items()}
    else:       # configurations without a reference-dumped manifest (the mini decoders): shapes of the product's state_dict
        from models.stylegan3.networks_stylegan3 import Generator
        man = {k: list(v.shape) for k, v in Generator(**CONFIGS[cfg]).state_dict().items() if not k.endswith('_filter')}
    sd = synth_state_dict(man, seed=seed, input_bandwidth=sched['input']['bandwidth'])
    sd.update(O.filter_taps_for(sched))
    return sd, sched


def maxabs(a, b):
    return float(np.abs(np.asarray(a, dtype=np.float64) - np.asarray(b, dtype=np.float64)).max())


def build_restyle_pair(cfg, device='cpu', encoder_type='BackboneEncoder', n_iters=5):
    """(product pSp net on `device`, opts, oracle encoder state dict, oracle generator state dict, schedule): the same
    seeded synthetic IR-SE50 encoder and decoder on both sides."""
    import types
    import torch
    from models.setgan.encoder.psp3 import pSp
    from synth_weights import synth_encoder_state_dict
    G = build_product_generator(cfg)
    opts = types.SimpleNamespace(encoder_type=encoder_type, input_nc=6, checkpoint_path=None, n_iters_per_batch=n_iters, resize_outputs=False)
    net = pSp(opts, decoder=G)
    man = {k: list(v.shape) for k, v in net.encoder.state_dict().items()}
    enc_sd = {k: np.asarray(v) for k, v in synth_encoder_state_dict(man, seed=0).items()}
    net.encoder.load_state_dict({k: torch.from_numpy(v) for k, v in enc_sd.items()})
    net = net.eval().requires_grad_(False).to(device)
    gen_sd, sched = build_oracle_generator(cfg)
    return net, opts, enc_sd, gen_sd, sched
