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
    man = {k: v for k, v in manifest()[cfg].items()}
    sd = synth_state_dict(man, seed=seed, input_bandwidth=sched['input']['bandwidth'])
    sd.update(O.filter_taps_for(sched))
    return sd, sched


def maxabs(a, b):
    return float(np.abs(np.asarray(a, dtype=np.float64) - np.asarray(b, dtype=np.float64)).max())
