"""Shared case builders for the caller tests (FOV expansion, video post-processing, StyleSpace sweep)."""
import types

import numpy as np

from synth_weights import make_user_transform


def styleclip_case(example_s, seed=31, clip_dim=24):
    """Synthetic stand-ins for delta_i_c / s_std / the CLIP text direction (CLIP weights are unavailable offline)."""
    r = np.random.RandomState(seed)
    channels = sum(int(v.shape[1]) for v in example_s.values())
    delta_i_c = (r.randn(channels, clip_dim) / np.sqrt(clip_dim)).astype(np.float32)
    delta_i = r.randn(clip_dim).astype(np.float32)
    delta_i /= np.linalg.norm(delta_i)
    s_std = {k: r.uniform(0.2, 1.0, size=int(v.shape[1])).astype(np.float32) for k, v in example_s.items()}
    return delta_i_c, delta_i, s_std


def sweep_opts(num_alphas=3, num_betas=2):
    return types.SimpleNamespace(alpha_min=-4.0, alpha_max=4.0, num_alphas=num_alphas, beta_min=0.15, beta_max=0.35,
                                 num_betas=num_betas, neutral_text='a face', target_text='a smiling face')


def landmark():
    return make_user_transform((0.05, -0.03), 10.0).astype(np.float64)
