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


def restyle_case(n, seed=4):
    """Frames in [-1, 1] and per-frame landmark transforms for the ReStyle loop fixtures."""
    x = np.random.RandomState(seed).uniform(-1, 1, size=(n, 3, 256, 256)).astype(np.float32)
    tr = np.stack([make_user_transform((0.04 * (i + 1), -0.03 * i), 6.0 * (i - 1)) for i in range(n)]).astype(np.float32)
    return x, tr


def tiny_encoder_weights(n_styles=16, w_dim=512, seed=21):
    r = np.random.RandomState(seed)
    return (r.randn(n_styles * w_dim, 6) * 0.3).astype(np.float32), (r.randn(n_styles * w_dim) * 0.05).astype(np.float32)


try:
    import torch

    class TinyEncoder(torch.nn.Module):
        """Stand-in for the IR-SE50 backbone in the ReStyle LOOP fixtures (the backbone has its own fixture): a fixed linear map
        of the six channel means to [N, 16, 512].  Deterministic numpy weights: the reference-side golden script, the product
        tests and the oracle tests all build the same one."""

        def __init__(self, n_styles=16, w_dim=512):
            super().__init__()
            w, b = tiny_encoder_weights(n_styles, w_dim)
            self.n_styles, self.w_dim = n_styles, w_dim
            self.lin = torch.nn.Linear(6, n_styles * w_dim)
            with torch.no_grad():
                self.lin.weight.copy_(torch.from_numpy(w)); self.lin.bias.copy_(torch.from_numpy(b))

        def forward(self, x):
            return self.lin(x.mean(dim=(2, 3))).view(-1, self.n_styles, self.w_dim)
except ImportError:          # numpy-only users of this module
    pass
