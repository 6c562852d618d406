"""Deterministic synthetic generator weights (numpy only; no reference, no torch RNG).

No pretrained StyleGAN3 weights exist offline, so golden vectors, the oracle,
the product and bench.py all use weights drawn here from a per-tensor seeded
`np.random.RandomState`.  The distributions follow the reference's initialisers
(models/stylegan3/networks_stylegan3.py:82-84, 180-195, 307-310) but with non-trivial
biases / magnitude_ema so that every term of the forward is exercised.
FIR filter buffers are NOT generated: each implementation designs its own and
tests compare the taps against the golden ones.
"""
import zlib

import numpy as np

# generator constructor kwargs for the named configurations
CONFIGS = {
    # BASELINE.json configs[1]: FFHQ-1024 StyleGAN3-T (reference models/stylegan3/model.py:29-40, "landscape" branch sizes)
    'T1024': dict(z_dim=512, c_dim=0, w_dim=512, img_resolution=1024, img_channels=3,
                  channel_base=32768, channel_max=512, magnitude_ema_beta=0.5 ** (32 / (20 * 1e3))),
    # BASELINE.json configs[0]: 256x256 config-T
    'T256': dict(z_dim=512, c_dim=0, w_dim=512, img_resolution=256, img_channels=3,
                 channel_base=32768, channel_max=512, magnitude_ema_beta=0.5 ** (32 / (20 * 1e3))),
    # config-R (reference models/stylegan3/model.py:42-54)
    'R1024': dict(z_dim=512, c_dim=0, w_dim=512, img_resolution=1024, img_channels=3,
                  channel_base=65536, channel_max=1024, conv_kernel=1, use_radial_filters=True,
                  magnitude_ema_beta=0.5 ** (32 / (20 * 1e3))),
    'R512': dict(z_dim=512, c_dim=0, w_dim=512, img_resolution=512, img_channels=3,
                 channel_base=65536, channel_max=1024, conv_kernel=1, use_radial_filters=True,
                 magnitude_ema_beta=0.5 ** (32 / (20 * 1e3))),
    # tiny variants for fast exhaustive tests (same 15-layer schedule, few channels)
    'Ttiny': dict(z_dim=32, c_dim=0, w_dim=32, img_resolution=64, img_channels=3,
                  channel_base=512, channel_max=12),
    'Rtiny': dict(z_dim=32, c_dim=0, w_dim=32, img_resolution=64, img_channels=3,
                  channel_base=1024, channel_max=16, conv_kernel=1, use_radial_filters=True),
    # the tiny decoders with the 512-wide latent the ReStyle encoder emits (ReStyle loop tests)
    'Tmini': dict(z_dim=512, c_dim=0, w_dim=512, img_resolution=64, img_channels=3,
                  channel_base=512, channel_max=12),
    'Rmini': dict(z_dim=512, c_dim=0, w_dim=512, img_resolution=64, img_channels=3,
                  channel_base=1024, channel_max=16, conv_kernel=1, use_radial_filters=True),
}


def _rs(seed, key):
    return np.random.RandomState((zlib.crc32(key.encode()) + 7919 * int(seed)) % (2 ** 32))


def synth_tensor(key, shape, seed=0, input_bandwidth=2.0):
    """One tensor of the synthetic state_dict; returns None for filter buffers."""
    r = _rs(seed, key)
    shape = tuple(int(s) for s in shape)
    leaf = key.split('.')[-1]
    if leaf in ('up_filter', 'down_filter'):
        return None
    if key == 'synthesis.input.transform':
        return np.eye(3, dtype=np.float32)
    if key == 'synthesis.input.freqs':
        f = r.randn(*shape)
        rad = np.sqrt((f ** 2).sum(axis=1, keepdims=True))
        f = f / (rad * np.exp(rad ** 2) ** 0.25) * input_bandwidth
        return f.astype(np.float32)
    if key == 'synthesis.input.phases':
        return (r.rand(*shape) - 0.5).astype(np.float32)
    if key == 'synthesis.input.affine.weight':
        return (r.randn(*shape) * 0.02).astype(np.float32)
    if key == 'synthesis.input.affine.bias':
        return (np.array([1, 0, 0, 0], np.float32) + r.randn(*shape).astype(np.float32) * 0.05)
    if key == 'mapping.w_avg':
        return (r.randn(*shape) * 0.1).astype(np.float32)
    if key.startswith('mapping.') and leaf == 'weight':
        return (r.randn(*shape) * 100.0).astype(np.float32)          # randn / lr_multiplier(0.01)
    if key.startswith('mapping.') and leaf == 'bias':
        return (r.randn(*shape) * 10.0).astype(np.float32)
    if leaf == 'magnitude_ema':
        return np.asarray(r.uniform(0.6, 1.8), dtype=np.float32).reshape(shape)
    if key.endswith('affine.bias'):
        return (1.0 + 0.1 * r.randn(*shape)).astype(np.float32)
    if key.endswith('affine.weight'):
        return r.randn(*shape).astype(np.float32)
    if leaf == 'bias':
        return (0.1 * r.randn(*shape)).astype(np.float32)
    if leaf == 'weight':
        return r.randn(*shape).astype(np.float32)
    raise KeyError(f'no synthetic rule for {key}')


def synth_state_dict(manifest, seed=0, input_bandwidth=2.0):
    """manifest: {key: shape}.  Returns {key: np.ndarray} without filter buffers."""
    out = {}
    for key in sorted(manifest):
        t = synth_tensor(key, manifest[key], seed=seed, input_bandwidth=input_bandwidth)
        if t is not None:
            out[key] = t
    return out


def synth_ws(n, num_ws, w_dim, seed=1):
    """Latents ws [n, num_ws, w_dim] ~ N(0,1)."""
    return np.random.RandomState(1000 + seed).randn(n, num_ws, w_dim).astype(np.float32)


def make_user_transform(translate=(0.1, -0.05), angle_deg=15.0):
    """Inverse of rotate+translate, as reference utils/common.py:9-27 builds and inverts it."""
    m = np.eye(3)
    s, c = np.sin(angle_deg / 360.0 * np.pi * 2), np.cos(angle_deg / 360.0 * np.pi * 2)
    m[0][0] = c; m[0][1] = s; m[0][2] = translate[0]
    m[1][0] = -s; m[1][1] = c; m[1][2] = translate[1]
    return np.linalg.inv(m).astype(np.float32)


# ---------------------------------------------------------------------------
# ReStyle encoder (IR-SE50 + 16 GradualStyleBlock heads): He-scaled weights and non-trivial BatchNorm statistics so
# that activations stay O(1) through 50 layers and every fused term (BN in front / behind, PReLU, SE) is exercised.

def synth_encoder_tensor(key, shape, seed=0):
    r = _rs(seed + 17, key)
    shape = tuple(int(s) for s in shape)
    leaf = key.split('.')[-1]
    if leaf == 'num_batches_tracked':
        return np.asarray(100, dtype=np.int64)
    if leaf == 'running_mean':
        return (0.1 * r.randn(*shape)).astype(np.float32)
    if leaf == 'running_var':
        return r.uniform(0.5, 1.5, size=shape).astype(np.float32)
    if len(shape) == 4:                                   # conv weight [O,I,k,k]
        fan_in = shape[1] * shape[2] * shape[3]
        return (r.randn(*shape) * np.sqrt(1.0 / fan_in)).astype(np.float32)
    if len(shape) == 2:                                   # EqualLinear weight (scaled by 1/sqrt(in) in forward)
        return r.randn(*shape).astype(np.float32)
    if 'linear' in key and leaf == 'bias':
        return (0.1 * r.randn(*shape)).astype(np.float32)
    if 'convs' in key and leaf == 'bias':
        return (0.05 * r.randn(*shape)).astype(np.float32)
    # 1-D: BatchNorm weight / bias or PReLU slope; told apart by the module position in the reference layout
    parts = key.split('.')
    if key == 'relu.weight':                              # PReLU of the ResNet34 stem
        return r.uniform(0.1, 0.4, size=shape).astype(np.float32)
    if leaf == 'weight' and (parts[-2] == '2' and parts[0] in ('input_layer',) or (len(parts) >= 3 and parts[-3] == 'res_layer' and parts[-2] == '2')):
        return r.uniform(0.1, 0.4, size=shape).astype(np.float32)        # PReLU
    if leaf == 'weight':
        return r.uniform(0.7, 1.3, size=shape).astype(np.float32)        # BatchNorm gamma
    if leaf == 'bias':
        return (0.1 * r.randn(*shape)).astype(np.float32)                # BatchNorm beta
    raise KeyError(key)


def synth_encoder_state_dict(manifest, seed=0):
    return {k: synth_encoder_tensor(k, manifest[k], seed=seed) for k in sorted(manifest)}
