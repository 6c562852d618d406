"""Case tables shared by tests/golden/make_golden.py (which runs the reference) and the parity tests.

Inputs are never stored: they are regenerated from `np.random.RandomState(seed)`; the golden files hold
only the reference outputs.  Filters named 'kaiser:<taps>:<cutoff>:<width>:<fs>[:radial]' are designed by the
implementation under test (oracle or product) so the design code is covered too; taps themselves are pinned
separately in filters.npz.
"""
import numpy as np


def rand(seed, *shape, dtype=np.float32):
    return np.random.RandomState(seed).randn(*shape).astype(dtype)


# --- filtered_lrelu: every distinct (up, fu, down, fd, padding, gain, slope, clamp) tuple used by the
#     T-1024 / R-1024 configs at C=3, N=2, plus edge cases ---------------------------------------
SG2 = float(np.sqrt(2))
FLRELU_CASES = {
    # name: dict(x shape, fu, fd, up, down, padding, gain, slope, clamp, flip, bias)
    't_up2_dn2':      dict(shape=(2, 3, 38, 38), fu='kaiser:12:2.0:4.6:32', fd='kaiser:12:2.8:5.2:32', up=2, down=2, padding=[9, 8, 9, 8], gain=SG2, slope=0.2, clamp=256, flip=False, bias=True),
    't_up4_dn2':      dict(shape=(2, 3, 38, 38), fu='kaiser:24:2.8:5.2:64', fd='kaiser:12:4.0:8.0:64', up=4, down=2, padding=[-6, -9, -6, -9], gain=SG2, slope=0.2, clamp=256, flip=False, bias=True),
    't_up2_dn2_54':   dict(shape=(2, 3, 54, 54), fu='kaiser:12:4.0:8.0:64', fd='kaiser:12:5.6:9.0:64', up=2, down=2, padding=[9, 8, 9, 8], gain=SG2, slope=0.2, clamp=256, flip=False, bias=True),
    't_crit_l13':     dict(shape=(1, 3, 70, 70), fu='kaiser:12:8.0:12.0:64', fd='kaiser:12:8.0:12.0:64', up=2, down=2, padding=[-11, -12, -11, -12], gain=SG2, slope=0.2, clamp=256, flip=False, bias=True),
    't_torgb':        dict(shape=(2, 3, 24, 24), fu=None, fd=None, up=1, down=1, padding=[0, 0, 0, 0], gain=1.0, slope=1.0, clamp=256, flip=False, bias=True),
    'r_up2_dnrad2':   dict(shape=(2, 3, 36, 36), fu='kaiser:12:2.0:4.6:32', fd='kaiser:12:2.8:5.2:32:radial', up=2, down=2, padding=[11, 10, 11, 10], gain=SG2, slope=0.2, clamp=256, flip=False, bias=True),
    'r_up4_dnrad2':   dict(shape=(2, 3, 36, 36), fu='kaiser:24:2.8:5.2:64', fd='kaiser:12:4.0:8.0:64:radial', up=4, down=2, padding=[-2, -5, -2, -5], gain=SG2, slope=0.2, clamp=256, flip=False, bias=True),
    'r_crit':         dict(shape=(1, 3, 64, 64), fu='kaiser:12:8.0:12.0:64', fd='kaiser:12:8.0:12.0:64', up=2, down=2, padding=[-9, -10, -9, -10], gain=SG2, slope=0.2, clamp=256, flip=False, bias=True),
    # edge cases
    'no_clamp':       dict(shape=(1, 2, 20, 22), fu='kaiser:12:2.0:4.6:32', fd='kaiser:12:2.8:5.2:32', up=2, down=2, padding=[9, 8, 7, 10], gain=SG2, slope=0.2, clamp=None, flip=False, bias=True),
    'no_bias':        dict(shape=(1, 2, 20, 22), fu='kaiser:12:2.0:4.6:32', fd='kaiser:12:2.8:5.2:32', up=2, down=2, padding=[9, 8, 9, 8], gain=SG2, slope=0.2, clamp=256, flip=False, bias=False),
    'flip':           dict(shape=(1, 2, 20, 22), fu='asym:12', fd='asym:12', up=2, down=2, padding=[9, 8, 9, 8], gain=SG2, slope=0.2, clamp=256, flip=True, bias=True),
    'asym_noflip':    dict(shape=(1, 2, 20, 22), fu='asym:12', fd='asym:12', up=2, down=2, padding=[9, 8, 9, 8], gain=SG2, slope=0.2, clamp=256, flip=False, bias=True),
    'tight_clamp':    dict(shape=(1, 2, 20, 22), fu='kaiser:12:2.0:4.6:32', fd='kaiser:12:2.8:5.2:32', up=2, down=2, padding=[9, 8, 9, 8], gain=SG2, slope=0.2, clamp=0.5, flip=False, bias=True),
    'nonsquare_neg':  dict(shape=(1, 2, 30, 41), fu='kaiser:24:2.8:5.2:64', fd='kaiser:12:4.0:8.0:64', up=4, down=2, padding=[-3, -8, 2, -5], gain=1.3, slope=0.1, clamp=2.0, flip=False, bias=True),
    'up1_dn2':        dict(shape=(1, 2, 40, 40), fu=None, fd='kaiser:12:4.0:8.0:64', up=1, down=2, padding=[5, 5, 5, 5], gain=SG2, slope=0.2, clamp=256, flip=False, bias=True),
    'up2_dn1':        dict(shape=(1, 2, 20, 20), fu='kaiser:12:2.0:4.6:32', fd=None, up=2, down=1, padding=[5, 6, 5, 6], gain=SG2, slope=0.2, clamp=256, flip=False, bias=True),
    'bwd_like_dn4':   dict(shape=(1, 2, 26, 26), fu='kaiser:12:2.8:5.2:32', fd='kaiser:24:2.8:5.2:64', up=2, down=4, padding=[13, 28, 13, 28], gain=0.35, slope=0.2, clamp=None, flip=True, bias=False),
    'full_up_2d':     dict(shape=(1, 2, 18, 18), fu='kaiser:12:2.0:4.6:32:radial', fd='kaiser:12:2.8:5.2:32', up=2, down=2, padding=[9, 8, 9, 8], gain=SG2, slope=0.2, clamp=256, flip=False, bias=True),
}

# gradient cases: d(sum(y*gy))/dx and /db through the reference `ref` composite
FLRELU_GRAD_CASES = ['t_up2_dn2', 't_up4_dn2', 'tight_clamp', 'no_clamp', 'r_up2_dnrad2', 't_torgb']

UPFIRDN_CASES = {
    'sep_up2':    dict(shape=(2, 3, 17, 19), f='kaiser:12:2.0:4.6:32', up=2, down=1, padding=[6, 5, 6, 5], flip=False, gain=4.0),
    'sep_dn2':    dict(shape=(2, 3, 33, 31), f='kaiser:12:2.8:5.2:32', up=1, down=2, padding=[5, 5, 5, 5], flip=False, gain=1.0),
    'full_rad':   dict(shape=(1, 2, 30, 30), f='kaiser:12:2.8:5.2:32:radial', up=1, down=2, padding=[0, 0, 0, 0], flip=False, gain=1.0),
    'asym_flip':  dict(shape=(1, 2, 15, 16), f='asym2d:5:3', up=[2, 3], down=[3, 2], padding=[2, 1, -1, 3], flip=True, gain=1.7),
    'asym_conv':  dict(shape=(1, 2, 15, 16), f='asym2d:5:3', up=[2, 3], down=[3, 2], padding=[2, 1, -1, 3], flip=False, gain=1.7),
    'identity':   dict(shape=(1, 2, 9, 9), f=None, up=1, down=1, padding=0, flip=False, gain=1.0),
    'crop_only':  dict(shape=(1, 2, 12, 12), f=None, up=2, down=1, padding=[-2, -1, -3, 0], flip=False, gain=1.0),
}

BIAS_ACT_NAMES = ['linear', 'relu', 'lrelu', 'tanh', 'sigmoid', 'elu', 'selu', 'softplus', 'swish']
BIAS_ACT_CASES = {}
for _a in BIAS_ACT_NAMES:
    BIAS_ACT_CASES[f'{_a}_default'] = dict(shape=(4, 7, 5), dim=1, act=_a, alpha=None, gain=None, clamp=None, bias=True)
BIAS_ACT_CASES['lrelu_clamp'] = dict(shape=(3, 6), dim=1, act='lrelu', alpha=0.3, gain=2.0, clamp=0.7, bias=True)
BIAS_ACT_CASES['lrelu_nobias'] = dict(shape=(3, 6), dim=1, act='lrelu', alpha=None, gain=None, clamp=None, bias=False)
BIAS_ACT_CASES['linear_dim0'] = dict(shape=(5, 4, 3), dim=0, act='linear', alpha=None, gain=1.5, clamp=None, bias=True)
BIAS_ACT_CASES['swish_dim2'] = dict(shape=(2, 3, 8), dim=2, act='swish', alpha=None, gain=None, clamp=1.0, bias=True)

MODCONV_CASES = {
    'k3_demod':      dict(n=2, ci=5, co=7, h=9, w=11, k=3, demodulate=True, input_gain=1.3),
    'k3_nodemod':    dict(n=2, ci=5, co=7, h=9, w=11, k=3, demodulate=False, input_gain=None),
    'k1_demod':      dict(n=3, ci=6, co=4, h=8, w=8, k=1, demodulate=True, input_gain=0.7),
    'k1_torgb':      dict(n=2, ci=6, co=3, h=10, w=10, k=1, demodulate=False, input_gain=1.1),
    'k3_odd_ch':     dict(n=1, ci=51, co=33, h=13, w=12, k=3, demodulate=True, input_gain=0.9),
    'k3_wide':       dict(n=2, ci=16, co=40, h=37, w=45, k=3, demodulate=True, input_gain=1.0),
}


def make_filter(spec, design):
    """spec string -> float32 numpy filter (or None).  `design(numtaps, cutoff, width, fs, radial)`."""
    if spec is None:
        return None
    parts = spec.split(':')
    if parts[0] == 'kaiser':
        radial = len(parts) > 5 and parts[5] == 'radial'
        return design(int(parts[1]), float(parts[2]), float(parts[3]), float(parts[4]), radial)
    if parts[0] == 'asym':       # deliberately asymmetric 1-D taps: exposes flip mistakes
        n = int(parts[1])
        f = np.random.RandomState(77).rand(n).astype(np.float32) + 0.1
        return (f / f.sum()).astype(np.float32)
    if parts[0] == 'asym2d':
        fh, fw = int(parts[1]), int(parts[2])
        f = np.random.RandomState(78).rand(fh, fw).astype(np.float32) + 0.1
        return (f / f.sum()).astype(np.float32)
    raise ValueError(spec)
