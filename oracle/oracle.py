"""CPU ORACLE for the StyleGAN3 synthesis hot path -- TEST INFRASTRUCTURE ONLY.

numpy + plain C (oracle/sg3_oracle.c) restatement of the reference algorithm.
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import
this module; the product package (stylegan3-editing_amd/) never does.

Parity status: PINNED against golden vectors produced by importing the
reference's own pure-PyTorch `ref` path in the build container
(tests/golden/make_golden.py, fixtures in tests/golden/*.npz).

Reference lines restated (relative to the reference tree):
  design_lowpass_filter   models/stylegan3/networks_stylegan3.py:370-391
                          (+ scipy.signal.firwin / kaiser_beta / kaiser_atten,
                          scipy 1.4.1 pinned by environment/sg3_env.yaml:27)
  layer schedule          models/stylegan3/networks_stylegan3.py:430-469, 294-333
  fully_connected         models/stylegan3/networks_stylegan3.py:86-100
  mapping                 models/stylegan3/networks_stylegan3.py:134-160
  synthesis_input         models/stylegan3/networks_stylegan3.py:197-249
  synthesis_layer         models/stylegan3/networks_stylegan3.py:335-368
  synthesis / W2S         models/stylegan3/networks_stylegan3.py:471-525
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build(force=False):
    """Compile oracle/sg3_oracle.c -> oracle/libsg3_oracle.so (gcc, OpenMP)."""
    so = os.path.join(_HERE, 'libsg3_oracle.so')
    src = os.path.join(_HERE, 'sg3_oracle.c')
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(['make', '-C', _HERE, '-s', 'libsg3_oracle.so'])
    return so


def lib():
    global _LIB
    if _LIB is None:
        _LIB = ctypes.CDLL(build())
        _LIB.sg3o_num_threads.restype = ctypes.c_int
        _LIB.sg3o_filtered_lrelu_f32.restype = ctypes.c_int
        _LIB.sg3o_filtered_lrelu_f64.restype = ctypes.c_int
    return _LIB


def num_threads():
    return int(lib().sg3o_num_threads())


def set_num_threads(n):
    lib().sg3o_set_num_threads(ctypes.c_int(int(n)))


def _ptr(a):
    return a.ctypes.data_as(ctypes.c_void_p) if a is not None else None


def _suffix(dtype):
    if dtype == np.float32:
        return 'f32'
    if dtype == np.float64:
        return 'f64'
    raise TypeError(f'oracle supports float32/float64, got {dtype}')


def _parse_padding(padding):
    if isinstance(padding, (int, np.integer)):
        padding = [int(padding)] * 2
    padding = [int(p) for p in padding]
    if len(padding) == 2:
        padding = [padding[0], padding[0], padding[1], padding[1]]
    return padding


def _parse_scaling(s):
    if isinstance(s, (int, np.integer)):
        return int(s), int(s)
    return int(s[0]), int(s[1])


# ----------------------------------------------------------------------------
# operators

def upfirdn2d(x, f, up=1, down=1, padding=0, flip_filter=False, gain=1):
    """torch_utils/ops/upfirdn2d.py:168-212."""
    x = np.ascontiguousarray(x)
    n, c, h, w = x.shape
    sfx = _suffix(x.dtype)
    upx, upy = _parse_scaling(up)
    dx, dy = _parse_scaling(down)
    px0, px1, py0, py1 = _parse_padding(padding)
    if f is None:
        f = np.ones([1, 1], np.float32)
    f = np.ascontiguousarray(f, dtype=np.float32)
    fn = getattr(lib(), 'sg3o_upfirdn2d_' + sfx)

    def run(src, filt2d, upx, upy, dx, dy, px0, px1, py0, py1, g):
        fh, fw = filt2d.shape
        sh, sw = src.shape[2:]
        oh = (sh * upy + py0 + py1 - fh) // dy + 1
        ow = (sw * upx + px0 + px1 - fw) // dx + 1
        assert oh >= 1 and ow >= 1
        dst = np.empty([n, c, oh, ow], x.dtype)
        fn(_ptr(src), _ptr(filt2d), _ptr(dst), n * c, sh, sw, fh, fw, upx, upy, dx, dy,
           px0, px1, py0, py1, int(bool(flip_filter)), ctypes.c_double(g))
        return dst

    if f.ndim == 2:
        return run(x, f, upx, upy, dx, dy, px0, px1, py0, py1, float(gain))
    g = float(gain) ** 0.5
    t = run(x, np.ascontiguousarray(f[None, :]), upx, 1, dx, 1, px0, px1, 0, 0, g)
    return run(t, np.ascontiguousarray(f[:, None]), 1, upy, 1, dy, 0, 0, py0, py1, g)


_ACT = {'linear': (1, 0.0, 1.0), 'relu': (2, 0.0, np.sqrt(2)), 'lrelu': (3, 0.2, np.sqrt(2)),
        'tanh': (4, 0.0, 1.0), 'sigmoid': (5, 0.0, 1.0), 'elu': (6, 0.0, 1.0),
        'selu': (7, 0.0, 1.0), 'softplus': (8, 0.0, 1.0), 'swish': (9, 0.0, np.sqrt(2))}


def bias_act(x, b=None, dim=1, act='linear', alpha=None, gain=None, clamp=None):
    """torch_utils/ops/bias_act.py:92-121."""
    x = np.ascontiguousarray(x)
    sfx = _suffix(x.dtype)
    idx, def_alpha, def_gain = _ACT[act]
    alpha = float(def_alpha if alpha is None else alpha)
    gain = float(def_gain if gain is None else gain)
    clamp = float(-1 if clamp is None else clamp)
    outer = int(np.prod(x.shape[:dim], dtype=np.int64)) if b is not None else 1
    nb = x.shape[dim] if b is not None else 1
    inner = int(np.prod(x.shape[dim + 1:], dtype=np.int64)) if b is not None else x.size
    if b is not None:
        b = np.ascontiguousarray(b, dtype=x.dtype)
        assert b.shape == (nb,)
    y = np.empty_like(x)
    getattr(lib(), 'sg3o_bias_act_' + sfx)(
        _ptr(x), _ptr(b), _ptr(y), ctypes.c_int64(outer), ctypes.c_int64(nb), ctypes.c_int64(inner),
        idx, ctypes.c_double(alpha), ctypes.c_double(gain), ctypes.c_double(clamp))
    return y


def filtered_lrelu(x, fu=None, fd=None, b=None, up=1, down=1, padding=0, gain=np.sqrt(2), slope=0.2,
                   clamp=None, flip_filter=False):
    """torch_utils/ops/filtered_lrelu.py:122-154."""
    x = np.ascontiguousarray(x)
    n, c, h, w = x.shape
    sfx = _suffix(x.dtype)
    px0, px1, py0, py1 = _parse_padding(padding)

    def prep(f):
        if f is None:
            return np.ones([1], np.float32), 1, 1
        f = np.ascontiguousarray(f, dtype=np.float32)
        if f.ndim == 1:
            return f, f.shape[0], 1
        assert f.shape[0] == f.shape[1]
        return f, f.shape[0], 0

    fu_a, fu_n, fu_sep = prep(fu)
    fd_a, fd_n, fd_sep = prep(fd)
    cw = w * up + px0 + px1 - (fu_n - 1)
    ch = h * up + py0 + py1 - (fu_n - 1)
    ow = (cw - fd_n) // down + 1
    oh = (ch - fd_n) // down + 1
    assert ow >= 1 and oh >= 1
    if b is not None:
        b = np.ascontiguousarray(b, dtype=x.dtype)
        assert b.shape == (c,)
    y = np.empty([n, c, oh, ow], x.dtype)
    rc = getattr(lib(), 'sg3o_filtered_lrelu_' + sfx)(
        _ptr(x), _ptr(fu_a), _ptr(fd_a), _ptr(b), _ptr(y), n * c, c, h, w, int(up), int(down),
        fu_n, fu_sep, fd_n, fd_sep, px0, px1, py0, py1,
        ctypes.c_double(float(gain)), ctypes.c_double(float(slope)),
        ctypes.c_double(-1.0 if clamp is None else float(clamp)), int(bool(flip_filter)))
    assert rc == 0, f'oracle filtered_lrelu failed rc={rc}'
    return y


def modulated_conv2d(x, w, s, demodulate=True, padding=0, input_gain=None):
    """models/stylegan3/networks_stylegan3.py:24-63 (scalar input_gain)."""
    x = np.ascontiguousarray(x)
    sfx = _suffix(x.dtype)
    w = np.ascontiguousarray(w, dtype=x.dtype)
    s = np.ascontiguousarray(s, dtype=x.dtype)
    n, ci, h, wd = x.shape
    co, ci2, k, k2 = w.shape
    assert ci == ci2 and k == k2 and s.shape == (n, ci)
    oh, ow = h + 2 * padding - k + 1, wd + 2 * padding - k + 1
    y = np.empty([n, co, oh, ow], x.dtype)
    getattr(lib(), 'sg3o_modulated_conv2d_' + sfx)(
        _ptr(x), _ptr(w), _ptr(s), _ptr(y), n, ci, co, h, wd, k, int(padding), int(bool(demodulate)),
        ctypes.c_double(1.0 if input_gain is None else float(input_gain)))
    return y


def conv2d(x, w, bias=None, stride=1, padding=0):
    x = np.ascontiguousarray(x, dtype=np.float32)
    w = np.ascontiguousarray(w, dtype=np.float32)
    n, ci, h, wd = x.shape
    co, _, k, _ = w.shape
    oh, ow = (h + 2 * padding - k) // stride + 1, (wd + 2 * padding - k) // stride + 1
    y = np.empty([n, co, oh, ow], np.float32)
    if bias is not None:
        bias = np.ascontiguousarray(bias, dtype=np.float32)
    lib().sg3o_conv2d_f32(_ptr(x), _ptr(w), _ptr(bias), _ptr(y), n, ci, co, h, wd, k, int(stride), int(padding))
    return y


# ----------------------------------------------------------------------------
# filter design (Kaiser-windowed sinc; radial jinc variant)

def kaiser_atten(numtaps, width):
    return 2.285 * (numtaps - 1) * np.pi * width + 7.95


def kaiser_beta(a):
    if a > 50:
        return 0.1102 * (a - 8.7)
    if a > 21:
        return 0.5842 * (a - 21) ** 0.4 + 0.07886 * (a - 21)
    return 0.0


def _bessel_j1(x):
    """J1 by its integral representation (fp64 trapezoid converges spectrally for a periodic integrand)."""
    x = np.asarray(x, dtype=np.float64)
    m = 4096
    t = (np.arange(m) + 0.5) * (np.pi / m)
    return np.cos(t[None, :] - x.reshape(-1, 1) * np.sin(t)[None, :]).mean(axis=1).reshape(x.shape)


def design_lowpass_filter(numtaps, cutoff, width, fs, radial=False):
    assert numtaps >= 1
    if numtaps == 1:
        return None
    nyq = fs / 2.0
    if not radial:
        beta = kaiser_beta(kaiser_atten(numtaps, width / nyq))
        win = np.kaiser(numtaps, beta)
        m = np.arange(numtaps) - 0.5 * (numtaps - 1)
        c = cutoff / nyq
        h = c * np.sinc(c * m) * win
        h = h / h.sum()
        return h.astype(np.float32)
    x = (np.arange(numtaps) - (numtaps - 1) / 2) / fs
    r = np.hypot(*np.meshgrid(x, x))
    f = _bessel_j1(2 * cutoff * (np.pi * r)) / (np.pi * r)
    beta = kaiser_beta(kaiser_atten(numtaps, width / nyq))
    w = np.kaiser(numtaps, beta)
    f = f * np.outer(w, w)
    f = f / f.sum()
    return f.astype(np.float32)


# ----------------------------------------------------------------------------
# network graph

def layer_schedule(w_dim=512, img_resolution=1024, img_channels=3, channel_base=32768, channel_max=512,
                   num_layers=14, num_critical=2, first_cutoff=2, first_stopband=2 ** 2.1,
                   last_stopband_rel=2 ** 0.3, margin_size=10, output_scale=0.25, num_fp16_res=4,
                   conv_kernel=3, filter_size=6, lrelu_upsampling=2, use_radial_filters=False,
                   conv_clamp=256, **_unused):
    """Per-layer geometry table (networks_stylegan3.py:430-469 and :294-333)."""
    last_cutoff = img_resolution / 2
    last_stopband = last_cutoff * last_stopband_rel
    exponents = np.minimum(np.arange(num_layers + 1) / (num_layers - num_critical), 1)
    cutoffs = first_cutoff * (last_cutoff / first_cutoff) ** exponents
    stopbands = first_stopband * (last_stopband / first_stopband) ** exponents
    sampling_rates = np.exp2(np.ceil(np.log2(np.minimum(stopbands * 2, img_resolution))))
    half_widths = np.maximum(stopbands, sampling_rates / 2) - cutoffs
    sizes = sampling_rates + margin_size * 2
    sizes[-2:] = img_resolution
    channels = np.rint(np.minimum((channel_base / 2) / cutoffs, channel_max))
    channels[-1] = img_channels
    layers = []
    for idx in range(num_layers + 1):
        prev = max(idx - 1, 0)
        is_torgb = idx == num_layers
        crit = idx >= num_layers - num_critical
        in_sr, out_sr = int(sampling_rates[prev]), int(sampling_rates[idx])
        tmp_sr = max(in_sr, out_sr) * (1 if is_torgb else lrelu_upsampling)
        k = 1 if is_torgb else conv_kernel
        up = int(np.rint(tmp_sr / in_sr))
        down = int(np.rint(tmp_sr / out_sr))
        up_taps = filter_size * up if up > 1 and not is_torgb else 1
        down_taps = filter_size * down if down > 1 and not is_torgb else 1
        radial = bool(use_radial_filters and not crit)
        in_size, out_size = int(sizes[prev]), int(sizes[idx])
        pad_total = (out_size - 1) * down + 1 - (in_size + k - 1) * up + up_taps + down_taps - 2
        pad_lo = (pad_total + up) // 2
        pad_hi = pad_total - pad_lo
        layers.append(dict(
            name=f'L{idx}_{out_size}_{int(channels[idx])}', is_torgb=is_torgb, in_channels=int(channels[prev]),
            out_channels=int(channels[idx]), in_size=in_size, out_size=out_size, conv_kernel=k,
            up=up, down=down, up_taps=up_taps, down_taps=down_taps, down_radial=radial,
            padding=[int(pad_lo), int(pad_hi), int(pad_lo), int(pad_hi)],
            up_filter_args=dict(numtaps=up_taps, cutoff=cutoffs[prev], width=half_widths[prev] * 2, fs=tmp_sr),
            down_filter_args=dict(numtaps=down_taps, cutoff=cutoffs[idx], width=half_widths[idx] * 2, fs=tmp_sr, radial=radial),
            conv_clamp=conv_clamp,
            use_fp16=bool(sampling_rates[idx] * (2 ** num_fp16_res) > img_resolution)))
    return dict(layers=layers, num_ws=num_layers + 2, w_dim=w_dim, img_resolution=img_resolution,
                img_channels=img_channels, output_scale=output_scale,
                input=dict(channels=int(channels[0]), size=int(sizes[0]), sampling_rate=float(sampling_rates[0]),
                           bandwidth=float(cutoffs[0])))


def fully_connected(x, weight, bias, activation='linear', lr_multiplier=1.0):
    """networks_stylegan3.py:86-100."""
    w = weight.astype(x.dtype) * np.asarray(lr_multiplier / np.sqrt(weight.shape[1]), x.dtype)
    y = x @ w.T
    b = None
    if bias is not None:
        b = bias.astype(x.dtype)
        if lr_multiplier != 1:
            b = b * np.asarray(lr_multiplier, x.dtype)
    if activation == 'linear':
        return y + b[None] if b is not None else y
    return bias_act(y, b, dim=1, act=activation)


def mapping(sd, z, num_ws, truncation_psi=1.0, truncation_cutoff=None, num_layers=2, lr_multiplier=0.01, prefix='mapping.'):
    """networks_stylegan3.py:134-160 (c_dim = 0)."""
    x = z.astype(np.float32)
    x = x * (1.0 / np.sqrt((x * x).mean(axis=1, keepdims=True) + 1e-8)).astype(np.float32)
    for i in range(num_layers):
        x = fully_connected(x, sd[f'{prefix}fc{i}.weight'], sd[f'{prefix}fc{i}.bias'], 'lrelu', lr_multiplier)
    x = np.repeat(x[:, None, :], num_ws, axis=1)
    if truncation_psi != 1:
        cut = num_ws if truncation_cutoff is None else truncation_cutoff
        w_avg = sd[f'{prefix}w_avg']
        x[:, :cut] = w_avg + (x[:, :cut] - w_avg) * np.float32(truncation_psi)
    return x


def _input_transform_params(sd, w, prefix):
    t = fully_connected(w, sd[prefix + 'affine.weight'], sd[prefix + 'affine.bias'])
    return t / np.sqrt((t[:, :2] ** 2).sum(axis=1, keepdims=True))


def synthesis_input(sd, sched, w=None, t=None, transform=None, prefix='synthesis.input.'):
    """networks_stylegan3.py:197-249.  `transform` overrides the stored buffer ([3,3] or [B,3,3])."""
    spec = sched['input']
    freqs = sd[prefix + 'freqs'].astype(np.float32)[None]            # [1,C,2]
    phases = sd[prefix + 'phases'].astype(np.float32)[None]          # [1,C]
    user = (sd[prefix + 'transform'] if transform is None else transform).astype(np.float32)
    if t is None:
        t = _input_transform_params(sd, w.astype(np.float32), prefix)
    t = t.astype(np.float32)
    bsz = t.shape[0]
    m_r = np.tile(np.eye(3, dtype=np.float32)[None], [bsz, 1, 1])
    m_r[:, 0, 0] = t[:, 0]; m_r[:, 0, 1] = -t[:, 1]; m_r[:, 1, 0] = t[:, 1]; m_r[:, 1, 1] = t[:, 0]
    m_t = np.tile(np.eye(3, dtype=np.float32)[None], [bsz, 1, 1])
    m_t[:, 0, 2] = -t[:, 2]; m_t[:, 1, 2] = -t[:, 3]
    transforms = m_r @ m_t @ user
    phases = phases + (freqs @ transforms[:, :2, 2:])[..., 0]
    freqs = freqs @ transforms[:, :2, :2]
    bw, sr = np.float32(spec['bandwidth']), np.float32(spec['sampling_rate'])
    amps = np.clip(1 - (np.sqrt((freqs ** 2).sum(axis=2)) - bw) / (sr / 2 - bw), 0, 1).astype(np.float32)
    size = spec['size']
    sx = np.float32(0.5 * size / spec['sampling_rate'])
    lin = ((2 * np.arange(size, dtype=np.float32) + 1) / np.float32(size) - 1).astype(np.float32)
    gx = (lin * sx)[None, :].repeat(size, 0)
    gy = (lin * sx)[:, None].repeat(size, 1)
    grid = np.stack([gx, gy], axis=-1).astype(np.float32)            # [H,W,2]
    x = np.einsum('hwk,bck->bhwc', grid, freqs).astype(np.float32) + phases[:, None, None, :]
    x = np.sin(x * np.float32(np.pi * 2)).astype(np.float32) * amps[:, None, None, :]
    weight = sd[prefix + 'weight'].astype(np.float32) / np.float32(np.sqrt(spec['channels']))
    x = x @ weight.T
    return np.ascontiguousarray(x.transpose(0, 3, 1, 2)).astype(np.float32)


def layer_styles(sd, layer, w):
    p = f"synthesis.{layer['name']}."
    s = fully_connected(w.astype(np.float32), sd[p + 'affine.weight'], sd[p + 'affine.bias'])
    if layer['is_torgb']:
        s = s * np.float32(1 / np.sqrt(layer['in_channels'] * layer['conv_kernel'] ** 2))
    return s


def _f16(a):
    """Round to float16 and back (the value a float16 tensor holds)."""
    return np.asarray(a, dtype=np.float32).astype(np.float16).astype(np.float32)


def modulated_conv2d_fp16(x, w, s, demodulate=True, padding=0, input_gain=None):
    """networks_stylegan3.py:24-63 as the reference executes it for a `use_fp16` layer on a GPU (x.dtype == float16, :355-357):
    the per-sample weights are formed in fp32 (:39-56) and cast to the activation dtype (:61 `w.to(x.dtype)`), the grouped
    convolution multiplies fp16 operands, accumulates in fp32 and returns fp16.  NOT pinned by a reference fixture (the reference's
    fp16 path needs a GPU): a restatement of its rounding points, used as an independent yardstick for the HIP fp16 path."""
    x = _f16(x); w = np.asarray(w, dtype=np.float32); s = np.asarray(s, dtype=np.float32)
    n = x.shape[0]
    if demodulate:
        w = w * (np.float32(1) / np.sqrt(np.mean(np.square(w), axis=(1, 2, 3), keepdims=True)))
        s = s * (np.float32(1) / np.sqrt(np.mean(np.square(s))))
    we = w[None] * s[:, None, :, None, None]                                         # [N,O,I,k,k]
    if demodulate:
        we = we * (np.float32(1) / np.sqrt(np.sum(np.square(we), axis=(2, 3, 4), keepdims=True) + np.float32(1e-8)))
    if input_gain is not None:
        we = we * np.float32(input_gain)
    we = _f16(we)
    y = np.concatenate([conv2d(np.ascontiguousarray(x[i:i + 1]), np.ascontiguousarray(we[i]), None, 1, padding) for i in range(n)], axis=0)
    return _f16(y)


def synthesis_layer(sd, layer, x, styles, fp16=False):
    """networks_stylegan3.py:335-368 (update_emas=False).  `fp16`: the layer runs as the reference runs a `use_fp16` layer on a GPU
    without force_fp32 (:355-366): fp16 activations and convolution operands, fp32 arithmetic inside filtered_lrelu, fp16 output."""
    p = f"synthesis.{layer['name']}."
    input_gain = 1.0 / np.sqrt(np.float32(sd[p + 'magnitude_ema']))
    conv = modulated_conv2d_fp16 if fp16 else modulated_conv2d
    x = conv(x, sd[p + 'weight'], styles, demodulate=not layer['is_torgb'],
             padding=layer['conv_kernel'] - 1, input_gain=input_gain)
    gain = 1.0 if layer['is_torgb'] else np.sqrt(2)
    slope = 1.0 if layer['is_torgb'] else 0.2
    fu = sd.get(p + 'up_filter')
    fd = sd.get(p + 'down_filter')
    b = sd[p + 'bias'].astype(x.dtype)
    y = filtered_lrelu(x, fu=fu, fd=fd, b=_f16(b) if fp16 else b, up=layer['up'], down=layer['down'],
                       padding=layer['padding'], gain=gain, slope=slope, clamp=layer['conv_clamp'])
    return _f16(y) if fp16 else y


def w2s(sd, sched, ws):
    """networks_stylegan3.py:503-525."""
    ws = ws.astype(np.float32)
    out = {'input': _input_transform_params(sd, ws[:, 0], 'synthesis.input.')}
    for layer, i in zip(sched['layers'], range(1, ws.shape[1])):
        out[layer['name']] = layer_styles(sd, layer, ws[:, i])
    return out


def synthesis(sd, sched, ws=None, all_s=None, transform=None, return_layers=False, mixed_fp16=False):
    """networks_stylegan3.py:471-494: W path or StyleSpace (all_s) path, fp32; `mixed_fp16`: the layers flagged use_fp16 run as on a
    GPU without force_fp32 (see synthesis_layer)."""
    feats = []
    if all_s is None:
        ws = ws.astype(np.float32)
        assert ws.shape[1:] == (sched['num_ws'], sched['w_dim'])
        x = synthesis_input(sd, sched, w=ws[:, 0], transform=transform)
        for i, layer in enumerate(sched['layers']):
            x = synthesis_layer(sd, layer, x, layer_styles(sd, layer, ws[:, i + 1]), fp16=mixed_fp16 and layer['use_fp16'])
            if return_layers:
                feats.append(x)
    else:
        x = synthesis_input(sd, sched, t=all_s['input'], transform=transform)
        for layer in sched['layers']:
            x = synthesis_layer(sd, layer, x, all_s[layer['name']].astype(np.float32), fp16=mixed_fp16 and layer['use_fp16'])
            if return_layers:
                feats.append(x)
    if sched['output_scale'] != 1:
        x = x * np.float32(sched['output_scale'])
    x = x.astype(np.float32)
    return (x, feats) if return_layers else x


def filter_taps_for(sched):
    """Design every layer's up/down filter (what SynthesisLayer.__init__ registers as buffers)."""
    out = {}
    for layer in sched['layers']:
        p = f"synthesis.{layer['name']}."
        fu = design_lowpass_filter(**layer['up_filter_args'])
        fd = design_lowpass_filter(**layer['down_filter_args'])
        if fu is not None:
            out[p + 'up_filter'] = fu
        if fd is not None:
            out[p + 'down_filter'] = fd
    return out


# ----------------------------------------------------------------------------
# ReStyle encoder (eval mode)

def _bn(x, sd, p, eps=1e-5):
    a = sd[p + 'weight'] / np.sqrt(sd[p + 'running_var'] + eps)
    b = sd[p + 'bias'] - sd[p + 'running_mean'] * a
    return (x * a[None, :, None, None] + b[None, :, None, None]).astype(np.float32)


def _prelu(x, slope):
    return np.where(x >= 0, x, x * slope[None, :, None, None]).astype(np.float32)


def bottleneck_ir_se(sd, p, x, stride):
    """models/setgan/encoder/encoders/helpers.py:98-120 (eval): BN, conv3x3, PReLU, conv3x3(stride), BN, SE; + shortcut."""
    r = _bn(x, sd, p + 'res_layer.0.')
    r = conv2d(r, sd[p + 'res_layer.1.weight'], None, 1, 1)
    r = _prelu(r, sd[p + 'res_layer.2.weight'])
    r = conv2d(r, sd[p + 'res_layer.3.weight'], None, stride, 1)
    r = _bn(r, sd, p + 'res_layer.4.')
    pooled = r.mean(axis=(2, 3))                                                      # helpers.py:66
    g = np.maximum(pooled @ sd[p + 'res_layer.5.fc1.weight'][:, :, 0, 0].T, 0)
    g = 1.0 / (1.0 + np.exp(-(g @ sd[p + 'res_layer.5.fc2.weight'][:, :, 0, 0].T)))
    r = r * g[:, :, None, None].astype(np.float32)
    if p + 'shortcut_layer.0.weight' in sd:
        s = conv2d(x, sd[p + 'shortcut_layer.0.weight'], None, stride, 0)
        s = _bn(s, sd, p + 'shortcut_layer.1.')
    else:
        s = x[:, :, ::stride, ::stride]                                               # MaxPool2d(1, stride)
    return (r + s).astype(np.float32)


def backbone_encoder(sd, x, num_layers=50, n_styles=16, return_feats=False):
    """models/setgan/encoder/encoders/restyle_psp_encoders.py:43-50 with helpers.get_blocks (:30-55), mode 'ir_se'."""
    units = {50: (3, 4, 14, 3), 100: (3, 13, 30, 3), 152: (3, 8, 36, 3)}[num_layers]
    h = conv2d(x, sd['input_layer.0.weight'], None, 1, 1)
    h = _prelu(_bn(h, sd, 'input_layer.1.'), sd['input_layer.2.weight'])
    feats = {'stem': h}
    i = 0
    for n_units in units:
        for u in range(n_units):
            h = bottleneck_ir_se(sd, f'body.{i}.', h, 2 if u == 0 else 1)
            feats[f'body{i}'] = h
            i += 1
    codes = _style_heads(sd, h, n_styles)
    return (codes, feats) if return_feats else codes


def _style_heads(sd, h, n_styles):
    """map2style.py:8-25 for every head on the trunk output h [N,512,16,16]; stacked (restyle_psp_encoders.py:46-50, :93-97)."""
    codes = []
    for j in range(n_styles):
        t = h
        for c in (0, 2, 4, 6):
            t = conv2d(t, sd[f'styles.{j}.convs.{c}.weight'], sd[f'styles.{j}.convs.{c}.bias'], 2, 1)
            t = np.where(t >= 0, t, t * np.float32(0.01)).astype(np.float32)
        t = t.reshape(-1, t.shape[1])
        w = sd[f'styles.{j}.linear.weight'] * np.float32(1.0 / np.sqrt(t.shape[1]))
        codes.append(t @ w.T + sd[f'styles.{j}.linear.bias'][None])
    return np.stack(codes, axis=1).astype(np.float32)


def resnet_backbone_encoder(sd, x, n_styles=16, return_feats=False):
    """models/setgan/encoder/encoders/restyle_psp_encoders.py:53-97 (eval): conv7x7 s2 + BN + PReLU, then resnet34's layer1..4
    flattened into `body` -- torchvision BasicBlock: relu(bn2(conv2(relu(bn1(conv1(x))))) + (downsample(x) | x)), conv1 strided
    (torchvision is absent here; its BasicBlock.forward is restated from the published definition) -- then the style heads."""
    h = conv2d(x, sd['conv1.weight'], None, 2, 3)
    h = _prelu(_bn(h, sd, 'bn1.'), sd['relu.weight'])
    feats = {'stem': h}
    i = 0
    for planes, count, stride in ((64, 3, 1), (128, 4, 2), (256, 6, 2), (512, 3, 2)):
        for u in range(count):
            p, st = f'body.{i}.', (stride if u == 0 else 1)
            r = np.maximum(_bn(conv2d(h, sd[p + 'conv1.weight'], None, st, 1), sd, p + 'bn1.'), 0)
            r = _bn(conv2d(r, sd[p + 'conv2.weight'], None, 1, 1), sd, p + 'bn2.')
            if p + 'downsample.0.weight' in sd:
                h = _bn(conv2d(h, sd[p + 'downsample.0.weight'], None, st, 0), sd, p + 'downsample.1.')
            h = np.maximum(r + h, 0).astype(np.float32)
            feats[f'body{i}'] = h
            i += 1
    codes = _style_heads(sd, h, n_styles)
    return (codes, feats) if return_feats else codes


# ----------------------------------------------------------------------------
# ReStyle wrapper and loop.  Restated from the reference text: psp3.py / inference_utils.py import
# torchvision / pyrallis (absent here), so these are pinned by this restatement; the encoder and the synthesis under them
# are pinned by reference-generated fixtures.

def adaptive_avg_pool(x, out_hw=(256, 256)):
    """torch.nn.AdaptiveAvgPool2d (psp3.py:18 `face_pool`): bin i covers [floor(i*n/o), ceil((i+1)*n/o))."""
    n, c, h, w = x.shape
    oh, ow = out_hw
    if (h, w) == (oh, ow):
        return x.astype(np.float32)
    def bins(n_in, n_out):
        return [(int(np.floor(i * n_in / n_out)), int(np.ceil((i + 1) * n_in / n_out))) for i in range(n_out)]
    if h % oh == 0 and w % ow == 0:                       # equal bins: a reshape-mean
        return x.reshape(n, c, oh, h // oh, ow, w // ow).mean(axis=(3, 5), dtype=np.float64).astype(np.float32)
    rows = np.stack([x[:, :, a:b, :].mean(axis=2, dtype=np.float64) for a, b in bins(h, oh)], axis=2)
    return np.stack([rows[:, :, :, a:b].mean(axis=3) for a, b in bins(w, ow)], axis=3).astype(np.float32)


def psp_forward(enc_sd, gen_sd, sched, x, latent=None, latent_avg=None, resize=True, input_code=False, landmarks_transform=None, encoder=None):
    """models/setgan/encoder/psp3.py:45-84.  Returns (aligned images, unaligned images or None, codes).  `encoder` (a callable
    x -> codes) replaces the IR-SE50 backbone built from `enc_sd` (the loop fixtures use a small stand-in encoder)."""
    if input_code:
        codes = x.astype(np.float32)
    else:
        codes = backbone_encoder(enc_sd, x) if encoder is None else np.asarray(encoder(x), dtype=np.float32)
        if x.shape[1] == 6 and latent is not None:
            codes = codes + latent                                          # :56-58 residual step
        else:
            codes = codes + np.asarray(latent_avg).reshape(1, -1, codes.shape[2])     # :59-60 first step: latent_avg.repeat(N, 1, 1)
        codes = codes.astype(np.float32)
    eye = np.repeat(np.eye(3, dtype=np.float32)[None], x.shape[0], axis=0)          # :63-66
    images = synthesis(gen_sd, sched, ws=codes, transform=eye)
    if resize:
        images = adaptive_avg_pool(images)
    unaligned = None
    if landmarks_transform is not None:                                     # :72-76
        unaligned = synthesis(gen_sd, sched, ws=codes, transform=landmarks_transform.astype(np.float32))
        if resize:
            unaligned = adaptive_avg_pool(unaligned)
    return images, unaligned, codes


def get_average_image(enc_sd, gen_sd, sched, latent_avg):
    """utils/inference_utils.py:59-64: the image of latent_avg repeated over the 16 styles, pooled (resize defaults to True)."""
    x = np.repeat(latent_avg.reshape(1, -1), 16, axis=0)[None]
    return psp_forward(enc_sd, gen_sd, sched, x, input_code=True)[0][0]


def run_on_batch(enc_sd, gen_sd, sched, inputs, latent_avg, avg_image, n_iters, landmarks_transform=None, resize_outputs=False, encoder=None):
    """utils/inference_utils.py:67-111.  Returns (per-step output images [steps][N,3,H,W], per-step latents [steps][N,16,512],
    the ALIGNED image of the last step).  With transforms the reference's last entry is the unaligned image (:96-100); every
    other quantity is the same with and without them, so one call pins both forms."""
    y_hat, latent = None, None
    step_images, step_latents = [], []
    for it in range(n_iters):
        second = np.repeat(avg_image[None], inputs.shape[0], axis=0) if it == 0 else y_hat          # :76-80
        x_input = np.concatenate([inputs, second], axis=1).astype(np.float32)
        aligned, unaligned, latent = psp_forward(enc_sd, gen_sd, sched, x_input, latent=latent, latent_avg=latent_avg,
                                                 resize=resize_outputs, landmarks_transform=landmarks_transform, encoder=encoder)
        # :92-100: aligned output, except that the LAST step returns the unaligned one when transforms are given
        y_hat = unaligned if (landmarks_transform is not None and it == n_iters - 1) else aligned
        step_images.append(y_hat)
        step_latents.append(latent)
        y_hat = adaptive_avg_pool(y_hat)                                     # :108 face_pool before the next step
    return step_images, step_latents, aligned


# ----------------------------------------------------------------------------
# Callers of the synthesis path: field-of-view expansion, video post-processing, StyleSpace edit sweep.
# Restated from the reference text; these modules cannot be imported here (imageio / clip / pyrallis are absent),
# so their parity is pinned by this restatement only.

def make_transform(translate, angle):
    """utils/common.py:9-19."""
    m = np.eye(3)
    s, c = np.sin(angle / 360.0 * np.pi * 2), np.cos(angle / 360.0 * np.pi * 2)
    m[0][0] = c; m[0][1] = s; m[0][2] = translate[0]
    m[1][0] = -s; m[1][1] = c; m[1][2] = translate[1]
    return m


def fov_transforms(res, pixels_right, pixels_left, pixels_top, pixels_bottom):
    """utils/fov_expansion.py:35-84: the nine tile transforms (None where no pixels are requested), already inverted."""
    def edge(e, n):
        if n == 0:
            return None
        return make_transform({'left': (n / res, 0), 'right': (-n / res, 0), 'top': (0, n / res), 'bottom': (0, -n / res)}[e], 0)

    def corner(cn, nh, nv):
        if nh == 0 or nv == 0:
            return None
        return make_transform({'top_left': (nh / res, nv / res), 'top_right': (-nh / res, nv / res),
                               'bottom_left': (nh / res, -nv / res), 'bottom_right': (-nh / res, -nv / res)}[cn], 0)
    ts = [make_transform((0, 0), 0), edge('left', pixels_left), edge('top', pixels_top), edge('right', pixels_right),
          edge('bottom', pixels_bottom), corner('top_left', pixels_left, pixels_top), corner('top_right', pixels_right, pixels_top),
          corner('bottom_right', pixels_right, pixels_bottom), corner('bottom_left', pixels_left, pixels_bottom)]
    return [None if t is None else np.linalg.inv(t) for t in ts]


def fov_merge(images, res, pr, pl, pt, pb):
    """utils/fov_expansion.py:86-108."""
    out = np.zeros((images[0].shape[0], 3, pt + res + pb, pl + res + pr), np.float32)
    out[:, :, pt:pt + res, pl:pl + res] = images[0]
    if pl > 0:
        out[:, :, pt:pt + res, :pl] = images[1][:, :, :, 0:pl]
    if pt > 0:
        out[:, :, :pt, pl:pl + res] = images[2][:, :, 0:pt, :]
    if pr > 0:
        out[:, :, pt:pt + res, pl + res:] = images[3][:, :, :, res - pr:]
    if pb > 0:
        out[:, :, pt + res:, pl:pl + res] = images[4][:, :, res - pb:, :]
    if pt > 0 and pl > 0:
        out[:, :, :pt, :pl] = images[5][:, :, :pt, :pl]
    if pt > 0 and pr > 0:
        out[:, :, :pt, res + pl:] = images[6][:, :, :pt, res - pr:]
    if pb > 0 and pr > 0:
        out[:, :, res + pt:, res + pl:] = images[7][:, :, res - pb:, res - pr:]
    if pb > 0 and pl > 0:
        out[:, :, res + pt:, :pl] = images[8][:, :, res - pb:, :pl]
    return out


def expand_fov(sd, sched, ws, landmark_t, pixels_right=0, pixels_left=0, pixels_top=0, pixels_bottom=0):
    """utils/fov_expansion.py:13-30: one synthesis pass per tile with transform = landmark_t @ tile transform."""
    res = sched['img_resolution']
    images = []
    for t in fov_transforms(res, pixels_right, pixels_left, pixels_top, pixels_bottom):
        images.append(None if t is None else synthesis(sd, sched, ws=ws, transform=(landmark_t @ t).astype(np.float32)))
    return fov_merge(images, res, pixels_right, pixels_left, pixels_top, pixels_bottom)


def smooth_ws(ws):
    """inversion/video/post_processing.py:49-52."""
    return (ws[2:-2] + 0.75 * ws[3:-1] + 0.75 * ws[1:-3] + 0.25 * ws[:-4] + 0.25 * ws[4:]) / 3


def postprocess_latents(result_latents):
    """inversion/video/post_processing.py:13-19: fine layers averaged over frames, then smoothed."""
    lat = np.array(result_latents)
    lat[:, 9:, :] = lat[:, 9:, :].mean(axis=0)
    return smooth_ws(lat)


def styleclip_delta_s(delta_i_c, delta_i, beta, s_std, example_s):
    """editing/styleclip_global_directions/global_direction.py:7-40."""
    r_c = delta_i_c @ delta_i
    delta_s = r_c.copy()
    delta_s[np.abs(r_c) < beta] = 0
    peak = np.abs(delta_s).max()
    if peak > 0:
        delta_s = delta_s / peak
    out, start = {}, 0
    for key in example_s:
        n = example_s[key].shape[1]
        out[key] = (delta_s[start:start + n] * s_std[key])[None]
        start += n
    return out


def styleclip_sweep(sd, sched, latent, delta_i_c, delta_i, s_std, alphas, betas, transform=None):
    """editing/styleclip_global_directions/edit.py:124-168: W2S, then one batch-1 synthesis per (beta, alpha)."""
    s = w2s(sd, sched, latent[None])
    base = {c: s[c][0][None] for c in s}
    images = []
    for beta in betas:
        d = styleclip_delta_s(delta_i_c, delta_i, beta, s_std, base)
        for alpha in alphas:
            edited = {c: (base[c] + np.float32(alpha) * d[c]).astype(np.float32) for c in base}
            images.append(synthesis(sd, sched, all_s=edited, transform=transform))
    return np.concatenate(images)
