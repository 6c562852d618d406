"""StyleGAN3 alias-free generator graph for the MI355X build.

Public surface, constructor arguments and parameter / buffer names are those of the reference graph
(reference models/stylegan3/networks_stylegan3.py: modulated_conv2d :24, FullyConnectedLayer :68, MappingNetwork
:108, SynthesisInput :168, SynthesisLayer :259, SynthesisNetwork :406 with the fork's StyleSpace `all_s` path and
`W2S` :503, Generator :531), so `.pt` state_dicts load unchanged and callers (pSp/e4e wrappers, PTI, FOV expansion,
StyleCLIP sweeps) keep working.  The implementation is organised differently:

  * the per-layer geometry (cutoffs, sampling rates, sizes, channels, up/down factors, taps, padding) is computed
    once by `synthesis_schedule()` as a table of `LayerGeometry` records that SynthesisNetwork hands to its layers
    (and that bench.py uses for byte / FLOP accounting);
  * the modulated convolution runs as one batch-wide MFMA implicit GEMM (torch_utils/ops/modulated_conv.py)
    instead of a grouped convolution over N-fold expanded weights;
  * bias + up-FIR + lrelu + clamp + down-FIR is the fused streaming HIP kernel behind
    torch_utils.ops.filtered_lrelu.

Small dense algebra that the reference also leaves to the BLAS library (the 512-wide affine / mapping layers and
the 36x36 Fourier-feature GEMM) stays on torch ops.
"""
import collections
import weakref

import numpy as np
import scipy.signal
import scipy.special
import torch

from torch_utils import misc, persistence
from torch_utils.ops import affine_batch, bias_act, filtered_lrelu, fourier_features
from torch_utils.ops import modulated_conv as _modconv

# Inference-time constants derived from parameters (scaled affine weights, the input's sampling grid and mixing matrix, the
# layers' input gains), kept per module OUTSIDE the module's __dict__ so that they are neither pickled nor deep-copied with it;
# every entry carries the `(data_ptr, _version)` of the parameters it was made from and is rebuilt when those change.
_derived = weakref.WeakKeyDictionary()


def derived_tensors(root):
    """Every tensor the tables above hold for `root` and its submodules (what a captured graph of `root` may read)."""
    found = []

    def walk(v):
        if isinstance(v, torch.Tensor):
            found.append(v)
        elif isinstance(v, dict):
            for x in v.values():
                walk(x)
        elif isinstance(v, (list, tuple)):
            for x in v:
                walk(x)
        elif isinstance(v, affine_batch.AffinePack):
            walk([getattr(v, n, None) for n in ('weight', 'bias', 'scale', 'row_start', 'ws_index')])
    for m in root.modules():
        walk(_derived.get(m))
    return found


def _derived_of(module):
    d = _derived.get(module)
    if d is None:
        d = _derived[module] = {}
    return d



def _keep(module, **fields):
    """Store constructor arguments as plain attributes (they are part of the pickled state and of the public surface)."""
    for name, value in fields.items():
        setattr(module, name, value)


def _xy(value):
    return np.broadcast_to(np.asarray(value), [2])


def _summary(module, *rows):
    """extra_repr text: one line per row of attribute names; floats in %g, sizes as lists, the rest via str()."""
    def show(name):
        v = getattr(module, name)
        if isinstance(v, np.ndarray):
            return f'{name}={[int(e) for e in v]}'
        return f'{name}={v:g}' if isinstance(v, (float, np.floating)) else f'{name}={v}'
    return '\n'.join(', '.join(show(n) for n in row.split()) + (',' if i + 1 < len(rows) else '') for i, row in enumerate(rows))


@misc.profiled_function
def modulated_conv2d(x, w, s, demodulate=True, padding=0, input_gain=None, x_bound=None, prepared=None, epilogue=None, align_rows=False):
    """x [N,I,H,W], w [O,I,kh,kw], s [N,I], input_gain [] | [I] | [N,I]  ->  [N,O,H',W'].

    Equals a per-sample convolution with weights  w * s[n] (unit-normalised and demodulated when `demodulate`) times
    `input_gain` (reference :24-63).  `x_bound` is an extension: a guaranteed bound on |x| that lets the HIP kernel use
    its split-precision matrix-core path (torch_utils/ops/modulated_conv.py)."""
    return _modconv.modulated_conv2d(x, w, s, demodulate=demodulate, padding=padding, input_gain=input_gain, x_bound=x_bound,
                                     prepared=prepared, epilogue=epilogue, align_rows=align_rows)



@persistence.persistent_class
class FullyConnectedLayer(torch.nn.Module):
    """y = act(x @ (W * lr_mul / sqrt(in)).T + b * lr_mul)   (equalised learning rate)."""

    def __init__(self, in_features, out_features, activation='linear', bias=True, lr_multiplier=1, weight_init=1, bias_init=0):
        super().__init__()
        _keep(self, in_features=in_features, out_features=out_features, activation=activation,
              weight_gain=lr_multiplier / np.sqrt(in_features), bias_gain=lr_multiplier)
        self.weight = torch.nn.Parameter(torch.randn([out_features, in_features]) * (weight_init / lr_multiplier))
        b0 = np.broadcast_to(np.asarray(bias_init, dtype=np.float32), [out_features])
        self.bias = torch.nn.Parameter(torch.from_numpy(b0 / lr_multiplier)) if bias else None

    def _scaled(self, dtype):
        w = self.weight.to(dtype) * self.weight_gain
        if self.bias is None:
            return w, None
        b = self.bias.to(dtype)
        return w, (b if self.bias_gain == 1 else b * self.bias_gain)

    def forward(self, x):
        w, b = self._scaled(x.dtype)
        if b is not None and self.activation == 'linear':
            return torch.addmm(b.unsqueeze(0), x, w.t())
        return bias_act.bias_act(x.matmul(w.t()), b, act=self.activation)

    def extra_repr(self):
        return _summary(self, 'in_features out_features activation')



@persistence.persistent_class
class MappingNetwork(torch.nn.Module):
    """z (and optional label c) -> num_ws copies of the intermediate latent w, with truncation towards w_avg."""

    def __init__(self, z_dim, c_dim, w_dim, num_ws, num_layers=2, lr_multiplier=0.01, w_avg_beta=0.998):
        super().__init__()
        _keep(self, z_dim=z_dim, c_dim=c_dim, w_dim=w_dim, num_ws=num_ws, num_layers=num_layers, w_avg_beta=w_avg_beta)
        self.embed = FullyConnectedLayer(c_dim, w_dim) if c_dim > 0 else None
        widths = [z_dim + (w_dim if c_dim > 0 else 0)] + [w_dim] * num_layers
        for i in range(num_layers):
            setattr(self, f'fc{i}', FullyConnectedLayer(widths[i], widths[i + 1], activation='lrelu', lr_multiplier=lr_multiplier))
        self.register_buffer('w_avg', torch.zeros([w_dim]))

    @staticmethod
    def _unit_rms(v):
        return v * (v.square().mean(1, keepdim=True) + 1e-8).rsqrt()

    def forward(self, z, c, truncation_psi=1, truncation_cutoff=None, update_emas=False):
        misc.assert_shape(z, [None, self.z_dim])
        x = self._unit_rms(z.to(torch.float32))
        if self.c_dim > 0:
            misc.assert_shape(c, [None, self.c_dim])
            x = torch.cat([x, self._unit_rms(self.embed(c.to(torch.float32)))], dim=1)
        for i in range(self.num_layers):
            x = getattr(self, f'fc{i}')(x)
        if update_emas:
            self.w_avg.copy_(x.detach().mean(dim=0).lerp(self.w_avg, self.w_avg_beta))
        ws = x.unsqueeze(1).repeat([1, self.num_ws, 1])
        if truncation_psi != 1:
            cut = self.num_ws if truncation_cutoff is None else truncation_cutoff
            ws[:, :cut] = self.w_avg.lerp(ws[:, :cut], truncation_psi)
        return ws

    def extra_repr(self):
        return _summary(self, 'z_dim c_dim w_dim num_ws')



@persistence.persistent_class
class SynthesisInput(torch.nn.Module):
    """Fourier-feature input: `channels` sinusoids with frequencies inside a disc of radius `bandwidth`, transformed by
    a learned rotation+translation (from w) and by the user-assignable `transform` buffer ([3,3] or [B,3,3])."""

    def __init__(self, w_dim, channels, size, sampling_rate, bandwidth):
        super().__init__()
        _keep(self, w_dim=w_dim, channels=channels, size=_xy(size), sampling_rate=sampling_rate, bandwidth=bandwidth)

        # directions uniform on the circle, radii distributed so that the spectrum is flat inside the band limit
        freqs = torch.randn([channels, 2])
        r = freqs.norm(dim=1, keepdim=True)
        freqs = freqs / (r * r.square().exp().pow(0.25)) * bandwidth
        phases = torch.rand([channels]) - 0.5

        self.weight = torch.nn.Parameter(torch.randn([channels, channels]))
        self.affine = FullyConnectedLayer(w_dim, 4, weight_init=0, bias_init=[1, 0, 0, 0])
        self.register_buffer('transform', torch.eye(3, 3))
        self.register_buffer('freqs', freqs)
        self.register_buffer('phases', phases)

    def output_bound(self):
        """|features| <= max_o sum_c |weight[o,c]| / sqrt(C)  (|sin| <= 1, amplitudes <= 1); cached per weight version."""
        key = (self.weight.data_ptr(), self.weight._version)
        if getattr(self, '_bound_key', None) != key:
            self._bound_key = key
            self._bound = float(self.weight.detach().abs().sum(dim=1).max()) / float(np.sqrt(self.channels)) * 1.01
        return self._bound

    def transform_params(self, w):
        """(r_c, r_s, t_x, t_y) predicted from w, rotation part normalised to unit length."""
        t = self.affine(w)
        return t / t[:, :2].norm(dim=1, keepdim=True)

    def sampling_grid(self, device):
        """The [1,H,W,2] grid of sampling positions (reference :231-234): a function of the layer's geometry only, kept per device
        in inference."""
        theta = torch.eye(2, 3, device=device)        # scaled in place by scalar kernels: no host->device copy, so the
        theta[0, 0].mul_(0.5 * self.size[0] / self.sampling_rate)   # forward can be captured into a HIP graph
        theta[1, 1].mul_(0.5 * self.size[1] / self.sampling_rate)
        return torch.nn.functional.affine_grid(theta.unsqueeze(0), [1, 1, self.size[1], self.size[0]], align_corners=False)

    def fast_path_ok(self):
        """GPU inference runs the features and their channel mix on HIP kernels; the mix kernel loads pixel pairs."""
        return (int(self.size[0]) * int(self.size[1])) % 2 == 0

    def _inference_constants(self, n, device):
        """What GPU inference would otherwise recompute every call: the sampling grid, the scaled mixing matrix (as a 1x1
        convolution weight; rebuilt when the parameter changes) and the all-ones style of that convolution."""
        key = (str(device), self.weight.data_ptr(), self.weight._version)
        store = _derived_of(self)
        c = store.get('input')
        if c is None or c['key'] != key:
            with torch.no_grad():
                c = dict(key=key, grid=self.sampling_grid(device)[0].contiguous(),
                         mix=(self.weight.detach() / np.sqrt(self.channels)).unsqueeze(2).unsqueeze(3).contiguous(), ones={})
            store['input'] = c
        if n not in c['ones']:
            c['ones'][n] = torch.ones([n, self.channels], device=device)
        return c['grid'], c['mix'], c['ones'][n]

    def mix_spec(self, n, device):
        """The channel mix (reference :243-244) as this network's first entry of `modulated_conv.prepare_batch`."""
        _, mix, ones = self._inference_constants(n, device)
        return dict(w=mix, s=ones, demodulate=False, padding=0, input_gain=None, x_bound=1.001, n=n,
                    h=int(self.size[1]), wd=int(self.size[0]), dtype=torch.float32)

    def forward_inference(self, t, normalise, prepared):
        """GPU inference (no autograd): transform algebra, features and channel mix as three launches (reference :204-244).
        t [N,4]: the affine output (`normalise`) or `transform_params` (already normalised); `prepared`: the entry of `mix_spec`."""
        n = int(t.shape[0])
        grid, mix, ones = self._inference_constants(n, t.device)
        freqs, phases, amps = fourier_features.input_transform(t.to(torch.float32), self.transform.to(torch.float32), self.freqs, self.phases,
                                                               self.bandwidth, self.sampling_rate, normalise)
        x = fourier_features.fourier_features(grid, freqs, phases, amps)
        x = modulated_conv2d(x, mix, ones, demodulate=False, x_bound=1.001, prepared=prepared)
        misc.assert_shape(x, [n, self.channels, int(self.size[1]), int(self.size[0])])
        return x

    def forward(self, w, t=None):
        if t is None:
            t = self.transform_params(w)
        device, n = t.device, t.shape[0]
        # inverse rotation, then inverse translation, then the user transform (all w.r.t. the output image)
        rot = torch.eye(3, device=device).repeat(n, 1, 1)
        rot[:, 0, 0], rot[:, 0, 1], rot[:, 1, 0], rot[:, 1, 1] = t[:, 0], -t[:, 1], t[:, 1], t[:, 0]
        trans = torch.eye(3, device=device).repeat(n, 1, 1)
        trans[:, 0, 2], trans[:, 1, 2] = -t[:, 2], -t[:, 3]
        m = rot @ trans @ self.transform

        freqs = self.freqs.unsqueeze(0)
        phases = self.phases.unsqueeze(0) + (freqs @ m[:, :2, 2:]).squeeze(2)
        freqs = freqs @ m[:, :2, :2]
        # fade out frequencies that the transform pushed beyond the band limit
        amps = (1 - (freqs.norm(dim=2) - self.bandwidth) / (self.sampling_rate / 2 - self.bandwidth)).clamp(0, 1)

        grid = self.sampling_grid(device)

        mix = self.weight / np.sqrt(self.channels)
        if grid.is_cuda and not torch.is_grad_enabled() and (int(self.size[0]) * int(self.size[1])) % 2 == 0:
            # GPU inference: the features in one kernel, channels-first and bit-identical to the torch ops below (whose K = 2
            # matmul is the BLAS library's worst case), then the channel mix as a 1x1 convolution on the split-precision
            # matrix-core kernel (|features| <= 1): 10 x faster than the fp32 GEMM picked for [N*H*W, C] x [C, C]
            x = fourier_features.fourier_features(grid[0], freqs, phases, amps)
            x = modulated_conv2d(x, mix.unsqueeze(2).unsqueeze(3), torch.ones([n, self.channels], device=device),
                                 demodulate=False, x_bound=1.001)
        else:
            x = (grid.unsqueeze(3) @ freqs.permute(0, 2, 1).unsqueeze(1).unsqueeze(2)).squeeze(3)   # [N,H,W,C]
            x = torch.sin((x + phases.unsqueeze(1).unsqueeze(2)) * (np.pi * 2)) * amps.unsqueeze(1).unsqueeze(2)
            x = (x @ mix.t()).permute(0, 3, 1, 2)
        misc.assert_shape(x, [n, self.channels, int(self.size[1]), int(self.size[0])])
        return x

    def extra_repr(self):
        return _summary(self, 'w_dim channels size', 'sampling_rate bandwidth')



def design_lowpass_filter(numtaps, cutoff, width, fs, radial=False):
    """Kaiser-window low-pass taps (float32 tensor), or None for the identity (numtaps == 1).

    Separable: scipy.signal.firwin(numtaps, cutoff, width, fs).  Radial: jinc(2*cutoff*r) sampled on the tap grid,
    windowed by the outer product of the same Kaiser window, normalised to unit DC gain (reference :370-391)."""
    assert numtaps >= 1
    if numtaps == 1:
        return None
    if not radial:
        return torch.as_tensor(scipy.signal.firwin(numtaps=numtaps, cutoff=cutoff, width=width, fs=fs), dtype=torch.float32)
    pos = (np.arange(numtaps) - (numtaps - 1) / 2) / fs
    r = np.hypot(*np.meshgrid(pos, pos))
    taps = scipy.special.j1(2 * cutoff * (np.pi * r)) / (np.pi * r)
    window = np.kaiser(numtaps, scipy.signal.kaiser_beta(scipy.signal.kaiser_atten(numtaps, width / (fs / 2))))
    taps = taps * np.outer(window, window)
    return torch.as_tensor(taps / np.sum(taps), dtype=torch.float32)


@persistence.persistent_class
class SynthesisLayer(torch.nn.Module):
    """affine(w) -> modulated conv -> fused {bias, upsample FIR, leaky ReLU, clamp, downsample FIR}."""

    def __init__(self, w_dim, is_torgb, is_critically_sampled, use_fp16,
                 in_channels, out_channels, in_size, out_size, in_sampling_rate, out_sampling_rate,
                 in_cutoff, out_cutoff, in_half_width, out_half_width,
                 conv_kernel=3, filter_size=6, lrelu_upsampling=2, use_radial_filters=False, conv_clamp=256,
                 magnitude_ema_beta=0.999):
        super().__init__()
        _keep(self, w_dim=w_dim, is_torgb=is_torgb, is_critically_sampled=is_critically_sampled, use_fp16=use_fp16,
              in_channels=in_channels, out_channels=out_channels, in_size=_xy(in_size), out_size=_xy(out_size),
              in_sampling_rate=in_sampling_rate, out_sampling_rate=out_sampling_rate,
              in_cutoff=in_cutoff, out_cutoff=out_cutoff, in_half_width=in_half_width, out_half_width=out_half_width,
              conv_kernel=(1 if is_torgb else conv_kernel), conv_clamp=conv_clamp, magnitude_ema_beta=magnitude_ema_beta,
              # the non-linearity runs at `lrelu_upsampling` times the faster of the two rates (ToRGB has none)
              tmp_sampling_rate=max(in_sampling_rate, out_sampling_rate) * (1 if is_torgb else lrelu_upsampling))

        self.affine = FullyConnectedLayer(w_dim, in_channels, bias_init=1)
        self.weight = torch.nn.Parameter(torch.randn([out_channels, in_channels, self.conv_kernel, self.conv_kernel]))
        self.bias = torch.nn.Parameter(torch.zeros([out_channels]))
        self.register_buffer('magnitude_ema', torch.ones([]))

        # resampling factors and FIR filters around the non-linearity
        self.up_factor = int(np.rint(self.tmp_sampling_rate / in_sampling_rate))
        assert in_sampling_rate * self.up_factor == self.tmp_sampling_rate
        self.up_taps = filter_size * self.up_factor if self.up_factor > 1 and not is_torgb else 1
        self.register_buffer('up_filter', self.design_lowpass_filter(
            numtaps=self.up_taps, cutoff=in_cutoff, width=in_half_width * 2, fs=self.tmp_sampling_rate))
        self.down_factor = int(np.rint(self.tmp_sampling_rate / out_sampling_rate))
        assert out_sampling_rate * self.down_factor == self.tmp_sampling_rate
        self.down_taps = filter_size * self.down_factor if self.down_factor > 1 and not is_torgb else 1
        self.down_radial = use_radial_filters and not is_critically_sampled
        self.register_buffer('down_filter', self.design_lowpass_filter(
            numtaps=self.down_taps, cutoff=out_cutoff, width=out_half_width * 2, fs=self.tmp_sampling_rate, radial=self.down_radial))

        # padding so that the decimated output has exactly out_size samples, centred per Appendix C.3 of the paper
        total = (self.out_size - 1) * self.down_factor + 1
        total = total - (self.in_size + self.conv_kernel - 1) * self.up_factor
        total = total + self.up_taps + self.down_taps - 2
        lo = (total + self.up_factor) // 2
        hi = total - lo
        self.padding = [int(lo[0]), int(hi[0]), int(lo[1]), int(hi[1])]

    design_lowpass_filter = staticmethod(design_lowpass_filter)

    def output_bound(self):
        """Upper bound on |output|: the clamp value times the L1 norm of the down filter (None when unclamped).
        Computed once from the filter buffer and cached; SynthesisNetwork hands it to the next layer's convolution."""
        if self.conv_clamp is None:
            return None
        f = self.down_filter
        key = None if f is None else (f.data_ptr(), f._version)
        if getattr(self, '_bound_key', 0) != key:
            l1 = 1.0 if f is None else float(f.abs().sum()) ** (2 if f.ndim == 1 else 1)
            self._bound_key, self._bound = key, float(self.conv_clamp) * max(l1, 1.0) * 1.01
        return self._bound

    def style_gain(self):
        """ToRGB styles carry the 1/sqrt(fan_in) weight gain (reference :350-352)."""
        return 1 / np.sqrt(self.in_channels * (self.conv_kernel ** 2)) if self.is_torgb else 1.0

    def styles_from_w(self, w):
        """Per-input-channel modulation for latent w."""
        styles = self.affine(w)
        if self.is_torgb:
            styles = styles * self.style_gain()
        return styles

    def compute_dtype(self, force_fp32, device_type):
        return torch.float16 if (self.use_fp16 and not force_fp32 and device_type == 'cuda') else torch.float32

    def conv_spec(self, styles, n, force_fp32, input_gain=None):
        """This layer's entry for `modulated_conv.prepare_batch`: everything its convolution's prep work depends on."""
        return dict(w=self.weight, s=styles, demodulate=not self.is_torgb, padding=self.conv_kernel - 1,
                    input_gain=self.magnitude_ema.rsqrt() if input_gain is None else input_gain,
                    x_bound=getattr(self, 'input_bound', None), n=n,
                    h=int(self.in_size[1]), wd=int(self.in_size[0]), dtype=self.compute_dtype(force_fp32, 'cuda'))

    def fuses_output(self, dtype):
        """ToRGB in inference: bias and clamp -- the layer's whole filtered_lrelu (up = down = 1, no filters, gain = slope = 1,
        no padding) -- fold into the convolution's stores; the layer's output is unchanged, one pass over the image is saved."""
        return (self.is_torgb and self.up_factor == 1 and self.down_factor == 1 and self.up_filter is None and self.down_filter is None
                and not any(self.padding) and _modconv.torgb_epilogue_ok(self.weight, self.conv_kernel - 1, dtype))

    def forward(self, x, w, styles=None, noise_mode='random', force_fp32=False, update_emas=False, prepared=None, out_scale=1.0, input_gain=None):
        assert noise_mode in ['random', 'const', 'none']  # kept for API compatibility; SG3 has no noise inputs
        in_w, in_h = (int(v) for v in self.in_size)
        out_w, out_h = (int(v) for v in self.out_size)
        misc.assert_shape(x, [None, self.in_channels, in_h, in_w])
        if update_emas:
            self._track_magnitude(x)
        # a prepared convolution carries the gain; with gradients recorded the backward's closed forms need its value as well
        # (`input_gain`: the network's cached rsqrt(magnitude_ema) of this layer, handed in with the batched preparation)
        if prepared is None:
            input_gain = self.magnitude_ema.rsqrt()
        elif not torch.is_grad_enabled():
            input_gain = None
        elif input_gain is None:
            input_gain = self.magnitude_ema.rsqrt()

        if styles is None:
            misc.assert_shape(w, [x.shape[0], self.w_dim])
            styles = self.styles_from_w(w)

        dtype = self.compute_dtype(force_fp32, x.device.type)
        epilogue = None
        if prepared is not None and not torch.is_grad_enabled() and self.fuses_output(dtype):
            epilogue = (self.bias, self.conv_clamp, float(out_scale))       # clamp(conv + bias) [* out_scale] inside the convolution
        elif out_scale != 1.0:
            raise RuntimeError('SynthesisLayer: out_scale rides in the fused ToRGB output stage only')
        x = modulated_conv2d(x=x.to(dtype), w=self.weight, s=styles, padding=self.conv_kernel - 1,
                             demodulate=(not self.is_torgb), input_gain=input_gain, x_bound=getattr(self, 'input_bound', None),
                             prepared=prepared, epilogue=epilogue,
                             align_rows=(x.is_cuda and not torch.is_grad_enabled()))   # consumed by filtered_lrelu, which honours strides
        if epilogue is not None:
            misc.assert_shape(x, [None, self.out_channels, out_h, out_w])
            return x
        x = filtered_lrelu.filtered_lrelu(
            x=x, fu=self.up_filter, fd=self.down_filter, b=self._bias_as(x.dtype), up=self.up_factor, down=self.down_factor,
            padding=self.padding, gain=(1 if self.is_torgb else np.sqrt(2)), slope=(1 if self.is_torgb else 0.2), clamp=self.conv_clamp)
        misc.assert_shape(x, [None, self.out_channels, out_h, out_w])
        assert x.dtype == dtype
        return x

    def _bias_as(self, dtype):
        """The bias in the activations' dtype; the float16 copy of inference is kept until the parameter changes."""
        b = self.bias
        if b.dtype == dtype or torch.is_grad_enabled():
            return b.to(dtype)
        store = _derived_of(self)
        key = (b.data_ptr(), b._version, dtype)
        c = store.get('bias')
        if c is None or c[0] != key:
            c = store['bias'] = (key, b.detach().to(dtype))
        return c[1]

    def _track_magnitude(self, x):
        """Running mean of the input's power; its rsqrt is the gain that keeps the convolution input at unit variance."""
        with torch.autograd.profiler.record_function('update_magnitude_ema'):
            power = x.detach().to(torch.float32).square().mean()
            self.magnitude_ema.copy_(power.lerp(self.magnitude_ema, self.magnitude_ema_beta))

    def extra_repr(self):
        return _summary(self, 'w_dim is_torgb', 'is_critically_sampled use_fp16', 'in_sampling_rate out_sampling_rate',
                        'in_cutoff out_cutoff', 'in_half_width out_half_width', 'in_size out_size', 'in_channels out_channels')


LayerGeometry = collections.namedtuple('LayerGeometry', [
    'index', 'is_torgb', 'is_critically_sampled', 'use_fp16', 'in_channels', 'out_channels', 'in_size', 'out_size',
    'in_sampling_rate', 'out_sampling_rate', 'in_cutoff', 'out_cutoff', 'in_half_width', 'out_half_width'])


def synthesis_schedule(img_resolution, img_channels, channel_base=32768, channel_max=512, num_layers=14, num_critical=2,
                       first_cutoff=2, first_stopband=2 ** 2.1, last_stopband_rel=2 ** 0.3, margin_size=10, num_fp16_res=4):
    """Geometric progression of cutoffs / stopbands and everything derived from it (reference :430-447, :458).

    Returns (input_spec, [LayerGeometry x (num_layers + 1)]): layer i reads the signal described by entry max(i-1, 0)
    of the progression and writes entry i; the last entry is the ToRGB layer."""
    last_cutoff = img_resolution / 2
    last_stopband = last_cutoff * last_stopband_rel
    expo = np.minimum(np.arange(num_layers + 1) / (num_layers - num_critical), 1)
    cutoffs = first_cutoff * (last_cutoff / first_cutoff) ** expo
    stopbands = first_stopband * (last_stopband / first_stopband) ** expo
    rates = np.exp2(np.ceil(np.log2(np.minimum(stopbands * 2, img_resolution))))
    half_widths = np.maximum(stopbands, rates / 2) - cutoffs
    sizes = rates + margin_size * 2
    sizes[-2:] = img_resolution
    channels = np.rint(np.minimum((channel_base / 2) / cutoffs, channel_max))
    channels[-1] = img_channels
    input_spec = dict(channels=int(channels[0]), size=int(sizes[0]), sampling_rate=rates[0], bandwidth=cutoffs[0])
    table = []
    for i in range(num_layers + 1):
        p = max(i - 1, 0)
        table.append(LayerGeometry(
            index=i, is_torgb=(i == num_layers), is_critically_sampled=(i >= num_layers - num_critical),
            use_fp16=bool(rates[i] * (2 ** num_fp16_res) > img_resolution),
            in_channels=int(channels[p]), out_channels=int(channels[i]), in_size=int(sizes[p]), out_size=int(sizes[i]),
            in_sampling_rate=int(rates[p]), out_sampling_rate=int(rates[i]), in_cutoff=cutoffs[p], out_cutoff=cutoffs[i],
            in_half_width=half_widths[p], out_half_width=half_widths[i]))
    return input_spec, table


@persistence.persistent_class
class SynthesisNetwork(torch.nn.Module):
    def __init__(self, w_dim, img_resolution, img_channels, channel_base=32768, channel_max=512, num_layers=14,
                 num_critical=2, first_cutoff=2, first_stopband=2 ** 2.1, last_stopband_rel=2 ** 0.3, margin_size=10,
                 output_scale=0.25, num_fp16_res=4, **layer_kwargs):
        super().__init__()
        _keep(self, w_dim=w_dim, num_ws=num_layers + 2, img_resolution=img_resolution, img_channels=img_channels,
              num_layers=num_layers, num_critical=num_critical, margin_size=margin_size, output_scale=output_scale,
              num_fp16_res=num_fp16_res)

        input_spec, table = synthesis_schedule(
            img_resolution, img_channels, channel_base=channel_base, channel_max=channel_max, num_layers=num_layers,
            num_critical=num_critical, first_cutoff=first_cutoff, first_stopband=first_stopband,
            last_stopband_rel=last_stopband_rel, margin_size=margin_size, num_fp16_res=num_fp16_res)
        self.input = SynthesisInput(w_dim=w_dim, **input_spec)
        self.layer_names = []
        for g in table:
            kw = g._asdict()
            kw.pop('index')
            layer = SynthesisLayer(w_dim=w_dim, **kw, **layer_kwargs)
            name = f'L{g.index}_{layer.out_size[0]}_{layer.out_channels}'
            setattr(self, name, layer)
            self.layer_names.append(name)

    batch_prep = True          # class-level switch (tests compare against per-layer preparation)

    def layers(self):
        return [getattr(self, n) for n in self.layer_names]

    def _propagate_bounds(self):
        """Each layer's input is the previous layer's clamped + low-pass filtered output (or, for the first layer, the
        Fourier features times a known matrix), so |x| has a cheap guaranteed bound."""
        prev = self.input
        for layer in self.layers():
            layer.input_bound = prev.output_bound()
            prev = layer

    def forward(self, ws, all_s=None, **layer_kwargs):
        """ws [N, num_ws, w_dim]  ->  image [N, img_channels, R, R] (fp32).
        With `all_s` (dict from W2S, possibly edited) the affine layers are bypassed: StyleSpace path."""
        self._propagate_bounds()
        layers = self.layers()
        given = ws if all_s is None else all_s['input']
        # Inference on the GPU: every layer's styles are known before the first convolution, so the weight / style preparation
        # of all convolutions is issued as one batch (two launches instead of thirty small, latency-bound ones)
        # With gradients recorded (pivotal tuning: thirty prep launches per forward otherwise) the preparation is batched the same way --
        # the styles then come from the layers' own differentiable affine modules, and each layer is handed its input gain for the backward
        grad = torch.is_grad_enabled()
        batched = (self.batch_prep and given.is_cuda and not layer_kwargs.get('update_emas', False)
                   and all(k in ('noise_mode', 'force_fp32', 'update_emas') for k in layer_kwargs))
        # modules whose own forward the merged kernels bypass keep it when somebody hooked them (hooks must keep firing)
        watched = bool(torch.nn.modules.module._global_forward_hooks) or bool(torch.nn.modules.module._global_forward_pre_hooks) or any(
            m._forward_hooks or m._forward_pre_hooks for m in [self.input, self.input.affine] + [layer.affine for layer in layers])
        t_in = None
        if all_s is None:
            misc.assert_shape(ws, [None, self.num_ws, self.w_dim])
            per_layer = ws.to(torch.float32).unbind(dim=1)
            styles = None
            if not (batched and self.input.fast_path_ok() and not watched and not grad):
                t_in = self.input.transform_params(per_layer[0])
        else:
            t_in = all_s['input']
            styles = [all_s[name] for name in self.layer_names]
        prepared = [None] * len(layers)
        x = None
        if batched:
            n = int(given.shape[0])
            fast_input = self.input.fast_path_ok() and not watched and not grad
            normalise = False
            if styles is None and (watched or grad):
                styles = [layer.styles_from_w(w) for layer, w in zip(layers, per_layer[1:])]
            if styles is None:
                # every affine layer (the input's and the 15 layers') in one launch instead of ~45 (affine_batch.py)
                ws32 = ws.to(torch.float32)
                outs = self._affine_pack()(ws32 if ws32.stride(2) == 1 else ws32.contiguous(),
                                           [self.input.affine] + [layer.affine for layer in layers], range(len(layers) + 1),
                                           [1.0] + [layer.style_gain() for layer in layers])
                styles = outs[1:]
                if fast_input:
                    t_in, normalise = outs[0], True          # the normalisation happens inside the input-transform kernel
            force_fp32 = bool(layer_kwargs.get('force_fp32', False))
            gains = self._input_gains(layers)
            specs = [layer.conv_spec(s, n, force_fp32, gains[j:j + 1]) for j, (layer, s) in enumerate(zip(layers, styles))]
            if fast_input:
                plans = _modconv.prepare_batch([self.input.mix_spec(n, given.device)] + specs)
                prepared = plans[1:]
                x = self.input.forward_inference(t_in, normalise, plans[0])
            else:
                prepared = _modconv.prepare_batch(specs)
        if x is None:
            x = self.input(None, t=t_in)
        scaled = False
        for j, layer in enumerate(layers):
            if styles is not None:
                # the network's output scale (reference :488-489) rides in the last layer's fused output stage
                # (not when somebody watches the layer's own output through a forward hook)
                hooked = bool(layer._forward_hooks) or bool(torch.nn.modules.module._global_forward_hooks)
                last = (j + 1 == len(layers) and prepared[j] is not None and self.output_scale != 1 and not hooked and not grad
                        and layer.fuses_output(layer.compute_dtype(bool(layer_kwargs.get('force_fp32', False)), 'cuda')))
                extra = dict(input_gain=gains[j:j + 1].reshape([])) if (grad and prepared[j] is not None) else {}
                x = layer(x, None, styles=styles[j], prepared=prepared[j], **extra,
                          **(dict(layer_kwargs, out_scale=self.output_scale) if last else layer_kwargs))
                scaled = last
            else:
                x = layer(x, per_layer[j + 1], **layer_kwargs)
        if self.output_scale != 1 and not scaled:
            x = x * self.output_scale
        misc.assert_shape(x, [None, self.img_channels, self.img_resolution, self.img_resolution])
        return x.to(torch.float32)

    def _affine_pack(self):
        store = _derived_of(self)
        if 'affine' not in store:
            store['affine'] = affine_batch.AffinePack()
        return store['affine']

    def _input_gains(self, layers):
        """rsqrt(magnitude_ema) of every layer as one [L] tensor, rebuilt when a buffer changes (two launches otherwise)."""
        key = tuple((layer.magnitude_ema.data_ptr(), layer.magnitude_ema._version) for layer in layers)
        store = _derived_of(self)
        c = store.get('gains')
        if c is None or c[0] != key:
            c = store['gains'] = (key, torch.stack([layer.magnitude_ema for layer in layers]).rsqrt())
        return c[1]

    def W2S(self, ws):
        """Latents -> StyleSpace: {'input': t [N,4], layer_name: styles [N, in_channels]} (reference :503-525)."""
        misc.assert_shape(ws, [None, self.num_ws, self.w_dim])
        per_layer = ws.to(torch.float32).unbind(dim=1)
        all_s = {'input': self.input.transform_params(per_layer[0])}
        for name, layer, w in zip(self.layer_names, self.layers(), per_layer[1:]):
            all_s[name] = layer.styles_from_w(w)
        return all_s

    def extra_repr(self):
        return _summary(self, 'w_dim num_ws', 'img_resolution img_channels', 'num_layers num_critical', 'margin_size num_fp16_res')



@persistence.persistent_class
class Generator(torch.nn.Module):
    def __init__(self, z_dim, c_dim, w_dim, img_resolution, img_channels, mapping_kwargs={}, **synthesis_kwargs):
        super().__init__()
        _keep(self, z_dim=z_dim, c_dim=c_dim, w_dim=w_dim, img_resolution=img_resolution, img_channels=img_channels)
        self.synthesis = SynthesisNetwork(w_dim=w_dim, img_resolution=img_resolution, img_channels=img_channels, **synthesis_kwargs)
        self.num_ws = self.synthesis.num_ws
        self.mapping = MappingNetwork(z_dim=z_dim, c_dim=c_dim, w_dim=w_dim, num_ws=self.num_ws, **mapping_kwargs)

    def forward(self, z, c, truncation_psi=1, truncation_cutoff=None, update_emas=False, **synthesis_kwargs):
        ws = self.mapping(z, c, truncation_psi=truncation_psi, truncation_cutoff=truncation_cutoff, update_emas=update_emas)
        return self.synthesis(ws, update_emas=update_emas, **synthesis_kwargs)
