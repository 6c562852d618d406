"""Modulated convolution on the MI355X matrix cores.

This is the op behind `modulated_conv2d()` of the generator graph (reference
models/stylegan3/networks_stylegan3.py:24-63).  The reference folds style, demodulation and input gain into
per-sample WEIGHTS [N*O, I, k, k] and calls one grouped cuDNN convolution (:59-62, via
torch_utils/ops/conv2d_gradfix.py:36-39).  Here the same result is produced by two launches of libsg3hip.so:

  1. sg3_modulated_conv2d_prep: normalised weights wn (once per call), per-sample input scales
     sIn[n,i] = s'[n,i] * input_gain and demodulation coefficients dcoef[n,o];
  2. sg3_modulated_conv2d: out[n,o] = dcoef[n,o] * conv(x[n] * sIn[n], wn) as ONE implicit GEMM for the whole
     batch (M = O, N = pixels, K = I*k*k) on fp32-exact MFMA -- the weights are shared by every sample instead
     of being expanded N-fold.

Moving the modulation from the weights to the activations is algebraically identical (the products are
re-associated; differences are fp32 rounding, ~1e-7 relative).  Backward (needed by PTI) re-expresses the op with
differentiable torch ops and lets autograd differentiate that composite, so first and higher order gradients
w.r.t. x, w and s are available.
"""
import ctypes
import os
import weakref

import torch

from .. import _sg3abi as abi
from .. import misc
from . import known_amax

# Arithmetic of the implicit GEMM when the caller supplies a bound on |x| (`x_bound`):
#   'f16x3' : fp16 hi/lo operand split, three fp16 MFMAs per K step, fp32 accumulation (fp32-equivalent: every
#             retained product is exact, the dropped lo*lo term is 2^-22 relative) at 5.3x the fp32 MFMA rate;
#   'fp32'  : v_mfma_f32_32x32x2_f32, exact fp32 products.
# Calls without a bound always use 'fp32'.  fp16 tensors (the reference's mixed-precision layers) take the plain fp16
# form: operands rounded to fp16 once, one MFMA per K step, fp32 accumulation -- what an fp16 cuDNN convolution does.
precision = 'f16x3'
# Transform-domain form of the split-precision 3x3 kernel (SG3_CONV_F16X3_F23, csrc/sg3_modconv_f23.hip: Winograd F(2,3) along x,
# two thirds of the matrix instructions).  'auto': layers whose channel counts fill its 64-channel x 16-channel tiles;
# 'on': every call the kernel supports; 'off': never.  SG3_CONV_F23 in the environment sets the initial value.
f23 = {'0': 'off', '1': 'on'}.get(os.environ.get('SG3_CONV_F23', ''), 'auto')
# fp16 tensors (SG3_CONV_F16_F23): with a third of the matrix instructions per K step the transform-domain kernel's per-chunk
# overhead weighs more, so it takes over from the direct fp16 kernel at larger I x O only (measured, see _f23_wanted)
_F23_FP16_MIN_IO = 4000
# `align_rows` requests are honoured unless SG3_CONV_DENSE_ROWS=1 (A/B timing of the padded row pitch)
_ALIGN_ROWS = os.environ.get('SG3_CONV_DENSE_ROWS', '0') != '1'


def _composite(x, w, s, demodulate, padding, input_gain):
    """The reference formulation with plain torch ops (differentiable; also the CPU / impl='ref' path)."""
    n = int(x.shape[0])
    o, i, kh, kw = w.shape
    if demodulate:
        w = w * w.square().mean([1, 2, 3], keepdim=True).rsqrt()
        s = s * s.square().mean().rsqrt()
    w = w.unsqueeze(0) * s.unsqueeze(1).unsqueeze(3).unsqueeze(4)           # [N,O,I,k,k]
    if demodulate:
        w = w * (w.square().sum(dim=[2, 3, 4]) + 1e-8).rsqrt().unsqueeze(2).unsqueeze(3).unsqueeze(4)
    if input_gain is not None:
        w = w * input_gain.expand(n, i).unsqueeze(1).unsqueeze(3).unsqueeze(4)
    y = torch.nn.functional.conv2d(x.reshape(1, -1, *x.shape[2:]), w.reshape(-1, i, kh, kw).to(x.dtype), padding=padding, groups=n)
    return y.reshape(n, -1, *y.shape[2:])


def _effective_weights(w, s, demodulate, input_gain, n):
    """Per-sample weights of the reference formulation (networks_stylegan3.py:39-56), [N,O,I,k,k]: a small tensor whose
    autograd graph carries the gradients of w, s and input_gain."""
    i = w.shape[1]
    if demodulate:
        w = w * w.square().mean([1, 2, 3], keepdim=True).rsqrt()
        s = s * s.square().mean().rsqrt()
    w = w.unsqueeze(0) * s.unsqueeze(1).unsqueeze(3).unsqueeze(4)
    if demodulate:
        w = w * (w.square().sum(dim=[2, 3, 4]) + 1e-8).rsqrt().unsqueeze(2).unsqueeze(3).unsqueeze(4)
    if input_gain is not None:
        w = w * input_gain.expand(n, i).unsqueeze(1).unsqueeze(3).unsqueeze(4)
    return w


def _f23_wanted(ci, co, h, wd, padding, fp16=False):
    """Heuristic of `f23 = 'auto'`: the transform-domain kernel works on 64 output channels x 16 input channels x 32 columns;
    thin layers (few K chunks per tile, HBM-bound) and narrow planes stay on the direct kernel."""
    if f23 == 'on':
        return True
    # measured at FFHQ-1024 config T, batch 8 (tools/bench_layer.py, same box, round 4 = staging on single-issue fp32): L3 328 -> 307 us
    # against round 3's kernel (direct kernel: 429), L5 791 -> 718, L6 1799 -> 1653-1713, L7 1454 -> 1263-1326, L8 1839 -> 1705,
    # L9 695 -> 627; since then the thin layers gain too: L10 (128 -> 81: 8 chunks per tile) 1527 direct -> 1398, L11 (81 -> 51) 2453 ->
    # 2293; L12 (51 -> 32: 4 chunks, half an M tile) loses, 1014 -> 1456, and the 38-column planes of L0-L2 fill 59 % of two column tiles
    ow = wd + 2 * padding - 2
    if fp16:
        return ci * co >= _F23_FP16_MIN_IO and co >= 48 and ow >= 48
    return ci * co >= 4000 and co >= 48 and ow >= 48


class _Prepared:
    """What the prep kernels produce for one (w, s) pair -- packed normalised weights, per-sample input scales and
    demodulation coefficients -- plus the settings the convolution launch must repeat."""
    __slots__ = ('wn', 'wsq', 's_in', 'dcoef', 'prec', 'key', 'params', 'keep')


# Inference keeps the packed (normalised, split, tile-ordered) weights and their per-(o, i) energies of a layer between calls:
# they depend on the weight tensor alone, and packing all layers' weights is 80 us of a 21 ms T-1024 step.  An entry is valid for
# the tensor object it was made from at the `(data_ptr, _version)` it was made at -- optimiser steps, `copy_`, `load_state_dict`
# all bump the version; writes through `.data` do not, and need `clear_weight_cache()`.  An entry is dropped when its weight
# tensor is collected (weakref.finalize), so deleted generators do not leave their packed weights resident.
_packed_weights = {}          # id(weight) -> (weakref to the weight, key, wn, wsq)


def clear_weight_cache():
    _packed_weights.clear()


def cached_weight_tensors():
    """The tensors the cache holds right now: a captured graph reads them, so its owner keeps these references (a later
    `clear_weight_cache()` or re-pack must not hand their memory back to the allocator under the graph)."""
    return [t for ent in _packed_weights.values() for t in ent[2:]]


def _cached_packed_weights(w, key):
    ent = _packed_weights.get(id(w))
    if ent is not None and ent[0]() is w and ent[1] == key:
        return ent[2], ent[3]
    return None


def _plan(w, s, demodulate, padding, input_gain, x_bound, x_bound_dev, n, h, wd, dtype, dev, reuse_weights=False):
    """Choose the arithmetic, allocate the prep outputs and fill the prep parameter block (nothing is launched).
    `reuse_weights`: take the packed weights from / leave them in the inference cache above."""
    co, ci, k, k2 = (int(v) for v in w.shape)
    if k != k2 or k not in (1, 3):
        raise RuntimeError(f'modulated_conv2d: unsupported weight shape {tuple(w.shape)}')
    if dtype not in (torch.float32, torch.float16):
        raise RuntimeError(f'modulated_conv2d: unsupported dtype {dtype}')
    lib = abi.load()
    w32 = w.detach().to(torch.float32).contiguous()
    s32 = s.detach().to(torch.float32).contiguous()
    gmode, gptr = 0, None
    if input_gain is not None:
        g = input_gain.detach().to(device=dev, dtype=torch.float32)
        if g.numel() == 1:
            gmode, g = 1, g.reshape(1)
        elif g.ndim <= 1 or (g.ndim == 2 and g.shape[0] == 1):
            gmode, g = 2, g.reshape(-1).contiguous()
            assert g.numel() == ci
        else:
            gmode, g = 3, g.expand(n, ci).contiguous()
        gptr = g
    if dtype == torch.float16 and x_bound is None and x_bound_dev is None:
        x_bound = 65504.0                                   # the dtype's own range
    bounded = x_bound_dev is not None or (x_bound is not None and x_bound > 0)
    # 1x1: ToRGB (O <= 4) is HBM-bound and has its own kernel; the GEMM kernel loads pixel pairs (even plane size)
    split = precision == 'f16x3' and (k == 3 or (padding == 0 and co > 4 and (h * wd) % 2 == 0)) and bounded
    prec = (abi.SG3_CONV_F16 if dtype == torch.float16 else abi.SG3_CONV_F16X3) if split else abi.SG3_CONV_FP32
    if prec == abi.SG3_CONV_F16X3 and dtype == torch.float32 and k == 3 and f23 != 'off' and _f23_wanted(ci, co, h, wd, int(padding)) \
            and lib.sg3_modconv_f23_supported(abi.SG3_F32, ci, co, h, wd, k, int(padding), 0):
        prec = abi.SG3_CONV_F16X3_F23
    elif prec == abi.SG3_CONV_F16 and dtype == torch.float16 and k == 3 and f23 != 'off' and _f23_wanted(ci, co, h, wd, int(padding), fp16=True) \
            and lib.sg3_modconv_f23_supported(abi.SG3_F16, ci, co, h, wd, k, int(padding), 0):
        prec = abi.SG3_CONV_F16_F23                         # the reference's use_fp16 layers: same transform domain, one product per K step
    pr = _Prepared()
    pr.prec = prec
    pr.key = (n, ci, co, k, h, wd, int(padding), dtype)
    wkey = (w.data_ptr(), w._version, tuple(w.shape), w.dtype, prec, bool(demodulate), str(dev))
    cached = _cached_packed_weights(w, wkey) if reuse_weights else None
    if cached is not None:
        pr.wn, pr.wsq = cached
    else:
        pr.wn = torch.empty([int(lib.sg3_modconv_packed_floats(co, ci, k, prec))], dtype=torch.float32, device=dev)
        pr.wsq = torch.empty([co, ci], dtype=torch.float32, device=dev)
        if reuse_weights:
            if id(w) not in _packed_weights:
                # the entry (and with it the packed copies on the GPU) goes when the weight tensor does
                weakref.finalize(w, _packed_weights.pop, id(w), None)
            _packed_weights[id(w)] = (weakref.ref(w), wkey, pr.wn, pr.wsq)
    pr.s_in = torch.empty([n, ci], dtype=torch.float32, device=dev)
    pr.dcoef = torch.empty([n, co], dtype=torch.float32, device=dev) if (demodulate or split) else None
    pr.keep = (w32, s32, gptr, x_bound_dev)                 # inputs of the prep launch stay alive with its outputs
    pp = abi.ModconvPrepParams()
    pp.w, pp.s, pp.wPacked, pp.wsq, pp.sIn, pp.dcoef = abi.ptr(w32), abi.ptr(s32), abi.ptr(pr.wn), abi.ptr(pr.wsq), abi.ptr(pr.s_in), abi.ptr(pr.dcoef)
    pp.inputGain, pp.inputGainMode = abi.ptr(gptr), gmode
    pp.N, pp.I, pp.O, pp.k, pp.demodulate = n, ci, co, k, int(bool(demodulate))
    pp.precision = prec
    pp.xBound = float(x_bound) if (split and x_bound_dev is None) else 0.0
    pp.xBoundDev = abi.ptr(x_bound_dev) if (split and x_bound_dev is not None) else None
    pp.reuseWeights = int(cached is not None)
    pr.params = pp
    return pr


def prepare_batch(specs):
    """Prep work of several modulated convolutions in two launches (sg3_modulated_conv2d_prep_batch).  `specs`: dicts with
    w, s, demodulate, padding, input_gain, x_bound, n, h, wd, dtype -- the layers of one synthesis pass, whose styles are
    all known before the first convolution.  Returns one `_Prepared` per spec, to be handed to `modulated_conv2d(prepared=)`."""
    if not specs:
        return []
    lib = abi.load()
    dev = specs[0]['w'].device
    plans = [_plan(sp['w'], sp['s'], sp['demodulate'], sp['padding'], sp.get('input_gain'), sp.get('x_bound'), None,
                   int(sp['n']), int(sp['h']), int(sp['wd']), sp['dtype'], dev, reuse_weights=not torch.is_grad_enabled())
             for sp in specs]
    block = (abi.ModconvPrepParams * len(plans))(*[pl.params for pl in plans])
    with torch.cuda.device(dev):
        abi.check(lib.sg3_modulated_conv2d_prep_batch(block, len(plans), abi.stream_ptr(dev)), 'sg3_modulated_conv2d_prep_batch')
    return plans


def _launch(x, w, s, demodulate, padding, input_gain, x_bound=None, x_bound_dev=None, out_scale=None, prepared=None, epilogue=None,
            align_rows=False):
    """prep + implicit-GEMM kernels.  Returns (out, sIn [N,I], dcoef [N,O] or None): the two per-sample scale vectors are
    what the data-gradient pass needs.  `x_bound_dev`: one-element device tensor holding the bound; `out_scale` [N,O]: extra
    per-sample output-channel scale folded into the epilogue coefficient; `prepared`: the outcome of `prepare_batch` for
    exactly this call (then no prep kernel is launched here); `epilogue` = (bias [O] float32, clamp or None, scale): the ToRGB
    kernel's fused  clamp(out + bias) * scale  (see `torgb_epilogue_ok`); `align_rows`: give the output a row pitch that is a
    multiple of 128 bytes and return the [..., :outW] view (3x3 split-precision / fp16 kernels: see `sg3_modconv_params.outRowStride`)."""
    n, ci, h, wd = (int(v) for v in x.shape)
    co, ci2, k, _ = (int(v) for v in w.shape)
    if ci != ci2:
        raise RuntimeError(f'modulated_conv2d: x has {ci} channels, w expects {ci2}')
    lib = abi.load()
    dev = x.device
    x = x.contiguous()
    pr = prepared
    if pr is not None and pr.key != (n, ci, co, k, h, wd, int(padding), x.dtype):
        raise RuntimeError(f'modulated_conv2d: prepared for {pr.key}, called with {(n, ci, co, k, h, wd, int(padding), x.dtype)}')
    stream = abi.stream_ptr(dev)
    with torch.cuda.device(dev):
        if pr is None:
            pr = _plan(w, s, demodulate, padding, input_gain, x_bound, x_bound_dev, n, h, wd, x.dtype, dev)
            abi.check(lib.sg3_modulated_conv2d_prep(ctypes.byref(pr.params), stream), 'sg3_modulated_conv2d_prep')
        oh, ow = h + 2 * padding - k + 1, wd + 2 * padding - k + 1
        pitch = ow
        if align_rows and _ALIGN_ROWS and k == 3 and pr.prec != abi.SG3_CONV_FP32 and ow >= 128:
            per_line = 128 // x.element_size()
            pitch = (ow + per_line - 1) // per_line * per_line
        full = torch.empty([n, co, oh, pitch], dtype=x.dtype, device=dev)
        out = full if pitch == ow else full[..., :ow]
        coef = pr.dcoef
        if out_scale is not None:
            coef = out_scale.to(torch.float32).contiguous() if pr.dcoef is None else pr.dcoef * out_scale
        cp = abi.ModconvParams()
        cp.x, cp.wPacked, cp.sIn, cp.dcoef, cp.out = abi.ptr(x), abi.ptr(pr.wn), abi.ptr(pr.s_in), abi.ptr(coef), abi.ptr(out)
        cp.dtype = abi.dtype_code(x.dtype)
        cp.N, cp.I, cp.O, cp.H, cp.W, cp.k, cp.pad = n, ci, co, h, wd, k, int(padding)
        cp.precision = pr.prec
        cp.outRowStride = pitch if pitch != ow else 0
        if epilogue is not None:
            bias, clamp, scale = epilogue
            bias = bias.detach().to(device=dev, dtype=torch.float32).contiguous()
            cp.epilogueBias, cp.epilogueClamp, cp.epilogueScale = abi.ptr(bias), float(-1.0 if clamp is None else clamp), float(scale)
        need = int(lib.sg3_modconv_split_scratch_floats(ctypes.byref(cp)))     # > 0: a grid smaller than the chip (batch 1, small maps)
        if need > 0:
            scratch = torch.empty([need], dtype=torch.float32, device=dev)
            cp.splitScratch, cp.splitScratchFloats = abi.ptr(scratch), need
        abi.check(lib.sg3_modulated_conv2d(ctypes.byref(cp), stream), 'sg3_modulated_conv2d')
    return out, pr.s_in, pr.dcoef


def _data_gradient(dy, w, s_in, dcoef, demodulate, padding, dy_amax=None):
    """dx of the modulated convolution on the same implicit-GEMM kernels: with out = d * conv(wn, x * sIn),
    dx = sIn * conv(flip(wn)^T, dy * d) with padding k-1-pad -- the forward kernel with the roles of the two per-sample
    scale vectors exchanged.  The bound the split-precision path needs is max |dy|, taken on the device."""
    n, co = int(dy.shape[0]), int(w.shape[0])
    k = int(w.shape[2])
    wt = _transposed_weights(w, demodulate)                                   # [I,O,k,k]: normalised, transposed, flipped
    mod = dcoef if dcoef is not None else torch.ones([n, co], dtype=torch.float32, device=dy.device)
    bound = dy_amax if dy_amax is not None else _amax(dy)
    dx, _, _ = _launch(dy, wt, mod, False, k - 1 - padding, None, x_bound_dev=bound, out_scale=s_in)
    return dx


def _transposed_weights(w, normalise):
    """wt[i,o,k-1-ky,k-1-kx] = w[o,i,ky,kx] * rsqrt(mean w[o]^2) (the pre-normalisation of networks_stylegan3.py:41-42 when
    `normalise`): the weights of the data-gradient convolution, one launch (sg3_modconv_transpose_weights) instead of the six of
    `wn = w * w.square().mean([1,2,3], keepdim=True).rsqrt(); wn.flip([2,3]).transpose(0,1).contiguous()`."""
    co, ci, k, _ = (int(v) for v in w.shape)
    w32 = w.detach().to(torch.float32).contiguous()
    wt = torch.empty([ci, co, k, k], dtype=torch.float32, device=w.device)
    with torch.cuda.device(w.device):
        abi.check(abi.load().sg3_modconv_transpose_weights(abi.ptr(w32), abi.ptr(wt), co, ci, k, int(bool(normalise)), abi.stream_ptr(w.device)),
                  'sg3_modconv_transpose_weights')
    return wt


def _amax(t):
    """max |t| as a one-element device tensor, in one reduction pass (no |t| temporary)."""
    return torch.linalg.vector_norm(t.detach().reshape(-1), ord=float('inf')).to(torch.float32).reshape(1)


_bound_scalars = {}


def _bound_scalar(value, device):
    """One-element float32 device tensor holding `value` (a layer's input bound), made once per (value, device)."""
    key = (float(value), str(device))
    t = _bound_scalars.get(key)
    if t is None:
        t = _bound_scalars[key] = torch.full([1], float(value), dtype=torch.float32, device=device)
    return t


def _weight_gradient(x, dy, k, padding, x_amax=None, dy_amax=None):
    """dW[n,o,i,ky,kx] = sum_pixels dy[n,o] * x[n,i] (shifted): the per-sample weight gradient on the matrix cores
    (csrc/sg3_wgrad.hip).  The pixel dimension is split over workgroups; the partial sums are added here."""
    lib = abi.load()
    x = x.contiguous(); dy = dy.contiguous()
    n, ci, h, w = (int(v) for v in x.shape)
    co = int(dy.shape[1])
    nb, ng = ctypes.c_int(), ctypes.c_int()
    abi.check(lib.sg3_conv2d_wgrad_splits(n, ci, co, h, w, k, int(padding), ctypes.byref(nb), ctypes.byref(ng)), 'sg3_conv2d_wgrad_splits')
    partial = torch.empty([nb.value * ng.value, n, k * k, co, ci], dtype=torch.float32, device=x.device)
    sx = x_amax if x_amax is not None else _amax(x)            # the kernel derives the power-of-two scales from the maxima
    sd = dy_amax if dy_amax is not None else _amax(dy)
    sx, sd = sx.to(torch.float32).reshape(1), sd.to(torch.float32).reshape(1)
    p = abi.WgradParams()
    p.x, p.dy, p.partial, p.scaleX, p.scaleDy = abi.ptr(x), abi.ptr(dy), abi.ptr(partial), abi.ptr(sx), abi.ptr(sd)
    p.scalesAreAmax = 1
    p.dtype = abi.dtype_code(x.dtype)
    p.N, p.I, p.O, p.H, p.W, p.k, p.pad = n, ci, co, h, w, int(k), int(padding)
    p.nBands, p.nSegGroups = nb.value, ng.value
    with torch.cuda.device(x.device):
        abi.check(lib.sg3_conv2d_wgrad(ctypes.byref(p), abi.stream_ptr(x.device)), 'sg3_conv2d_wgrad')
    return partial.sum(dim=0).permute(0, 2, 3, 1).reshape(n, co, ci, k, k)


# First-order gradients of w and s go through the closed-form kernels of csrc/sg3_modgrad.hip unless SG3_MODGRAD_AUTOGRAD=1
# (then, as for higher-order gradients, autograd differentiates the reference's weight algebra op by op: ~70 launches per layer)
_MODGRAD_KERNELS = os.environ.get('SG3_MODGRAD_AUTOGRAD', '0') != '1'


def _modulation_grads(dw_eff, w, s, input_gain, demodulate):
    """dL/dw [O,I,k,k] and dL/ds [N,I] from dL/dw_eff [N,O,I,k,k] (sg3_modulation_backward; `dw_eff` is consumed).
    `input_gain` is a constant here ([] | [I] | [N,I] or None)."""
    lib = abi.load()
    n, co, ci, k, _ = (int(v) for v in dw_eff.shape)
    dev = dw_eff.device
    g_eff = dw_eff.to(torch.float32).contiguous()
    w32 = w.detach().to(torch.float32).contiguous()
    s32 = s.detach().to(torch.float32).contiguous()
    gmode, gptr = 0, None
    if input_gain is not None:
        g = input_gain.detach().to(device=dev, dtype=torch.float32)
        if g.numel() == 1:
            gmode, gptr = 1, g.reshape(1)
        elif g.ndim <= 1 or (g.ndim == 2 and g.shape[0] == 1):
            gmode, gptr = 2, g.reshape(-1).contiguous()
        else:
            gmode, gptr = 3, g.expand(n, ci).contiguous()
    dw = torch.empty([co, ci, k, k], dtype=torch.float32, device=dev)
    ds = torch.empty([n, ci], dtype=torch.float32, device=dev)
    scratch = torch.empty([co + n * ci], dtype=torch.float32, device=dev)
    p = abi.ModgradParams()
    p.G, p.w, p.s, p.inputGain, p.inputGainMode = abi.ptr(g_eff), abi.ptr(w32), abi.ptr(s32), abi.ptr(gptr), gmode
    p.dW, p.dS, p.a, p.dSn = abi.ptr(dw), abi.ptr(ds), abi.ptr(scratch), abi.ptr(scratch[co:])
    p.N, p.O, p.I, p.T, p.demodulate = n, co, ci, k * k, int(bool(demodulate))
    with torch.cuda.device(dev):
        abi.check(lib.sg3_modulation_backward(ctypes.byref(p), abi.stream_ptr(dev)), 'sg3_modulation_backward')
    return dw, ds


class _ModulatedConv2dHip(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, s, input_gain, demodulate, padding, x_bound, prepared=None, epilogue=None, align_rows=False):  # pylint: disable=arguments-differ
        out, s_in, dcoef = _launch(x, w, s, demodulate, padding, input_gain, x_bound, prepared=prepared, epilogue=epilogue, align_rows=align_rows)
        ctx.save_for_backward(x, w, s, input_gain if input_gain is not None else torch.empty(0), s_in,
                              dcoef if dcoef is not None else torch.empty(0))
        ctx.cfg = (demodulate, padding, input_gain is not None)
        ctx.x_bound = x_bound
        return out

    @staticmethod
    def backward(ctx, dy):  # pylint: disable=arguments-differ
        x, w, s, g, s_in, dcoef = ctx.saved_tensors
        demodulate, padding, has_gain = ctx.cfg
        need = ctx.needs_input_grad
        out = [None] * 10
        if torch.is_grad_enabled():
            # higher-order gradients: differentiate the reference formulation itself
            ins, idx = [], []
            with torch.enable_grad():
                xd = x.detach().requires_grad_(need[0]); wd = w.detach().requires_grad_(need[1]); sd = s.detach().requires_grad_(need[2])
                gd = g.detach().requires_grad_(need[3]) if has_gain else None
                for j, t in enumerate((xd, wd, sd, gd)):
                    if t is not None and need[j]:
                        ins.append(t); idx.append(j)
                y = _composite(xd, wd, sd, demodulate, padding, gd)
                grads = torch.autograd.grad(y, ins, dy, create_graph=True, allow_unused=True) if ins else []
            for j, gr in zip(idx, grads):
                out[j] = gr
            return tuple(out)
        dy_amax = known_amax.lookup(dy)                       # left by the adjoint filtered_lrelu launch that wrote dy, if it did
        dy = dy.contiguous()
        if dy_amax is None:
            dy_amax = _amax(dy)                               # shared by both gradient kernels' operand scaling
        n = int(x.shape[0])
        co, ci, k, _ = (int(v) for v in w.shape)
        if need[0]:
            out[0] = _data_gradient(dy, w, s_in, dcoef if dcoef.numel() else None, demodulate, padding, dy_amax=dy_amax)
        if (need[1] or need[2]) and not (has_gain and need[3]) and _MODGRAD_KERNELS:
            # gradient of the per-sample effective weights (weight-gradient kernel), then the chain rule to w and s in closed form
            x_amax = None
            if ctx.x_bound is not None and ctx.x_bound > 0:
                x_amax = _bound_scalar(ctx.x_bound, x.device)            # the layer's own bound: no pass over x
            dw_eff = _weight_gradient(x, dy, k, padding, x_amax=x_amax, dy_amax=dy_amax)
            dw, ds = _modulation_grads(dw_eff, w, s, g if has_gain else None, demodulate)
            if need[1]:
                out[1] = dw.to(w.dtype)
            if need[2]:
                out[2] = ds.to(s.dtype)
        elif need[1] or need[2] or (has_gain and need[3]):
            # the same through autograd (input_gain needs a gradient too, or the kernels are switched off): the chain rule
            # through the small [N,O,I,k,k] tensor for w, s and input_gain
            with torch.enable_grad():
                wd = w.detach().requires_grad_(need[1]); sd = s.detach().requires_grad_(need[2])
                gd = g.detach().requires_grad_(need[3]) if has_gain else None
                w_eff = _effective_weights(wd.float(), sd.float(), demodulate, gd, n)
            x_amax = None
            if ctx.x_bound is not None and ctx.x_bound > 0:
                x_amax = _bound_scalar(ctx.x_bound, x.device)            # the layer's own bound: no pass over x
            dw_eff = _weight_gradient(x, dy, k, padding, x_amax=x_amax, dy_amax=dy_amax)
            ins, idx = [], []
            for j, t in ((1, wd), (2, sd), (3, gd)):
                if t is not None and need[j]:
                    ins.append(t); idx.append(j)
            grads = torch.autograd.grad(w_eff, ins, dw_eff.to(w_eff.dtype), allow_unused=True)
            for j, gr in zip(idx, grads):
                out[j] = gr
        return tuple(out)


def torgb_epilogue_ok(w, padding, dtype):
    """True when a call with these weights runs on the ToRGB kernel (1x1, at most 4 output channels), the one kernel that can
    fuse  clamp(out + bias) * scale  into its stores.  float16 tensors (the reference's mixed-precision default) take it too:
    bias, clamp and scale are then applied to the fp32 accumulator and the result is rounded to fp16 once, where the separate
    ops round three times."""
    co, ci, k, _ = (int(v) for v in w.shape)
    return k == 1 and int(padding) == 0 and co <= 4 and ci * 16 <= 48 * 1024 and dtype in (torch.float32, torch.float16)


@misc.profiled_function
def modulated_conv2d(x, w, s, demodulate=True, padding=0, input_gain=None, impl='cuda', x_bound=None, prepared=None, epilogue=None,
                     align_rows=False):
    """x [N,I,H,W], w [O,I,k,k], s [N,I]; input_gain [], [I] or [N,I].  Returns [N,O,H+2p-k+1,W+2p-k+1] in x.dtype.
    `x_bound` (optional float): a guaranteed upper bound on |x|; enables the split-precision MFMA path (see `precision`).
    `prepared` (optional): this call's entry of `prepare_batch`, made from the same w, s, input_gain and x_bound.
    `epilogue` (optional, inference only): (bias, clamp, scale) fused into the ToRGB kernel, see `torgb_epilogue_ok`.
    `align_rows` (optional, inference only): the result may be a [..., :W'] view of a buffer whose rows start on 128-byte lines
    (faster stores for the 1046-wide layers; consumers must honour strides, as filtered_lrelu does)."""
    assert impl in ['ref', 'cuda']
    with misc.suppress_tracer_warnings():
        n = int(x.shape[0])
    o, i, kh, kw = w.shape
    misc.assert_shape(w, [o, i, kh, kw])
    misc.assert_shape(x, [n, i, None, None])
    misc.assert_shape(s, [n, i])
    if impl == 'cuda' and x.device.type == 'cuda':
        if epilogue is not None and (torch.is_grad_enabled() or not torgb_epilogue_ok(w, padding, x.dtype)):
            raise RuntimeError('modulated_conv2d: the fused epilogue is for ToRGB-shaped inference calls only')
        return _ModulatedConv2dHip.apply(x, w, s, input_gain, bool(demodulate), int(padding), x_bound, prepared, epilogue,
                                         bool(align_rows) and not torch.is_grad_enabled())
    return _composite(x, w, s, demodulate, padding, input_gain)
