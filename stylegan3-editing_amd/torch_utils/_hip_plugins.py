"""Tensor-level plugin objects with the reference's pybind signatures, backed by the libsg3hip.so C ABI.

`custom_ops.get_plugin('filtered_lrelu_plugin' | 'upfirdn2d_plugin' | 'bias_act_plugin', ...)` returns one of these,
so code written against the reference's `_plugin.<fn>(...)` calls (including module source embedded in official
pickles) runs unchanged:
  filtered_lrelu_plugin.filtered_lrelu / filtered_lrelu_act_   reference torch_utils/ops/filtered_lrelu.cpp:16-18, 213, 294-298
  upfirdn2d_plugin.upfirdn2d                                   reference torch_utils/ops/upfirdn2d.cpp:16, 102-105
  bias_act_plugin.bias_act                                     reference torch_utils/ops/bias_act.cpp:32, 94-97
Argument checks that the reference does with TORCH_CHECK are done here and raise RuntimeError, like the
reference; the shared library itself never throws.  Output / sign tensors are allocated here (caller-owned in
ABI terms) with the size formulas of the reference host code.
"""
import ctypes
import weakref

import torch

from . import _sg3abi as abi

INT_MAX = 2 ** 31 - 1
planes_per_wave_log = None      # set to a list to record sg3_filtered_lrelu_planes_per_wave of every filtered_lrelu call


def _require(cond, msg):
    if not cond:
        raise RuntimeError(msg)


_mirror_cache = {}      # id(filter tensor) -> (weakref, version, bool)


def _mirror_symmetric(f):
    """f[r][c] == f[r][W-1-c] bit for bit?  One device comparison per filter tensor and version (the filters are module
    buffers, so the same tensor comes back on every call); the entry dies with the tensor."""
    key = id(f)
    hit = _mirror_cache.get(key)
    if hit is not None and hit[0]() is f and hit[1] == f._version:
        return hit[2]
    same = bool(torch.equal(f, f.flip(-1)))
    _mirror_cache[key] = (weakref.ref(f, lambda _, k=key: _mirror_cache.pop(k, None)), f._version, same)
    return same


def _no_kernel(return_sum, return_amax=False):
    base = (torch.empty(0), torch.empty(0), -1)
    if return_sum:
        base = base + (None,)
        if return_amax:
            base = base + (None,)
    return base


class FilteredLreluPlugin:
    name = 'filtered_lrelu_plugin'

    @staticmethod
    def filtered_lrelu(x, fu, fd, b, si, up, down, px0, px1, py0, py1, sx, sy, gain, slope, clamp, flip_filters, writeSigns, return_sum=False, return_amax=False):
        """-> (y, so, return_code).  return_code -1 (with empty tensors) = no fused kernel for this configuration.
        `return_sum` (extension): also return y.sum([0,2,3]) per channel, accumulated by the kernel (-> (y, so, rc, ysum));
        with `return_amax` besides: max |y| as a one-element float32 tensor, from the same launch (-> (y, so, rc, ysum, amax); None
        when the kernel did not accumulate it)."""
        _require(x.is_cuda, 'x must reside on CUDA device')
        _require(fu.device == x.device and fd.device == x.device and b.device == x.device, 'all input tensors must reside on the same device')
        _require(fu.dtype == torch.float32 and fd.dtype == torch.float32, 'fu and fd must be float32')
        _require(b.dtype == x.dtype, 'x and b must have the same dtype')
        _require(x.dtype in (torch.float16, torch.float32), 'x and b must be float16 or float32')
        _require(x.ndim == 4, 'x must be rank 4')
        _require(x.shape[0] * x.shape[1] <= INT_MAX and x.shape[2] <= INT_MAX and x.shape[3] <= INT_MAX, 'x is too large')
        _require(x.numel() > 0, 'x is empty')
        _require(fu.ndim in (1, 2) and fd.ndim in (1, 2), 'fu and fd must be rank 1 or 2')
        _require(fu.numel() > 0, 'fu is empty')
        _require(fd.numel() > 0, 'fd is empty')
        _require(b.ndim == 1 and b.shape[0] == x.shape[1], 'b must be a vector with the same number of channels as x')
        _require(up >= 1 and down >= 1, 'up and down must be at least 1')
        lib = abi.load()
        fuW, fuH = int(fu.shape[-1]), (int(fu.shape[0]) if fu.ndim == 2 else 0)
        fdW, fdH = int(fd.shape[-1]), (int(fd.shape[0]) if fd.ndim == 2 else 0)
        if not lib.sg3_filtered_lrelu_has_kernel(int(up), int(down), fuW, fuH, fdW, fdH):
            return _no_kernel(return_sum, return_amax)

        N, C, xH, xW = (int(v) for v in x.shape)
        yH, yW, sH, sWb, swl = (ctypes.c_int() for _ in range(5))
        rc = lib.sg3_filtered_lrelu_shape(xH, xW, int(up), int(down), fuW, fuH, fdW, fdH, int(px0), int(px1), int(py0), int(py1),
                                          ctypes.byref(yH), ctypes.byref(yW), ctypes.byref(sH), ctypes.byref(sWb), ctypes.byref(swl))
        _require(rc == abi.SG3_OK, abi.last_error())
        cl = x.ndim == 4 and x.stride(1) == 1 and x.shape[1] > 1
        y = torch.empty([N, C, yH.value, yW.value], dtype=x.dtype, device=x.device,
                        memory_format=torch.channels_last if cl else torch.contiguous_format)
        so = torch.empty(0)
        s = si
        readSigns = si.numel() > 0
        sw_limit = 0
        if writeSigns:
            s = so = torch.empty([N, C, sH.value, sWb.value], dtype=torch.uint8, device=x.device)
            sw_limit = swl.value
        elif readSigns:
            sw_limit = (int(s.shape[3]) * 4 + 3) >> 2
        if readSigns or writeSigns:
            _require(s.is_contiguous(), 'signs must be contiguous')
            _require(s.dtype == torch.uint8, 'signs must be uint8')
            _require(s.device == x.device, 'signs must reside on the same device as x')
            _require(s.ndim == 4, 'signs must be rank 4')
            _require(s.shape[0] == N and s.shape[1] == C, 'signs must have same batch & channels as x')
        # fused kernels read taps with unit stride
        fu_c, fd_c = fu.contiguous(), fd.contiguous()
        p = abi.FilteredLreluParams()
        p.x, p.y, p.b = abi.ptr(x), abi.ptr(y), abi.ptr(b)
        p.s = abi.ptr(s) if (readSigns or writeSigns) else None
        p.fu, p.fd = abi.ptr(fu_c), abi.ptr(fd_c)
        p.dtype = abi.dtype_code(x.dtype)
        p.N, p.C, p.xH, p.xW, p.yH, p.yW = N, C, xH, xW, yH.value, yW.value
        p.xStride, p.yStride, p.bStride = abi.strides4(x), abi.strides4(y), int(b.stride(0))
        p.up, p.down, p.fuW, p.fuH, p.fdW, p.fdH = int(up), int(down), fuW, fuH, fdW, fdH
        p.px0, p.py0 = int(px0), int(py0)
        p.sH = int(s.shape[2]) if (readSigns or writeSigns) else 0
        p.sWbytes = int(s.shape[3]) if (readSigns or writeSigns) else 0
        p.sx, p.sy, p.swLimit = int(sx), int(sy), int(sw_limit)
        p.gain, p.slope, p.clamp = float(gain), float(slope), float(clamp)
        p.flip, p.writeSigns, p.readSigns = int(bool(flip_filters)), int(bool(writeSigns)), int(bool(readSigns))
        p.fdMirror = int(fd.ndim == 2 and fd.shape[0] > 1 and _mirror_symmetric(fd))
        partial = partial_max = None
        if return_sum:
            slots = int(lib.sg3_filtered_lrelu_sum_slots(N, C, yH.value, yW.value, int(down)))
            if slots > 0:
                partial = torch.empty([N, C, slots], dtype=torch.float32, device=x.device)
                p.ySumPartial = abi.ptr(partial)
                if return_amax:
                    partial_max = torch.empty([N, C, slots], dtype=torch.float32, device=x.device)
                    p.yAbsMaxPartial = abi.ptr(partial_max)
        if planes_per_wave_log is not None:            # tests: which form of the streaming kernel this call takes (host-only query)
            planes_per_wave_log.append(int(lib.sg3_filtered_lrelu_planes_per_wave(ctypes.byref(p))))
        with torch.cuda.device(x.device):
            rc = lib.sg3_filtered_lrelu(ctypes.byref(p), abi.stream_ptr(x.device))
        if abi.check(rc, 'sg3_filtered_lrelu', allow_no_kernel=True) == abi.SG3_NO_KERNEL:
            return _no_kernel(return_sum, return_amax)
        if return_sum:
            ysum = amax = None
            if partial is not None:                    # fold the per-workgroup values: one small launch for both
                ysum = torch.empty([C], dtype=torch.float32, device=x.device)
                amax = torch.empty([1], dtype=torch.float32, device=x.device) if partial_max is not None else None
                with torch.cuda.device(x.device):
                    abi.check(lib.sg3_filtered_lrelu_finish_partials(abi.ptr(partial), abi.ptr(partial_max) if partial_max is not None else None,
                                                                     N, C, int(partial.shape[2]), abi.ptr(ysum), abi.ptr(amax) if amax is not None else None,
                                                                     abi.stream_ptr(x.device)), 'sg3_filtered_lrelu_finish_partials')
                ysum = ysum.to(x.dtype)
            return (y, so, 0, ysum, amax) if return_amax else (y, so, 0, ysum)
        return y, so, 0

    @staticmethod
    def filtered_lrelu_act_(x, si, sx, sy, gain, slope, clamp, writeSigns):
        """In-place gain * lrelu * clamp on x; returns the written sign tensor (or an empty tensor)."""
        _require(x.is_cuda, 'x must reside on CUDA device')
        _require(x.ndim == 4, 'x must be rank 4')
        _require(x.numel() > 0, 'x is empty')
        _require(x.dtype in (torch.float16, torch.float32, torch.float64), 'x must be float16, float32 or float64')
        N, C, H, W = (int(v) for v in x.shape)
        so = torch.empty(0)
        s = si
        readSigns = si.numel() > 0
        if writeSigns:
            sw = (W + 15) & ~15
            s = so = torch.empty([N, C, H, sw >> 2], dtype=torch.uint8, device=x.device)
        if readSigns or writeSigns:
            _require(s.is_contiguous(), 'signs must be contiguous')
            _require(s.dtype == torch.uint8, 'signs must be uint8')
            _require(s.device == x.device, 'signs must reside on the same device as x')
            _require(s.ndim == 4, 'signs must be rank 4')
            _require(s.shape[0] == N and s.shape[1] == C, 'signs must have same batch & channels as x')
        p = abi.FilteredLreluActParams()
        p.x = abi.ptr(x)
        p.s = abi.ptr(s) if (readSigns or writeSigns) else None
        p.dtype = abi.dtype_code(x.dtype)
        p.N, p.C, p.H, p.W = N, C, H, W
        p.xStride = abi.strides4(x)
        p.sH = int(s.shape[2]) if (readSigns or writeSigns) else 0
        p.sW = (int(s.shape[3]) << 2) if (readSigns or writeSigns) else 0
        p.sx, p.sy = int(sx), int(sy)
        p.gain, p.slope, p.clamp = float(gain), float(slope), float(clamp)
        p.writeSigns, p.readSigns = int(bool(writeSigns)), int(bool(readSigns and not writeSigns))
        with torch.cuda.device(x.device):
            rc = abi.load().sg3_filtered_lrelu_act(ctypes.byref(p), abi.stream_ptr(x.device))
        abi.check(rc, 'sg3_filtered_lrelu_act')
        return so


class Upfirdn2dPlugin:
    name = 'upfirdn2d_plugin'

    @staticmethod
    def upfirdn2d(x, f, upx, upy, downx, downy, padx0, padx1, pady0, pady1, flip, gain):
        _require(x.is_cuda, 'x must reside on CUDA device')
        _require(f.device == x.device, 'f must reside on the same device as x')
        _require(f.dtype == torch.float32, 'f must be float32')
        _require(x.numel() <= INT_MAX, 'x is too large')
        _require(x.numel() > 0, 'x has zero size')
        _require(f.numel() > 0, 'f has zero size')
        _require(x.ndim == 4, 'x must be rank 4')
        _require(f.ndim == 2, 'f must be rank 2')
        _require(upx >= 1 and upy >= 1, 'upsampling factor must be at least 1')
        _require(downx >= 1 and downy >= 1, 'downsampling factor must be at least 1')
        _require(x.dtype in (torch.float16, torch.float32, torch.float64), 'x must be float16, float32 or float64')
        lib = abi.load()
        N, C, xH, xW = (int(v) for v in x.shape)
        fH, fW = int(f.shape[0]), int(f.shape[1])
        yH, yW = ctypes.c_int(), ctypes.c_int()
        rc = lib.sg3_upfirdn2d_shape(xH, xW, fH, fW, int(upx), int(upy), int(downx), int(downy),
                                     int(padx0), int(padx1), int(pady0), int(pady1), ctypes.byref(yH), ctypes.byref(yW))
        _require(rc == abi.SG3_OK, abi.last_error() if rc != abi.SG3_OK else '')
        cl = x.stride(1) == 1 and x.shape[1] > 1
        y = torch.empty([N, C, yH.value, yW.value], dtype=x.dtype, device=x.device,
                        memory_format=torch.channels_last if cl else torch.contiguous_format)
        _require(y.numel() <= INT_MAX, 'output is too large')
        p = abi.Upfirdn2dParams()
        p.x, p.f, p.y = abi.ptr(x), abi.ptr(f), abi.ptr(y)
        p.dtype = abi.dtype_code(x.dtype)
        p.N, p.C, p.xH, p.xW, p.yH, p.yW = N, C, xH, xW, yH.value, yW.value
        p.xStride, p.yStride = abi.strides4(x), abi.strides4(y)
        p.fH, p.fW = fH, fW
        p.fStride = (ctypes.c_int64 * 2)(int(f.stride(0)), int(f.stride(1)))
        p.upx, p.upy, p.downx, p.downy = int(upx), int(upy), int(downx), int(downy)
        p.padx0, p.pady0 = int(padx0), int(pady0)
        p.flip, p.gain = int(bool(flip)), float(gain)
        with torch.cuda.device(x.device):
            rc = lib.sg3_upfirdn2d(ctypes.byref(p), abi.stream_ptr(x.device))
        abi.check(rc, 'sg3_upfirdn2d')
        return y


def _same_layout(a, b):
    if a.ndim != b.ndim:
        return False
    for i in range(a.ndim):
        if a.shape[i] != b.shape[i]:
            return False
        if a.shape[i] >= 2 and a.stride(i) != b.stride(i):
            return False
    return True


class BiasActPlugin:
    name = 'bias_act_plugin'

    @staticmethod
    def bias_act(x, b, xref, yref, dy, grad, dim, act, alpha, gain, clamp):
        _require(x.is_cuda, 'x must reside on CUDA device')
        _require(b.numel() == 0 or (b.dtype == x.dtype and b.device == x.device), 'b must have the same dtype and device as x')
        for t, nm in ((xref, 'xref'), (yref, 'yref'), (dy, 'dy')):
            _require(t.numel() == 0 or (t.shape == x.shape and t.dtype == x.dtype and t.device == x.device),
                     f'{nm} must have the same shape, dtype, and device as x')
        _require(x.numel() <= INT_MAX, 'x is too large')
        _require(b.ndim == 1, 'b must have rank 1')
        _require(b.numel() == 0 or (0 <= dim < x.ndim), 'dim is out of bounds')
        _require(b.numel() == 0 or b.numel() == x.shape[dim], 'b has wrong number of elements')
        _require(grad >= 0, 'grad must be non-negative')
        _require(_dense_non_overlapping(x), 'x must be non-overlapping and dense')
        _require(b.is_contiguous(), 'b must be contiguous')
        for t, nm in ((xref, 'xref'), (yref, 'yref'), (dy, 'dy')):
            _require(t.numel() == 0 or _same_layout(t, x), f'{nm} must have the same layout as x')
        _require(x.dtype in (torch.float16, torch.float32, torch.float64), 'no kernel found for the specified dtype')
        _require(1 <= act <= 9, 'no CUDA kernel found for the specified activation func')
        y = torch.empty_like(x)
        _require(_same_layout(y, x), 'y must have the same layout as x')
        if x.numel() == 0:
            return y
        p = abi.BiasActParams()
        p.x, p.b, p.xref, p.yref, p.dy, p.y = abi.ptr(x), abi.ptr(b), abi.ptr(xref), abi.ptr(yref), abi.ptr(dy), abi.ptr(y)
        p.dtype = abi.dtype_code(x.dtype)
        p.grad, p.act = int(grad), int(act)
        p.alpha, p.gain, p.clamp = float(alpha), float(gain), float(clamp)
        p.sizeX, p.sizeB = int(x.numel()), int(b.numel())
        p.stepB = int(x.stride(dim)) if b.numel() else 1
        with torch.cuda.device(x.device):
            rc = abi.load().sg3_bias_act(ctypes.byref(p), abi.stream_ptr(x.device))
        abi.check(rc, 'sg3_bias_act')
        return y


def _dense_non_overlapping(t):
    """True when t's elements tile a contiguous block of memory exactly once (any dimension order)."""
    if t.numel() <= 1:
        return True
    dims = sorted((d for d in range(t.ndim) if t.shape[d] > 1), key=lambda d: t.stride(d))
    expect = 1
    for d in dims:
        if t.stride(d) != expect:
            return False
        expect *= t.shape[d]
    return True


PLUGINS = {p.name: p for p in (FilteredLreluPlugin, Upfirdn2dPlugin, BiasActPlugin)}
