"""ctypes binding of libsg3hip.so (C ABI declared in include/sg3_ops.h).

This is the only place that touches the shared library.  Tensors cross the boundary as raw device pointers +
sizes + element strides, launches go to the caller's current HIP stream.  There is NO fallback here: if the
library is missing, or a call fails, a RuntimeError is raised.
"""
import ctypes
import os

import torch

_PKG_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# SG3_LIB: another build of the same library (same-box A/B timing of kernel variants from tools/); the product loads its own lib/ copy
LIB_PATH = os.environ.get('SG3_LIB') or os.path.join(_PKG_ROOT, 'lib', 'libsg3hip.so')

SG3_OK, SG3_NO_KERNEL, SG3_BAD_ARG, SG3_HIP_ERROR = 0, -1, -2, -3
SG3_F32, SG3_F16, SG3_F64 = 0, 1, 2
SG3_CONV_FP32, SG3_CONV_F16X3, SG3_CONV_F16, SG3_CONV_F16X3_F23, SG3_CONV_F16_F23 = 0, 1, 2, 3, 4
_DTYPE = {torch.float32: SG3_F32, torch.float16: SG3_F16, torch.float64: SG3_F64}

c_i32, c_i64, c_f32, c_vp = ctypes.c_int32, ctypes.c_int64, ctypes.c_float, ctypes.c_void_p


class FilteredLreluParams(ctypes.Structure):
    _fields_ = [('x', c_vp), ('y', c_vp), ('b', c_vp), ('s', c_vp), ('fu', c_vp), ('fd', c_vp),
                ('dtype', c_i32), ('N', c_i32), ('C', c_i32), ('xH', c_i32), ('xW', c_i32), ('yH', c_i32), ('yW', c_i32),
                ('xStride', c_i64 * 4), ('yStride', c_i64 * 4), ('bStride', c_i64),
                ('up', c_i32), ('down', c_i32), ('fuW', c_i32), ('fuH', c_i32), ('fdW', c_i32), ('fdH', c_i32),
                ('px0', c_i32), ('py0', c_i32), ('sH', c_i32), ('sWbytes', c_i32), ('sx', c_i32), ('sy', c_i32),
                ('swLimit', c_i32), ('gain', c_f32), ('slope', c_f32), ('clamp', c_f32),
                ('flip', c_i32), ('writeSigns', c_i32), ('readSigns', c_i32), ('ySumPartial', c_vp), ('fdMirror', c_i32), ('yAbsMaxPartial', c_vp)]


class FilteredLreluActParams(ctypes.Structure):
    _fields_ = [('x', c_vp), ('s', c_vp), ('dtype', c_i32), ('N', c_i32), ('C', c_i32), ('H', c_i32), ('W', c_i32),
                ('xStride', c_i64 * 4), ('sH', c_i32), ('sW', c_i32), ('sx', c_i32), ('sy', c_i32),
                ('gain', c_f32), ('slope', c_f32), ('clamp', c_f32), ('writeSigns', c_i32), ('readSigns', c_i32)]


class Upfirdn2dParams(ctypes.Structure):
    _fields_ = [('x', c_vp), ('f', c_vp), ('y', c_vp), ('dtype', c_i32),
                ('N', c_i32), ('C', c_i32), ('xH', c_i32), ('xW', c_i32), ('yH', c_i32), ('yW', c_i32),
                ('xStride', c_i64 * 4), ('yStride', c_i64 * 4), ('fH', c_i32), ('fW', c_i32), ('fStride', c_i64 * 2),
                ('upx', c_i32), ('upy', c_i32), ('downx', c_i32), ('downy', c_i32), ('padx0', c_i32), ('pady0', c_i32),
                ('flip', c_i32), ('gain', c_f32)]


class BiasActParams(ctypes.Structure):
    _fields_ = [('x', c_vp), ('b', c_vp), ('xref', c_vp), ('yref', c_vp), ('dy', c_vp), ('y', c_vp),
                ('dtype', c_i32), ('grad', c_i32), ('act', c_i32), ('alpha', c_f32), ('gain', c_f32), ('clamp', c_f32),
                ('sizeX', c_i64), ('sizeB', c_i32), ('stepB', c_i32)]


class ModconvParams(ctypes.Structure):
    _fields_ = [('x', c_vp), ('wPacked', c_vp), ('sIn', c_vp), ('dcoef', c_vp), ('out', c_vp), ('dtype', c_i32),
                ('N', c_i32), ('I', c_i32), ('O', c_i32), ('H', c_i32), ('W', c_i32), ('k', c_i32), ('pad', c_i32), ('precision', c_i32),
                ('epilogueBias', c_vp), ('epilogueClamp', c_f32), ('epilogueScale', c_f32), ('outRowStride', c_i32),
                ('splitScratch', c_vp), ('splitScratchFloats', c_i64)]


class FourierParams(ctypes.Structure):
    _fields_ = [('grid', c_vp), ('freqs', c_vp), ('phases', c_vp), ('amps', c_vp), ('out', c_vp),
                ('N', c_i32), ('C', c_i32), ('H', c_i32), ('W', c_i32)]


class InputTransformParams(ctypes.Structure):
    _fields_ = [('t', c_vp), ('user', c_vp), ('freqs', c_vp), ('phases', c_vp), ('outFreqs', c_vp), ('outPhases', c_vp), ('outAmps', c_vp),
                ('N', c_i32), ('C', c_i32), ('normalise', c_i32), ('userStrideN', c_i32), ('bandwidth', c_f32), ('samplingRate', c_f32)]


class ModgradParams(ctypes.Structure):
    _fields_ = [('G', c_vp), ('w', c_vp), ('s', c_vp), ('inputGain', c_vp), ('inputGainMode', c_i32), ('dW', c_vp), ('dS', c_vp),
                ('a', c_vp), ('dSn', c_vp), ('N', c_i32), ('O', c_i32), ('I', c_i32), ('T', c_i32), ('demodulate', c_i32)]


class SeParams(ctypes.Structure):
    _fields_ = [('res', c_vp), ('shortcut', c_vp), ('scStride', c_i64 * 4), ('fc1', c_vp), ('fc2', c_vp), ('mean', c_vp), ('out', c_vp),
                ('N', c_i32), ('C', c_i32), ('H', c_i32), ('W', c_i32), ('R', c_i32)]


class UnfoldParams(ctypes.Structure):
    _fields_ = [('src', c_vp), ('srcStride', c_i64 * 5), ('dst', c_vp), ('G', c_i32), ('N', c_i32), ('C', c_i32), ('IH', c_i32), ('IW', c_i32),
                ('slope', c_f32)]


class HeadGemmParams(ctypes.Structure):
    _fields_ = [('a', c_vp), ('wPacked', c_vp), ('bias', c_vp), ('c', c_vp), ('rangeFlag', c_vp),
                ('G', c_i32), ('M', c_i32), ('K', c_i32), ('N', c_i32), ('slope', c_f32)]


class AffineBatchParams(ctypes.Structure):
    _fields_ = [('ws', c_vp), ('wsStrideN', c_i64), ('wsStrideL', c_i64), ('weight', c_vp), ('bias', c_vp), ('scale', c_vp),
                ('rowStart', c_vp), ('wsIndex', c_vp), ('out', c_vp), ('N', c_i32), ('wDim', c_i32), ('layers', c_i32), ('rows', c_i32)]


class ModconvPrepParams(ctypes.Structure):
    _fields_ = [('w', c_vp), ('s', c_vp), ('wPacked', c_vp), ('wsq', c_vp), ('sIn', c_vp), ('dcoef', c_vp),
                ('inputGain', c_vp), ('inputGainMode', c_i32),
                ('N', c_i32), ('I', c_i32), ('O', c_i32), ('k', c_i32), ('demodulate', c_i32), ('precision', c_i32), ('xBound', c_f32),
                ('xBoundDev', c_vp), ('reuseWeights', c_i32)]


class WgradParams(ctypes.Structure):
    _fields_ = [('x', c_vp), ('dy', c_vp), ('partial', c_vp), ('scaleX', c_vp), ('scaleDy', c_vp), ('dtype', c_i32),
                ('N', c_i32), ('I', c_i32), ('O', c_i32), ('H', c_i32), ('W', c_i32), ('k', c_i32), ('pad', c_i32),
                ('nBands', c_i32), ('nSegGroups', c_i32), ('scalesAreAmax', c_i32)]


class Conv2dParams(ctypes.Structure):
    _fields_ = [('x', c_vp), ('wPacked', c_vp), ('inScale', c_vp), ('inShift', c_vp), ('bias', c_vp), ('slope', c_vp), ('out', c_vp),
                ('N', c_i32), ('I', c_i32), ('O', c_i32), ('H', c_i32), ('W', c_i32), ('k', c_i32), ('stride', c_i32), ('pad', c_i32),
                ('act', c_i32), ('precision', c_i32), ('rangeFlag', c_vp)]


# every symbol include/sg3_ops.h declares: (name, restype, argtypes)
EXPORTS = [
    ('sg3_abi_version', ctypes.c_int, []),
    ('sg3_last_error', ctypes.c_char_p, []),
    ('sg3_device_count', ctypes.c_int, []),
    ('sg3_filtered_lrelu', ctypes.c_int, [ctypes.POINTER(FilteredLreluParams), c_vp]),
    ('sg3_filtered_lrelu_planes_per_wave', ctypes.c_int, [ctypes.POINTER(FilteredLreluParams)]),
    ('sg3_filtered_lrelu_has_kernel', ctypes.c_int, [ctypes.c_int] * 6),
    ('sg3_filtered_lrelu_sum_slots', ctypes.c_int, [ctypes.c_int] * 5),
    ('sg3_filtered_lrelu_finish_partials', ctypes.c_int, [c_vp, c_vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, c_vp, c_vp, c_vp]),
    ('sg3_filtered_lrelu_shape', ctypes.c_int, [ctypes.c_int] * 12 + [ctypes.POINTER(ctypes.c_int)] * 5),
    ('sg3_filtered_lrelu_act', ctypes.c_int, [ctypes.POINTER(FilteredLreluActParams), c_vp]),
    ('sg3_upfirdn2d', ctypes.c_int, [ctypes.POINTER(Upfirdn2dParams), c_vp]),
    ('sg3_upfirdn2d_shape', ctypes.c_int, [ctypes.c_int] * 12 + [ctypes.POINTER(ctypes.c_int)] * 2),
    ('sg3_bias_act', ctypes.c_int, [ctypes.POINTER(BiasActParams), c_vp]),
    ('sg3_modconv_packed_floats', ctypes.c_int64, [ctypes.c_int] * 4),
    ('sg3_modulated_conv2d', ctypes.c_int, [ctypes.POINTER(ModconvParams), c_vp]),
    ('sg3_modconv_split_scratch_floats', ctypes.c_int64, [ctypes.POINTER(ModconvParams)]),
    ('sg3_modconv_f23_supported', ctypes.c_int, [ctypes.c_int] * 8),
    ('sg3_modconv_f23_force_rows', ctypes.c_int, [ctypes.c_int]),
    ('sg3_fourier_features', ctypes.c_int, [ctypes.POINTER(FourierParams), c_vp]),
    ('sg3_input_transform', ctypes.c_int, [ctypes.POINTER(InputTransformParams), c_vp]),
    ('sg3_affine_batch', ctypes.c_int, [ctypes.POINTER(AffineBatchParams), c_vp]),
    ('sg3_se_residual', ctypes.c_int, [ctypes.POINTER(SeParams), c_vp]),
    ('sg3_unfold3x3s2', ctypes.c_int, [ctypes.POINTER(UnfoldParams), c_vp]),
    ('sg3_head_gemm_packed_halfs', ctypes.c_int64, [ctypes.c_int] * 3),
    ('sg3_head_gemm_pack', ctypes.c_int, [c_vp, c_vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, c_vp, c_vp]),
    ('sg3_head_gemm', ctypes.c_int, [ctypes.POINTER(HeadGemmParams), c_vp]),
    ('sg3_modulation_backward', ctypes.c_int, [ctypes.POINTER(ModgradParams), c_vp]),
    ('sg3_modconv_transpose_weights', ctypes.c_int, [c_vp, c_vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, c_vp]),
    ('sg3_modulated_conv2d_prep', ctypes.c_int, [ctypes.POINTER(ModconvPrepParams), c_vp]),
    ('sg3_modulated_conv2d_prep_batch', ctypes.c_int, [ctypes.POINTER(ModconvPrepParams), ctypes.c_int, c_vp]),
    ('sg3_conv2d', ctypes.c_int, [ctypes.POINTER(Conv2dParams), c_vp]),
    ('sg3_conv2d_wgrad_splits', ctypes.c_int, [ctypes.c_int] * 7 + [ctypes.POINTER(ctypes.c_int)] * 2),
    ('sg3_conv2d_wgrad', ctypes.c_int, [ctypes.POINTER(WgradParams), c_vp]),
    ('sg3_conv2d_pack', ctypes.c_int, [c_vp, c_vp, c_vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, c_vp]),
]

_lib = None
launch_count = 0      # number of successful kernel-launching ABI calls (bench / smoke assert the HIP path ran)


def load():
    """Load libsg3hip.so (after torch, so both share torch's libamdhip64.so.7).  Raises if it is not built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f'libsg3hip.so not found at {LIB_PATH}: build it with `python -c "import __graft_entry__ as g; g.build()"` '
                f'(hipcc --offload-arch=gfx950).  There is no CPU or PyTorch fallback for GPU tensors.')
        lib = ctypes.CDLL(LIB_PATH, mode=ctypes.RTLD_GLOBAL)
        for name, res, args in EXPORTS:
            fn = getattr(lib, name)          # AttributeError here = header / library mismatch
            fn.restype, fn.argtypes = res, args
        if lib.sg3_abi_version() != 1:
            raise RuntimeError(f'libsg3hip.so ABI version {lib.sg3_abi_version()} != 1')
        _lib = lib
    return _lib


def is_built():
    return os.path.exists(LIB_PATH)


def last_error():
    msg = load().sg3_last_error()
    return msg.decode() if msg else ''


def stream_ptr(device):
    return c_vp(torch.cuda.current_stream(device).cuda_stream)


def dtype_code(dtype):
    try:
        return _DTYPE[dtype]
    except KeyError:
        raise RuntimeError(f'unsupported dtype {dtype}') from None


def ptr(t):
    return c_vp(t.data_ptr()) if t is not None and t.numel() > 0 else c_vp(0)


def strides4(t):
    return (c_i64 * 4)(*[int(s) for s in t.stride()])


def check(rc, what, allow_no_kernel=False):
    """Translate an ABI return code: 0 ok; -1 passes through when allowed; everything else raises RuntimeError."""
    global launch_count
    if rc == SG3_OK:
        launch_count += 1
        return rc
    if rc == SG3_NO_KERNEL and allow_no_kernel:
        return rc
    raise RuntimeError(f'{what} failed (rc={rc}): {last_error()}')
