"""CPU: libsg3hip.so loads without a GPU, exports every symbol include/sg3_ops.h declares, and its host-side
geometry helpers agree with the reference formulas.  No compute kernels are launched here."""
import ctypes
import os
import re

import pytest

from helpers import HERE

ROOT = os.path.dirname(HERE)


def _declared_symbols():
    with open(os.path.join(ROOT, 'include', 'sg3_ops.h')) as f:
        src = f.read()
    return sorted(set(re.findall(r'SG3_API\s+[\w\*\s]+?\b(sg3_\w+)\s*\(', src)))


def test_header_symbols_exported():
    import torch  # noqa: F401
    from torch_utils import _sg3abi
    lib = _sg3abi.load()
    declared = _declared_symbols()
    assert len(declared) >= 13
    for name in declared:
        assert hasattr(lib, name), f'{name} declared in include/sg3_ops.h but not exported'
    assert sorted(n for n, _, _ in _sg3abi.EXPORTS) == declared
    assert lib.sg3_abi_version() == 1
    assert lib.sg3_device_count() >= 0


def test_struct_layout_matches_header():
    """ctypes mirrors of the parameter blocks have the field order of the C structs."""
    from torch_utils import _sg3abi
    with open(os.path.join(ROOT, 'include', 'sg3_ops.h')) as f:
        src = f.read()
    for cname, cls in (('sg3_filtered_lrelu_params', _sg3abi.FilteredLreluParams), ('sg3_filtered_lrelu_act_params', _sg3abi.FilteredLreluActParams),
                       ('sg3_upfirdn2d_params', _sg3abi.Upfirdn2dParams), ('sg3_bias_act_params', _sg3abi.BiasActParams),
                       ('sg3_modconv_params', _sg3abi.ModconvParams), ('sg3_modconv_prep_params', _sg3abi.ModconvPrepParams),
                       ('sg3_conv2d_params', _sg3abi.Conv2dParams), ('sg3_wgrad_params', _sg3abi.WgradParams),
                       ('sg3_fourier_params', _sg3abi.FourierParams), ('sg3_se_params', _sg3abi.SeParams), ('sg3_unfold_params', _sg3abi.UnfoldParams)):
        body = re.search(r'typedef struct ' + cname + r' \{(.*?)\} ' + cname + ';', src, re.S).group(1)
        body = re.sub(r'/\*.*?\*/', '', body, flags=re.S)
        names = []
        for decl in body.split(';'):
            decl = decl.strip()
            if not decl:
                continue
            parts = decl.split(',')
            first = parts[0].split()[-1]
            for nm in [first] + [q.strip() for q in parts[1:]]:
                names.append(re.sub(r'\[\d+\]|\*', '', nm))
        assert names == [n for n, _ in cls._fields_], cname


@pytest.mark.parametrize('xs,up,down,fu,fd,pad', [
    (38, 2, 2, 12, 12, (9, 8)), (38, 4, 2, 24, 12, (-6, -9)), (1046, 2, 2, 12, 12, (-11, -12)), (1024, 1, 1, 1, 1, (0, 0)),
    (26, 2, 4, 12, 24, (13, 28))])
def test_filtered_lrelu_shape(xs, up, down, fu, fd, pad):
    from torch_utils import _sg3abi
    lib = _sg3abi.load()
    out = [ctypes.c_int() for _ in range(5)]
    rc = lib.sg3_filtered_lrelu_shape(xs, xs, up, down, fu, 0, fd, 0, pad[0], pad[1], pad[0], pad[1], *[ctypes.byref(o) for o in out])
    assert rc == 0
    # python-side formula of the reference op (torch_utils/ops/filtered_lrelu.py:142-143)
    ow = (xs * up + (pad[0] + pad[1]) - (fu - 1) - (fd - 1) + (down - 1)) // down
    assert out[0].value == ow and out[1].value == ow
    sw_active = ow * down - (down - 1) + fd - 1
    assert out[2].value == sw_active                      # sign rows
    assert out[3].value == ((sw_active + 15) & ~15) >> 2  # sign row pitch in bytes
    assert out[4].value == (sw_active + 3) >> 2


def test_shape_errors_and_kernel_table():
    from torch_utils import _sg3abi
    lib = _sg3abi.load()
    o = ctypes.c_int()
    # upsampled buffer smaller than the down filter -> reference raises; here SG3_BAD_ARG + message
    rc = lib.sg3_filtered_lrelu_shape(2, 2, 1, 2, 1, 0, 12, 0, 0, 0, 0, 0, ctypes.byref(o), ctypes.byref(o), None, None, None)
    assert rc == _sg3abi.SG3_BAD_ARG and b'downsampling filter' in lib.sg3_last_error()
    assert lib.sg3_filtered_lrelu_has_kernel(2, 2, 12, 0, 12, 0) == 1      # config-T up2/down2
    assert lib.sg3_filtered_lrelu_has_kernel(4, 2, 24, 0, 12, 0) == 1      # config-T up4/down2
    assert lib.sg3_filtered_lrelu_has_kernel(1, 1, 1, 1, 1, 1) == 1        # ToRGB
    assert lib.sg3_filtered_lrelu_has_kernel(2, 2, 12, 0, 12, 12) == 1     # config-R radial 12x12 down filter
    assert lib.sg3_filtered_lrelu_has_kernel(2, 2, 12, 12, 12, 0) == 1     # adjoint of the radial up-2 layers (12x12 up filter)
    assert lib.sg3_filtered_lrelu_has_kernel(2, 4, 12, 12, 24, 0) == 1     # adjoint of the radial up-4 layers
    assert lib.sg3_filtered_lrelu_has_kernel(4, 2, 24, 24, 12, 0) == 0     # any other 2-D up filter -> generic composition
    assert lib.sg3_filtered_lrelu_has_kernel(2, 4, 12, 0, 24, 0) == 1      # adjoint of the up-4 layers (sign-read calls)
    assert lib.sg3_modconv_packed_floats(512, 512, 3, _sg3abi.SG3_CONV_FP32) == 512 * 64 * 9 * 8
    assert lib.sg3_modconv_packed_floats(81, 128, 3, _sg3abi.SG3_CONV_FP32) == 81 * 16 * 9 * 8
    assert lib.sg3_modconv_packed_floats(3, 32, 1, _sg3abi.SG3_CONV_FP32) == 3 * 2 * 16
    assert lib.sg3_modconv_packed_floats(81, 128, 3, _sg3abi.SG3_CONV_F16X3) == 81 * 8 * 9 * 16     # hi|lo halfs
    assert lib.sg3_modconv_packed_floats(645, 406, 1, _sg3abi.SG3_CONV_F16X3) == 645 * 26 * 16           # 1x1: chunk count padded to even
    assert lib.sg3_modconv_packed_floats(81, 128, 3, _sg3abi.SG3_CONV_F16) == 81 * 8 * 9 * 16       # same packing as f16x3


def test_gpu_tensor_without_library_raises(monkeypatch, tmp_path):
    """The product never falls back silently: a missing library is a RuntimeError."""
    from torch_utils import _sg3abi
    monkeypatch.setattr(_sg3abi, '_lib', None)
    monkeypatch.setattr(_sg3abi, 'LIB_PATH', str(tmp_path / 'nope.so'))
    with pytest.raises(RuntimeError, match='libsg3hip.so not found'):
        _sg3abi.load()
    from torch_utils import custom_ops
    monkeypatch.setattr(custom_ops, '_cached_plugins', {})
    monkeypatch.setattr(custom_ops, 'verbosity', 'none')
    with pytest.raises(RuntimeError):
        custom_ops.get_plugin('filtered_lrelu_plugin', sources=[])
    with pytest.raises(RuntimeError, match='unknown plugin'):
        monkeypatch.setattr(_sg3abi, 'LIB_PATH', os.path.join(ROOT, 'stylegan3-editing_amd', 'lib', 'libsg3hip.so'))
        custom_ops.get_plugin('no_such_plugin', sources=[])
