"""Helper functions used by the StyleGAN3 graph (API of reference torch_utils/misc.py).

Provided: constant (:24-44), nan_to_num (:49-60), symbolic_assert (:65-68), suppress_tracer_warnings (:73-79),
assert_shape (:84-97), profiled_function (:102-107), params_and_buffers / named_params_and_buffers /
copy_params_and_buffers (:149-166), print_module_summary (:198-268, simplified text table).
Training-only pieces (InfiniteSampler, ddp_sync, check_ddp_consistency) belong to the SetGAN trainer and are out
of scope.
"""
import contextlib
import warnings

import numpy as np
import torch

_constant_cache = {}


def constant(value, shape=None, dtype=None, device=None, memory_format=None):
    """Cached constant tensor (avoids a host->device copy per use)."""
    value = np.asarray(value)
    shape = None if shape is None else tuple(shape)
    dtype = torch.get_default_dtype() if dtype is None else dtype
    device = torch.device('cpu') if device is None else torch.device(device)
    memory_format = torch.contiguous_format if memory_format is None else memory_format
    key = (value.shape, value.dtype, value.tobytes(), shape, dtype, device, memory_format)
    t = _constant_cache.get(key)
    if t is None:
        t = torch.as_tensor(value.copy(), dtype=dtype, device=device)
        if shape is not None:
            t = t.expand(shape)
        t = t.contiguous(memory_format=memory_format)
        _constant_cache[key] = t
    return t


nan_to_num = torch.nan_to_num
symbolic_assert = torch._assert  # pylint: disable=protected-access


@contextlib.contextmanager
def suppress_tracer_warnings():
    """Silence torch.jit.TracerWarning inside the block."""
    flt = ('ignore', None, torch.jit.TracerWarning, None, 0)
    warnings.filters.insert(0, flt)
    try:
        yield
    finally:
        if flt in warnings.filters:
            warnings.filters.remove(flt)


def assert_shape(tensor, ref_shape):
    """Raise AssertionError unless tensor.shape matches ref_shape (None = any size)."""
    if tensor.ndim != len(ref_shape):
        raise AssertionError(f'Wrong number of dimensions: got {tensor.ndim}, expected {len(ref_shape)}')
    for idx, (size, ref) in enumerate(zip(tensor.shape, ref_shape)):
        if ref is None:
            continue
        if isinstance(ref, torch.Tensor) or isinstance(size, torch.Tensor):
            with suppress_tracer_warnings():
                symbolic_assert(torch.equal(torch.as_tensor(size), torch.as_tensor(ref)), f'Wrong size for dimension {idx}')
        elif size != ref:
            raise AssertionError(f'Wrong size for dimension {idx}: got {size}, expected {ref}')


def profiled_function(fn):
    """Decorator: run fn inside a torch profiler range named after it (shows up in rocprofv3 marker traces)."""
    def wrapper(*args, **kwargs):
        with torch.autograd.profiler.record_function(fn.__name__):
            return fn(*args, **kwargs)
    wrapper.__name__ = fn.__name__
    wrapper.__doc__ = fn.__doc__
    return wrapper


def params_and_buffers(module):
    assert isinstance(module, torch.nn.Module)
    return list(module.parameters()) + list(module.buffers())


def named_params_and_buffers(module):
    assert isinstance(module, torch.nn.Module)
    return list(module.named_parameters()) + list(module.named_buffers())


def copy_params_and_buffers(src_module, dst_module, require_all=False):
    """Copy every same-named parameter / buffer from src to dst (detached, grad flag preserved)."""
    assert isinstance(src_module, torch.nn.Module) and isinstance(dst_module, torch.nn.Module)
    src = dict(named_params_and_buffers(src_module))
    for name, tensor in named_params_and_buffers(dst_module):
        assert (name in src) or (not require_all), f'{name} missing from source module'
        if name in src:
            with torch.no_grad():
                tensor.copy_(src[name].detach())


def print_module_summary(module, inputs, max_nesting=3, skip_redundant=True):
    """Run module(*inputs) once and print a per-submodule table of parameter / buffer counts and output shapes."""
    assert isinstance(module, torch.nn.Module) and isinstance(inputs, (tuple, list))
    rows, hooks, nesting = [], [], [0]

    def pre(_m, _i):
        nesting[0] += 1

    def post(m, _i, out):
        nesting[0] -= 1
        if nesting[0] <= max_nesting:
            outs = [t for t in (list(out) if isinstance(out, (tuple, list)) else [out]) if isinstance(t, torch.Tensor)]
            rows.append((m, outs))

    for m in module.modules():
        hooks.append(m.register_forward_pre_hook(pre))
        hooks.append(m.register_forward_hook(post))
    outputs = module(*inputs)
    for h in hooks:
        h.remove()
    names = {m: n for n, m in module.named_modules()}
    seen = set()
    lines = [('Module', 'Parameters', 'Buffers', 'Output shape', 'Datatype')]
    tot_p = tot_b = 0
    for m, outs in rows:
        ps = [p for p in m.parameters() if id(p) not in seen]
        bs = [b for b in m.buffers() if id(b) not in seen]
        seen |= {id(t) for t in ps + bs}
        if skip_redundant and not ps and not bs and len(outs) == 0:
            continue
        np_, nb_ = sum(p.numel() for p in ps), sum(b.numel() for b in bs)
        tot_p += np_; tot_b += nb_
        name = '<top-level>' if m is module else names.get(m, '?')
        shape = str(list(outs[0].shape)) if outs else '-'
        dt = str(outs[0].dtype).split('.')[-1] if outs else '-'
        lines.append((name, str(np_) if np_ else '-', str(nb_) if nb_ else '-', shape, dt))
    lines.append(('Total', str(tot_p), str(tot_b), '-', '-'))
    widths = [max(len(r[i]) for r in lines) for i in range(5)]
    print()
    for r in lines:
        print('  '.join(c.ljust(w) for c, w in zip(r, widths)))
    print()
    return outputs
