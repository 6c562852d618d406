"""Plugin loader (API of reference torch_utils/custom_ops.py:58-154).

The reference JIT-compiles its CUDA plugins with `torch.utils.cpp_extension.load` at first use.  Here the three
plugins are entry points of ONE prebuilt HIP shared library (libsg3hip.so, built ahead of time for gfx950 by
`__graft_entry__.build()`), so `get_plugin` keeps its signature -- the op modules and pickled NVIDIA module
source call it unchanged -- but ignores the source lists and returns the matching tensor-level shim from
`_hip_plugins`.  A missing library raises (no silent fallback), like a failed build does in the reference.
"""
from . import _hip_plugins, _sg3abi

verbosity = 'brief'  # 'none' | 'brief' | 'full'

_cached_plugins = dict()


def get_plugin(module_name, sources=None, headers=None, source_dir=None, **build_kwargs):  # pylint: disable=unused-argument
    assert verbosity in ['none', 'brief', 'full']
    if module_name in _cached_plugins:
        return _cached_plugins[module_name]
    if verbosity != 'none':
        print(f'Setting up PyTorch plugin "{module_name}"... ', end='', flush=True)
    try:
        if module_name not in _hip_plugins.PLUGINS:
            raise RuntimeError(f'unknown plugin "{module_name}": libsg3hip provides {sorted(_hip_plugins.PLUGINS)}')
        _sg3abi.load()
        plugin = _hip_plugins.PLUGINS[module_name]
    except Exception:
        if verbosity != 'none':
            print('Failed!', flush=True)
        raise
    if verbosity != 'none':
        print('Done.', flush=True)
    _cached_plugins[module_name] = plugin
    return plugin
