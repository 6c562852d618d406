"""Source-carrying pickles for network classes (API of reference torch_utils/persistence.py).

Official StyleGAN3 `.pkl` files store, for every persistent object, a dict
    {type:'class', version:6, module_src:<python source of the defining module>, class_name, state}
and name `torch_utils.persistence._reconstruct_persistent_obj` as the unpickling function (reference
torch_utils/persistence.py:119-127, 180-203).  Unpickling therefore execs the embedded NVIDIA module source, which
imports `torch_utils.misc`, `torch_utils.persistence`, `torch_utils.ops.*` and `dnnlib` BY THOSE NAMES -- that is
what makes this package layout part of the drop-in contract: the embedded graph code then runs on the HIP kernels.
"""
import copy
import inspect
import io
import pickle
import sys
import types
import uuid

import dnnlib

_version = 6
_decorators = set()
_import_hooks = []
_module_to_src_dict = {}
_src_to_module_dict = {}


def persistent_class(orig_class):
    """Class decorator: instances remember their constructor arguments and pickle together with the source code of
    the module that defines the class (reference torch_utils/persistence.py:36-131)."""
    assert isinstance(orig_class, type)
    if is_persistent(orig_class):
        return orig_class
    assert orig_class.__module__ in sys.modules
    module = sys.modules[orig_class.__module__]
    module_src = _module_to_src(module)

    class Decorator(orig_class):
        _orig_module_src = module_src
        _orig_class_name = orig_class.__name__

        def __init__(self, *args, **kwargs):
            super().__init__(*args, **kwargs)
            self._init_args = copy.deepcopy(args)
            self._init_kwargs = copy.deepcopy(kwargs)
            assert orig_class.__name__ in module.__dict__
            _check_pickleable(self.__reduce__())

        @property
        def init_args(self):
            return copy.deepcopy(self._init_args)

        @property
        def init_kwargs(self):
            return dnnlib.EasyDict(copy.deepcopy(self._init_kwargs))

        def __reduce__(self):
            fields = list(super().__reduce__())
            fields += [None] * max(3 - len(fields), 0)
            if fields[0] is not _reconstruct_persistent_obj:
                meta = dict(type='class', version=_version, module_src=self._orig_module_src,
                            class_name=self._orig_class_name, state=fields[2])
                fields[0], fields[1], fields[2] = _reconstruct_persistent_obj, (meta,), None
            return tuple(fields)

    Decorator.__name__ = orig_class.__name__
    Decorator.__qualname__ = orig_class.__qualname__
    _decorators.add(Decorator)
    return Decorator


def is_persistent(obj):
    try:
        if obj in _decorators:
            return True
    except TypeError:
        pass
    return type(obj) in _decorators


def import_hook(hook):
    """Register hook(meta) -> meta, called for every persistent object being unpickled."""
    assert callable(hook)
    _import_hooks.append(hook)
    return hook


def _reconstruct_persistent_obj(meta):
    meta = dnnlib.EasyDict(meta)
    meta.state = dnnlib.EasyDict(meta.state)
    for hook in _import_hooks:
        meta = hook(meta)
        assert meta is not None
    assert meta.version == _version
    module = _src_to_module(meta.module_src)
    assert meta.type == 'class'
    cls = persistent_class(module.__dict__[meta.class_name])
    obj = cls.__new__(cls)
    setstate = getattr(obj, '__setstate__', None)
    if callable(setstate):
        setstate(meta.state)
    else:
        obj.__dict__.update(meta.state)
    return obj


def _module_to_src(module):
    src = _module_to_src_dict.get(module)
    if src is None:
        src = inspect.getsource(module)
        _module_to_src_dict[module] = src
        _src_to_module_dict[src] = module
    return src


def _src_to_module(src):
    module = _src_to_module_dict.get(src)
    if module is None:
        name = '_imported_module_' + uuid.uuid4().hex
        module = types.ModuleType(name)
        sys.modules[name] = module
        _module_to_src_dict[module] = src
        _src_to_module_dict[src] = module
        exec(src, module.__dict__)  # pylint: disable=exec-used
    return module


def _check_pickleable(obj):
    """Cheap structural check that obj can be pickled (containers, primitives, tensors, persistent objects,
    functions); anything else is test-pickled for real."""
    def strip(o):
        if isinstance(o, (list, tuple, set)):
            return [strip(x) for x in o]
        if isinstance(o, dict):
            return [[strip(k), strip(v)] for k, v in o.items()]
        if isinstance(o, (str, int, float, bool, bytes, bytearray)) or o is None:
            return None
        if f'{type(o).__module__}.{type(o).__name__}' in ('numpy.ndarray', 'torch.Tensor', 'torch.nn.parameter.Parameter'):
            return None
        if is_persistent(o) or inspect.isfunction(o) or inspect.isclass(o):
            return None
        return o
    with io.BytesIO() as f:
        pickle.dump(strip(obj), f)
