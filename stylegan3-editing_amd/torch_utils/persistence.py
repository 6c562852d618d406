"""Source-carrying pickles for network classes (API of reference torch_utils/persistence.py).

Official StyleGAN3 `.pkl` files store, for every persistent object, a dict
    {type:'class', version:6, module_src:<python source of the defining module>, class_name, state}
and name `torch_utils.persistence._reconstruct_persistent_obj` as the unpickling function (reference
torch_utils/persistence.py:119-127, 180-203).  Unpickling therefore execs the embedded NVIDIA module source, which
imports `torch_utils.misc`, `torch_utils.persistence`, `torch_utils.ops.*` and `dnnlib` BY THOSE NAMES -- that is
what makes this package layout part of the drop-in contract: the embedded graph code then runs on the HIP kernels.

Layout of this module: `_Sources` is the two-way registry module <-> source text; `_PersistentMixin` carries the
behaviour (remembered constructor arguments, `__reduce__` in the wire format above) and `persistent_class` derives
`type(name, (_PersistentMixin, cls), ...)` from the decorated class.
"""
import copy
import inspect
import io
import pickle
import sys
import types
import uuid

import dnnlib

_version = 6            # wire-format version written into / required from every record
_import_hooks = []      # callables meta -> meta applied while unpickling


class _Sources:
    """Two-way map between live modules and their source text; unseen source text is exec'd into a fresh module."""

    def __init__(self):
        self.text_of = {}
        self.module_of = {}

    def _bind(self, module, text):
        self.text_of[module] = text
        self.module_of[text] = module

    def text(self, module):
        if module not in self.text_of:
            self._bind(module, inspect.getsource(module))
        return self.text_of[module]

    def module(self, text):
        if text not in self.module_of:
            mod = types.ModuleType('_imported_module_' + uuid.uuid4().hex)
            sys.modules[mod.__name__] = mod
            self._bind(mod, text)
            exec(text, mod.__dict__)  # pylint: disable=exec-used
        return self.module_of[text]


_sources = _Sources()
_module_to_src = _sources.text          # names used by the reference's legacy loader
_src_to_module = _sources.module
_derived = set()                        # every class produced by `persistent_class`


class _PersistentMixin:
    _orig_module_src = None
    _orig_class_name = None

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self._init_args = copy.deepcopy(args)
        self._init_kwargs = copy.deepcopy(kwargs)
        _check_pickleable(self.__reduce__())

    @property
    def init_args(self):
        return copy.deepcopy(self._init_args)

    @property
    def init_kwargs(self):
        return dnnlib.EasyDict(copy.deepcopy(self._init_kwargs))

    def __reduce__(self):
        func, args, state, *rest = (*super().__reduce__(), None, None)[:5]
        if func is _reconstruct_persistent_obj:         # a base class is persistent too: already in wire format
            return (func, args, state, *rest)
        record = dict(type='class', version=_version, module_src=self._orig_module_src,
                      class_name=self._orig_class_name, state=state)
        return (_reconstruct_persistent_obj, (record,), None, *rest)


def persistent_class(orig_class):
    """Class decorator: instances remember their constructor arguments and pickle together with the source code of
    the module that defines the class (reference torch_utils/persistence.py:36-131)."""
    assert isinstance(orig_class, type)
    if is_persistent(orig_class):
        return orig_class
    home = sys.modules[orig_class.__module__]
    derived = type(orig_class.__name__, (_PersistentMixin, orig_class), dict(
        _orig_module_src=_sources.text(home), _orig_class_name=orig_class.__name__,
        __module__=orig_class.__module__, __qualname__=orig_class.__qualname__, __doc__=orig_class.__doc__))
    _derived.add(derived)
    return derived


def is_persistent(obj):
    """True for a class made by `persistent_class` and for instances of one."""
    try:
        if obj in _derived:
            return True
    except TypeError:       # unhashable instance
        pass
    return type(obj) in _derived


def import_hook(hook):
    """Register hook(meta) -> meta, called for every persistent object being unpickled."""
    assert callable(hook)
    _import_hooks.append(hook)
    return hook


def _reconstruct_persistent_obj(meta):
    """Unpickling entry point named inside the pickles: rebuild the class from its source, then restore the state."""
    meta = dnnlib.EasyDict(meta)
    meta.state = dnnlib.EasyDict(meta.state)
    for hook in _import_hooks:
        meta = hook(meta)
        assert meta is not None
    assert meta.version == _version and meta.type == 'class'
    cls = persistent_class(getattr(_sources.module(meta.module_src), meta.class_name))
    obj = cls.__new__(cls)
    restore = getattr(obj, '__setstate__', None)
    if callable(restore):
        restore(meta.state)
    else:
        vars(obj).update(meta.state)
    return obj


_PLAIN = (str, int, float, bool, bytes, bytearray, type(None))
_ARRAYS = ('numpy.ndarray', 'torch.Tensor', 'torch.nn.parameter.Parameter')


def _check_pickleable(obj):
    """Fail early (at construction) when something in `obj` cannot be pickled.  Containers are walked; primitives,
    arrays / tensors, persistent objects, functions and classes are accepted as they are; whatever remains is
    test-pickled for real."""
    def residue(o):
        if isinstance(o, dict):
            return [[residue(k), residue(v)] for k, v in o.items()]
        if isinstance(o, (list, tuple, set)):
            return [residue(v) for v in o]
        kind = f'{type(o).__module__}.{type(o).__name__}'
        if isinstance(o, _PLAIN) or kind in _ARRAYS or is_persistent(o) or inspect.isfunction(o) or inspect.isclass(o):
            return None
        return o
    pickle.dump(residue(obj), io.BytesIO())
