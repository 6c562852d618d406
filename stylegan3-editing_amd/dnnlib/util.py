"""Attribute-style dict and by-name construction helpers.

Mirrors the behaviour of reference dnnlib/util.py:40-56 (EasyDict) and :262-306 (object lookup /
construct_class_by_name).  Networking helpers (open_url) are out of scope: there is no egress.
"""
import importlib
from typing import Any


class EasyDict(dict):
    """dict whose items are also reachable as attributes (`d.key` == `d['key']`)."""

    def __getattr__(self, name: str) -> Any:
        try:
            return self[name]
        except KeyError:
            raise AttributeError(name) from None

    def __setattr__(self, name: str, value: Any) -> None:
        self[name] = value

    def __delattr__(self, name: str) -> None:
        try:
            del self[name]
        except KeyError:
            raise AttributeError(name) from None


def get_obj_by_name(name: str) -> Any:
    """Resolve 'package.module.attr[.attr...]' to the Python object it names."""
    parts = name.split('.')
    for split in range(len(parts) - 1, 0, -1):
        mod_name, attr_path = '.'.join(parts[:split]), parts[split:]
        try:
            obj = importlib.import_module(mod_name)
        except ImportError:
            continue
        try:
            for a in attr_path:
                obj = getattr(obj, a)
            return obj
        except AttributeError:
            continue
    raise ImportError(f'cannot resolve {name!r}')


def call_func_by_name(*args, func_name: str = None, **kwargs) -> Any:
    assert func_name is not None
    fn = get_obj_by_name(func_name)
    assert callable(fn)
    return fn(*args, **kwargs)


def construct_class_by_name(*args, class_name: str = None, **kwargs) -> Any:
    return call_func_by_name(*args, func_name=class_name, **kwargs)
