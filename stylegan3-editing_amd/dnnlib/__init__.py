"""Minimal `dnnlib` namespace: the pieces of the reference's dnnlib that the synthesis hot path and
official `.pkl` pickles touch (reference dnnlib/util.py:40-56 EasyDict; :396 class lookup helpers)."""
from .util import EasyDict, call_func_by_name, construct_class_by_name, get_obj_by_name  # noqa: F401
