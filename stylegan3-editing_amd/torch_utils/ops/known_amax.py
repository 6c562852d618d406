"""max |t| of a gradient tensor, handed from the kernel that wrote it to the kernels that read it next.

The split-precision gradient kernels of `modulated_conv2d` scale their operands by a power of two derived from max |dy|
(torch_utils/ops/modulated_conv.py: _data_gradient / _weight_gradient); finding it is a pass over dy (0.42 ms per PTI step at T-1024).
dy is the `dx` the adjoint `filtered_lrelu` launch of the layer wrote a moment earlier (reference torch_utils/ops/filtered_lrelu.py:
246-269 -> networks_stylegan3.py:59-62 in backward order), and that launch can keep the running maximum of what it stores.  The
value travels as an attribute of the tensor object -- autograd hands a backward's result to the next backward as the same object --
together with the tensor's version counter and address: anything that touched the tensor in between (an in-place clip, a hook that
replaced it) makes the lookup miss, and the reader falls back to its own reduction.  A miss costs a pass, never a wrong bound."""

_ATTR = '_sg3_known_amax'
hits = 0           # lookups that found a valid value (tests / profiling)
enabled = True


def attach(t, amax):
    """amax: one-element float32 device tensor holding max |t|, valid for t as it is now."""
    setattr(t, _ATTR, (amax, t._version, t.data_ptr(), tuple(t.shape)))


def lookup(t):
    """The attached max |t| if `t` is still the tensor (object, storage, version, shape) it was attached to, else None."""
    global hits
    if not enabled:
        return None
    v = getattr(t, _ATTR, None)
    if v is None or v[1] != t._version or v[2] != t.data_ptr() or v[3] != tuple(t.shape) or v[0].device != t.device:
        return None
    hits += 1
    return v[0]
