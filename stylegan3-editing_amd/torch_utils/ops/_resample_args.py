"""Argument normalisation shared by `upfirdn2d` and `filtered_lrelu`: integer pairs, four-sided padding and the
extent of a FIR filter tensor.  Conventions are those of the reference ops (torch_utils/ops/upfirdn2d.py:27-59)."""
import numbers

import torch


def _ints(values, what):
    out = []
    for v in values:
        if isinstance(v, bool) or not isinstance(v, numbers.Integral):
            raise AssertionError(f'{what} must be made of integers, got {v!r}')
        out.append(int(v))
    return out


def xy_factor(value):
    """int | [x, y]  ->  (x, y), both >= 1."""
    pair = _ints([value, value] if isinstance(value, numbers.Integral) else list(value), 'scaling')
    assert len(pair) == 2 and min(pair) >= 1
    return pair[0], pair[1]


def four_sided(value):
    """int | [x, y] | [x0, x1, y0, y1]  ->  (x0, x1, y0, y1); negative entries mean cropping."""
    sides = _ints([value] * 2 if isinstance(value, numbers.Integral) else list(value), 'padding')
    if len(sides) == 2:
        sides = [sides[0], sides[0], sides[1], sides[1]]
    assert len(sides) == 4
    return tuple(sides)


def fir_extent(f):
    """(width, height) of a filter: (1, 1) for None, (n, n) for separable 1-D taps."""
    if f is None:
        return 1, 1
    assert isinstance(f, torch.Tensor) and f.ndim in (1, 2) and f.numel() > 0
    return int(f.shape[-1]), int(f.shape[0])
