"""CPU: torch_utils/ops/known_amax.py -- the hand-off of max |t| between the kernel that wrote a gradient tensor and the kernels that
read it next must miss whenever the tensor is no longer what the value was computed for (a miss costs a reduction pass; a stale hit
would give the split-precision gradient kernels a wrong operand bound)."""
import torch

from torch_utils.ops import known_amax


def test_lookup_hits_only_the_tensor_as_it_was():
    t = torch.randn(2, 3, 4, 5)
    a = t.abs().max().reshape(1)
    assert known_amax.lookup(t) is None                      # nothing attached
    known_amax.attach(t, a)
    h0 = known_amax.hits
    assert known_amax.lookup(t) is a and known_amax.hits == h0 + 1
    assert known_amax.lookup(t.contiguous()) is a             # contiguous() of a contiguous tensor is the same object
    assert known_amax.lookup(t[:, 1:]) is None                # a view: another object, another address
    assert known_amax.lookup(t.clone()) is None
    assert known_amax.lookup(t.view(6, 20)) is None           # same storage, another object (and shape)
    t.mul_(2.0)                                               # in-place change: version counter moved
    assert known_amax.lookup(t) is None


def test_lookup_survives_autograd_handing_the_tensor_on():
    """A custom Function's backward result reaches the next backward as the same Python object (attributes and all)."""
    seen = {}

    class Producer(torch.autograd.Function):
        @staticmethod
        def forward(ctx, x):
            return x * 2.0

        @staticmethod
        def backward(ctx, g):
            out = g * 2.0
            known_amax.attach(out, out.abs().max().reshape(1))
            return out

    class Consumer(torch.autograd.Function):
        @staticmethod
        def forward(ctx, x):
            return x + 1.0

        @staticmethod
        def backward(ctx, g):
            seen['amax'] = known_amax.lookup(g)
            seen['true'] = g.abs().max()
            return g

    x = torch.randn(3, 4, requires_grad=True)
    Producer.apply(Consumer.apply(x)).sum().backward()
    assert seen['amax'] is not None and float(seen['amax']) == float(seen['true'])


def test_switch():
    t = torch.randn(4)
    known_amax.attach(t, t.abs().max().reshape(1))
    saved, known_amax.enabled = known_amax.enabled, False
    try:
        assert known_amax.lookup(t) is None
    finally:
        known_amax.enabled = saved
    assert known_amax.lookup(t) is not None
