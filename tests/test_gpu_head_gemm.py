"""GPU: sg3_head_gemm (csrc/sg3_head_gemm.hip), the batched small-M GEMMs of the GradualStyleBlock heads (reference
models/setgan/encoder/encoders/map2style.py:8-25), against a float64 evaluation of the same sums.

Tolerance: split precision keeps 22 significand bits of every operand and accumulates in fp32 over K <= 4608 terms, like the fp32
GEMM it replaces: 2e-6 of the largest |term sum| per output (sum_k |a| |w|)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def _case(g, m, k, n, seed, scale=1.0):
    r = np.random.RandomState(seed)
    a = (r.standard_normal((g, m, k)) * scale).astype(np.float32)
    w = (r.standard_normal((g, k, n)) / np.sqrt(k)).astype(np.float32)
    b = r.standard_normal((g, n)).astype(np.float32)
    return a, w, b


def _ref(a, w, b, slope):
    a64 = a.astype(np.float64)
    act = np.where(a64 < 0, a64 * np.float64(np.float32(slope)), a64)
    out = np.einsum('gmk,gkn->gmn', act, w.astype(np.float64)) + (0 if b is None else b.astype(np.float64)[:, None, :])
    mag = np.einsum('gmk,gkn->gmn', np.abs(act), np.abs(w.astype(np.float64))) + 1.0
    return out, mag


@pytest.mark.parametrize('g,m,k,n,slope,bias', [
    (16, 8, 512, 512, 0.01, True),        # the EqualLinear at batch 8
    (16, 8, 4608, 512, 1.0, True),        # level 4 at batch 8
    (16, 32, 4608, 512, 1.0, True),       # level 3 at batch 8
    (3, 33, 512, 64, 0.2, True),          # two row blocks, the second nearly empty
    (2, 64, 144, 32, 1.0, False),         # K = 9 k steps (not a multiple of the pipeline depth), one column block, no bias
    (16, 128, 4608, 512, 1.0, True),      # level 2 at batch 8
    (5, 130, 48, 96, 0.01, True),         # more than one 128-row block, ragged
    (1, 1, 16, 32, 1.0, True),            # smallest shape
    (2, 300, 512, 512, 1.0, True),        # three 128-row blocks
])
def test_head_gemm_matches_float64(g, m, k, n, slope, bias):
    from torch_utils.ops import plain_conv
    from torch_utils.ops.head_gemm import PackedHeadWeights
    a, w, b = _case(g, m, k, n, seed=g * 1000 + m)
    pw = PackedHeadWeights(torch.from_numpy(w).to(DEV), torch.from_numpy(b).to(DEV) if bias else None)
    assert pw.usable
    plain_conv.reset_overflow(DEV)
    out = pw.run(torch.from_numpy(a).to(DEV), slope).cpu().numpy().astype(np.float64)
    assert not plain_conv.overflowed(DEV)
    ref, mag = _ref(a, w, b if bias else None, slope)
    assert out.shape == ref.shape
    assert np.max(np.abs(out - ref) / mag) < 2e-6


def test_head_gemm_is_deterministic_and_rows_are_independent():
    from torch_utils.ops.head_gemm import PackedHeadWeights
    a, w, b = _case(4, 40, 512, 128, seed=7)
    pw = PackedHeadWeights(torch.from_numpy(w).to(DEV), torch.from_numpy(b).to(DEV))
    x = torch.from_numpy(a).to(DEV)
    o1, o2 = pw.run(x), pw.run(x)
    assert torch.equal(o1, o2)
    # a row's result does not depend on the other rows of the launch (the row-block template differs: 40 rows vs 8 rows)
    assert torch.equal(pw.run(x[:, 16:24].contiguous()), o1[:, 16:24])


def test_head_gemm_range_guard():
    from torch_utils.ops import plain_conv
    from torch_utils.ops.head_gemm import PackedHeadWeights
    a, w, b = _case(2, 8, 64, 32, seed=3)
    pw = PackedHeadWeights(torch.from_numpy(w).to(DEV), None)
    big = a.copy(); big[1, 3, 17] = 7.0e4
    plain_conv.reset_overflow(DEV)
    pw.run(torch.from_numpy(big).to(DEV))
    assert plain_conv.overflowed(DEV)
    plain_conv.reset_overflow(DEV)
    neg = a.copy(); neg[0, 0, 0] = -7.0e6                   # inside the range after LeakyReLU(0.001)... still 7e3: no flag
    pw.run(torch.from_numpy(neg).to(DEV), 0.001)
    assert not plain_conv.overflowed(DEV)
    nan = a.copy(); nan[0, 2, 5] = np.nan
    out = pw.run(torch.from_numpy(nan).to(DEV))
    assert torch.isnan(out[0, 2]).all() and torch.isfinite(out[0, :2]).all() and torch.isfinite(out[1]).all()   # loud in its own row only
    plain_conv.reset_overflow(DEV)
    wbig = w.copy(); wbig[1, 5, 5] = 1.0e5
    assert not PackedHeadWeights(torch.from_numpy(wbig).to(DEV), None).usable


def test_head_gemm_rejects_what_it_cannot_run():
    from torch_utils.ops.head_gemm import PackedHeadWeights
    with pytest.raises(RuntimeError):
        PackedHeadWeights(torch.zeros([2, 24, 32], device=DEV))           # K not a multiple of 16
    with pytest.raises(RuntimeError):
        PackedHeadWeights(torch.zeros([2, 32, 48], device=DEV))           # N not a multiple of 32
    pw = PackedHeadWeights(torch.zeros([2, 32, 32], device=DEV))
    with pytest.raises(RuntimeError):
        pw.run(torch.zeros([2, 4, 16], device=DEV))
    with pytest.raises(RuntimeError):
        pw.run(torch.zeros([2, 4, 32], device=DEV, dtype=torch.float16))


def test_encoder_heads_take_the_own_gemm_for_few_rows_and_match_baddbmm():
    """The encoder's head path with sg3_head_gemm (batch 2: every level has <= 16 rows... level 2 has 32) against the same
    forward with the row limits at zero (all torch.baddbmm)."""
    from helpers import build_restyle_pair
    from models.setgan.encoder.encoders import restyle_psp_encoders as E
    from torch_utils import _sg3abi
    net, *_ = build_restyle_pair('Rmini', device=DEV, n_iters=1)
    enc = net.encoder.eval()
    x = torch.from_numpy(np.random.RandomState(0).uniform(-1, 1, size=(2, 6, 256, 256)).astype(np.float32)).to(DEV)
    with torch.no_grad():
        n0 = _sg3abi.launch_count
        own = enc(x)
        launches_own = _sg3abi.launch_count - n0
        saved = E._HEAD_GEMM_MAX_ROWS
        E._HEAD_GEMM_MAX_ROWS = (0, 0)
        try:
            n0 = _sg3abi.launch_count
            blas = enc(x)
            launches_blas = _sg3abi.launch_count - n0
        finally:
            E._HEAD_GEMM_MAX_ROWS = saved
    assert launches_own > launches_blas                       # the own GEMM did run
    assert float((own - blas).abs().max()) <= 2e-5 * float(blas.abs().max())
