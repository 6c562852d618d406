"""Fourier features of `SynthesisInput` as one HIP kernel, written channels-first (libsg3hip: csrc/sg3_fourier.hip).

Replaces, for GPU inference, the chain  K=2 matmul -> + phases -> * 2pi -> sin -> * amplitudes -> permute  of the reference's
`SynthesisInput.forward` (models/stylegan3/networks_stylegan3.py:236-241, :245) with the same arithmetic in the same order
(bit-identical features), one launch instead of six and no [N,H,W,C] intermediate."""
import ctypes

import torch

from .. import _sg3abi as abi


def fourier_features(grid, freqs, phases, amps):
    """grid [H,W,2] (from F.affine_grid), freqs [N,C,2], phases [N,C], amps [N,C], all float32 on one GPU  ->  [N,C,H,W]."""
    if not (grid.is_cuda and grid.dtype == freqs.dtype == phases.dtype == amps.dtype == torch.float32):
        raise RuntimeError('fourier_features: float32 CUDA tensors expected')
    h, w, two = (int(v) for v in grid.shape)
    n, c, two2 = (int(v) for v in freqs.shape)
    if two != 2 or two2 != 2 or tuple(phases.shape) != (n, c) or tuple(amps.shape) != (n, c):
        raise RuntimeError(f'fourier_features: shapes {tuple(grid.shape)}, {tuple(freqs.shape)}, {tuple(phases.shape)}, {tuple(amps.shape)}')
    lib = abi.load()
    grid, freqs, phases, amps = grid.contiguous(), freqs.contiguous(), phases.contiguous(), amps.contiguous()
    out = torch.empty([n, c, h, w], dtype=torch.float32, device=grid.device)
    p = abi.FourierParams()
    p.grid, p.freqs, p.phases, p.amps, p.out = abi.ptr(grid), abi.ptr(freqs), abi.ptr(phases), abi.ptr(amps), abi.ptr(out)
    p.N, p.C, p.H, p.W = n, c, h, w
    with torch.cuda.device(grid.device):
        abi.check(lib.sg3_fourier_features(ctypes.byref(p), abi.stream_ptr(grid.device)), 'sg3_fourier_features')
    return out
