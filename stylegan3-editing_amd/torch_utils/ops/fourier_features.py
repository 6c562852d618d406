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


def input_transform(t, user, freqs, phases, bandwidth, sampling_rate, normalise):
    """The transform algebra of `SynthesisInput.forward` (reference :204-230) in one launch.
    t [N,4] (straight from the affine layer when `normalise`, else already divided by |t[:2]|), user [3,3] or [N,3,3],
    freqs [C,2], phases [C]  ->  (freqs [N,C,2], phases [N,C], amplitudes [N,C]) for `fourier_features`."""
    if not (t.is_cuda and t.dtype == user.dtype == freqs.dtype == phases.dtype == torch.float32):
        raise RuntimeError('input_transform: float32 CUDA tensors expected')
    n, c = int(t.shape[0]), int(freqs.shape[0])
    if user.ndim == 3 and user.shape[0] == 1:
        user = user[0]                                      # a [1,3,3] transform broadcasts over the batch, as in the reference's matmul
    if tuple(t.shape) != (n, 4) or tuple(freqs.shape) != (c, 2) or tuple(phases.shape) != (c,) or tuple(user.shape) not in ((3, 3), (n, 3, 3)):
        raise RuntimeError(f'input_transform: shapes {tuple(t.shape)}, {tuple(user.shape)}, {tuple(freqs.shape)}, {tuple(phases.shape)}')
    lib = abi.load()
    t, user, freqs, phases = t.contiguous(), user.contiguous(), freqs.contiguous(), phases.contiguous()
    out_f = torch.empty([n, c, 2], dtype=torch.float32, device=t.device)
    out_p = torch.empty([n, c], dtype=torch.float32, device=t.device)
    out_a = torch.empty([n, c], dtype=torch.float32, device=t.device)
    p = abi.InputTransformParams()
    p.t, p.user, p.freqs, p.phases = abi.ptr(t), abi.ptr(user), abi.ptr(freqs), abi.ptr(phases)
    p.outFreqs, p.outPhases, p.outAmps = abi.ptr(out_f), abi.ptr(out_p), abi.ptr(out_a)
    p.N, p.C, p.normalise, p.userStrideN = n, c, int(bool(normalise)), (9 if user.ndim == 3 else 0)
    p.bandwidth, p.samplingRate = float(bandwidth), float(sampling_rate)
    with torch.cuda.device(t.device):
        abi.check(lib.sg3_input_transform(ctypes.byref(p), abi.stream_ptr(t.device)), 'sg3_input_transform')
    return out_f, out_p, out_a
