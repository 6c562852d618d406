"""The affine (style) layers of every synthesis layer as ONE launch of libsg3hip (csrc/sg3_affine.hip: sg3_affine_batch).

The reference runs `styles = self.affine(w)` layer by layer (models/stylegan3/networks_stylegan3.py:349, the input's at :205):
per layer a `weight * weight_gain` kernel, a bias kernel and an addmm (:88-96) -- ~45 launches of ~5 us per synthesis forward.
GPU inference packs all layers' scaled weights and biases once (cached on the parameters' versions) and evaluates them together."""
import ctypes

import numpy as np
import torch

from .. import _sg3abi as abi


class AffinePack:
    """Concatenated `weight * weight_gain` / `bias * bias_gain` / post-scale of a list of linear FullyConnectedLayers, rebuilt when
    any parameter changed (`(data_ptr, _version)` of every weight and bias; writes through `.data` need `invalidate()`)."""

    def __init__(self):
        self.key = None

    def invalidate(self):
        self.key = None

    def _build(self, fcs, ws_index, post_scale, device):
        rows = [int(fc.weight.shape[0]) for fc in fcs]
        with torch.no_grad():
            self.weight = torch.cat([fc.weight.detach().to(device=device, dtype=torch.float32) * fc.weight_gain for fc in fcs]).contiguous()
            self.bias = torch.cat([(fc.bias.detach().to(device=device, dtype=torch.float32) * fc.bias_gain) if fc.bias is not None
                                   else torch.zeros([r], device=device) for fc, r in zip(fcs, rows)]).contiguous()
            scale = np.concatenate([np.full([r], float(sc), dtype=np.float32) for r, sc in zip(rows, post_scale)])
            self.scale = None if bool((scale == 1).all()) else torch.from_numpy(scale).to(device)
        start = np.concatenate([[0], np.cumsum(rows)]).astype(np.int32)
        self.row_start_host = [int(v) for v in start]
        self.row_start = torch.from_numpy(start).to(device)
        self.ws_index = torch.tensor([int(i) for i in ws_index], dtype=torch.int32, device=device)
        self.w_dim = int(fcs[0].weight.shape[1])

    def __call__(self, ws, fcs, ws_index, post_scale):
        """ws [N, num_ws, w_dim] float32 on the GPU; fcs[j] reads ws[:, ws_index[j]]; result j = (affine_j(...)) * post_scale[j]
        as a dense [N, C_j] tensor (views of one buffer)."""
        if not (ws.is_cuda and ws.dtype == torch.float32 and ws.ndim == 3 and ws.stride(2) == 1):
            raise RuntimeError('affine_batch: ws must be a float32 CUDA tensor [N, num_ws, w_dim] with unit innermost stride')
        if any(fc.activation != 'linear' or int(fc.weight.shape[1]) != int(ws.shape[2]) for fc in fcs):
            raise RuntimeError('affine_batch: linear layers on w_dim inputs only')
        key = tuple((fc.weight.data_ptr(), fc.weight._version, None if fc.bias is None else (fc.bias.data_ptr(), fc.bias._version))
                    for fc in fcs) + (tuple(int(i) for i in ws_index), tuple(float(s) for s in post_scale), str(ws.device))
        if key != self.key:
            self._build(fcs, ws_index, post_scale, ws.device)
            self.key = key
        n = int(ws.shape[0])
        rows = self.row_start_host[-1]
        out = torch.empty([n * rows], dtype=torch.float32, device=ws.device)
        p = abi.AffineBatchParams()
        p.ws, p.wsStrideN, p.wsStrideL = abi.ptr(ws), int(ws.stride(0)), int(ws.stride(1))
        p.weight, p.bias, p.scale = abi.ptr(self.weight), abi.ptr(self.bias), (abi.ptr(self.scale) if self.scale is not None else None)
        p.rowStart, p.wsIndex, p.out = abi.ptr(self.row_start), abi.ptr(self.ws_index), abi.ptr(out)
        p.N, p.wDim, p.layers, p.rows = n, self.w_dim, len(fcs), rows
        with torch.cuda.device(ws.device):
            abi.check(abi.load().sg3_affine_batch(ctypes.byref(p), abi.stream_ptr(ws.device)), 'sg3_affine_batch')
        return [out[n * a: n * b].view(n, b - a) for a, b in zip(self.row_start_host[:-1], self.row_start_host[1:])]
