"""Golden vectors for the ReStyle encoder, generated with the REFERENCE's importable pieces.

The reference's `BackboneEncoder` module cannot be imported here (it imports torchvision and a non-existent
`inversion.models` package), but its residual units can: `models/setgan/encoder/encoders/helpers.py`
(`get_blocks`, `bottleneck_IR_SE`, `SEModule`).  This script assembles them exactly as
`restyle_psp_encoders.py:16-50` does (input_layer / body / styles, same module names => same state_dict keys) and
restates the two small modules that cannot be imported -- `GradualStyleBlock` (map2style.py:8-25) and `EqualLinear`
(models/stylegan2/model.py:129-158, the non-activated branch) -- with plain torch ops.  Parity of the assembled encoder
is therefore pinned on the reference's own block code; the glue is pinned on the reference text.

    python tests/golden/make_golden_encoder.py      ->  tests/golden/encoder.npz, encoder_manifest.json
"""
import json
import math
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, '/root/reference')

import torch  # noqa: E402
from torch import nn  # noqa: E402
from torch.nn import BatchNorm2d, Conv2d, PReLU, Sequential  # noqa: E402

from models.setgan.encoder.encoders.helpers import bottleneck_IR_SE, get_blocks  # noqa: E402  (reference code)
from synth_weights import synth_encoder_state_dict  # noqa: E402

torch.set_grad_enabled(False)


class EqualLinear(nn.Module):                      # reference models/stylegan2/model.py:129-158, activation=None
    def __init__(self, in_dim, out_dim, lr_mul=1):
        super().__init__()
        self.weight = nn.Parameter(torch.randn(out_dim, in_dim).div_(lr_mul))
        self.bias = nn.Parameter(torch.zeros(out_dim))
        self.scale = (1 / math.sqrt(in_dim)) * lr_mul
        self.lr_mul = lr_mul

    def forward(self, x):
        return torch.nn.functional.linear(x, self.weight * self.scale, bias=self.bias * self.lr_mul)


class GradualStyleBlock(nn.Module):                # reference map2style.py:8-25
    def __init__(self, in_c, out_c, spatial):
        super().__init__()
        self.out_c = out_c
        num_pools = int(np.log2(spatial))
        modules = [Conv2d(in_c, out_c, kernel_size=3, stride=2, padding=1), nn.LeakyReLU()]
        for _ in range(num_pools - 1):
            modules += [Conv2d(out_c, out_c, kernel_size=3, stride=2, padding=1), nn.LeakyReLU()]
        self.convs = nn.Sequential(*modules)
        self.linear = EqualLinear(out_c, out_c, lr_mul=1)

    def forward(self, x):
        return self.linear(self.convs(x).view(-1, self.out_c))


class RefBackboneEncoder(nn.Module):               # reference restyle_psp_encoders.py:16-50 (mode 'ir_se')
    def __init__(self, num_layers=50, n_styles=16, input_nc=6):
        super().__init__()
        self.input_layer = Sequential(Conv2d(input_nc, 64, (3, 3), 1, 1, bias=False), BatchNorm2d(64), PReLU(64))
        self.body = Sequential(*[bottleneck_IR_SE(b.in_channel, b.depth, b.stride) for blk in get_blocks(num_layers) for b in blk])
        self.styles = nn.ModuleList([GradualStyleBlock(512, 512, 16) for _ in range(n_styles)])
        self.style_count = n_styles

    def forward(self, x):
        x = self.body(self.input_layer(x))
        return torch.stack([s(x) for s in self.styles], dim=1)


def main():
    enc = RefBackboneEncoder().eval()
    man = {k: list(v.shape) for k, v in enc.state_dict().items()}
    with open(os.path.join(HERE, 'encoder_manifest.json'), 'w') as f:
        json.dump(man, f, indent=0, sort_keys=True)
    sd = synth_encoder_state_dict(man, seed=0)
    enc.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, strict=True)
    x = torch.from_numpy(np.random.RandomState(3).uniform(-1, 1, size=(2, 6, 256, 256)).astype(np.float32))
    out = {}
    feats = {}
    h = enc.input_layer(x)
    feats['stem'] = h
    for i, unit in enumerate(enc.body):
        h = unit(h)
        if i in (0, 2, 3, 6, 7, 20, 21, 23):
            feats[f'body{i}'] = h
    codes = enc(x)
    out['codes'] = codes.numpy()
    for k, v in feats.items():
        out[f'feat/{k}'] = v[:, :4, :12, :12].numpy()
        out[f'stat/{k}'] = np.asarray([v.mean().item(), v.std().item(), v.abs().max().item()], np.float64)
    # single units at small size, for op-level tests: (in, depth, stride) incl. the projection shortcut
    for name, (ci, co, st) in dict(unit_same=(64, 64, 1), unit_down=(64, 64, 2), unit_proj=(64, 128, 2)).items():
        u = bottleneck_IR_SE(ci, co, st).eval()
        uman = {k: list(v.shape) for k, v in u.state_dict().items()}
        usd = synth_encoder_state_dict({('body.0.' + k): s for k, s in uman.items()}, seed=5)
        u.load_state_dict({k[len('body.0.'):]: torch.from_numpy(np.asarray(v)) for k, v in usd.items()})
        xu = torch.from_numpy(np.random.RandomState(7).randn(2, ci, 20, 24).astype(np.float32))
        out[f'{name}/y'] = u(xu).numpy()
    np.savez_compressed(os.path.join(HERE, 'encoder.npz'), **out)
    print('encoder.npz', {k: v.shape for k, v in out.items() if k.startswith('codes') or k.startswith('unit')}, codes.abs().mean().item(), codes.std().item())


if __name__ == '__main__':
    main()
