"""CPU: ReStyle encoder -- oracle and product (plain PyTorch path) against golden vectors built from the reference's
own residual-unit code (tests/golden/make_golden_encoder.py), state_dict key compatibility, pSp / run_on_batch plumbing."""
import json
import os
import types

import numpy as np
import pytest
import torch

from helpers import GOLD, golden, maxabs
from synth_weights import synth_encoder_state_dict


def _manifest():
    with open(os.path.join(GOLD, 'encoder_manifest.json')) as f:
        return json.load(f)


def _input():
    return np.random.RandomState(3).uniform(-1, 1, size=(2, 6, 256, 256)).astype(np.float32)


def build_product_encoder(device='cpu'):
    from models.setgan.encoder.encoders.restyle_psp_encoders import BackboneEncoder
    enc = BackboneEncoder(50, 'ir_se', 16, types.SimpleNamespace(input_nc=6)).eval().requires_grad_(False)
    man = {k: list(v.shape) for k, v in enc.state_dict().items()}
    assert man == _manifest()                     # same keys / shapes as the reference-assembled encoder
    sd = synth_encoder_state_dict(man, seed=0)
    enc.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, strict=True)
    return enc.to(device)


def test_oracle_encoder():
    from oracle import oracle as O
    g = golden('encoder')
    sd = {k: np.asarray(v) for k, v in synth_encoder_state_dict(_manifest(), seed=0).items()}
    codes, feats = O.backbone_encoder(sd, _input(), return_feats=True)
    for k in [f for f in g.files if f.startswith('feat/')]:
        assert maxabs(feats[k[5:]][:, :4, :12, :12], g[k]) <= 2e-4, k
    assert maxabs(codes, g['codes']) <= 2e-4


def test_product_encoder_torch_path():
    g = golden('encoder')
    enc = build_product_encoder()
    with torch.no_grad():
        codes = enc(torch.from_numpy(_input()))
    assert tuple(codes.shape) == (2, 16, 512)
    assert maxabs(codes.numpy(), g['codes']) <= 1e-5


@pytest.mark.parametrize('name,ci,co,st', [('unit_same', 64, 64, 1), ('unit_down', 64, 64, 2), ('unit_proj', 64, 128, 2)])
def test_product_units(name, ci, co, st):
    from models.setgan.encoder.encoders.helpers import bottleneck_IR_SE
    u = bottleneck_IR_SE(ci, co, st).eval()
    man = {('body.0.' + k): list(v.shape) for k, v in u.state_dict().items()}
    sd = synth_encoder_state_dict(man, seed=5)
    u.load_state_dict({k[len('body.0.'):]: torch.from_numpy(np.asarray(v)) for k, v in sd.items()})
    x = torch.from_numpy(np.random.RandomState(7).randn(2, ci, 20, 24).astype(np.float32))
    with torch.no_grad():
        assert maxabs(u(x).numpy(), golden('encoder')[name + '/y']) <= 1e-5


def test_psp_and_run_on_batch_plumbing():
    """pSp.forward / get_average_image / run_on_batch semantics on a tiny decoder (CPU, plain PyTorch paths):
    latent_avg on step 0, residual latents afterwards, second synthesis with the landmarks transform on request."""
    from helpers import build_product_generator
    from models.setgan.encoder.psp3 import pSp
    from utils.inference_utils import get_average_image, run_on_batch

    class TinyEncoder(torch.nn.Module):           # stands in for the 186 M parameter backbone in this plumbing test
        def __init__(self):
            super().__init__()
            self.lin = torch.nn.Linear(6, 16 * 32)

        def forward(self, x):
            return self.lin(x.mean(dim=(2, 3))).view(-1, 16, 32) * 0.1

    G = build_product_generator('Ttiny')
    opts = types.SimpleNamespace(encoder_type='BackboneEncoder', input_nc=6, checkpoint_path=None, n_iters_per_batch=3, resize_outputs=False)
    torch.manual_seed(0)
    net = pSp.__new__(pSp)
    torch.nn.Module.__init__(net)
    net.opts, net.n_styles = opts, 16
    net.encoder = TinyEncoder()
    net.face_pool = torch.nn.AdaptiveAvgPool2d((256, 256))
    net.decoder = G
    net.latent_avg = G.mapping.w_avg
    net.eval()
    with torch.no_grad():
        avg = get_average_image(net)
        assert tuple(avg.shape) == (3, 256, 256)              # face_pool(resize=True default) of the 64x64 image
        inputs = torch.rand(2, 3, 256, 256) * 2 - 1
        imgs, lats = run_on_batch(inputs, net, opts, avg)
        assert len(imgs[0]) == 3 and len(lats[1]) == 3 and lats[0][0].shape == (16, 32)
        # step 0 adds latent_avg, later steps add the previous latent
        x0 = torch.cat([inputs, avg.unsqueeze(0).repeat(2, 1, 1, 1)], dim=1)
        c0 = net.encoder(x0) + G.mapping.w_avg.repeat(2, 1, 1)
        assert maxabs(lats[0][0], c0[0].numpy()) <= 1e-6
        y0 = net.face_pool(G.synthesis(c0, noise_mode='const', force_fp32=True))
        c1 = net.encoder(torch.cat([inputs, y0], dim=1)) + c0
        assert maxabs(lats[1][1], c1[1].numpy()) <= 1e-5
        # landmarks transform: a second synthesis; the last step returns the unaligned image
        tr = torch.eye(3).repeat(2, 1, 1)
        tr[:, 0, 2] = 0.2
        imgs_u, _ = run_on_batch(inputs, net, opts, avg, landmarks_transform=tr)
        assert maxabs(imgs_u[0][1].numpy(), imgs[0][1].numpy()) <= 1e-5       # non-final steps: aligned output
        assert maxabs(imgs_u[0][2].numpy(), imgs[0][2].numpy()) > 1e-3        # final step: shifted (unaligned) output


def test_e4e_progressive_encoder_combination():
    """ProgressiveBackboneEncoder: same trunk/heads as the pSp backbone, combined as w0 + delta_i
    (reference restyle_e4e_encoders.py:79-89); state_dict keys are the pSp backbone's."""
    from models.setgan.encoder.e4e3 import e4e
    from models.setgan.encoder.encoders.restyle_e4e_encoders import ProgressiveBackboneEncoder
    enc = build_product_encoder()
    prog = ProgressiveBackboneEncoder(50, 'ir_se', 16, types.SimpleNamespace(input_nc=6)).eval().requires_grad_(False)
    prog.load_state_dict(enc.state_dict(), strict=True)
    x = torch.from_numpy(_input()[:1])
    with torch.no_grad():
        a, b = enc(x), prog(x)
        assert maxabs(b[:, 0].numpy(), a[:, 0].numpy()) <= 1e-6
        assert maxabs(b[:, 5].numpy(), (a[:, 0] + a[:, 5]).numpy()) <= 1e-6
        prog.set_progressive_stage(3)
        c = prog(x)
        assert maxabs(c[:, 3].numpy(), (a[:, 0] + a[:, 3]).numpy()) <= 1e-6 and maxabs(c[:, 4].numpy(), a[:, 0].numpy()) <= 1e-6
    assert issubclass(e4e, torch.nn.Module) and e4e.forward is not None
