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


def _resnet34_expected_keys(n_styles):
    """state_dict keys of the reference ResNetBackboneEncoder: its own stem + torchvision resnet34's layer1..4 BasicBlocks
    flattened into `body` (reference restyle_psp_encoders.py:61-83) + the style heads."""
    bn = ('weight', 'bias', 'running_mean', 'running_var', 'num_batches_tracked')
    keys = ['conv1.weight'] + [f'bn1.{b}' for b in bn] + ['relu.weight']
    i = 0
    for count, projected in ((3, False), (4, True), (6, True), (3, True)):
        for u in range(count):
            keys += [f'body.{i}.conv1.weight'] + [f'body.{i}.bn1.{b}' for b in bn] + [f'body.{i}.conv2.weight'] + [f'body.{i}.bn2.{b}' for b in bn]
            if projected and u == 0:
                keys += [f'body.{i}.downsample.0.weight'] + [f'body.{i}.downsample.1.{b}' for b in bn]
            i += 1
    for j in range(n_styles):
        for c in (0, 2, 4, 6):
            keys += [f'styles.{j}.convs.{c}.weight', f'styles.{j}.convs.{c}.bias']
        keys += [f'styles.{j}.linear.weight', f'styles.{j}.linear.bias']
    return keys


def build_resnet_encoder(n_styles=4, device='cpu', progressive=False):
    from models.setgan.encoder.encoders.restyle_e4e_encoders import ResNetProgressiveBackboneEncoder
    from models.setgan.encoder.encoders.restyle_psp_encoders import ResNetBackboneEncoder
    opts = types.SimpleNamespace(input_nc=6)
    enc = (ResNetProgressiveBackboneEncoder(n_styles, opts) if progressive else ResNetBackboneEncoder(n_styles, opts)).eval().requires_grad_(False)
    man = {k: list(v.shape) for k, v in enc.state_dict().items()}
    sd = synth_encoder_state_dict(man, seed=2)
    enc.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, strict=True)
    return enc.to(device), {k: np.asarray(v) for k, v in sd.items()}


def test_resnet34_encoder_keys_and_oracle():
    """ResNetBackboneEncoder (reference restyle_psp_encoders.py:53-97): checkpoint-compatible keys and shapes, product
    (plain PyTorch path) against the oracle restatement, e4e's progressive combination on the same trunk."""
    from oracle import oracle as O
    enc, sd = build_resnet_encoder(4)
    assert list(enc.state_dict().keys()) == _resnet34_expected_keys(4)
    shapes = {k: tuple(v.shape) for k, v in enc.state_dict().items()}
    assert shapes['conv1.weight'] == (64, 6, 7, 7) and shapes['body.3.downsample.0.weight'] == (128, 64, 1, 1)
    assert shapes['body.7.conv1.weight'] == (256, 128, 3, 3) and shapes['body.15.conv2.weight'] == (512, 512, 3, 3)
    x = _input()[:1]
    with torch.no_grad():
        codes = enc(torch.from_numpy(x))
    ref, feats = O.resnet_backbone_encoder(sd, x, n_styles=4, return_feats=True)
    assert feats['stem'].shape == (1, 64, 128, 128) and feats['body15'].shape == (1, 512, 16, 16)
    assert tuple(codes.shape) == (1, 4, 512)
    assert maxabs(codes.numpy(), ref) <= 2e-4 * max(1.0, float(np.abs(ref).max()))
    prog, _ = build_resnet_encoder(4, progressive=True)
    with torch.no_grad():
        b = prog(torch.from_numpy(x))
    assert maxabs(b[:, 0].numpy(), codes[:, 0].numpy()) <= 1e-6 and maxabs(b[:, 2].numpy(), (codes[:, 0] + codes[:, 2]).numpy()) <= 1e-5


def test_inference_iterative_writes_latents_and_stats(tmp_path):
    """The on-disk contract of reference inversion/scripts/inference_iterative.py:61-101: `latents.npy` holds
    {image name: [one [16,512] latent per ReStyle step]} and `stats.txt` 'Runtime mean+-std' of the per-batch time."""
    import re
    from helpers import build_product_generator
    from inversion.scripts.inference_iterative import run_inference
    from models.setgan.encoder.psp3 import pSp
    from utils.inference_utils import get_average_image, run_on_batch

    class TinyEncoder(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.lin = torch.nn.Linear(6, 16 * 32)

        def forward(self, x):
            return self.lin(x.mean(dim=(2, 3))).view(-1, 16, 32) * 0.1

    G = build_product_generator('Ttiny')
    opts = types.SimpleNamespace(encoder_type='BackboneEncoder', input_nc=6, checkpoint_path=None, n_iters_per_batch=3, resize_outputs=False, test_batch_size=2)
    torch.manual_seed(0)
    net = pSp.__new__(pSp)
    torch.nn.Module.__init__(net)
    net.opts, net.n_styles, net.encoder, net.decoder = opts, 16, TinyEncoder(), G
    net.face_pool = torch.nn.AdaptiveAvgPool2d((256, 256))
    net.latent_avg = G.mapping.w_avg
    net.eval()
    images = torch.rand(5, 3, 256, 256) * 2 - 1
    names = [f'{i:05d}.png' for i in range(5)]
    results, latents, line = run_inference(net, opts, images, names, str(tmp_path), n_images=None)
    assert re.fullmatch(r'Runtime \d+\.\d{4}\+-\d+\.\d{4}', line)
    assert open(tmp_path / 'stats.txt').read() == line
    loaded = np.load(tmp_path / 'latents.npy', allow_pickle=True).item()
    assert list(loaded.keys()) == names                          # insertion order = dataset order
    assert all(len(v) == 3 and v[0].shape == (16, 32) and v[0].dtype == np.float32 for v in loaded.values())
    with torch.no_grad():
        _, direct = run_on_batch(images[4:5], net, opts, get_average_image(net))     # the ragged last batch (5 = 2 + 2 + 1)
    assert maxabs(loaded['00004.png'][2], direct[0][2]) <= 1e-6
    assert len(results['00000.png']) == 3 and tuple(results['00000.png'][0].shape) == (3, 64, 64)


@pytest.mark.parametrize('family,encoder_type,n_styles', [('pSp', 'ResNetBackboneEncoder', 16), ('e4e', 'ProgressiveBackboneEncoder', 3)])
def test_load_encoder_checkpoint_roundtrip(family, encoder_type, n_styles, tmp_path):
    """The encoder checkpoint format of the reference (psp3.py:34-38, :108-114; inference_utils.py:28-56): `state_dict` with
    `encoder.*` / `decoder.*` entries, `latent_avg`, `opts`; `load_encoder` picks pSp or e4e from `opts['encoder_type']`
    (utils/model_utils.py), merges the test options over the stored ones and restores every tensor."""
    from models.setgan.encoder.e4e3 import e4e
    from models.setgan.encoder.psp3 import pSp
    from models.stylegan3.model import SG3Generator
    from utils.inference_utils import load_encoder
    wrapper = {'pSp': pSp, 'e4e': e4e}[family]
    stored = dict(encoder_type=encoder_type, input_nc=6, n_styles=n_styles, n_iters_per_batch=3, resize_outputs=False, stylegan_weights=None,
                  checkpoint_path=None)
    torch.manual_seed(3)
    decoder = SG3Generator(checkpoint_path=None, device='cpu').decoder          # the reference's default: config R at 1024
    src = wrapper(types.SimpleNamespace(**stored), decoder=decoder).eval()
    with torch.no_grad():
        for prm in src.encoder.parameters():
            prm.add_(0.01 * torch.randn_like(prm))                              # away from the constructor's values
        decoder.synthesis.input.transform = torch.eye(3).repeat(2, 1, 1)        # a batched call leaves this behind (psp3.py:37)
    latent_avg = torch.randn(n_styles, 512)
    state = {'encoder.' + k: v for k, v in src.encoder.state_dict().items()}
    state.update({'decoder.' + k: v for k, v in decoder.state_dict().items()})
    path = tmp_path / 'restyle.pt'
    torch.save({'state_dict': state, 'latent_avg': latent_avg, 'opts': stored}, path)
    net, opts = load_encoder(path, test_opts={'n_iters_per_batch': 2, 'test_batch_size': 4}, device='cpu')
    assert type(net) is wrapper and type(net.encoder).__name__ == encoder_type and not net.training
    assert opts.n_iters_per_batch == 2 and opts.test_batch_size == 4 and opts.encoder_type == encoder_type and opts.checkpoint_path == path
    assert torch.equal(net.latent_avg, latent_avg) and net.n_styles == n_styles
    for k, v in src.encoder.state_dict().items():
        assert torch.equal(net.encoder.state_dict()[k], v), k
    got = net.decoder.state_dict()
    for k, v in decoder.state_dict().items():
        if k != 'synthesis.input.transform':
            assert torch.equal(got[k], v), k
    assert tuple(got['synthesis.input.transform'].shape) == (3, 3)              # the shape-dependent buffer is not loaded
    x = torch.from_numpy(_input()[:1])
    with torch.no_grad():
        assert maxabs(net.encoder(x).numpy(), src.encoder(x).numpy()) <= 1e-6
