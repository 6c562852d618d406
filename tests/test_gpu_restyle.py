"""GPU: the ReStyle loop (BASELINE configs[2]) on the HIP path -- `pSp.forward`, `get_average_image`,
`run_on_batch` with the IR-SE50 encoder -- against the oracle's restatement of reference models/setgan/encoder/psp3.py:45-84
and utils/inference_utils.py:59-111, and at full size (R-1024 decoder, batch 16, 5 steps) through golden encoder codes and
batch-independence.

Tolerances: the loop feeds its own output back five times through a 50-layer encoder with random (untrained) weights, so a
1e-7 difference in step k's image reaches the step k+1 latent amplified; latents are O(1..10).  Images: 1e-4 (the
BASELINE bound).  Latents: 2e-4 * max|latent| per step."""
import numpy as np
import pytest
import torch

from helpers import build_product_generator, build_restyle_pair, golden, maxabs

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def _frames(n, seed=4):
    return np.random.RandomState(seed).uniform(-1, 1, size=(n, 3, 256, 256)).astype(np.float32)


def _landmarks(n):
    from synth_weights import make_user_transform
    return np.stack([make_user_transform((0.04 * (i + 1), -0.03 * i), 6.0 * (i - 1)) for i in range(n)]).astype(np.float32)


@pytest.mark.parametrize('cfg,batch,steps', [('Rmini', 3, 5), ('Tmini', 2, 3)])
def test_run_on_batch_matches_oracle(cfg, batch, steps):
    from oracle import oracle as O
    from torch_utils import _sg3abi
    from utils.inference_utils import get_average_image, run_on_batch
    net, opts, enc_sd, gen_sd, sched = build_restyle_pair(cfg, device=DEV, n_iters=steps)
    lat_avg = gen_sd['mapping.w_avg']
    x, lt = _frames(batch), _landmarks(batch)
    avg_o = O.get_average_image(enc_sd, gen_sd, sched, lat_avg)
    imgs_o, lats_o, aligned_last_o = O.run_on_batch(enc_sd, gen_sd, sched, x, lat_avg, avg_o, steps, landmarks_transform=lt)
    n0 = _sg3abi.launch_count
    with torch.no_grad():
        avg = get_average_image(net)
        assert avg.is_cuda and maxabs(avg.cpu().numpy(), avg_o) <= 1e-5
        xt = torch.from_numpy(x).to(DEV)
        imgs_on, lats_on = run_on_batch(xt, net, opts, avg, landmarks_transform=torch.from_numpy(lt).to(DEV))
        imgs_off, lats_off = run_on_batch(xt, net, opts, avg, landmarks_transform=None)
    assert _sg3abi.launch_count - n0 > 100 * steps, 'the loop did not run on the HIP kernels'
    for it in range(steps):
        lat_on = np.stack([lats_on[i][it] for i in range(batch)])
        lat_off = np.stack([lats_off[i][it] for i in range(batch)])
        tol = 2e-4 * float(np.abs(lats_o[it]).max())
        assert maxabs(lat_on, lats_o[it]) <= tol, (it, maxabs(lat_on, lats_o[it]), tol)
        assert maxabs(lat_off, lats_o[it]) <= tol, it                  # the latents never see the landmark transforms
        img_on = torch.stack([imgs_on[i][it] for i in range(batch)]).cpu().numpy()
        img_off = torch.stack([imgs_off[i][it] for i in range(batch)]).cpu().numpy()
        assert maxabs(img_on, imgs_o[it]) <= 1e-4, (it, maxabs(img_on, imgs_o[it]))
        # without transforms every step returns the aligned image: the oracle's per-step output except for the last step
        assert maxabs(img_off, aligned_last_o if it == steps - 1 else imgs_o[it]) <= 1e-4, it
    # the last step with transforms is the UNALIGNED render: it differs visibly from the aligned one
    assert maxabs(imgs_o[-1], aligned_last_o) > 1e-3


def test_restyle_full_size_first_step_and_batch_independence():
    """BASELINE configs[2] shape: IR-SE50 encoder + FFHQ-1024 config-R decoder, batch 16, 5 steps.  (1) step 0 of
    pSp.forward on the golden encoder input = golden encoder codes (reference residual units) + latent_avg; (2) the batch-16
    loop gives every frame what a batch-1 loop gives it (frames are independent units: the sharding argument of SURVEY 8e)."""
    import types
    from models.setgan.encoder.psp3 import pSp
    from synth_weights import synth_encoder_state_dict
    from utils.inference_utils import get_average_image, run_on_batch
    g = golden('encoder')
    G = build_product_generator('R1024')
    opts = types.SimpleNamespace(encoder_type='BackboneEncoder', input_nc=6, checkpoint_path=None, n_iters_per_batch=5, resize_outputs=False)
    net = pSp(opts, decoder=G)
    man = {k: list(v.shape) for k, v in net.encoder.state_dict().items()}
    net.encoder.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in synth_encoder_state_dict(man, seed=0).items()})
    net = net.eval().requires_grad_(False).to(DEV)
    x6 = np.random.RandomState(3).uniform(-1, 1, size=(2, 6, 256, 256)).astype(np.float32)       # the golden encoder input
    with torch.no_grad():
        img, codes = net.forward(torch.from_numpy(x6).to(DEV), latent=None, return_latents=True, resize=False)
        assert tuple(img.shape) == (2, 3, 1024, 1024) and bool(torch.isfinite(img).all())
        want = g['codes'] + G.mapping.w_avg.cpu().numpy()[None, None]
        assert maxabs(codes.cpu().numpy(), want) <= 2e-4 * float(np.abs(want).max())
        avg = get_average_image(net)
        x = torch.from_numpy(_frames(16, seed=11)).to(DEV)
        imgs, lats = run_on_batch(x, net, opts, avg)
        assert len(imgs[15]) == 5 and tuple(imgs[15][4].shape) == (3, 1024, 1024)
        for i in (0, 7, 15):
            imgs1, lats1 = run_on_batch(x[i:i + 1], net, opts, avg)
            for it in range(5):
                tol = 2e-4 * max(1.0, float(np.abs(lats1[0][it]).max()))
                assert maxabs(lats[i][it], lats1[0][it]) <= tol, (i, it, maxabs(lats[i][it], lats1[0][it]))
            assert maxabs(imgs[i][4].cpu().numpy(), imgs1[0][4].cpu().numpy()) <= 1e-4, i


def test_graphed_restyle_step_equals_eager_loop():
    """The hipGraph-replayed ReStyle loop (sg3_runtime.GraphedReStyleStep through run_on_batch) against the eager loop: same
    latents and images, with and without landmark transforms, and eager fallback for a batch the graph was not captured for."""
    from sg3_runtime import GraphedReStyleStep
    from utils.inference_utils import get_average_image, run_on_batch
    net, opts, *_ = build_restyle_pair('Rmini', device=DEV, n_iters=4)
    x = torch.from_numpy(_frames(3, seed=6)).to(DEV)
    lt = torch.from_numpy(_landmarks(3)).to(DEV)
    with torch.no_grad():
        avg = get_average_image(net)
        eager = {key: run_on_batch(x, net, opts, avg, landmarks_transform=t) for key, t in (('off', None), ('on', lt))}
        net.graphed_step = GraphedReStyleStep(net, 3)
        for key, t in (('off', None), ('on', lt)):
            imgs, lats = run_on_batch(x, net, opts, avg, landmarks_transform=t)
            for i in range(3):
                for it in range(4):
                    assert maxabs(lats[i][it], eager[key][1][i][it]) <= 1e-5 * max(1.0, float(np.abs(eager[key][1][i][it]).max())), (key, i, it)
                    assert maxabs(imgs[i][it].cpu().numpy(), eager[key][0][i][it].cpu().numpy()) <= 1e-5, (key, i, it)
        imgs2, lats2 = run_on_batch(x[:2], net, opts, avg)              # other batch size: eager path
        assert maxabs(lats2[1][3], eager['off'][1][1][3]) <= 1e-4 * max(1.0, float(np.abs(lats2[1][3]).max()))


@pytest.mark.parametrize('cfg', ['Tmini', 'Rmini'])
def test_psp_forward_and_restyle_loop_match_reference_fixture(cfg):
    """pSp.forward and run_on_batch on the HIP path against the fixture the REFERENCE's own psp3.forward / run_on_batch produced
    (tests/golden/make_golden_callers.py): latent_avg on step 0, residual steps, landmark transforms, face_pool."""
    from test_callers_golden_cpu import build_loop_net, check_loop_against_golden
    from torch_utils import _sg3abi
    net, opts = build_loop_net(cfg, device=DEV)
    n0 = _sg3abi.launch_count
    check_loop_against_golden(net, opts, cfg, DEV, tol_img=1e-4, tol_lat=1e-5)
    assert _sg3abi.launch_count - n0 > 300, 'the decoder did not run on the HIP kernels'


def test_run_on_batch_with_resnet34_encoder_matches_oracle():
    """The ReStyle loop with the ResNet34 backbone (`encoder_type='ResNetBackboneEncoder'`, reference restyle_psp_encoders.py:53-97)
    on the HIP path against the oracle loop driven by the oracle's ResNet34 restatement: 2 steps, landmark transforms on."""
    from oracle import oracle as O
    from utils.inference_utils import get_average_image, run_on_batch
    net, opts, enc_sd, gen_sd, sched = build_restyle_pair('Rmini', device=DEV, encoder_type='ResNetBackboneEncoder', n_iters=2)
    assert type(net.encoder).__name__ == 'ResNetBackboneEncoder'
    lat_avg = gen_sd['mapping.w_avg']
    x, lt = _frames(2, seed=8), _landmarks(2)
    enc = lambda x6: O.resnet_backbone_encoder(enc_sd, x6, n_styles=16)  # noqa: E731
    avg_o = O.get_average_image(None, gen_sd, sched, lat_avg)
    imgs_o, lats_o, _ = O.run_on_batch(None, gen_sd, sched, x, lat_avg, avg_o, 2, landmarks_transform=lt, encoder=enc)
    with torch.no_grad():
        avg = get_average_image(net)
        imgs, lats = run_on_batch(torch.from_numpy(x).to(DEV), net, opts, avg, landmarks_transform=torch.from_numpy(lt).to(DEV))
    for it in range(2):
        got_l = np.stack([lats[i][it] for i in range(2)])
        assert maxabs(got_l, lats_o[it]) <= 2e-4 * max(1.0, float(np.abs(lats_o[it]).max())), it
        assert maxabs(torch.stack([imgs[i][it] for i in range(2)]).cpu().numpy(), imgs_o[it]) <= 1e-4, it
