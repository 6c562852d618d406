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


@pytest.mark.parametrize('stage', [99, 5])
def test_e4e_forward_and_restyle_loop_match_oracle(stage):
    """`e4e(opts, decoder=G)` with the ProgressiveBackboneEncoder on the HIP path (reference models/setgan/encoder/e4e3.py:45-87,
    encoders/restyle_e4e_encoders.py:79-89): e4e.forward (first step on latent_avg, residual step, landmark transforms, pooling)
    and a 3-step run_on_batch against the oracle loop driven by the progressive combination w[:, i] = w0 + delta_i (i <= stage)
    of the oracle's IR-SE50 heads."""
    import types
    from models.setgan.encoder.e4e3 import e4e
    from oracle import oracle as O
    from synth_weights import synth_encoder_state_dict
    from torch_utils import _sg3abi
    from utils.inference_utils import get_average_image, run_on_batch
    from helpers import build_oracle_generator
    G = build_product_generator('Rmini')
    opts = types.SimpleNamespace(encoder_type='ProgressiveBackboneEncoder', input_nc=6, n_styles=int(G.num_ws), checkpoint_path=None,
                                 n_iters_per_batch=3, resize_outputs=False, sgxl=False)
    net = e4e(opts, decoder=G)
    assert type(net.encoder).__name__ == 'ProgressiveBackboneEncoder' and type(net).forward is type(net).__mro__[1].forward
    man = {k: list(v.shape) for k, v in net.encoder.state_dict().items()}
    enc_sd = {k: np.asarray(v) for k, v in synth_encoder_state_dict(man, seed=0).items()}
    net.encoder.load_state_dict({k: torch.from_numpy(v) for k, v in enc_sd.items()})
    net.encoder.progressive_stage = stage
    net = net.eval().requires_grad_(False).to(DEV)
    gen_sd, sched = build_oracle_generator('Rmini')
    lat_avg = gen_sd['mapping.w_avg']
    n_styles = int(G.num_ws)

    def enc(x6):                                                            # restyle_e4e_encoders.py:79-89
        per = O.backbone_encoder(enc_sd, x6, n_styles=n_styles)
        w = np.repeat(per[:, :1], n_styles, axis=1).copy()
        for i in range(1, min(stage + 1, n_styles)):
            w[:, i] += per[:, i]
        return w

    x, lt = _frames(2, seed=9), _landmarks(2)
    avg_o = O.get_average_image(None, gen_sd, sched, lat_avg)
    imgs_o, lats_o, aligned_last_o = O.run_on_batch(None, gen_sd, sched, x, lat_avg, avg_o, 3, landmarks_transform=lt, encoder=enc)
    n0 = _sg3abi.launch_count
    with torch.no_grad():
        avg = get_average_image(net)
        assert maxabs(avg.cpu().numpy(), avg_o) <= 1e-5
        xt, ltt = torch.from_numpy(x).to(DEV), torch.from_numpy(lt).to(DEV)
        # e4e.forward itself: first step (6 channels, no latent) and a residual step, all three return forms
        x6 = torch.cat([xt, avg.unsqueeze(0).repeat(2, 1, 1, 1)], dim=1)
        first = net.forward(x6, latent=None, resize=True, return_latents=True)
        want0 = enc(x6.cpu().numpy()) + lat_avg.reshape(1, 1, -1)
        assert tuple(first[0].shape) == (2, 3, 256, 256)
        assert maxabs(first[1].cpu().numpy(), want0) <= 2e-4 * float(np.abs(want0).max())
        al, un, codes = net.forward(x6, latent=first[1], landmarks_transform=ltt, return_aligned_and_unaligned=True, return_latents=True, resize=False)
        want1 = enc(x6.cpu().numpy()) + first[1].cpu().numpy()
        assert maxabs(codes.cpu().numpy(), want1) <= 2e-4 * float(np.abs(want1).max())
        assert al.shape == un.shape and maxabs(al.cpu().numpy(), un.cpu().numpy()) > 1e-3          # the transforms move the image
        only = net.forward(first[1], input_code=True, resize=False)
        assert torch.is_tensor(only) and maxabs(only.cpu().numpy(), O.psp_forward(None, gen_sd, sched, first[1].cpu().numpy(), input_code=True, resize=False)[0]) <= 1e-4
        imgs, lats = run_on_batch(xt, net, opts, avg, landmarks_transform=ltt)
    assert _sg3abi.launch_count - n0 > 300, 'the loop did not run on the HIP kernels'
    for it in range(3):
        got = np.stack([lats[i][it] for i in range(2)])
        assert maxabs(got, lats_o[it]) <= 2e-4 * float(np.abs(lats_o[it]).max()), (it, maxabs(got, lats_o[it]))
        img = torch.stack([imgs[i][it] for i in range(2)]).cpu().numpy()
        assert maxabs(img, imgs_o[it]) <= 1e-4, (it, maxabs(img, imgs_o[it]))
    if stage < n_styles - 1:                                                # styles beyond the stage carry the base code only
        assert np.array_equal(lats[0][0][stage + 1] - lat_avg.reshape(-1), lats[0][0][n_styles - 1] - lat_avg.reshape(-1))


def test_run_on_batch_survives_weight_updates_after_graph_capture():
    """a captured ReStyle step goes stale when the decoder is tuned (PTI / load_state_dict / an optimiser step).
    run_on_batch must then drop the graph and give the eager loop's result for the NEW weights instead of raising; a
    ShardedInversion built afterwards re-captures."""
    from sg3_runtime import GraphedReStyleStep
    from sg3_runtime.sharded import ShardedInversion
    from utils.inference_utils import get_average_image, run_on_batch
    net, opts, *_ = build_restyle_pair('Rmini', device=DEV, n_iters=3)
    x = torch.from_numpy(_frames(3, seed=6)).to(DEV)
    with torch.no_grad():
        avg = get_average_image(net)
        net.graphed_step = GraphedReStyleStep(net, 3)
        assert not net.graphed_step.is_stale()
        before = run_on_batch(x, net, opts, avg)
    w = getattr(net.decoder.synthesis, net.decoder.synthesis.layer_names[2]).weight
    opt = torch.optim.SGD([w.requires_grad_(True)], lr=0.5)
    w.grad = torch.ones_like(w) * 0.05
    opt.step()                                                                          # in-place update: _version moves on
    w.requires_grad_(False)
    assert net.graphed_step.is_stale()
    with torch.no_grad():
        after = run_on_batch(x, net, opts, avg)                                        # must not raise
        assert net.graphed_step is None                                                # dropped, logged once
        eager = run_on_batch(x, net, opts, avg)
    for i in range(3):
        assert np.array_equal(after[1][i][2], eager[1][i][2])
        assert torch.equal(after[0][i][2], eager[0][i][2])
    assert maxabs(after[0][0][2].cpu().numpy(), before[0][0][2].cpu().numpy()) > 1e-4   # the new weights are what was rendered
    # the device-resident form used by the sharded path: final latents only, equal to the loop's last step
    with torch.no_grad():
        sh = ShardedInversion(net, opts, batch_size=3)
        assert net.graphed_step is not None and not net.graphed_step.is_stale()        # re-captured on the new weights
        lat, (a, b) = sh.invert(x.cpu())
        eager = run_on_batch(x, net, opts, sh.avg_image)                                   # the average image of the NEW weights
    assert (a, b) == (0, 3) and lat.is_cuda
    assert maxabs(lat.cpu().numpy(), np.stack([eager[1][i][2] for i in range(3)])) <= 1e-5 * max(1.0, float(np.abs(eager[1][0][2]).max()))


@pytest.mark.parametrize('with_transforms', [False, True])
def test_sharded_inversion_ragged_batches_match_the_eager_loop(with_transforms):
    """ShardedInversion.invert on 5 frames at batch size 2: two full batches replay the captured ReStyle step (latents stay on the
    device, the encoder's range flag is read once after them), the ragged last frame runs the eager loop; every frame's latent
    equals the per-frame eager run_on_batch result (reference inversion/video/inference_on_video.py:119-145 inverts frame by frame),
    with and without per-frame landmark transforms."""
    from sg3_runtime.sharded import ShardedInversion
    from utils.inference_utils import run_on_batch
    net, opts, *_ = build_restyle_pair('Rmini', device=DEV, n_iters=3)
    frames = torch.from_numpy(_frames(5, seed=11))
    lts = None
    if with_transforms:
        r = np.random.RandomState(5)
        lts = torch.from_numpy(np.stack([np.array([[np.cos(a), -np.sin(a), tx], [np.sin(a), np.cos(a), ty], [0, 0, 1]], dtype=np.float32)
                                         for a, tx, ty in zip(r.uniform(-0.2, 0.2, 5), r.uniform(-0.1, 0.1, 5), r.uniform(-0.1, 0.1, 5))]))
    with torch.no_grad():
        sh = ShardedInversion(net, opts, batch_size=2)
        assert net.graphed_step is not None and net.graphed_step.batch == 2
        lat, span = sh.invert(frames, lts)
        assert span == (0, 5) and tuple(lat.shape)[0] == 5 and lat.is_cuda
        keep, net.graphed_step = net.graphed_step, None                 # the eager loop, frame by frame
        try:
            for i in range(5):
                lt = None if lts is None else lts[i:i + 1].to(DEV)
                _, ref = run_on_batch(frames[i:i + 1].to(DEV), net, opts, sh.avg_image, landmarks_transform=lt)
                want = ref[0][-1]
                assert maxabs(lat[i].cpu().numpy(), want) <= 2e-5 * max(1.0, float(np.abs(want).max())), i
        finally:
            net.graphed_step = keep


def test_graph_replay_after_a_ragged_tail_and_after_eval_reads_live_memory():
    """ADVICE r3 (high / medium): the captured ReStyle step bakes the addresses of the encoder's strip image and packed weights.
    (i) An eager call at another batch size (the ragged tail of invert(), run_on_batch(x[:1])) must not replace the strip the graph
    replays into; (ii) `net.eval()` / `invalidate_packed()` after the capture drops the encoder's references to its packs while
    is_stale() stays False (no weight changed), so the graph must hold them itself.  Both: invert -> eager odd batch -> (eval,
    empty_cache, allocations that would reuse freed blocks) -> invert again == the per-frame eager loop."""
    from sg3_runtime.sharded import ShardedInversion
    from utils.inference_utils import run_on_batch
    net, opts, *_ = build_restyle_pair('Rmini', device=DEV, n_iters=3)
    frames = torch.from_numpy(_frames(5, seed=23))
    with torch.no_grad():
        sh = ShardedInversion(net, opts, batch_size=2)
        g = net.graphed_step
        assert g is not None and g.batch == 2
        first, _ = sh.invert(frames)                                    # two replays, then the ragged frame eagerly (batch 1: another strip)
        first = first.clone()
        run_on_batch(frames[:1].to(DEV), net, opts, sh.avg_image)       # one more eager call at batch 1
        net.eval()                                                      # -> invalidate_packed(): the encoder forgets its packs
        assert net.encoder._packed is None and not g.is_stale()
        run_on_batch(frames[:1].to(DEV), net, opts, sh.avg_image)       # re-packs (new tensors, new strips)
        torch.cuda.synchronize()
        torch.cuda.empty_cache()
        junk = [torch.full([1 << 20], float('nan'), device=DEV) for _ in range(64)]     # anything freed would be reused and poisoned here
        assert net.graphed_step is g
        second, _ = sh.invert(frames)
        torch.cuda.synchronize()
        del junk
        assert torch.isfinite(second).all()
        assert maxabs(second.cpu().numpy(), first.cpu().numpy()) <= 1e-6
        keep, net.graphed_step = net.graphed_step, None
        try:
            for i in range(5):
                _, ref = run_on_batch(frames[i:i + 1].to(DEV), net, opts, sh.avg_image)
                want = ref[0][-1]
                assert maxabs(second[i].cpu().numpy(), want) <= 2e-5 * max(1.0, float(np.abs(want).max())), i
        finally:
            net.graphed_step = keep
