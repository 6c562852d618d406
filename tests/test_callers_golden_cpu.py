"""CPU: the callers of the synthesis path -- FOV Expander, video post-processing, StyleCLIP StyleSpace sweep, pSp.forward and the
ReStyle loop -- against fixtures produced by the REFERENCE's own caller code (tests/golden/make_golden_callers.py ->
callers.npz).  Both the product (plain PyTorch path) and the oracle restatement are held to them, which pins the oracle functions
the GPU parity tests use as their checker ("caller arithmetic unpinned")."""
import types

import numpy as np
import pytest
import torch

from callers_common import TinyEncoder, landmark, restyle_case, styleclip_case, sweep_opts, tiny_encoder_weights
from helpers import build_oracle_generator, build_product_generator, golden, maxabs
from synth_weights import make_user_transform, synth_ws


def _video_case():
    lat = synth_ws(7, 16, 32, seed=9)
    tr = [np.linalg.inv(_make_transform((0.02 * i, -0.01 * i), 2.0 * i)) for i in range(7)]
    return lat, tr


def _make_transform(translate, angle):
    from oracle import oracle as O
    return O.make_transform(translate, angle)


@pytest.mark.parametrize('cfg', ['Ttiny', 'Rtiny'])
def test_fov_expander(cfg):
    from oracle import oracle as O
    from utils.fov_expansion import Expander
    g = golden('callers')
    G = build_product_generator(cfg)
    ws = synth_ws(2, G.num_ws, G.w_dim, seed=4)
    got = Expander(G, force_fp32=True).generate_expanded_image(ws=torch.from_numpy(ws), landmark_t=landmark(), pixels_right=8, pixels_left=4, pixels_top=6, pixels_bottom=0)
    assert maxabs(got.numpy(), g[f'fov/{cfg}/img']) <= 1e-5
    sd, sched = build_oracle_generator(cfg)
    ref = O.expand_fov(sd, sched, ws, landmark(), pixels_right=8, pixels_left=4, pixels_top=6, pixels_bottom=0)
    assert maxabs(ref, g[f'fov/{cfg}/img']) <= 1e-5


def test_video_postprocessing():
    from oracle import oracle as O
    from inversion.video import post_processing as pp
    g = golden('callers')
    G = build_product_generator('Ttiny')
    lat, tr = _video_case()
    assert maxabs(pp.smooth_ws(lat.copy()), g['video/smooth_ws']) <= 1e-6 and maxabs(O.smooth_ws(lat.copy()), g['video/smooth_ws']) <= 1e-6
    results = {'result_latents': {f'{i:04d}': lat[i] for i in range(7)}, 'landmarks_transforms': [torch.from_numpy(t) for t in tr]}
    opts = types.SimpleNamespace(expansion_amounts=[4, 2, 0, 3], landmarks_transforms_path='given')
    _, sm_t = pp.smooth_latents_and_transforms(lat.copy(), results['landmarks_transforms'], opts, device='cpu')
    assert maxabs(sm_t.numpy(), g['video/smooth_transforms']) <= 1e-12
    frames = pp.postprocess_and_smooth_inversions(results, types.SimpleNamespace(decoder=G), opts, frames_per_batch=2, force_fp32=True)
    want = g['video/frames']
    assert len(frames) == want.shape[0] == 3
    for f, w in zip(frames, want):
        assert f.shape == w.shape and f.dtype == np.uint8 and np.abs(f.astype(np.int32) - w.astype(np.int32)).max() <= 1
    # oracle: fine-layer mean + smoothing, then the expander per frame under the smoothed transform
    sd, sched = build_oracle_generator('Ttiny')
    sm = O.postprocess_latents(lat)
    sm_tr = O.smooth_ws(np.stack(tr))
    for i in range(3):
        ref = O.expand_fov(sd, sched, sm[i][None].astype(np.float32), sm_tr[i], pixels_left=4, pixels_right=2, pixels_top=0, pixels_bottom=3)[0]
        ref = (np.clip((ref.transpose(1, 2, 0) + 1) / 2, 0, 1) * 255).astype(np.uint8)
        assert np.abs(ref.astype(np.int32) - want[i].astype(np.int32)).max() <= 1


@pytest.mark.parametrize('cfg', ['Ttiny', 'Rtiny'])
def test_styleclip_direction_and_sweep(cfg):
    from oracle import oracle as O
    from editing.styleclip_global_directions.edit import edit_image
    from editing.styleclip_global_directions.global_direction import StyleCLIPGlobalDirection
    g = golden('callers')
    G = build_product_generator(cfg)
    lat = synth_ws(1, G.num_ws, G.w_dim, seed=12)[0]
    with torch.no_grad():
        s_avg = G.synthesis.W2S(G.mapping.w_avg.unsqueeze(0).repeat(1, G.num_ws, 1))
    delta_i_c, delta_i, s_std = styleclip_case(s_avg)
    calc = StyleCLIPGlobalDirection(torch.from_numpy(delta_i_c), {k: torch.from_numpy(v) for k, v in s_std.items()}, ['{}'], s_avg,
                                    text_encoder=None)
    d = calc.get_delta_s_from_delta_i(torch.from_numpy(delta_i), 0.25)
    d_o = O.styleclip_delta_s(delta_i_c, delta_i, 0.25, s_std, {k: v.numpy() for k, v in s_avg.items()})
    for k in d:
        assert maxabs(d[k].numpy(), g[f'styleclip/{cfg}/delta_s/{k}']) <= 1e-6, k
        assert maxabs(d_o[k], g[f'styleclip/{cfg}/delta_s/{k}']) <= 1e-6, k
    opts = sweep_opts()
    calc.get_delta_i = lambda prompts: torch.from_numpy(delta_i)             # the text direction is an input (no CLIP weights offline)
    lt = landmark().astype(np.float32)
    results, latents = edit_image(lat, lt, G, calc, opts, max_batch=4, force_fp32=True)
    assert maxabs(results.numpy(), g[f'styleclip/{cfg}/results']) <= 1e-5
    assert maxabs(latents[-1]['input'].numpy(), g[f'styleclip/{cfg}/last_latent_input']) <= 1e-6
    sd, sched = build_oracle_generator(cfg)
    ref = O.styleclip_sweep(sd, sched, lat, delta_i_c, delta_i, s_std, np.linspace(opts.alpha_min, opts.alpha_max, opts.num_alphas),
                            np.linspace(opts.beta_min, opts.beta_max, opts.num_betas), transform=lt)
    assert maxabs(ref, g[f'styleclip/{cfg}/results']) <= 1e-5


def _tiny_encoder_numpy():
    w, b = tiny_encoder_weights()
    return lambda x: (x.mean(axis=(2, 3), dtype=np.float64).astype(np.float32) @ w.T + b).reshape(-1, 16, 512)


def build_loop_net(cfg, device='cpu', n_iters=3):
    """Product pSp wrapper around the stand-in encoder of the loop fixtures and a mini decoder."""
    from models.setgan.encoder.psp3 import pSp
    G = build_product_generator(cfg)
    opts = types.SimpleNamespace(encoder_type='BackboneEncoder', input_nc=6, checkpoint_path=None, n_iters_per_batch=n_iters, resize_outputs=False)
    net = pSp.__new__(pSp)
    torch.nn.Module.__init__(net)
    net.opts, net.n_styles, net.encoder, net.decoder = opts, 16, TinyEncoder(), G
    net.face_pool = torch.nn.AdaptiveAvgPool2d((256, 256))
    net.latent_avg = G.mapping.w_avg
    return net.eval().requires_grad_(False).to(device), opts


def check_loop_against_golden(net, opts, cfg, device, tol_img=1e-5, tol_lat=1e-5):
    from utils.inference_utils import get_average_image, run_on_batch
    g = golden('callers')
    x, tr = restyle_case(2)
    xt, trt = torch.from_numpy(x).to(device), torch.from_numpy(tr).to(device)
    with torch.no_grad():
        avg = get_average_image(net)
        assert maxabs(avg.cpu().numpy()[:, ::4, ::4], g[f'restyle/{cfg}/avg_image']) <= tol_img
        x6 = torch.cat([xt, avg.unsqueeze(0).repeat(2, 1, 1, 1)], dim=1)
        img0, lat0 = net.forward(x6, latent=None, return_latents=True, resize=False)
        img1, un1, lat1 = net.forward(x6, latent=lat0, landmarks_transform=trt, return_aligned_and_unaligned=True, resize=True)
        assert maxabs(img0.cpu().numpy(), g[f'psp/{cfg}/img0']) <= tol_img and maxabs(lat0.cpu().numpy(), g[f'psp/{cfg}/lat0']) <= tol_lat
        assert maxabs(img1.cpu().numpy()[:, :, ::4, ::4], g[f'psp/{cfg}/img1']) <= tol_img
        assert maxabs(un1.cpu().numpy()[:, :, ::4, ::4], g[f'psp/{cfg}/un1']) <= tol_img
        assert maxabs(lat1.cpu().numpy(), g[f'psp/{cfg}/lat1']) <= tol_lat
        for key, lt in (('off', None), ('on', trt)):
            imgs, lats = run_on_batch(xt, net, opts, avg, landmarks_transform=lt)
            got_i = np.stack([np.stack([imgs[i][it].cpu().numpy() for i in range(2)]) for it in range(3)])
            got_l = np.stack([np.stack([lats[i][it] for i in range(2)]) for it in range(3)])
            assert maxabs(got_i, g[f'restyle/{cfg}/{key}/images']) <= tol_img, key
            assert maxabs(got_l, g[f'restyle/{cfg}/{key}/latents']) <= tol_lat * max(1.0, float(np.abs(got_l).max())), key


@pytest.mark.parametrize('cfg', ['Tmini', 'Rmini'])
def test_psp_forward_and_restyle_loop_product(cfg):
    net, opts = build_loop_net(cfg)
    check_loop_against_golden(net, opts, cfg, 'cpu')


@pytest.mark.parametrize('cfg', ['Tmini', 'Rmini'])
def test_psp_forward_and_restyle_loop_oracle(cfg):
    from oracle import oracle as O
    g = golden('callers')
    gen_sd, sched = build_oracle_generator(cfg)
    enc = _tiny_encoder_numpy()
    lat_avg = gen_sd['mapping.w_avg']
    x, tr = restyle_case(2)
    avg = O.get_average_image(None, gen_sd, sched, lat_avg)
    assert maxabs(avg[:, ::4, ::4], g[f'restyle/{cfg}/avg_image']) <= 1e-5
    x6 = np.concatenate([x, np.repeat(avg[None], 2, axis=0)], axis=1)
    img0, _, lat0 = O.psp_forward(None, gen_sd, sched, x6, latent=None, latent_avg=lat_avg, resize=False, encoder=enc)
    img1, un1, lat1 = O.psp_forward(None, gen_sd, sched, x6, latent=lat0, latent_avg=lat_avg, resize=True, landmarks_transform=tr, encoder=enc)
    assert maxabs(img0, g[f'psp/{cfg}/img0']) <= 1e-5 and maxabs(lat0, g[f'psp/{cfg}/lat0']) <= 1e-5
    assert maxabs(img1[:, :, ::4, ::4], g[f'psp/{cfg}/img1']) <= 1e-5 and maxabs(un1[:, :, ::4, ::4], g[f'psp/{cfg}/un1']) <= 1e-5
    assert maxabs(lat1, g[f'psp/{cfg}/lat1']) <= 1e-5
    imgs_on, lats_on, aligned_last = O.run_on_batch(None, gen_sd, sched, x, lat_avg, avg, 3, landmarks_transform=tr, encoder=enc)
    imgs_off, lats_off, _ = O.run_on_batch(None, gen_sd, sched, x, lat_avg, avg, 3, landmarks_transform=None, encoder=enc)
    assert maxabs(np.stack(imgs_on), g[f'restyle/{cfg}/on/images']) <= 1e-5 and maxabs(np.stack(imgs_off), g[f'restyle/{cfg}/off/images']) <= 1e-5
    assert maxabs(np.stack(lats_on), g[f'restyle/{cfg}/on/latents']) <= 1e-4 and maxabs(np.stack(lats_off), g[f'restyle/{cfg}/off/latents']) <= 1e-4
    assert maxabs(aligned_last, g[f'restyle/{cfg}/off/images'][-1]) <= 1e-5      # the aligned last step IS the transform-free result


def test_tensor2im_matches_reference_frames():
    """utils.common.tensor2im (reference :39-45) on the expanded frames of the video fixture: the same uint8 pixels the reference
    wrote (the fixture's frames ARE np.array(tensor2im(...)) of the reference)."""
    from utils.common import tensor2im
    from utils.fov_expansion import Expander
    from oracle import oracle as O
    g = golden('callers')
    G = build_product_generator('Ttiny')
    lat, tr = _video_case()
    sm = O.postprocess_latents(lat)
    sm_tr = O.smooth_ws(np.stack(tr))
    im = Expander(G, force_fp32=True).generate_expanded_image(ws=torch.from_numpy(sm[:1].astype(np.float32)), landmark_t=sm_tr[0], pixels_left=4, pixels_right=2, pixels_top=0, pixels_bottom=3)
    frame = np.array(tensor2im(im[0]))
    assert frame.dtype == np.uint8 and frame.shape == g['video/frames'][0].shape
    assert np.abs(frame.astype(np.int32) - g['video/frames'][0].astype(np.int32)).max() <= 1
