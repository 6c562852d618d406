"""GPU: callers of the synthesis path through the HIP kernels against the oracle restatement of the reference
(FOV Expander, video post-processing, StyleCLIP StyleSpace sweep); full-size properties for the batched forms."""
import types

import numpy as np
import pytest
import torch

from callers_common import landmark, styleclip_case, sweep_opts
from helpers import build_oracle_generator, build_product_generator, maxabs
from synth_weights import synth_ws

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


@pytest.mark.parametrize('cfg', ['Ttiny', 'Rtiny'])
def test_fov_expander_matches_oracle(cfg):
    from oracle import oracle as O
    from utils.fov_expansion import Expander
    G = build_product_generator(cfg, device=DEV)
    sd, sched = build_oracle_generator(cfg)
    ws = synth_ws(2, G.num_ws, G.w_dim, seed=4)
    lt = landmark()
    got = Expander(G, force_fp32=True).generate_expanded_image(ws=torch.from_numpy(ws).to(DEV), landmark_t=lt, pixels_right=8, pixels_left=4, pixels_top=6, pixels_bottom=0)
    ref = O.expand_fov(sd, sched, ws, lt, pixels_right=8, pixels_left=4, pixels_top=6, pixels_bottom=0)
    assert got.is_cuda and tuple(got.shape) == ref.shape
    assert maxabs(got.cpu().numpy(), ref) <= 1e-4


def test_video_postprocessing_matches_oracle():
    from oracle import oracle as O
    from inversion.video import post_processing as pp
    G = build_product_generator('Ttiny', device=DEV)
    sd, sched = build_oracle_generator('Ttiny')
    lat = synth_ws(7, G.num_ws, G.w_dim, seed=9)
    results = {'result_latents': {f'{i:04d}': lat[i] for i in range(7)}, 'landmarks_transforms': [None] * 7}
    opts = types.SimpleNamespace(expansion_amounts=[4, 2, 0, 3], landmarks_transforms_path=None)
    frames = pp.postprocess_and_smooth_inversions(results, types.SimpleNamespace(decoder=G), opts, frames_per_batch=2, force_fp32=True)
    smoothed = O.postprocess_latents(lat)
    ident = np.linalg.inv(O.make_transform((0, 0), 0))
    assert len(frames) == 3
    for f, w in zip(frames, smoothed):
        ref = O.expand_fov(sd, sched, w[None].astype(np.float32), ident, pixels_left=4, pixels_right=2, pixels_top=0, pixels_bottom=3)[0]
        ref = (np.clip((ref.transpose(1, 2, 0) + 1) / 2, 0, 1) * 255).astype(np.uint8)
        assert np.abs(f.astype(np.int32) - ref.astype(np.int32)).max() <= 1


@pytest.mark.parametrize('cfg', ['Ttiny', 'Rtiny'])
def test_styleclip_sweep_matches_oracle(cfg):
    from oracle import oracle as O
    from editing.styleclip_global_directions.edit import edit_image
    from editing.styleclip_global_directions.global_direction import StyleCLIPGlobalDirection
    G = build_product_generator(cfg, device=DEV)
    sd, sched = build_oracle_generator(cfg)
    lat = synth_ws(1, G.num_ws, G.w_dim, seed=12)[0]
    with torch.no_grad():
        s_avg = G.synthesis.W2S(G.mapping.w_avg.unsqueeze(0).repeat(1, G.num_ws, 1))
    delta_i_c, delta_i, s_std = styleclip_case(s_avg)
    calc = StyleCLIPGlobalDirection(torch.from_numpy(delta_i_c).to(DEV), {k: torch.from_numpy(v).to(DEV) for k, v in s_std.items()}, ['{}'], s_avg)
    opts = sweep_opts()
    betas = np.linspace(opts.beta_min, opts.beta_max, opts.num_betas)
    alphas = np.linspace(opts.alpha_min, opts.alpha_max, opts.num_alphas)
    dirs = [calc.get_delta_s_from_delta_i(torch.from_numpy(delta_i).to(DEV), beta) for beta in betas]
    lt = landmark().astype(np.float32)
    results, latents = edit_image(lat, lt, G, calc, opts, directions=dirs, max_batch=4, force_fp32=True)
    ref = O.styleclip_sweep(sd, sched, lat, delta_i_c, delta_i, s_std, alphas, betas, transform=lt)
    assert results.is_cuda and tuple(results.shape) == ref.shape
    assert maxabs(results.cpu().numpy(), ref) <= 1e-4


def test_batched_forms_equal_per_item_forms_full_size():
    """R-512 (the StyleCLIP config): the batched sweep equals batch-1 calls, and the batched 9-tile FOV expansion equals
    per-tile synthesis (tile seams continue the centre image: alias-free translation equivariance)."""
    from editing.styleclip_global_directions.edit import render_sweep
    from utils.fov_expansion import Expander
    G = build_product_generator('R512', device=DEV)
    ws = torch.from_numpy(synth_ws(3, G.num_ws, G.w_dim, seed=5)).to(DEV)
    with torch.no_grad():
        sweep = G.synthesis.W2S(ws)
        sweep['L5_84_1024'] = sweep['L5_84_1024'] * 1.5
        batched = render_sweep(G, sweep, max_batch=3, force_fp32=True)
        single = torch.cat([G.synthesis(None, all_s={c: v[i:i + 1] for c, v in sweep.items()}, force_fp32=True) for i in range(3)])
    assert maxabs(batched.cpu().numpy(), single.cpu().numpy()) <= 1e-5
    ident = np.eye(3)
    ex = Expander(G, force_fp32=True)
    big = ex.generate_expanded_image(ws=ws[:1], landmark_t=ident, pixels_left=64, pixels_right=32, pixels_top=16, pixels_bottom=48)
    assert tuple(big.shape) == (1, 3, 512 + 64, 512 + 96)
    tiles = Expander._get_transforms(512, 32, 64, 16, 48)
    with torch.no_grad():
        G.synthesis.input.transform = torch.from_numpy(tiles[1]).float().to(DEV)          # left tile on its own
        left = G.synthesis(ws[:1], None, force_fp32=True)
    assert maxabs(big[:, :, 16:16 + 512, :64].cpu().numpy(), left[:, :, :, :64].cpu().numpy()) <= 1e-5
    # the left tile is the centre image shifted by 64 px: where both exist they agree (interior, away from the margin)
    centre = big[:, :, 16:16 + 512, 64:64 + 512]
    assert maxabs(left[:, :, 100:400, 64 + 100:64 + 300].cpu().numpy(), centre[:, :, 100:400, 100:300].cpu().numpy()) <= 5e-3


def test_styleclip_sweep_batch32_equals_single_renders():
    """BASELINE configs[4] shape: AFHQ-512 config-R, the 5 x 11 edit sweep rendered in StyleSpace batches of 32 (the reference
    renders the 55 edits one by one, edit.py:136-160): every image of the batched sweep equals its batch-1 render."""
    from editing.styleclip_global_directions.edit import render_sweep
    G = build_product_generator('R512', device=DEV)
    w = torch.from_numpy(synth_ws(1, G.num_ws, G.w_dim, seed=8)).to(DEV)
    with torch.no_grad():
        base = G.synthesis.W2S(w)
        r = np.random.RandomState(13)
        sweep = {c: v.repeat(55, 1) + torch.from_numpy((0.3 * r.randn(55, 1) * (r.rand(1, v.shape[1]) < 0.05)).astype(np.float32)).to(DEV) for c, v in base.items()}
        batched = render_sweep(G, sweep, max_batch=32, force_fp32=True)
        assert tuple(batched.shape) == (55, 3, 512, 512)
        for i in (0, 31, 32, 54):          # both batches (32 + 23), first and last of each
            single = G.synthesis(None, all_s={c: v[i:i + 1] for c, v in sweep.items()}, noise_mode='const', force_fp32=True)
            assert maxabs(batched[i:i + 1].cpu().numpy(), single.cpu().numpy()) <= 1e-5, i
        assert maxabs(batched[0].cpu().numpy(), batched[54].cpu().numpy()) > 1e-3


@pytest.mark.parametrize('cfg', ['Ttiny', 'Rtiny'])
def test_callers_match_reference_fixtures(cfg):
    """FOV Expander and the StyleCLIP sweep on the HIP path against the outputs of the reference's own Expander / edit_image
    (tests/golden/callers.npz); the video post-processing frames likewise (config T)."""
    from callers_common import styleclip_case
    from editing.styleclip_global_directions.edit import edit_image
    from editing.styleclip_global_directions.global_direction import StyleCLIPGlobalDirection
    from helpers import golden
    from inversion.video import post_processing as pp
    from utils.fov_expansion import Expander
    g = golden('callers')
    G = build_product_generator(cfg, device=DEV)
    ws = synth_ws(2, G.num_ws, G.w_dim, seed=4)
    got = Expander(G, force_fp32=True).generate_expanded_image(ws=torch.from_numpy(ws).to(DEV), landmark_t=landmark(), pixels_right=8, pixels_left=4, pixels_top=6, pixels_bottom=0)
    assert maxabs(got.cpu().numpy(), g[f'fov/{cfg}/img']) <= 1e-4
    lat = synth_ws(1, G.num_ws, G.w_dim, seed=12)[0]
    G.synthesis.input.transform = torch.eye(3, device=DEV)
    with torch.no_grad():
        s_avg = G.synthesis.W2S(G.mapping.w_avg.unsqueeze(0).repeat(1, G.num_ws, 1))
    delta_i_c, delta_i, s_std = styleclip_case(s_avg)
    calc = StyleCLIPGlobalDirection(torch.from_numpy(delta_i_c).to(DEV), {k: torch.from_numpy(v).to(DEV) for k, v in s_std.items()}, ['{}'], s_avg)
    calc.get_delta_i = lambda prompts: torch.from_numpy(delta_i).to(DEV)
    results, _ = edit_image(lat, landmark().astype(np.float32), G, calc, sweep_opts(), max_batch=4, force_fp32=True)
    assert maxabs(results.cpu().numpy(), g[f'styleclip/{cfg}/results']) <= 1e-4
    if cfg == 'Ttiny':
        lat7 = synth_ws(7, G.num_ws, G.w_dim, seed=9)
        from oracle import oracle as O
        tr = [torch.from_numpy(np.linalg.inv(O.make_transform((0.02 * i, -0.01 * i), 2.0 * i))) for i in range(7)]
        results7 = {'result_latents': {f'{i:04d}': lat7[i] for i in range(7)}, 'landmarks_transforms': tr}
        vopts = types.SimpleNamespace(expansion_amounts=[4, 2, 0, 3], landmarks_transforms_path='given')
        frames = pp.postprocess_and_smooth_inversions(results7, types.SimpleNamespace(decoder=G), vopts, frames_per_batch=2, force_fp32=True)
        for f, w in zip(frames, g['video/frames']):
            assert np.abs(f.astype(np.int32) - w.astype(np.int32)).max() <= 1
