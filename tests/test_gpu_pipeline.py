"""GPU: the callers chained the way the reference's scripts chain them (inference_on_video -> post-processing / FOV
expansion, run_pti_images on the inverted latent, StyleCLIP sweep on the tuned generator), on one set of synthetic
frames with a 256x256 config-T generator and the IR-SE50 encoder.  Every stage is checked for shape, finiteness and the
stage-specific invariant; numerical parity of each stage has its own test file."""
import types

import numpy as np
import pytest
import torch

from helpers import build_product_generator
from synth_weights import synth_encoder_state_dict

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def test_invert_postprocess_tune_expand_edit():
    from models.setgan.encoder.psp3 import pSp
    from sg3_runtime.sharded import ShardedInversion
    from inversion.video.post_processing import postprocess_and_smooth_inversions
    from inversion.scripts.run_pti_images import PTI, default_opts
    from utils.fov_expansion import Expander
    from editing.styleclip_global_directions.edit import edit_image
    from editing.styleclip_global_directions.global_direction import StyleCLIPGlobalDirection
    from torch_utils import _sg3abi

    G = build_product_generator('T256', device=DEV)
    opts = types.SimpleNamespace(encoder_type='BackboneEncoder', input_nc=6, checkpoint_path=None, n_iters_per_batch=2, resize_outputs=False)
    net = pSp(opts, decoder=G)
    man = {k: list(v.shape) for k, v in net.encoder.state_dict().items()}
    net.encoder.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in synth_encoder_state_dict(man, seed=0).items()})
    net = net.eval().requires_grad_(False).to(DEV)
    n0 = _sg3abi.launch_count

    # 1. ReStyle inversion of 6 frames (world size 1: the all-gather is the identity)
    frames = torch.from_numpy(np.random.RandomState(2).uniform(-1, 1, size=(6, 3, 256, 256)).astype(np.float32))
    latents, span = ShardedInversion(net, opts, batch_size=3).invert(frames)
    assert span == (0, 6) and tuple(latents.shape) == (6, 16, 512) and bool(torch.isfinite(latents).all())

    # 2. video post-processing: fine layers averaged, 5-tap smoothing, expanded frames
    results = {'result_latents': {f'{i:03d}': latents[i].cpu().numpy() for i in range(6)}, 'landmarks_transforms': [None] * 6}
    vopts = types.SimpleNamespace(expansion_amounts=[16, 8, 0, 4], landmarks_transforms_path=None)
    out_frames = postprocess_and_smooth_inversions(results, net, vopts, frames_per_batch=2, force_fp32=True)
    assert len(out_frames) == 2 and out_frames[0].shape == (256 + 4, 256 + 24, 3) and out_frames[0].dtype == np.uint8

    # 3. pivotal tuning of the generator on frame 0 (target at the generator's resolution): the loss goes down.  The
    # reference tunes a freshly loaded generator (run_pti_images.py:91); here the shared one gets its identity transform back
    # (pSp.forward leaves a per-batch identity behind, psp3.py:61-64)
    G.synthesis.input.transform = torch.eye(3, device=DEV)
    G.requires_grad_(True)
    pti = PTI(default_opts(device=DEV, steps=4, learning_rate=3e-3, lpips_lambda=0.0))
    pti.optimize_model(G, latents[0].cpu().numpy(), frames[0])
    losses = [h[1] for h in pti.history]
    assert len(losses) == 4 and losses[-1] < losses[0] and all(np.isfinite(losses))
    G.requires_grad_(False)

    # 4. field-of-view expansion with the tuned generator: the centre of the canvas is the plain synthesis output
    G.synthesis.input.transform = torch.eye(3, device=DEV)
    ws = latents[:1].to(DEV)
    with torch.no_grad():
        plain = G.synthesis(ws, noise_mode='const', force_fp32=True)
    big = Expander(G, force_fp32=True).generate_expanded_image(ws=ws, landmark_t=np.eye(3), pixels_left=32, pixels_right=0, pixels_top=0, pixels_bottom=16)
    assert tuple(big.shape) == (1, 3, 256 + 16, 256 + 32)
    assert float((big[:, :, :256, 32:] - plain).abs().max()) <= 1e-5

    # 5. StyleSpace sweep: alpha = 0 reproduces the unedited image, the other edits differ from it
    with torch.no_grad():
        s_avg = G.synthesis.W2S(G.mapping.w_avg.unsqueeze(0).repeat(1, G.num_ws, 1))
    channels = sum(int(v.shape[1]) for v in s_avg.values())
    r = np.random.RandomState(5)
    calc = StyleCLIPGlobalDirection(torch.from_numpy((r.randn(channels, 16) / 4).astype(np.float32)).to(DEV),
                                    {k: torch.ones(int(v.shape[1]), device=DEV) for k, v in s_avg.items()}, ['{}'], s_avg)
    eopts = types.SimpleNamespace(alpha_min=-3.0, alpha_max=3.0, num_alphas=3, beta_min=0.1, beta_max=0.2, num_betas=2, neutral_text='a', target_text='b')
    di = torch.from_numpy(r.randn(16).astype(np.float32)).to(DEV)
    dirs = [calc.get_delta_s_from_delta_i(di / di.norm(), b) for b in np.linspace(0.1, 0.2, 2)]
    edits, lat = edit_image(latents[0].cpu().numpy(), np.eye(3, dtype=np.float32), G, calc, eopts, directions=dirs, max_batch=4, force_fp32=True)
    assert tuple(edits.shape) == (6, 3, 256, 256) and len(lat) == 6
    assert float((edits[1] - plain[0]).abs().max()) <= 1e-5 and float((edits[4] - plain[0]).abs().max()) <= 1e-5      # alpha = 0
    assert float((edits[0] - plain[0]).abs().max()) > 1e-3
    assert _sg3abi.launch_count - n0 > 500, 'the pipeline did not run on the HIP kernels'
