"""GPU: pivotal tuning through the HIP kernels (forward: fused kernels with sign write; backward: sign-read adjoints and
convolution gradients) against the golden vectors produced with the reference generator and autograd on CPU."""
import numpy as np
import pytest
import torch

from helpers import golden, maxabs
from synth_weights import synth_ws
from test_pti_cpu import check_weights, pti_target, tunable_generator, video_case

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


@pytest.mark.parametrize('cfg', ['Ttiny', 'Rtiny'])
def test_pti_single_image_matches_reference(cfg):
    from inversion.scripts.run_pti_images import PTI, default_opts
    from torch_utils import _sg3abi
    g = golden('pti')
    G = tunable_generator(cfg, device=DEV)
    pti = PTI(default_opts(device=DEV, steps=4, learning_rate=3e-3, lpips_lambda=0.0))
    code = synth_ws(1, G.num_ws, G.w_dim, seed=6)[0]
    n0 = _sg3abi.launch_count
    pti.optimize_model(G, code, pti_target(G.img_resolution, 40))
    assert _sg3abi.launch_count - n0 >= 4 * 30, 'HIP kernels did not run'
    losses = np.asarray([h[1] for h in pti.history])
    assert np.abs(losses - g[f'{cfg}/image/losses']).max() <= 1e-5, (losses, g[f'{cfg}/image/losses'])
    with torch.no_grad():
        final = G.synthesis(torch.from_numpy(code)[None].to(DEV), noise_mode='const', force_fp32=True).cpu().numpy()
    assert maxabs(final, g[f'{cfg}/image/final']) <= 5e-4
    sd = {k: v.cpu() for k, v in G.state_dict().items()}
    wkey = [k for k in g.files if k.startswith(f'{cfg}/image/synthesis.')][0]
    check_weights(sd[wkey.split('/')[-1]].numpy(), g[wkey], 3e-3, 4)
    check_weights(sd[[k for k in sd if k.startswith('synthesis.L13_') and k.endswith('.bias')][0]].numpy(), g[f'{cfg}/image/bias_L13'], 3e-3, 4)
    check_weights(sd[[k for k in sd if k.startswith('synthesis.L5_') and k.endswith('affine.weight')][0]].numpy()[:4], g[f'{cfg}/image/affine_L5'], 3e-3, 4)


def test_video_pti_matches_reference():
    from inversion.scripts.run_pti_images import default_opts
    from inversion.video import run_pti_video as rv
    g = golden('pti')
    real = rv.epoch_order
    rv.epoch_order = lambda n, device: torch.arange(n)
    try:
        G = tunable_generator('Ttiny', device=DEV)
        codes, targets, tr = video_case()
        v = rv.VideoPTI(default_opts(device=DEV, steps=4, learning_rate=3e-3, lpips_lambda=0.0, batch_size=2))
        v.optimize_model(G, codes, targets, landmarks_transforms=tr)
    finally:
        rv.epoch_order = real
    losses = np.asarray([h[1] for h in v.history])
    assert np.abs(losses - g['Ttiny/video/losses']).max() <= 1e-5
    with torch.no_grad():
        G.synthesis.input.transform = torch.from_numpy(tr[:1]).float().to(DEV)
        final = G.synthesis(torch.from_numpy(codes[:1]).to(DEV), noise_mode='const', force_fp32=True).cpu().numpy()
    assert maxabs(final, g['Ttiny/video/final']) <= 5e-4
    check_weights(G.state_dict()['synthesis.L0_36_12.weight'].cpu().numpy(), g['Ttiny/video/synthesis.L0_36_12.weight'], 3e-3, 4)
