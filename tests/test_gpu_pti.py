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


def _layer_cases(cfg, names):
    """(name, x side, up, padding, fu, fd) of the named layers of a full-size generator (filters as the product designs them:
    pinned against the reference's taps in test_product_cpu / golden filters.npz)."""
    from helpers import build_product_generator
    G = build_product_generator(cfg)
    out = []
    for name in G.synthesis.layer_names:
        if name.split('_')[0] in names:
            L = getattr(G.synthesis, name)
            out.append((name, int(L.in_size[0]) + L.conv_kernel - 1, L.up_factor, list(L.padding), L.up_filter.numpy(), L.down_filter.numpy()))
    return out


@pytest.mark.parametrize('cfg,names', [('T1024', ('L10', 'L11', 'L13')), ('R1024', ('L10', 'L11', 'L13'))])
def test_pti_filter_kernels_at_1024_geometry(cfg, names):
    """The 1024^2 sign-writing forward and sign-reading adjoint (the PTI path of BASELINE configs[3]) on a cropped plane set
    (2 channels of the real layer geometry: the kernels treat planes independently): output, dx and db against autograd through
    the reference formulation on the CPU (`impl='ref'`, reference filtered_lrelu.py:122-154)."""
    from golden_cases import rand
    from torch_utils.ops import filtered_lrelu as fl
    for name, side, up, pad, fu, fd in _layer_cases(cfg, names):
        xn, bn = rand(21, 1, 2, side, side), rand(22, 2)
        kw = dict(up=up, down=2, padding=pad, gain=float(np.sqrt(2)), slope=0.2, clamp=256, flip_filter=False)
        xr = torch.from_numpy(xn).requires_grad_(True); br = torch.from_numpy(bn).requires_grad_(True)
        yr = fl.filtered_lrelu(xr, torch.from_numpy(fu), torch.from_numpy(fd), br, impl='ref', **kw)
        gy = rand(23, *yr.shape)
        loss_r = (yr * torch.from_numpy(gy)).sum()
        loss_r.backward()
        x = torch.from_numpy(xn).to(DEV).requires_grad_(True); b = torch.from_numpy(bn).to(DEV).requires_grad_(True)
        y = fl.filtered_lrelu(x, torch.from_numpy(fu).to(DEV), torch.from_numpy(fd).to(DEV), b, **kw)
        loss = (y * torch.from_numpy(gy).to(DEV)).sum()
        loss.backward()
        assert tuple(y.shape) == tuple(yr.shape), name
        assert maxabs(y.detach().cpu().numpy(), yr.detach().numpy()) <= 2e-5, name
        assert abs(loss.item() - loss_r.item()) <= 1e-4 * max(1.0, abs(loss_r.item())), name
        assert maxabs(x.grad.cpu().numpy(), xr.grad.numpy()) <= 5e-5, name
        assert maxabs(b.grad.cpu().numpy(), br.grad.numpy()) <= 1e-4 * max(1.0, float(br.grad.abs().max())), name


@pytest.mark.parametrize('cfg', ['T1024', 'R1024'])
def test_pti_step_at_full_size(cfg):
    """One pivotal-tuning step at FFHQ-1024 on the HIP path: the sign-writing training forward reproduces the inference forward,
    the loss gradient agrees with central differences of the loss along a random direction of the last layers' biases and
    weights, and one Adam step lowers the loss."""
    from helpers import build_product_generator
    G = build_product_generator(cfg, device=DEV)
    w = torch.from_numpy(synth_ws(1, G.num_ws, G.w_dim, seed=3)).to(DEV)
    with torch.no_grad():
        ref_img = G.synthesis(w, noise_mode='const', force_fp32=True)
    target = (0.5 * ref_img + 0.1).detach()
    G.requires_grad_(True)
    params = list(G.synthesis.parameters())[3:]
    out = G.synthesis(w, noise_mode='const', force_fp32=True)
    assert maxabs(out.detach().cpu().numpy(), ref_img.cpu().numpy()) <= 2e-5          # training forward == inference forward
    loss = torch.nn.functional.mse_loss(out, target)
    grads = torch.autograd.grad(loss, params)
    assert all(bool(torch.isfinite(g).all()) for g in grads)
    # directional derivative along a random direction over the parameters of the last three layers
    names = [n for n, _ in G.synthesis.named_parameters()][3:]
    pick = [i for i, n in enumerate(names) if n.split('.')[0].split('_')[0] in ('L12', 'L13', 'L14')]
    r = torch.Generator(device='cpu').manual_seed(5)
    dirs = {i: torch.randn(params[i].shape, generator=r).to(DEV) for i in pick}
    analytic = sum(float((grads[i].double() * dirs[i].double()).sum()) for i in pick)

    def loss_at(eps):
        with torch.no_grad():
            for i in pick:
                params[i].add_(dirs[i], alpha=eps)
            val = torch.nn.functional.mse_loss(G.synthesis(w, noise_mode='const', force_fp32=True).double(), target.double()).item()
            for i in pick:
                params[i].sub_(dirs[i], alpha=eps)
        return val
    eps = 1e-3
    numeric = (loss_at(eps) - loss_at(-eps)) / (2 * eps)
    assert abs(numeric - analytic) <= 2e-2 * max(abs(analytic), 1e-6), (numeric, analytic)
    opt = torch.optim.Adam(params, lr=3e-4)
    opt.zero_grad()
    for p_, g_ in zip(params, grads):
        p_.grad = g_
    opt.step()
    with torch.no_grad():
        after = torch.nn.functional.mse_loss(G.synthesis(w, noise_mode='const', force_fp32=True), target).item()
    assert after < loss.item(), (after, loss.item())


def test_fused_adam_step_advances_the_version_counters_the_inference_caches_key_on():
    """`tuning_optimizer` fuses Adam into one launch on a GPU; torch's fused update leaves the parameters' version counters alone, so
    the subclass advances them -- otherwise the packed-weight caches of the inference path (and the graphs' staleness check) would
    keep serving the weights from before the step.  Checked end to end: inference, one tuning step, inference again == the same
    weights loaded into a fresh generator."""
    from helpers import build_product_generator
    from inversion.scripts.run_pti_images import tuning_optimizer
    G = build_product_generator('Tmini', device=DEV)
    ws = torch.from_numpy(np.random.RandomState(2).randn(1, G.num_ws, G.w_dim).astype(np.float32)).to(DEV)
    with torch.no_grad():
        before = G.synthesis(ws, noise_mode='const', force_fp32=True).clone()          # fills the inference caches
    G.requires_grad_(True)
    params = list(G.synthesis.parameters())[3:]
    opt = tuning_optimizer(params, lr=1e-2)
    assert type(opt).__name__ == '_FusedAdam'
    versions = [p._version for p in params]
    out = G.synthesis(ws, noise_mode='const', force_fp32=True)
    out.square().mean().backward()
    opt.step()
    assert all(p._version > v for p, v in zip(params, versions))
    G.requires_grad_(False)
    with torch.no_grad():
        after = G.synthesis(ws, noise_mode='const', force_fp32=True)
    fresh = build_product_generator('Tmini', device=DEV)
    fresh.load_state_dict(G.state_dict())
    with torch.no_grad():
        ref = fresh.synthesis(ws, noise_mode='const', force_fp32=True)
    assert float((after - before).abs().max()) > 1e-4                                  # the step did change the image
    assert float((after - ref).abs().max()) <= 1e-5
