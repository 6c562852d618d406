"""CPU: pivotal tuning (PTI / VideoPTI) against golden vectors produced with the reference generator and autograd graph
(tests/golden/make_golden_pti.py); data-parallel VideoPTI over gloo (world_size 2) equals the single-process run."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from helpers import build_product_generator, golden, maxabs
from synth_weights import make_user_transform, synth_ws


def pti_target(res, seed):
    r = np.random.RandomState(seed)
    yy, xx = np.meshgrid(np.linspace(-1, 1, res), np.linspace(-1, 1, res), indexing='ij')
    base = np.stack([np.sin(3 * xx + s) * np.cos(2 * yy - s) for s in (0.0, 0.7, 1.9)])
    return (0.6 * base + 0.05 * r.randn(3, res, res)).astype(np.float32)


def tunable_generator(cfg, device='cpu'):
    G = build_product_generator(cfg, device=device)
    G.requires_grad_(True)
    return G


def check_weights(got, ref, lr, steps):
    """Adam moves every weight by <= lr per step; where the gradient is ~0 its sign is noise, so a small share of the
    entries may differ by up to 2 * lr * steps while the bulk agrees tightly."""
    d = np.abs(np.asarray(got, np.float64) - np.asarray(ref, np.float64))
    assert d.max() <= 2 * lr * steps + 1e-6
    assert np.mean(d > 2e-5) <= 0.02, float(np.mean(d > 2e-5))


@pytest.mark.parametrize('cfg', ['Ttiny', 'Rtiny'])
def test_pti_single_image_matches_reference(cfg):
    from inversion.scripts.run_pti_images import PTI, default_opts
    g = golden('pti')
    G = tunable_generator(cfg)
    opts = default_opts(device='cpu', steps=4, learning_rate=3e-3, lpips_lambda=0.0)
    pti = PTI(opts)
    assert len(list(G.synthesis.parameters())) - len(pti.get_optimizer(G).param_groups[0]['params']) == 3
    code = synth_ws(1, G.num_ws, G.w_dim, seed=6)[0]
    pti.optimize_model(G, code, pti_target(G.img_resolution, 40), image_name='x.jpg')
    losses = np.asarray([h[1] for h in pti.history])
    assert np.abs(losses - g[f'{cfg}/image/losses']).max() <= 2e-6, (losses, g[f'{cfg}/image/losses'])
    with torch.no_grad():
        final = G.synthesis(torch.from_numpy(code)[None], noise_mode='const', force_fp32=True).numpy()
    assert maxabs(final, g[f'{cfg}/image/final']) <= 2e-4
    sd = G.state_dict()
    wkey = [k for k in g.files if k.startswith(f'{cfg}/image/synthesis.')][0]
    check_weights(sd[wkey.split('/')[-1]].numpy(), g[wkey], 3e-3, 4)
    check_weights(sd[[k for k in sd if k.startswith('synthesis.L13_') and k.endswith('.bias')][0]].numpy(), g[f'{cfg}/image/bias_L13'], 3e-3, 4)
    check_weights(sd[[k for k in sd if k.startswith('synthesis.L5_') and k.endswith('affine.weight')][0]].numpy()[:4], g[f'{cfg}/image/affine_L5'], 3e-3, 4)
    # the Fourier-feature input is left alone (run_pti_images.py:113-114)
    ref = build_product_generator(cfg).state_dict()
    for k in ('synthesis.input.weight', 'synthesis.input.affine.weight', 'synthesis.input.affine.bias'):
        assert torch.equal(sd[k], ref[k])


def test_pti_needs_lpips_callable_and_uses_it():
    from inversion.scripts.run_pti_images import PTI, default_opts
    with pytest.raises(RuntimeError):
        PTI(default_opts(device='cpu'))
    G = tunable_generator('Ttiny')
    calls = []

    def fake_lpips(a, b):
        calls.append(1)
        return (a - b).abs().mean()
    pti = PTI(default_opts(device='cpu', steps=3, lpips_threshold=0.0), lpips_loss=fake_lpips)
    pti.optimize_model(G, synth_ws(1, G.num_ws, G.w_dim, seed=6)[0], pti_target(64, 40))
    assert len(calls) == 3 and pti.history[0][2] is not None and pti.history[-1][1] < pti.history[0][1]
    pti = PTI(default_opts(device='cpu', steps=3, lpips_threshold=1e9), lpips_loss=fake_lpips)     # already "good enough": no step
    pti.optimize_model(G, synth_ws(1, G.num_ws, G.w_dim, seed=6)[0], pti_target(64, 40))
    assert pti.history == []
    assert 'LPIPS' in PTI.get_description(1, torch.tensor(0.5), torch.tensor(0.25), None)


def video_case():
    codes = synth_ws(3, 16, 32, seed=6)
    targets = np.stack([pti_target(64, 40 + i) for i in range(3)])
    tr = np.stack([make_user_transform((0.02 * i, -0.01 * i), 5.0 * i) for i in range(3)])
    return codes, targets, tr


def run_video(monkeypatch_order=True, steps=4):
    from inversion.scripts.run_pti_images import default_opts
    from inversion.video import run_pti_video as rv
    real = rv.epoch_order
    if monkeypatch_order:
        rv.epoch_order = lambda n, device: torch.arange(n)
    try:
        G = tunable_generator('Ttiny')
        codes, targets, tr = video_case()
        v = rv.VideoPTI(default_opts(device='cpu', steps=steps, learning_rate=3e-3, lpips_lambda=0.0, batch_size=2))
        v.optimize_model(G, codes, targets, landmarks_transforms=tr)
    finally:
        rv.epoch_order = real
    return G, v


def test_video_pti_matches_reference():
    g = golden('pti')
    G, v = run_video()
    losses = np.asarray([h[1] for h in v.history])
    assert np.abs(losses - g['Ttiny/video/losses']).max() <= 2e-6
    codes, _, tr = video_case()
    with torch.no_grad():
        G.synthesis.input.transform = torch.from_numpy(tr[:1]).float()
        final = G.synthesis(torch.from_numpy(codes[:1]), noise_mode='const', force_fp32=True).numpy()
    assert maxabs(final, g['Ttiny/video/final']) <= 2e-4
    check_weights(G.state_dict()['synthesis.L0_36_12.weight'].numpy(), g['Ttiny/video/synthesis.L0_36_12.weight'], 3e-3, 4)


def test_epoch_order_is_a_permutation():
    from inversion.video.run_pti_video import epoch_order
    torch.manual_seed(3)
    o = epoch_order(11, 'cpu')
    assert sorted(o.tolist()) == list(range(11)) and o.tolist() != list(range(11))
    torch.manual_seed(3)
    assert o.tolist() == list(torch.utils.data.RandomSampler(range(11)))         # what DataLoader(shuffle=True) would draw


def _worker(rank, world, port, out_dir, paths):
    for p in paths:
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ['MASTER_ADDR'], os.environ['MASTER_PORT'] = '127.0.0.1', str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    torch.set_num_threads(2)
    torch.manual_seed(1234)                       # shuffle drawn on rank 0 and broadcast
    G, v = run_video(monkeypatch_order=False)
    np.save(os.path.join(out_dir, f'w_{rank}.npy'), G.state_dict()['synthesis.L0_36_12.weight'].numpy())
    np.save(os.path.join(out_dir, f'loss_{rank}.npy'), np.asarray([h[1] for h in v.history]))
    dist.barrier()
    dist.destroy_process_group()


def test_data_parallel_video_pti_equals_single_process(tmp_path):
    """gloo, world_size 2: batches of 2 are split 1 + 1 and the odd last batch 1 + 0 (one rank idles but still joins
    the all-reduce); weights and the loss curve equal the single-process run with the same shuffle."""
    paths = [p for p in sys.path if 'stylegan3-editing_amd' in p or p.endswith('tests') or p.endswith('repo')]
    port = 33500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(2, port, str(tmp_path), paths), nprocs=2, join=True)
    torch.manual_seed(1234)
    G, v = run_video(monkeypatch_order=False)
    w0, w1 = np.load(tmp_path / 'w_0.npy'), np.load(tmp_path / 'w_1.npy')
    assert np.array_equal(w0, w1)                                  # replicas stay identical
    l0 = np.load(tmp_path / 'loss_0.npy')
    assert np.abs(l0 - np.asarray([h[1] for h in v.history])).max() <= 2e-6
    check_weights(w0, G.state_dict()['synthesis.L0_36_12.weight'].numpy(), 3e-3, 4)
