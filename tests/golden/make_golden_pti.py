"""Golden vectors for pivotal tuning (PTI), generated with the REFERENCE generator and autograd graph on CPU.

The reference's PTI scripts (inversion/scripts/run_pti_images.py, inversion/video/run_pti_video.py) cannot be imported
here (pyrallis / lpips / torchvision are absent), so the optimisation loop is restated below exactly as
run_pti_images.py:111-139 defines it -- Adam over list(generator.synthesis.parameters())[3:], MSE loss, fp32 synthesis --
while every forward/backward runs through the reference's own Generator and torch_utils ops (_ref paths).

    python tests/golden/make_golden_pti.py         (needs /root/reference; writes tests/golden/pti.npz)
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, '/root/reference')

from synth_weights import CONFIGS, make_user_transform, synth_state_dict, synth_ws  # noqa: E402
from models.stylegan3.networks_stylegan3 import Generator  # noqa: E402  (reference)


def build(cfg):
    G = Generator(**CONFIGS[cfg]).eval()
    man = {k: list(v.shape) for k, v in G.state_dict().items()}
    sd = synth_state_dict(man, seed=0, input_bandwidth=float(G.synthesis.input.bandwidth))
    G.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
    return G


def pti_target(cfg_res, seed):
    r = np.random.RandomState(seed)
    yy, xx = np.meshgrid(np.linspace(-1, 1, cfg_res), np.linspace(-1, 1, cfg_res), indexing='ij')
    base = np.stack([np.sin(3 * xx + s) * np.cos(2 * yy - s) for s in (0.0, 0.7, 1.9)])
    return (0.6 * base + 0.05 * r.randn(3, cfg_res, cfg_res)).astype(np.float32)


def run(cfg, n_frames, steps, lr, batch_size, transforms):
    G = build(cfg)
    params = list(G.synthesis.parameters())[3:]
    opt = torch.optim.Adam(params, lr=lr)
    ws = torch.from_numpy(synth_ws(n_frames, G.num_ws, G.w_dim, seed=6))
    targets = torch.from_numpy(np.stack([pti_target(G.img_resolution, 40 + i) for i in range(n_frames)]))
    mse = torch.nn.MSELoss()
    losses = []
    order = list(range(n_frames))                           # fixed order: batches [0:bs], [bs:2bs], ...
    step = 0
    while step < steps:
        for b0 in range(0, n_frames, batch_size):
            idx = order[b0:b0 + batch_size]
            if transforms is not None:
                G.synthesis.input.transform = torch.from_numpy(np.stack([transforms[i] for i in idx])).float()
            out = G.synthesis(ws[idx], noise_mode='const', force_fp32=True)
            loss = mse(out, targets[idx])
            losses.append(float(loss))
            opt.zero_grad()
            loss.backward()
            opt.step()
            step += 1
            if step == steps:
                break
    with torch.no_grad():
        if transforms is not None:
            G.synthesis.input.transform = torch.from_numpy(np.stack(transforms[:1])).float()
        final = G.synthesis(ws[:1], noise_mode='const', force_fp32=True).numpy()
    sd = G.state_dict()
    return losses, final, sd


def main():
    out = {}
    for cfg in ('Ttiny', 'Rtiny'):
        # single image, identity transform (run_pti_images.py), 4 steps
        losses, final, sd = run(cfg, 1, 4, 3e-3, 1, None)
        out[f'{cfg}/image/losses'] = np.asarray(losses, np.float64)
        out[f'{cfg}/image/final'] = final
        for k in ('synthesis.L0_36_12.weight' if cfg == 'Ttiny' else 'synthesis.L0_36_16.weight',):
            out[f'{cfg}/image/{k}'] = sd[k].numpy()
        out[f'{cfg}/image/bias_L13'] = sd[[k for k in sd if k.startswith('synthesis.L13_') and k.endswith('.bias')][0]].numpy()
        out[f'{cfg}/image/affine_L5'] = sd[[k for k in sd if k.startswith('synthesis.L5_') and k.endswith('affine.weight')][0]].numpy()[:4]
    # video: 3 frames, batch 2, per-frame landmarks transforms, 4 steps (two epochs of [0,1], [2])
    tr = [make_user_transform((0.02 * i, -0.01 * i), 5.0 * i) for i in range(3)]
    losses, final, sd = run('Ttiny', 3, 4, 3e-3, 2, tr)
    out['Ttiny/video/losses'] = np.asarray(losses, np.float64)
    out['Ttiny/video/final'] = final
    out['Ttiny/video/synthesis.L0_36_12.weight'] = sd['synthesis.L0_36_12.weight'].numpy()
    np.savez_compressed(os.path.join(HERE, 'pti.npz'), **out)
    for k, v in out.items():
        print(k, v.shape, float(np.abs(v).max()))


if __name__ == '__main__':
    main()
