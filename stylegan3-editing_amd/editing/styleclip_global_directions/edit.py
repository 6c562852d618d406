"""StyleSpace edit sweep (reference editing/styleclip_global_directions/edit.py:124-191).

For one inverted latent the reference renders num_betas x num_alphas edits with batch-1 synthesis calls (55 for the
default 5 x 11 sweep).  Here the whole sweep is assembled as ONE StyleSpace batch and rendered `max_batch` edits per
synthesis forward; under torch.distributed the sweep items are sharded over the ranks and the rendered images
all-gathered in sweep order.  Results (order: beta-major, alpha-minor) equal the reference's `torch.cat(results)`.
"""
import pickle

import numpy as np
import torch

from sg3_runtime.sharded import all_gather_ragged, shard_range


def build_sweep(latent_code_i, directions, alphas):
    """{c: [1, C_c]} + per-beta direction dicts -> {c: [num_betas * num_alphas, C_c]} (beta-major)."""
    a = torch.as_tensor(np.asarray(alphas), dtype=torch.float32)
    out = {}
    for c, base in latent_code_i.items():
        d = torch.cat([directions[b][c] for b in range(len(directions))], dim=0).to(base)        # [num_betas, C]
        out[c] = (base.unsqueeze(0) + a.to(base).view(1, -1, 1) * d.unsqueeze(1)).reshape(-1, base.shape[-1])
    return out


def render_sweep(stylegan_model, sweep, max_batch=32, shard=False, **synthesis_kwargs):
    """Render every item of a StyleSpace batch; returns [items, 3, R, R] (on every rank when sharded)."""
    n = int(sweep['input'].shape[0])
    start, stop = 0, n
    distributed = shard and torch.distributed.is_available() and torch.distributed.is_initialized()
    if distributed:
        start, stop = shard_range(n, torch.distributed.get_rank(), torch.distributed.get_world_size())
    outs = []
    with torch.no_grad():
        for b0 in range(start, stop, max_batch):
            b1 = min(b0 + max_batch, stop)
            outs.append(stylegan_model.synthesis(None, all_s={c: v[b0:b1] for c, v in sweep.items()}, **synthesis_kwargs))
    res = stylegan_model.img_resolution
    local = torch.cat(outs) if outs else torch.zeros([0, stylegan_model.img_channels, res, res], device=sweep['input'].device)
    return all_gather_ragged(local, n) if distributed else local


def edit_image(latent, landmarks_transform, stylegan_model, global_direction_calculator, opts,
               image_name=None, save=False, max_batch=32, shard=False, directions=None, **synthesis_kwargs):
    """latent: [16,512] array.  opts: alpha_min/alpha_max/num_alphas, beta_min/beta_max/num_betas, neutral_text,
    target_text.  `directions` (optional list of per-beta direction dicts) bypasses the text encoder; extra keyword
    arguments (e.g. force_fp32=True) go to Generator.synthesis.
    Returns (results [num_betas*num_alphas,3,R,R], latents_results list of per-edit StyleSpace dicts)."""
    device = next(stylegan_model.parameters()).device
    latent_code = torch.from_numpy(np.asarray(latent)).to(device).unsqueeze(0)
    if landmarks_transform is not None:
        stylegan_model.synthesis.input.transform = torch.from_numpy(np.asarray(landmarks_transform)).to(device).float()
    with torch.no_grad():
        latent_code_s = stylegan_model.synthesis.W2S(latent_code)
    latent_code_i = {c: latent_code_s[c][0].unsqueeze(0) for c in latent_code_s}
    alphas = np.linspace(opts.alpha_min, opts.alpha_max, opts.num_alphas)
    betas = np.linspace(opts.beta_min, opts.beta_max, opts.num_betas)
    if directions is None:
        directions = [global_direction_calculator.get_delta_s(opts.neutral_text, opts.target_text, beta) for beta in betas]
    sweep = build_sweep(latent_code_i, directions, alphas)
    results = render_sweep(stylegan_model, sweep, max_batch=max_batch, shard=shard, **synthesis_kwargs)
    latents_results = [{c: sweep[c][i:i + 1] for c in sweep} for i in range(int(sweep['input'].shape[0]))]
    if save:
        raise RuntimeError('edit_image: image-grid writing (torchvision.utils.save_image) is outside this package; use the returned tensor')
    return results, latents_results


def load_direction_calculator(stylegan_model, opts, text_encoder=None):
    """On-disk formats of the reference (:176-191): delta_i_c .npy [channels, 512]; s_stats pickle [transform, s_mean, s_std]."""
    from editing.styleclip_global_directions.global_direction import StyleCLIPGlobalDirection
    device = next(stylegan_model.parameters()).device
    delta_i_c = torch.from_numpy(np.load(opts.delta_i_c)).float().to(device)
    with open(opts.s_statistics, "rb") as f:
        _, s_mean, s_std = pickle.load(f)
    s_std = {c: torch.from_numpy(np.asarray(v)).float().to(device) for c, v in s_std.items()}
    with open(opts.text_prompt_templates, "r") as f:
        templates = f.readlines()
    with torch.no_grad():
        s_avg = stylegan_model.synthesis.W2S(stylegan_model.mapping.w_avg.unsqueeze(0).repeat(1, stylegan_model.num_ws, 1))
    return StyleCLIPGlobalDirection(delta_i_c, s_std, templates, s_avg, text_encoder=text_encoder)
