"""ReStyle inference harness (API of reference utils/inference_utils.py:59-111): `get_average_image`, `run_on_batch`.

`load_encoder` of the reference rebuilds pyrallis option dataclasses from a checkpoint (:28-56); pyrallis is not
installed here, so `load_encoder` takes / returns a plain namespace with the same fields
(`encoder_type`, `input_nc`, `n_iters_per_batch`, `resize_outputs`, `checkpoint_path`, `stylegan_weights`).
"""
import types

import torch

from models.setgan.encoder.psp3 import pSp


def load_encoder(checkpoint_path, test_opts=None, generator_path=None, device='cuda'):
    ckpt = torch.load(checkpoint_path, map_location='cpu')
    opts = dict(ckpt['opts'])
    opts['checkpoint_path'] = checkpoint_path
    if test_opts is not None:
        opts.update(test_opts if isinstance(test_opts, dict) else vars(test_opts))
    opts = types.SimpleNamespace(**opts)
    net = pSp(opts)
    if generator_path is not None:
        from models.stylegan3.model import SG3Generator
        net.decoder = SG3Generator(checkpoint_path=generator_path, device='cpu').decoder
    net.eval().to(device)
    return net, opts


def get_average_image(net):
    """Image of the average latent: net(latent_avg repeated over the 16 styles, input_code=True)."""
    device = next(net.parameters()).device
    avg = net(net.latent_avg.to(device).repeat(16, 1).unsqueeze(0), input_code=True, return_latents=False)[0]
    return avg.float().detach()


def run_on_batch(inputs, net, opts, avg_image, landmarks_transform=None):
    """The ReStyle loop: `opts.n_iters_per_batch` encoder + synthesis steps; every step feeds [input, previous output
    pooled to 256] (step 0: the average image) and refines the latent additively.  Returns per-sample lists of output
    images and latents (numpy), one entry per step, exactly like the reference."""
    results_batch = {idx: [] for idx in range(inputs.shape[0])}
    results_latent = {idx: [] for idx in range(inputs.shape[0])}
    y_hat, latent = None, None
    step_latents = []
    resize_outputs = getattr(opts, 'resize_outputs', False)
    for it in range(opts.n_iters_per_batch):
        if it == 0:
            x_input = torch.cat([inputs, avg_image.unsqueeze(0).repeat(inputs.shape[0], 1, 1, 1)], dim=1)
        else:
            x_input = torch.cat([inputs, y_hat], dim=1)
        is_last = it == opts.n_iters_per_batch - 1
        res = net.forward(x_input, latent=latent, landmarks_transform=landmarks_transform,
                          return_aligned_and_unaligned=True, return_latents=True, resize=resize_outputs)
        if landmarks_transform is None:
            y_hat, latent = res
        elif is_last:
            _, y_hat, latent = res
        else:
            y_hat, _, latent = res
        for idx in range(inputs.shape[0]):
            results_batch[idx].append(y_hat[idx])
        step_latents.append(latent)
        y_hat = net.face_pool(y_hat)
    # one device -> host copy for all steps (the reference copies every sample of every step as it goes, which
    # drains the GPU queue 16 x 5 times per batch)
    all_latents = torch.stack(step_latents).cpu().numpy()               # [steps, N, 16, 512]
    for idx in range(inputs.shape[0]):
        results_latent[idx] = [all_latents[it, idx] for it in range(len(step_latents))]
    return results_batch, results_latent
