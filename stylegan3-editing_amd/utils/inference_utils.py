"""ReStyle inference harness (API of reference utils/inference_utils.py:59-111): `get_average_image`, `run_on_batch`.

`load_encoder` of the reference rebuilds pyrallis option dataclasses from a checkpoint (:28-56); pyrallis is not
installed here, so `load_encoder` takes / returns a plain namespace with the same fields
(`encoder_type`, `input_nc`, `n_iters_per_batch`, `resize_outputs`, `checkpoint_path`, `stylegan_weights`).
"""
import types

import torch

from models.setgan.encoder.e4e3 import e4e
from models.setgan.encoder.psp3 import pSp
from utils.model_utils import ENCODER_TYPES


def load_encoder(checkpoint_path, test_opts=None, generator_path=None, device='cuda'):
    ckpt = torch.load(checkpoint_path, map_location='cpu')
    opts = dict(ckpt['opts'])
    opts['checkpoint_path'] = checkpoint_path
    if test_opts is not None:
        opts.update(test_opts if isinstance(test_opts, dict) else vars(test_opts))
    opts = types.SimpleNamespace(**opts)
    # the checkpoint's encoder_type selects the wrapper (reference inference_utils.py:38-47)
    net = pSp(opts) if opts.encoder_type in ENCODER_TYPES['pSp'] else e4e(opts)
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


def run_on_batch(inputs, net, opts, avg_image, landmarks_transform=None, final_latents_only=False, check_range=True):
    """The ReStyle loop: `opts.n_iters_per_batch` encoder + synthesis steps; every step feeds [input, previous output
    pooled to 256] (step 0: the average image) and refines the latent additively.  Returns per-sample lists of output
    images and latents (numpy), one entry per step, exactly like the reference.

    Extensions for the sharded video path (sg3_runtime.ShardedInversion), both off by default:
    `final_latents_only`: return the last step's latents as ONE device tensor [N,n_styles,512] -- no host copy, no sync;
    `check_range=False` : on the graph-replay path, leave the encoder's split-precision range flag alone (the caller reset it
    before its first batch and reads it once after its last: one host sync per video instead of one per batch)."""
    results_batch = {idx: [] for idx in range(inputs.shape[0])}
    results_latent = {idx: [] for idx in range(inputs.shape[0])}
    y_hat, latent = None, None
    step_latents = []
    resize_outputs = getattr(opts, 'resize_outputs', False)
    graphed = getattr(net, 'graphed_step', None)
    if graphed is not None and graphed.is_stale():
        # the weights were tuned / reloaded after the capture (PTI, load_state_dict, an EMA update): the graph renders the old
        # ones.  Drop it and run the eager loop, as the reference does; ShardedInversion re-captures when it is built again.
        print('[run_on_batch] encoder / decoder weights changed since the ReStyle step was captured: dropping the hipGraph, running eagerly')
        net.graphed_step = graphed = None
    if (graphed is not None and inputs.is_cuda and inputs.shape[0] == graphed.batch and not resize_outputs and not torch.is_grad_enabled()
            and not net.training):
        done = _run_on_batch_graphed(inputs, net, opts, avg_image, landmarks_transform, graphed, final_latents_only, check_range)
        if done is not None:
            return done
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
    if final_latents_only:
        return step_latents[-1]
    # one device -> host copy for all steps (the reference copies every sample of every step as it goes, which
    # drains the GPU queue 16 x 5 times per batch)
    all_latents = torch.stack(step_latents).cpu().numpy()               # [steps, N, 16, 512]
    for idx in range(inputs.shape[0]):
        results_latent[idx] = [all_latents[it, idx] for it in range(len(step_latents))]
    return results_batch, results_latent


def _run_on_batch_graphed(inputs, net, opts, avg_image, landmarks_transform, graphed, final_latents_only=False, check_range=True):
    """The same loop with every step replayed from ONE captured hipGraph (sg3_runtime.GraphedReStyleStep): step 0 is the graph
    fed with the average image and latent_avg.  With landmark transforms the last step adds the unaligned render eagerly
    (inference_utils.py:96-100 returns it instead of the aligned one).  Returns None when the encoder's split-precision range
    guard fired: the caller then runs the eager loop, which owns the fp32 fallback."""
    from torch_utils.ops import plain_conv
    n, steps = inputs.shape[0], opts.n_iters_per_batch
    if check_range:
        plain_conv.reset_overflow(inputs.device)
    prev_image = avg_image.unsqueeze(0).expand(n, -1, -1, -1)
    prev_latent = net.latent_avg.to(inputs.device)
    images, latents = [], []
    for it in range(steps):
        image, latent, pooled = graphed(inputs, prev_image, prev_latent, checked=True)      # run_on_batch asked is_stale() for this batch
        if not final_latents_only:
            if landmarks_transform is not None and it == steps - 1:
                y_hat = net._render(latent, landmarks_transform.float(), False)
            else:
                y_hat = image.clone()                # the graph's output buffers are overwritten by the next replay
            images.append(y_hat)
        if not final_latents_only or it == steps - 1:
            latents.append(latent.clone())
        prev_image, prev_latent = pooled, latent
    if check_range and plain_conv.overflowed(inputs.device):
        return None
    if final_latents_only:
        return latents[-1]
    all_latents = torch.stack(latents).cpu().numpy()
    results_batch = {idx: [images[it][idx] for it in range(steps)] for idx in range(n)}
    results_latent = {idx: [all_latents[it, idx] for it in range(steps)] for idx in range(n)}
    return results_batch, results_latent
