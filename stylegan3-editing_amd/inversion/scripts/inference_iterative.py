"""Batched ReStyle inversion of a set of images with the reference's on-disk outputs
(reference inversion/scripts/inference_iterative.py:22-101).

The reference script is a pyrallis CLI around a torchvision dataset; both are unavailable here and out of scope,
so the harness is a function over tensors.  What it keeps is the data contract downstream tools read:

  <output_path>/latents.npy   np.save of {image_name: [latent_step_0, ..., latent_step_{n-1}]}, each [n_styles, 512] float32
                              (:88-89, :101; load with np.load(path, allow_pickle=True).item())
  <output_path>/stats.txt     'Runtime {mean:.4f}+-{std:.4f}' over the per-batch wall time of run_on_batch (:62-68, :93-98)

Images stay on the device and are returned to the caller (per image, one tensor per ReStyle step); writing PNGs is the
caller's business.
"""
import os
import time

import numpy as np
import torch

from utils.inference_utils import get_average_image, run_on_batch


def run_inference(net, opts, images, names, output_path, landmarks_transforms=None, batch_size=None, n_images=None):
    """images: [F,3,256,256] tensor (any device; batches are moved to the net's device); names: F file names (dict keys of
    latents.npy); landmarks_transforms: optional [F,3,3].  Returns (results {name: [image per step]}, latents dict, runtime string)."""
    device = next(net.parameters()).device
    total = int(images.shape[0]) if n_images is None else min(int(n_images), int(images.shape[0]))
    assert len(names) >= total
    bs = int(batch_size or getattr(opts, 'test_batch_size', 2))
    os.makedirs(output_path, exist_ok=True)
    with torch.no_grad():
        avg_image = get_average_image(net)
    batch_times, all_latents, all_results = [], {}, {}
    for b0 in range(0, total, bs):
        b1 = min(b0 + bs, total)
        with torch.no_grad():
            x = images[b0:b1].to(device).float()
            lt = None if landmarks_transforms is None else landmarks_transforms[b0:b1].to(device).float()
            tic = time.time()
            result_batch, result_latents = run_on_batch(inputs=x, net=net, opts=opts, avg_image=avg_image, landmarks_transform=lt)
            if device.type == 'cuda':
                torch.cuda.synchronize(device)          # run_on_batch ends with a device -> host copy; this makes the stamp explicit
            batch_times.append(time.time() - tic)
        for i in range(b1 - b0):
            all_results[names[b0 + i]] = result_batch[i]
            all_latents[names[b0 + i]] = result_latents[i]
    result_str = f'Runtime {np.mean(batch_times):.4f}+-{np.std(batch_times):.4f}'
    print(result_str)
    with open(os.path.join(output_path, 'stats.txt'), 'w') as f:
        f.write(result_str)
    np.save(os.path.join(output_path, 'latents.npy'), all_latents)
    return all_results, all_latents, result_str
