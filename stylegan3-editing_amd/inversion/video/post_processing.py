"""Video post-processing after inversion (reference inversion/video/post_processing.py:12-67): average the fine layers
over all frames, smooth latents and landmark transforms over time, and render the smoothed frames through the
field-of-view Expander.

Differences in execution, not in results: frames are rendered `frames_per_batch` at a time (each Expander call is one
batched synthesis forward over frames x tiles) and, under torch.distributed, every rank renders only its contiguous
range of the smoothed frames (the latents were all-gathered by sg3_runtime.sharded.ShardedInversion beforehand).
"""
import numpy as np
import torch

from utils.fov_expansion import Expander
from utils.common import get_identity_transform
from sg3_runtime.sharded import shard_range


def tensor2im_array(var):
    """HWC uint8 array of one CHW image in [-1, 1] (what np.array(utils.common.tensor2im(var)) yields, :33-40)."""
    var = var.detach().float().permute(1, 2, 0)
    var = ((var + 1) / 2).clamp(0, 1) * 255
    return var.to(torch.uint8).cpu().numpy()          # truncation, as ndarray.astype('uint8') on values in [0, 255]


def smooth_ws(ws):
    ws_p = ws[2:-2] + 0.75 * ws[3:-1] + 0.75 * ws[1:-3] + 0.25 * ws[:-4] + 0.25 * ws[4:]
    return ws_p / 3


def smooth_s(s):
    """Temporal smoothing of a list of per-frame style dicts (:55-67)."""
    batched = {c: torch.cat([frame[c] for frame in s]) for c in s[0]}
    smoothed = {c: smooth_ws(v) for c, v in batched.items()}
    return [{c: smoothed[c][i].unsqueeze(0) for c in smoothed} for i in range(smoothed['input'].shape[0])]


def smooth_latents_and_transforms(result_latents, result_landmarks_transforms, opts, device=None):
    device = device if device is not None else ('cuda' if torch.cuda.is_available() else 'cpu')
    smoothed_latents = torch.from_numpy(smooth_ws(np.asarray(result_latents))).float().to(device)
    if getattr(opts, 'landmarks_transforms_path', None) is not None:
        smoothed_transforms = smooth_ws(torch.cat([torch.as_tensor(t).unsqueeze(0) for t in result_landmarks_transforms]))
    else:
        smoothed_transforms = [None] * len(smoothed_latents)
    return smoothed_latents, smoothed_transforms


def postprocess_and_smooth_inversions(results, net, opts, frames_per_batch=4, shard=False, **synthesis_kwargs):
    """results: {'result_latents': {name: [16,512]}, 'landmarks_transforms': [...]}.  Returns the list of smoothed,
    expanded frames as HWC uint8 arrays (this rank's range when shard=True)."""
    result_latents = np.array(list(results["result_latents"].values()))
    result_latents[:, 9:, :] = result_latents[:, 9:, :].mean(axis=0)                 # average fine layers (:14-15)
    device = next(net.decoder.parameters()).device
    smoothed_latents, smoothed_transforms = smooth_latents_and_transforms(result_latents, results["landmarks_transforms"], opts, device=device)
    n = len(smoothed_latents)
    start, stop = 0, n
    if shard and torch.distributed.is_available() and torch.distributed.is_initialized():
        start, stop = shard_range(n, torch.distributed.get_rank(), torch.distributed.get_world_size())
    expander = Expander(G=net.decoder, **synthesis_kwargs)
    left, right, top, bottom = opts.expansion_amounts
    frames = []
    for b0 in range(start, stop, frames_per_batch):
        b1 = min(b0 + frames_per_batch, stop)
        trans = np.stack([np.asarray(get_identity_transform() if smoothed_transforms[i] is None
                                     else torch.as_tensor(smoothed_transforms[i]).cpu().numpy(), dtype=np.float64) for i in range(b0, b1)])
        with torch.no_grad():
            im = expander.generate_expanded_image(ws=smoothed_latents[b0:b1], landmark_t=trans, pixels_left=left,
                                                  pixels_right=right, pixels_top=top, pixels_bottom=bottom)
        frames.extend(tensor2im_array(im[i]) for i in range(b1 - b0))
    return frames
