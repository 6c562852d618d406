"""One-process-per-GPU sharding of independent images / video frames, with the single exchange the inversion
pipeline needs: an all-gather of the final latents (RCCL over xGMI on MI355X; `backend='nccl'` IS RCCL on ROCm).

The reference inverts a video frame by frame on one GPU (inversion/video/inference_on_video.py:119-145) and then needs
ALL latents on the host: fine layers 9..15 are averaged across frames (inversion/video/post_processing.py:13-15), the
latents are smoothed with a 5-tap temporal window (:49-52) and `latents.npy` is written in frame order
(inference_on_video.py:64-65).  Frames are independent units through encoder and synthesis (eval-mode BatchNorm,
per-sample modulation), so rank r takes the contiguous range [start_r, stop_r) and the only collective is one
all-gather of [frames_r, 16, 512] fp32 (1 MiB per rank for 256 frames on 8 GPUs) plus the optional [frames_r, 3, 3]
transforms: latency-bound, one call per video.  Weights are replicated (encoder 186 M + generator 22 M parameters).
"""
import torch
import torch.distributed as dist


def shard_range(n_items, rank, world_size):
    """Contiguous, balanced, rank-ordered partition: the first (n_items % world_size) ranks get one extra item."""
    base, extra = divmod(int(n_items), int(world_size))
    start = rank * base + min(rank, extra)
    return start, start + base + (1 if rank < extra else 0)


def shard_sizes(n_items, world_size):
    return [shard_range(n_items, r, world_size)[1] - shard_range(n_items, r, world_size)[0] for r in range(world_size)]


def all_gather_ragged(local, n_items, group=None):
    """All-gather per-rank shards (first dimension = this rank's items, possibly different per rank) into the full,
    rank-ordered tensor on every rank.  Equal shards use one `all_gather_into_tensor`; ragged shards are padded to
    the largest shard for the collective and trimmed afterwards."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        assert local.shape[0] == n_items
        return local
    world = dist.get_world_size(group)
    sizes = shard_sizes(n_items, world)
    assert local.shape[0] == sizes[dist.get_rank(group)], (local.shape, sizes)
    biggest = max(sizes)
    local = local.contiguous()
    if min(sizes) == biggest:
        out = torch.empty([n_items, *local.shape[1:]], dtype=local.dtype, device=local.device)
        dist.all_gather_into_tensor(out, local, group=group)
        return out
    pad = torch.zeros([biggest, *local.shape[1:]], dtype=local.dtype, device=local.device)
    pad[:local.shape[0]] = local
    bufs = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(bufs, pad, group=group)
    return torch.cat([b[:s] for b, s in zip(bufs, sizes)], dim=0)


class ShardedInversion:
    """ReStyle inversion of a set of frames sharded over the ranks of the default process group.

    net        : pSp / e4e style module with .forward(...), .face_pool, .latent_avg (already on this rank's device)
    opts       : namespace with n_iters_per_batch (and optionally resize_outputs)
    batch_size : frames per run_on_batch call on each rank
    """

    def __init__(self, net, opts, batch_size=16, use_graph=True):
        from utils.inference_utils import get_average_image
        self.net, self.opts, self.batch_size = net, opts, int(batch_size)
        self.rank = dist.get_rank() if dist.is_initialized() else 0
        self.world = dist.get_world_size() if dist.is_initialized() else 1
        with torch.no_grad():
            self.avg_image = get_average_image(net)
        # full batches replay one captured ReStyle step (encoder + decoder + pooling); ragged last batches run eagerly
        device = next(net.parameters()).device
        if use_graph and device.type == 'cuda' and not getattr(opts, 'resize_outputs', False):
            from .graphed import GraphedReStyleStep
            have = getattr(net, 'graphed_step', None)
            if have is None or have.batch != self.batch_size or have.is_stale():      # stale: weights changed since that capture
                try:
                    net.graphed_step = GraphedReStyleStep(net, self.batch_size)
                except RuntimeError as err:                      # capture refused: the eager loop is always there
                    print(f'[ShardedInversion] hipGraph capture of the ReStyle step failed, running eagerly: {err}')
                    net.graphed_step = None

    def invert(self, frames, landmarks_transforms=None):
        """frames: [F,3,256,256] tensor (or any indexable returning such slices) holding ALL frames; every rank reads only
        its own range.  Returns (latents [F,16,512] of the last ReStyle step, identical on every rank; this rank's
        frame range)."""
        from utils.inference_utils import run_on_batch
        from torch_utils.ops import plain_conv
        n = int(frames.shape[0])
        start, stop = shard_range(n, self.rank, self.world)
        device = next(self.net.parameters()).device

        def batch(b0, b1, check_range):
            x = frames[b0:b1].to(device, non_blocking=True).float()
            lt = None if landmarks_transforms is None else landmarks_transforms[b0:b1].to(device).float()
            return run_on_batch(x, self.net, self.opts, self.avg_image, landmarks_transform=lt, final_latents_only=True, check_range=check_range)

        # The latents stay on the device from the encoder to the all-gather.  Graph-replayed (full) batches cannot read the encoder's
        # split-precision range flag per batch without a host sync: it is reset once here and read once after the last full batch; if
        # any of them raised it (operands beyond the fp16 range), they are repeated on the eager path, which owns the fp32 form.
        # A ragged last batch runs eagerly (its encoder forward resets and reads the flag itself), after that read.
        bounds = [(b0, min(b0 + self.batch_size, stop)) for b0 in range(start, stop, self.batch_size)]
        with torch.no_grad():
            graphed = device.type == 'cuda' and getattr(self.net, 'graphed_step', None) is not None and not self.net.graphed_step.is_stale()
            full = [bd for bd in bounds if graphed and bd[1] - bd[0] == self.batch_size]
            if full:
                plain_conv.reset_overflow(device)
            local = {bd: batch(*bd, check_range=False) for bd in full}
            if full and plain_conv.overflowed(device):
                keep, self.net.graphed_step = self.net.graphed_step, None
                try:
                    local = {bd: batch(*bd, check_range=True) for bd in full}
                finally:
                    self.net.graphed_step = keep
            for bd in bounds:
                if bd not in local:
                    local[bd] = batch(*bd, check_range=True)
            local = [local[bd] for bd in bounds]
        width = (int(self.net.n_styles), int(self.net.latent_avg.shape[-1]))
        local = torch.cat(local, dim=0) if local else torch.zeros([0, *width], device=device)
        return all_gather_ragged(local, n), (start, stop)


def smooth_ws(ws):
    """5-tap temporal smoothing of the reference (inversion/video/post_processing.py:49-52): weights
    [0.25, 0.75, 1, 0.75, 0.25] / 3 over frames t-2..t+2; the result has F - 4 frames."""
    return (ws[2:-2] + 0.75 * ws[3:-1] + 0.75 * ws[1:-3] + 0.25 * ws[:-4] + 0.25 * ws[4:]) / 3


def postprocess_latents(ws, fine_from=9):
    """Average the fine layers over all frames (post_processing.py:13-15), then smooth over time (:17-19)."""
    ws = ws.clone()
    ws[:, fine_from:, :] = ws[:, fine_from:, :].mean(dim=0)
    return smooth_ws(ws)
