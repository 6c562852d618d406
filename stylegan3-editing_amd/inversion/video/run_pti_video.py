"""Joint pivotal tuning over all frames of a video (reference inversion/video/run_pti_video.py:91-168): one generator,
shuffled mini-batches of (target frame, inverted latent, landmarks transform), Adam on the synthesis weights.

Data parallel form (one process per GPU, torch.distributed over RCCL): every rank holds a generator replica and takes
a contiguous slice of each mini-batch; the per-rank losses are weighted so that their sum is the reference's batch
mean, and the gradients are summed with ONE all-reduce per step over a single flat bucket that all parameter
gradients alias (21 M parameters = 84 MB fp32 for T-1024; no per-tensor collectives, no copy in or out).  The
optimizer step is replicated, so the replicas stay bit-identical without broadcasting weights.
"""
import numpy as np
import torch
import torch.distributed as dist

from inversion.scripts.run_pti_images import PTI, as_image_batch
from sg3_runtime.sharded import shard_range


class FlatGradBucket:
    """All gradients of `params` as views into one contiguous fp32 buffer (the all-reduce operand)."""

    def __init__(self, params):
        self.params = [p for p in params if p.requires_grad]
        total = sum(p.numel() for p in self.params)
        self.flat = torch.zeros(total, dtype=torch.float32, device=self.params[0].device)
        ofs = 0
        for p in self.params:
            p.grad = self.flat[ofs:ofs + p.numel()].view_as(p)
            ofs += p.numel()

    def zero(self):
        self.flat.zero_()

    def all_reduce(self, group=None):
        if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
            dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=group)


def epoch_order(n, device):
    """One shuffled pass over n frames, identical on every rank: drawn on rank 0 the way DataLoader(shuffle=True)'s
    RandomSampler does and broadcast."""
    distributed = dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1
    if not distributed or dist.get_rank() == 0:
        order = torch.tensor(list(torch.utils.data.RandomSampler(range(n))), dtype=torch.int64)
    else:
        order = torch.empty(n, dtype=torch.int64)
    if distributed:
        order = order.to(device)
        dist.broadcast(order, src=0)
        order = order.cpu()
    return order


class VideoPTI(PTI):

    def __init__(self, opts, lpips_loss=None):
        super().__init__(opts, lpips_loss=lpips_loss)

    def optimize_model(self, generator, codes, target_images, landmarks_transforms=None, image_name=None):
        rank = dist.get_rank() if dist.is_available() and dist.is_initialized() else 0
        world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
        optimizer = self.get_optimizer(generator)
        bucket = FlatGradBucket(optimizer.param_groups[0]['params'])
        codes = torch.as_tensor(np.asarray(codes)).float()
        n = int(codes.shape[0])
        if landmarks_transforms is not None:
            landmarks_transforms = torch.as_tensor(np.asarray(landmarks_transforms)).float()
        self.history = []
        outputs = None
        step = 0
        early_exits = 0
        n_batches = (n + self.opts.batch_size - 1) // self.opts.batch_size
        # the reference leaves the epoch when a batch is already below the LPIPS threshold and starts a new shuffled
        # one (:137-138); it would spin forever once every batch is below it, so that case ends the tuning here
        while step < self.opts.steps and early_exits <= n_batches:
            order = epoch_order(n, self.device)
            for b0 in range(0, n, self.opts.batch_size):
                indices = order[b0:b0 + self.opts.batch_size]
                lo, hi = shard_range(len(indices), rank, world)
                mine = indices[lo:hi]
                bucket.zero()
                stats = torch.zeros(3, dtype=torch.float64, device=self.device)        # loss, lpips, l2 (weighted parts)
                if len(mine) > 0:
                    targets = torch.cat([as_image_batch(target_images[int(i)], self.device) for i in mine])
                    latents = codes[mine].to(self.device)
                    if landmarks_transforms is not None:
                        generator.synthesis.input.transform = landmarks_transforms[mine].to(self.device)
                    outputs = generator.synthesis(latents, noise_mode='const', force_fp32=True)
                    loss, lpips_loss, l2_loss_val = self.calc_loss(outputs, targets)
                    share = len(mine) / len(indices)             # batch means -> this rank's part of the global mean
                    (loss * share).backward()
                    stats[0] = float(loss.detach()) * share
                    stats[1] = (float(lpips_loss.detach()) if lpips_loss is not None else 0.0) * share
                    stats[2] = (float(l2_loss_val.detach()) if l2_loss_val is not None else 0.0) * share
                bucket.all_reduce()
                if world > 1:
                    dist.all_reduce(stats, op=dist.ReduceOp.SUM)
                if self.opts.lpips_lambda > 0 and float(stats[1]) < self.opts.lpips_threshold:
                    early_exits += 1
                    break
                early_exits = 0
                self.history.append((step, float(stats[0]), float(stats[1]) if self.opts.lpips_lambda > 0 else None,
                                     float(stats[2]) if self.opts.l2_lambda > 0 else None))
                optimizer.step()
                step += 1
                if step == self.opts.steps:
                    break
        if self.opts.save_final_model and self.opts.output_path is not None and rank == 0:
            torch.save(generator.state_dict(), self.opts.output_path / 'final_pti_model.pt')
        return outputs
