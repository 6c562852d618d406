"""CPU, world_size 2 over gloo: frames sharded across ranks give, after the all-gather, exactly the latents of a single
process, in frame order, for equal and ragged shard sizes; plus the video post-processing arithmetic."""
import os
import sys
import types

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from helpers import build_product_generator, maxabs


class TinyEncoder(torch.nn.Module):
    def __init__(self):
        super().__init__()
        g = torch.Generator().manual_seed(11)
        self.w = torch.nn.Parameter(torch.randn(6, 16 * 32, generator=g) * 0.1)

    def forward(self, x):
        return (x.mean(dim=(2, 3)) @ self.w).view(-1, 16, 32)


def make_net():
    from models.setgan.encoder.psp3 import pSp
    G = build_product_generator('Ttiny')
    net = pSp.__new__(pSp)
    torch.nn.Module.__init__(net)
    net.opts = types.SimpleNamespace(encoder_type='BackboneEncoder', input_nc=6, n_iters_per_batch=2, resize_outputs=False)
    net.n_styles = 16
    net.encoder = TinyEncoder()
    net.face_pool = torch.nn.AdaptiveAvgPool2d((256, 256))
    net.decoder = G
    net.latent_avg = G.mapping.w_avg
    return net.eval()


def frames(n):
    return torch.from_numpy(np.random.RandomState(21).uniform(-1, 1, size=(n, 3, 256, 256)).astype(np.float32))


def _worker(rank, world, port, n_frames, out_dir):
    for p in sys.path_extra:
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ['MASTER_ADDR'], os.environ['MASTER_PORT'] = '127.0.0.1', str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    torch.set_num_threads(2)
    from sg3_runtime.sharded import ShardedInversion, shard_range
    net = make_net()
    inv = ShardedInversion(net, net.opts, batch_size=2)
    lat, (a, b) = inv.invert(frames(n_frames))
    assert (a, b) == shard_range(n_frames, rank, world)
    np.save(os.path.join(out_dir, f'lat_{rank}.npy'), lat.numpy())
    dist.barrier()
    dist.destroy_process_group()


sys.path_extra = [p for p in sys.path if 'stylegan3-editing_amd' in p or p.endswith('tests') or p.endswith('repo')]


@pytest.mark.parametrize('n_frames', [4, 5])
def test_sharded_inversion_matches_single_process(n_frames, tmp_path):
    from sg3_runtime.sharded import ShardedInversion
    net = make_net()
    ref, _ = ShardedInversion(net, net.opts, batch_size=2).invert(frames(n_frames))
    assert tuple(ref.shape) == (n_frames, 16, 32)
    port = 29500 + (os.getpid() % 2000) + n_frames
    mp.spawn(_worker, args=(2, port, n_frames, str(tmp_path)), nprocs=2, join=True)
    l0, l1 = np.load(tmp_path / 'lat_0.npy'), np.load(tmp_path / 'lat_1.npy')
    assert np.array_equal(l0, l1)                                 # every rank holds the full, ordered result
    assert maxabs(l0, ref.numpy()) <= 1e-5


def test_shard_ranges_cover_everything():
    from sg3_runtime.sharded import shard_range, shard_sizes
    for n in (0, 1, 7, 8, 256, 1000):
        for w in (1, 2, 3, 8):
            spans = [shard_range(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(w - 1))
            assert max(shard_sizes(n, w)) - min(shard_sizes(n, w)) <= 1


def test_postprocess_latents_matches_reference_formula():
    from sg3_runtime.sharded import postprocess_latents
    ws = np.random.RandomState(4).randn(9, 16, 8).astype(np.float32)
    ref = ws.copy()
    ref[:, 9:, :] = ref[:, 9:, :].mean(axis=0)                    # post_processing.py:15
    ref = (ref[2:-2] + 0.75 * ref[3:-1] + 0.75 * ref[1:-3] + 0.25 * ref[:-4] + 0.25 * ref[4:]) / 3   # :49-52
    out = postprocess_latents(torch.from_numpy(ws)).numpy()
    assert out.shape == (5, 16, 8) and maxabs(out, ref) <= 1e-6
