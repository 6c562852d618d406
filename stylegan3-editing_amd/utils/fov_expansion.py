"""Field-of-view expansion (reference utils/fov_expansion.py:8-108): the generator is evaluated under translated input
transforms and the translated renderings are stitched around the centre image.

The reference runs up to nine batch-1 synthesis passes, one per tile.  Here the tiles are ONE batched forward: the
Fourier-feature input takes a per-sample transform [B,3,3] (networks_stylegan3.py SynthesisInput), so the N samples x
T tiles go through every kernel launch together, and the canvas is filled with slice copies on the device.
"""
import numpy as np
import torch

from utils.common import make_transform

# tile order of the reference (:48-49): centre, left, top, right, bottom, top-left, top-right, bottom-right, bottom-left
# (dx, dy): direction of the tile in canvas space; the content shift is +dx * pixels / res (translate of make_transform)
_TILES = [('center', 0, 0), ('left', 1, 0), ('top', 0, 1), ('right', -1, 0), ('bottom', 0, -1),
          ('top_left', 1, 1), ('top_right', -1, 1), ('bottom_right', -1, -1), ('bottom_left', 1, -1)]


class Expander:

    def __init__(self, G, **synthesis_kwargs):
        self.G = G
        self.synthesis_kwargs = synthesis_kwargs      # e.g. force_fp32=True; the reference always uses the defaults

    def generate_expanded_image(self, ws=None, all_s=None, landmark_t=None,
                                pixels_right=0, pixels_left=0, pixels_top=0, pixels_bottom=0):
        assert landmark_t is not None, "Expected to receive landmarks transforms! Received None!"
        G = self.G
        res = G.img_resolution
        device = next(G.parameters()).device
        transforms = Expander._get_transforms(res, pixels_right, pixels_left, pixels_top, pixels_bottom)
        live = [i for i, t in enumerate(transforms) if t is not None]
        n = int(ws.shape[0]) if ws is not None else int(next(iter(all_s.values())).shape[0])
        landmark_t = np.asarray(landmark_t.detach().cpu() if isinstance(landmark_t, torch.Tensor) else landmark_t, dtype=np.float64)
        lt = np.broadcast_to(landmark_t, (n, 3, 3)) if landmark_t.ndim == 2 else landmark_t
        # sample-major inside each tile: batch index = tile * n + sample
        tr = np.stack([lt[j] @ transforms[i] for i in live for j in range(n)])
        # the per-sample tile transforms are put back afterwards: a [B*T,3,3] buffer left behind would break the next
        # caller's batch size (the reference leaves its last [3,3] tile transform in place)
        previous = G.synthesis.input.transform
        G.synthesis.input.transform = torch.from_numpy(tr).float().to(device)
        rep = len(live)
        try:
            with torch.no_grad():
                if all_s is not None:
                    imgs = G.synthesis(None, {k: v.repeat(rep, *([1] * (v.ndim - 1))) for k, v in all_s.items()}, **self.synthesis_kwargs)
                else:
                    imgs = G.synthesis(ws.repeat(rep, 1, 1), None, **self.synthesis_kwargs)
        finally:
            G.synthesis.input.transform = previous
        images = [None] * len(transforms)
        for k, i in enumerate(live):
            images[i] = imgs[k * n:(k + 1) * n]
        return Expander._merge_images(images, res, pixels_right, pixels_left, pixels_top, pixels_bottom)

    @staticmethod
    def _get_transforms(res, pixels_right, pixels_left, pixels_top, pixels_bottom):
        hor = {1: pixels_left, -1: pixels_right, 0: None}
        ver = {1: pixels_top, -1: pixels_bottom, 0: None}
        out = []
        for _, dx, dy in _TILES:
            ph, pv = hor[dx], ver[dy]
            if (ph is not None and ph == 0) or (pv is not None and pv == 0):
                out.append(None)                      # no pixels requested on that side (:54-55, :69-70)
                continue
            t = make_transform((dx * (ph or 0) / res, dy * (pv or 0) / res), 0)
            out.append(np.linalg.inv(t))
        return out

    @staticmethod
    def _get_transform_single_edge(res, edge, num_pixels):
        if num_pixels == 0:
            return None
        shift = {'left': (1, 0), 'right': (-1, 0), 'top': (0, 1), 'bottom': (0, -1)}
        if edge not in shift:
            raise ValueError("Invalid edge for transform")
        dx, dy = shift[edge]
        return make_transform((dx * num_pixels / res, dy * num_pixels / res), 0)

    @staticmethod
    def _get_transform_corner(res, corner, num_pixels_hor, num_pixels_ver):
        if num_pixels_hor == 0 or num_pixels_ver == 0:
            return None
        shift = {'top_left': (1, 1), 'top_right': (-1, 1), 'bottom_left': (1, -1), 'bottom_right': (-1, -1)}
        if corner not in shift:
            raise ValueError("Invalid corner for transform")
        dx, dy = shift[corner]
        return make_transform((dx * num_pixels_hor / res, dy * num_pixels_ver / res), 0)

    @staticmethod
    def _merge_images(images, res, pixels_right, pixels_left, pixels_top, pixels_bottom):
        c = images[0]
        canvas = torch.zeros(c.shape[0], 3, pixels_top + res + pixels_bottom, pixels_left + res + pixels_right, device=c.device)
        # per axis: (canvas slice, source slice) of the band a tile with direction d fills
        def band(d, before, after):
            if d == 0:
                return slice(before, before + res), slice(0, res)
            if d == 1:
                return slice(0, before), slice(0, before)
            return slice(before + res, before + res + after), slice(res - after, res)
        for (_, dx, dy), img in zip(_TILES, images):
            if img is None:
                continue
            cx, sx = band(dx, pixels_left, pixels_right)
            cy, sy = band(dy, pixels_top, pixels_bottom)
            canvas[:, :, cy, cx] = img[:, :, sy, sx]
        return canvas
