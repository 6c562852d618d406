"""Shared body of the two ReStyle wrappers (`pSp`, `e4e`): an encoder that refines a W+ code and the StyleGAN3 decoder
that renders it.  Contract of the reference wrappers (models/setgan/encoder/psp3.py:45-84, e4e3.py:45-87):

  forward(x, latent=None, resize=True, input_code=False, landmarks_transform=None, return_latents=False,
          return_aligned_and_unaligned=False)
    code    = x                                   if input_code
            = encoder(x) + latent                 if x carries 6 channels and a previous code is given (residual step)
            = encoder(x) + latent_avg             otherwise (first step)
    aligned = synthesis(code) under a per-sample identity transform, fp32, const noise; pooled to 256^2 if `resize`
    unaligned (only with landmarks_transform [B,3,3]) = a second synthesis under that transform
    returns (aligned, unaligned, code) | (aligned, code) | aligned  according to the two return flags.

The device follows the module (the reference hard-codes .cuda()), the decoder may be handed in, and checkpoints are the
reference's format: {'state_dict': {'encoder.*', 'decoder.*'}, 'latent_avg', 'opts'}.
"""
import torch
from torch import nn

from models.stylegan3.model import SG3Generator
from utils.common import get_identity_transform

_DROP_ON_LOAD = ('synthesis.input.transform',)        # buffer whose shape depends on the last call (psp3.py:37)


def _sub_state_dict(ckpt, prefix, drop=()):
    """Entries of ckpt['state_dict'] (or ckpt itself) under `prefix.`, with the prefix removed."""
    sd = ckpt.get('state_dict', ckpt)
    cut = len(prefix) + 1
    return {k[cut:]: v for k, v in sd.items() if k.startswith(prefix + '.') and k[cut:] not in drop}


class ReStyleNet(nn.Module):
    encoders = {}                                   # encoder_type -> factory(n_styles, opts); filled by the subclasses
    pool_to = (256, 256)

    def __init__(self, opts, n_styles, decoder=None):
        super().__init__()
        self.opts = opts
        self.n_styles = n_styles
        self.encoder = self.set_encoder()
        self.face_pool = nn.AdaptiveAvgPool2d(self.pool_to)
        self.latent_avg = None
        if decoder is not None:
            self._attach(decoder)
        self.load_weights()

    # ---- construction -------------------------------------------------------------------------------------------
    def set_opts(self, opts):
        self.opts = opts

    def set_encoder(self):
        kind = self.opts.encoder_type
        if kind not in self.encoders:
            raise Exception(f'{kind} is not a valid encoders')
        return self.encoders[kind](self.n_styles, self.opts)

    def _attach(self, decoder):
        self.decoder = decoder
        self.latent_avg = decoder.mapping.w_avg

    def load_weights(self):
        path = getattr(self.opts, 'checkpoint_path', None)
        if path is None:
            # untrained encoder on a generator given by the caller / by opts.stylegan_weights (the reference additionally
            # seeds the encoder from an ArcFace IR-SE50 checkpoint that is not available offline)
            if not hasattr(self, 'decoder'):
                self._attach(SG3Generator(checkpoint_path=getattr(self.opts, 'stylegan_weights', None), device='cpu').decoder)
            return
        print(f'Loading ReStyle {type(self).__name__} from checkpoint: {path}')
        ckpt = torch.load(path, map_location='cpu')
        self.encoder.load_state_dict(_sub_state_dict(ckpt, 'encoder'), strict=True)
        if not hasattr(self, 'decoder'):
            self.decoder = SG3Generator(checkpoint_path=None, device='cpu').decoder
        self.decoder.load_state_dict(_sub_state_dict(ckpt, 'decoder', drop=_DROP_ON_LOAD), strict=False)
        self._load_latent_avg(ckpt)

    def _load_latent_avg(self, ckpt, repeat=None):
        avg = ckpt.get('latent_avg')
        self.latent_avg = avg if (avg is None or repeat is None) else avg.repeat(repeat, 1)

    _get_keys = staticmethod(_sub_state_dict)           # name used by the reference's scripts

    # ---- forward ------------------------------------------------------------------------------------------------
    def _code(self, x, latent, input_code):
        if input_code:
            return x
        delta = self.encoder(x)
        if x.shape[1] == 6 and latent is not None:
            return delta + latent
        return delta + self.latent_avg.to(delta.device).repeat(delta.shape[0], 1, 1)

    def _render(self, codes, transform, resize):
        self.decoder.synthesis.input.transform = transform
        img = self.decoder.synthesis(codes, noise_mode='const', force_fp32=True)
        return self.face_pool(img) if resize else img

    def forward(self, x, latent=None, resize=True, input_code=False, landmarks_transform=None,
                return_latents=False, return_aligned_and_unaligned=False):
        codes = self._code(x, latent, input_code)
        eye = torch.from_numpy(get_identity_transform()).to(codes.device).float()
        images = self._render(codes, eye.unsqueeze(0).repeat(x.shape[0], 1, 1), resize)
        unaligned = None if landmarks_transform is None else self._render(codes, landmarks_transform.float(), resize)
        if unaligned is not None and return_aligned_and_unaligned:
            return images, unaligned, codes
        return (images, codes) if return_latents else images
