"""`pSp`: ReStyle encoder + StyleGAN3 decoder (API of reference models/setgan/encoder/psp3.py:10-114).

forward(x, latent, resize, input_code, landmarks_transform, return_latents, return_aligned_and_unaligned):
  codes = encoder(x) + (latent if x has 6 channels and latent is given else latent_avg)           (:53-60)
  images = decoder.synthesis(codes) with the identity transform                                    (:63-66)
  unaligned images = a SECOND synthesis with `landmarks_transform` assigned to synthesis.input.transform  (:72-76)
Differences from the reference, none visible to callers: the device is taken from the module instead of hard-coded
`.cuda()`, the decoder is passed in or built by SG3Generator without hard-coded checkpoint paths, and
`ResNetBackboneEncoder` is unavailable offline (see restyle_psp_encoders.py).
"""
import torch
from torch import nn

from models.setgan.encoder.encoders import restyle_psp_encoders
from models.stylegan3.model import SG3Generator
from utils import common


class pSp(nn.Module):  # noqa: N801  (reference class name)
    def __init__(self, opts, decoder=None):
        super().__init__()
        self.opts = opts
        self.n_styles = 16
        self.encoder = self.set_encoder()
        self.face_pool = torch.nn.AdaptiveAvgPool2d((256, 256))
        self.latent_avg = None
        if decoder is not None:
            self.decoder = decoder
            self.latent_avg = decoder.mapping.w_avg
        self.load_weights()

    def set_encoder(self):
        if self.opts.encoder_type == 'BackboneEncoder':
            return restyle_psp_encoders.BackboneEncoder(50, 'ir_se', self.n_styles, self.opts)
        if self.opts.encoder_type == 'ResNetBackboneEncoder':
            return restyle_psp_encoders.ResNetBackboneEncoder(self.n_styles, self.opts)
        raise Exception(f'{self.opts.encoder_type} is not a valid encoders')

    def load_weights(self):
        ckpt_path = getattr(self.opts, 'checkpoint_path', None)
        if ckpt_path is None:
            if not hasattr(self, 'decoder'):
                self.decoder = SG3Generator(checkpoint_path=getattr(self.opts, 'stylegan_weights', None), device='cpu').decoder
                self.latent_avg = self.decoder.mapping.w_avg
            return
        print(f'Loading ReStyle pSp from checkpoint: {ckpt_path}')
        ckpt = torch.load(ckpt_path, map_location='cpu')
        self.encoder.load_state_dict(self._get_keys(ckpt, 'encoder'), strict=True)
        if not hasattr(self, 'decoder'):
            self.decoder = SG3Generator(checkpoint_path=None, device='cpu').decoder
        self.decoder.load_state_dict(self._get_keys(ckpt, 'decoder', remove=["synthesis.input.transform"]), strict=False)
        self._load_latent_avg(ckpt)

    def forward(self, x, latent=None, resize=True, input_code=False, landmarks_transform=None,
                return_latents=False, return_aligned_and_unaligned=False):
        unaligned_images = None
        if input_code:
            codes = x
        else:
            codes = self.encoder(x)
            if x.shape[1] == 6 and latent is not None:
                codes = codes + latent                                    # residual w.r.t. the previous ReStyle step
            else:
                codes = codes + self.latent_avg.to(codes.device).repeat(codes.shape[0], 1, 1)

        device = codes.device
        identity = torch.from_numpy(common.get_identity_transform()).unsqueeze(0).repeat(x.shape[0], 1, 1).to(device).float()
        self.decoder.synthesis.input.transform = identity
        images = self.decoder.synthesis(codes, noise_mode='const', force_fp32=True)
        if resize:
            images = self.face_pool(images)

        if landmarks_transform is not None:
            self.decoder.synthesis.input.transform = landmarks_transform.float()      # [batch, 3, 3]
            unaligned_images = self.decoder.synthesis(codes, noise_mode='const', force_fp32=True)
            if resize:
                unaligned_images = self.face_pool(unaligned_images)

        if landmarks_transform is not None and return_aligned_and_unaligned:
            return images, unaligned_images, codes
        if return_latents:
            return images, codes
        return images

    def set_opts(self, opts):
        self.opts = opts

    def _load_latent_avg(self, ckpt, repeat=None):
        if 'latent_avg' in ckpt:
            self.latent_avg = ckpt['latent_avg']
            if repeat is not None:
                self.latent_avg = self.latent_avg.repeat(repeat, 1)
        else:
            self.latent_avg = None

    @staticmethod
    def _get_keys(d, name, remove=()):
        if 'state_dict' in d:
            d = d['state_dict']
        return {k[len(name) + 1:]: v for k, v in d.items() if k[:len(name)] == name and k[len(name) + 1:] not in remove}
