"""`e4e`: ReStyle-e4e encoder + StyleGAN3 decoder (API of reference models/setgan/encoder/e4e3.py:10-117).

Identical call contract and forward to `pSp` (models/setgan/encoder/psp3.py:45-84 == e4e3.py:45-87); differences:
`n_styles` comes from the options (:16) and the encoder is the progressive backbone (:22-29).  The `opts.sgxl`
branch (:66-69) belongs to the out-of-scope StyleGAN-XL decoder and is not carried over."""
import torch

from models.setgan.encoder.encoders import restyle_e4e_encoders
from models.setgan.encoder.psp3 import pSp


class e4e(pSp):  # noqa: N801  (reference class name)
    def __init__(self, opts, decoder=None):
        torch.nn.Module.__init__(self)
        self.opts = opts
        self.n_styles = opts.n_styles
        self.encoder = self.set_encoder()
        self.face_pool = torch.nn.AdaptiveAvgPool2d((256, 256))
        self.latent_avg = None
        if decoder is not None:
            self.decoder = decoder
            self.latent_avg = decoder.mapping.w_avg
        self.load_weights()

    def set_encoder(self):
        if self.opts.encoder_type == 'ProgressiveBackboneEncoder':
            return restyle_e4e_encoders.ProgressiveBackboneEncoder(50, 'ir_se', self.n_styles, self.opts)
        if self.opts.encoder_type == 'ResNetProgressiveBackboneEncoder':
            return restyle_e4e_encoders.ResNetProgressiveBackboneEncoder(self.n_styles, self.opts)
        raise Exception(f'{self.opts.encoder_type} is not a valid encoders')
