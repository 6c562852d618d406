"""`e4e`: ReStyle-e4e encoder + StyleGAN3 decoder (reference models/setgan/encoder/e4e3.py:10-117): the number of styles
comes from the options (:16) and the encoder is the progressive backbone (:22-29).  Same call contract as `pSp`
(restyle_net.ReStyleNet); the `opts.sgxl` branch (:66-69) belongs to the out-of-scope StyleGAN-XL decoder."""
from models.setgan.encoder.encoders import restyle_e4e_encoders
from models.setgan.encoder.restyle_net import ReStyleNet


class e4e(ReStyleNet):  # noqa: N801  (reference class name)
    encoders = {
        'ProgressiveBackboneEncoder': lambda n_styles, opts: restyle_e4e_encoders.ProgressiveBackboneEncoder(50, 'ir_se', n_styles, opts),
        'ResNetProgressiveBackboneEncoder': lambda n_styles, opts: restyle_e4e_encoders.ResNetProgressiveBackboneEncoder(n_styles, opts),
    }

    def __init__(self, opts, decoder=None):
        super().__init__(opts, n_styles=opts.n_styles, decoder=decoder)
