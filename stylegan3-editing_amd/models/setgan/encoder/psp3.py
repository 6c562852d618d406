"""`pSp`: ReStyle-pSp encoder + StyleGAN3 decoder (reference models/setgan/encoder/psp3.py:10-114): 16 style vectors,
`BackboneEncoder` (IR-SE50) or `ResNetBackboneEncoder`.  The shared behaviour lives in restyle_net.ReStyleNet."""
from models.setgan.encoder.encoders import restyle_psp_encoders
from models.setgan.encoder.restyle_net import ReStyleNet


class pSp(ReStyleNet):  # noqa: N801  (reference class name)
    encoders = {
        'BackboneEncoder': lambda n_styles, opts: restyle_psp_encoders.BackboneEncoder(50, 'ir_se', n_styles, opts),
        'ResNetBackboneEncoder': lambda n_styles, opts: restyle_psp_encoders.ResNetBackboneEncoder(n_styles, opts),
    }

    def __init__(self, opts, decoder=None):
        super().__init__(opts, n_styles=16, decoder=decoder)
