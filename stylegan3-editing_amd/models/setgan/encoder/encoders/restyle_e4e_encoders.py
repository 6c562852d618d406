"""ReStyle-e4e encoder backbone (reference models/setgan/encoder/encoders/restyle_e4e_encoders.py:11-91).

Same IR-SE50 trunk and GradualStyleBlock heads as the pSp backbone; the heads are combined as one base code plus
per-style deltas: w[:, i] = w0 + delta_i for 1 <= i <= progressive_stage (:79-89).  Inherits the fused eval-mode HIP
forward of `BackboneEncoder`.  `ResNetProgressiveBackboneEncoder` (:91-140) is the same combination on the ResNet34 trunk.
Note: the reference constructor takes `input_nc` as its 4th positional argument while the e4e wrapper passes the options
object there; both forms are accepted here.
"""
from enum import Enum

from models.setgan.encoder.encoders.restyle_psp_encoders import BackboneEncoder, ResNetBackboneEncoder


class ProgressiveStage(Enum):
    WTraining = 0
    Delta1Training = 1
    Delta2Training = 2
    Delta3Training = 3
    Delta4Training = 4
    Delta5Training = 5
    Delta6Training = 6
    Delta7Training = 7
    Delta8Training = 8
    Delta9Training = 9
    Delta10Training = 10
    Delta11Training = 11
    Delta12Training = 12
    Delta13Training = 13
    Delta14Training = 14
    Delta15Training = 15
    Inference = -1


class _NC:
    def __init__(self, input_nc):
        self.input_nc = input_nc


class _Progressive:
    """w[:, 0] = w0, w[:, i] = w0 + delta_i for 1 <= i <= progressive_stage (reference :79-89, :128-140)."""
    progressive_stage = 99

    def get_deltas_starting_dimensions(self):
        return list(range(self.style_count))

    def set_progressive_stage(self, new_stage):
        self.progressive_stage = new_stage
        print('Changed progressive stage to: ', new_stage)

    def _combine(self, per_style):
        w0 = per_style[0]
        w = w0.repeat(self.style_count, 1, 1).permute(1, 0, 2).clone()
        for i in range(1, min(self.progressive_stage + 1, self.style_count)):
            w[:, i] += per_style[i]
        return w


class ProgressiveBackboneEncoder(_Progressive, BackboneEncoder):
    def __init__(self, num_layers, mode='ir', n_styles=16, input_nc=3):
        opts = input_nc if hasattr(input_nc, 'input_nc') else _NC(int(input_nc))
        BackboneEncoder.__init__(self, num_layers, mode, n_styles, opts)
        self.progressive_stage = 99


class ResNetProgressiveBackboneEncoder(_Progressive, ResNetBackboneEncoder):
    def __init__(self, n_styles=16, input_nc=3):
        opts = input_nc if hasattr(input_nc, 'input_nc') else _NC(int(input_nc))
        ResNetBackboneEncoder.__init__(self, n_styles, opts)
        self.progressive_stage = 99
