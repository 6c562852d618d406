"""`GradualStyleBlock`: a feature map of side `spatial` is reduced to 1x1 by log2(spatial) stride-2 3x3 convolutions
(each followed by a default leaky ReLU) and mapped to one style vector by an equalised linear layer
(reference models/setgan/encoder/encoders/map2style.py:8-25).  Sub-module names (`convs.{0,2,4,..}`, `linear`) are the
reference's, so encoder checkpoints load; in eval mode on a GPU the owning encoder runs `convs` on libsg3hip's
matrix-core convolution with the leaky ReLU fused (restyle_psp_encoders.BackboneEncoder._forward_hip).
"""
import math

from torch import nn

from models.stylegan2.model import EqualLinear


def _halving_stack(in_c, out_c, steps):
    """[conv3x3 s2, LeakyReLU] x steps; only the first convolution changes the channel count."""
    layers = []
    for i in range(steps):
        layers.append(nn.Conv2d(in_c if i == 0 else out_c, out_c, kernel_size=3, stride=2, padding=1))
        layers.append(nn.LeakyReLU())
    return nn.Sequential(*layers)


class GradualStyleBlock(nn.Module):
    def __init__(self, in_c, out_c, spatial):
        super().__init__()
        self.out_c, self.spatial = out_c, spatial
        self.convs = _halving_stack(in_c, out_c, int(math.log2(spatial)))
        self.linear = EqualLinear(out_c, out_c, lr_mul=1)

    def forward(self, x):
        return self.linear(self.convs(x).view(-1, self.out_c))
